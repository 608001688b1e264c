import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.stats.svEM import SVEM_PyTorch
from svgpfa_amd.utils import synthetic
prob = synthetic.make_config("cfg3")
step = dict(max_iter=20, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
optim = {"em_max_iter": 10}
for name in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
    optim[f"{name}_estimate"] = True
    optim[f"{name}_optim_params"] = dict(step)
model = build_model(prob)
t0 = time.time()
hist, el, info, _ = SVEM_PyTorch().maximize(model=model, optim_params=optim, method="ECM", out=io.StringIO(), verbose=False)
torch.cuda.synchronize()
print("10 ECM iterations at cfg3: %.2f s; %s" % (time.time() - t0, info.message))
print("lower bound:", ["%.1f" % h for h in hist])
print("kernel params:", [t.cpu().numpy().round(4).tolist() for t in model.getKernelsParams()])
