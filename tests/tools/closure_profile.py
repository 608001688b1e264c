"""cProfile of the E-step closure (model.eval() + backward) on BASELINE config 1: where the host time goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model, load_golden
prob, _ = load_golden("cfg1_bundled")
model = build_model(prob)
x = model.getSVPosteriorOnIndPointsParams()
for t in x:
    t.requires_grad_(True)
def closure():
    for t in x:
        t.grad = None
    loss = -model.eval()
    loss.backward()
    return loss
for _ in range(20):
    closure()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(200):
    closure()
torch.cuda.synchronize()
print("closure: %.3f ms" % (1e3 * (time.time() - t0) / 200))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    closure()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
