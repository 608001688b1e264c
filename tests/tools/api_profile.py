"""Every caller-side entry point once at the north-star size (for rocprofv3 --kernel-trace --stats: no kernel of the
prediction / getter / statistics API should stand out against the evaluation itself)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.utils import synthetic

prob = synthetic.make_config("cfg3")
model = build_model(prob)
def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    print(f"{name:45s} {1e3 * (time.perf_counter() - t0) / reps:9.3f} ms", flush=True)
    return out
times = torch.linspace(0.0, 1.0, 1000, dtype=torch.double).reshape(1, -1, 1).repeat(prob["R"], 1, 1).cuda()
timed("eval", lambda: model.eval())
timed("computeSVPosteriorOnLatentsStats", lambda: model.computeSVPosteriorOnLatentsStats())
timed("freezeLatentsStats", lambda: model.freezeLatentsStats())
stats = model.computeSVPosteriorOnLatentsStats()
timed("evalELLSumAcrossTrialsAndNeurons", lambda: model.evalELLSumAcrossTrialsAndNeurons(svPosteriorOnLatentsStats=stats))
timed("predictLatents (1000 times / trial)", lambda: model.predictLatents(times=times))
timed("predictEmbedding", lambda: model.predictEmbedding(times=times))
timed("computeExpectedPosteriorCIFs", lambda: model.computeExpectedPosteriorCIFs(times=times))
timed("lastTerms", lambda: model.lastTerms())
import pickle
timed("pickle round trip", lambda: pickle.loads(pickle.dumps({"model": model})), reps=1)
