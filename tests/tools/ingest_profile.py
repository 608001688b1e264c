"""Where setParamsAndData (spike ingest + engine set-up) spends its time at the north-star size: nested lists of 100 000
spike-time arrays -> CSR -> chunks -> device.  One-off per model, not part of the timed path."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from helpers import build_model
from svgpfa_amd.utils import synthetic
prob = synthetic.make_config("cfg3")
build_model(prob)                      # first touch of every code path
torch.cuda.synchronize()
t0 = time.perf_counter()
model = build_model(prob)
model.eval()
torch.cuda.synchronize()
print(f"build_model + first eval: {time.perf_counter() - t0:.3f} s for {int(prob['spike_offsets'][-1])} spikes")
pr = cProfile.Profile()
pr.enable()
model = build_model(prob)
model.eval()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
