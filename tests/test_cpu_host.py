"""CPU-only tests: C-ABI surface, loud failure modes, host-side sharding logic, 2-rank gloo exchange."""
import os
import re
import json
import socket
import subprocess
import time
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """The library loads without a GPU and exports exactly what include/svgpfa.h declares."""
    from svgpfa_amd import _lib
    hdr = open(os.path.join(REPO, "include", "svgpfa.h")).read()
    declared = set(re.findall(r"\b(svgpfa_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)      # signatures only: plain pointers and sizes, no torch types
    assert "torch" not in code and "at::" not in code and "Tensor" not in code


def test_missing_library_fails_loudly(monkeypatch):
    from svgpfa_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsvgpfa_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_and_model_refuse_to_run_without_gpu():
    from svgpfa_amd.engine import Engine
    from svgpfa_amd.stats.svLowerBound import SVLowerBound
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(2, 3, 8, [4], [0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SVLowerBound()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under svgpfa_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "svgpfa_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert "svgpfa_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_synthetic_is_seed_reproducible_and_neuron_sorted():
    from svgpfa_amd.utils import synthetic
    a = synthetic.make_config("cfg2")
    b = synthetic.make_config("cfg2")
    assert np.array_equal(a["spike_times"], b["spike_times"]) and np.array_equal(a["C"], b["C"])
    off, n, t = a["spike_offsets"], a["spike_neuron"], a["spike_times"]
    for r in range(a["R"]):
        nn, tt = n[off[r]:off[r + 1]], t[off[r]:off[r + 1]]
        assert np.all(np.diff(nn) >= 0)
        same = np.diff(nn) == 0
        assert np.all(np.diff(tt)[same] >= 0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_trials_partitions_and_balances(world):
    from svgpfa_amd.utils import synthetic
    prob = synthetic.make_problem(R=41, N=6, K=2, M=5, seed=3, n_quad=8, rate=9.0)
    seen, spikes = [], []
    for rank in range(world):
        sh = synthetic.shard_trials(prob, rank, world)
        lo, hi = sh["trial_range"]
        seen.extend(range(lo, hi))
        assert sh["R"] == hi - lo > 0
        assert sh["spike_offsets"][0] == 0 and sh["spike_offsets"][-1] == len(sh["spike_times"])
        assert np.array_equal(sh["Z"][0], prob["Z"][0][lo:hi])
        spikes.append(len(sh["spike_times"]))
    assert seen == list(range(prob["R"]))
    assert sum(spikes) == len(prob["spike_times"])
    assert max(spikes) - min(spikes) <= 2 * max(np.diff(prob["spike_offsets"]))


def _pack_shared(out, K):
    return torch.cat([out["elbo"].reshape(1), out["ell"].reshape(1), out["kl"].reshape(1),
                      out["g_C"].reshape(-1), out["g_d"].reshape(-1)] + [g.reshape(-1) for g in out["g_theta"]])


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from oracle import svgpfa_oracle as orc
    from svgpfa_amd import distributed
    from svgpfa_amd.utils import synthetic
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    prob = synthetic.make_problem(R=7, N=5, K=2, M=[4, 5], kernels="mix", seed=21, n_quad=12, rate=6.0, jitter=1.0)
    shard = synthetic.shard_trials(prob, rank, world)
    out = orc.elbo_and_grads(orc.from_numpy_problem(shard), 15)     # stand-in evaluator for the CPU test
    shared = _pack_shared(out, prob["K"])
    distributed.allreduce_shared(shared)
    if rank == 0:
        full = _pack_shared(orc.elbo_and_grads(orc.from_numpy_problem(prob), 15), prob["K"])
        q.put(float((shared - full).abs().max() / full.abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_sum_equals_full_problem():
    """N > 1 path on CPU: trials sharded over 2 ranks, one all-reduce of the shared vector (gloo)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) < 1e-12


def _rank_async(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from svgpfa_amd import distributed
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # bench.py's N > 1 loop: alternating output buffers, the exchange of evaluation i waited for only before buffer
    # i % 2 is written again (and at the end)
    bufs, works, got = [torch.zeros(6, dtype=torch.float64) for _ in range(2)], [None, None], []
    for i in range(7):
        b = i % 2
        if works[b] is not None:
            works[b].wait()
            got.append(bufs[b].clone())
        bufs[b].copy_(torch.arange(6, dtype=torch.float64) * (i + 1) + rank)      # this rank's "evaluation" i
        works[b] = distributed.allreduce_shared_async(bufs[b])
    for b in ((7 % 2), (8 % 2)):                                                   # drain in the order they were enqueued
        if works[b] is not None:
            works[b].wait()
            got.append(bufs[b].clone())
    if rank == 0:
        want = [torch.arange(6, dtype=torch.float64) * (i + 1) * world + sum(range(world)) for i in range(7)]
        q.put(max(float((g - w).abs().max()) for g, w in zip(got, want)) if len(got) == 7 else -1.0)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_async_exchange_with_alternating_buffers():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_async, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 0.0


# ---- bench.py as its own launcher (`python bench.py --gpus N` without torch.distributed.run) -------------------------------
_FAKE_RANK = """
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(3)
if mode == "fail" and rank == 0:
    time.sleep(60)          # waits "in a collective" for the rank that died: the launcher must end it
if rank == 0:
    print("noise before the line")
    print(json.dumps({"n_gpus": 1 if mode == "wrong" else world, "value": 1.0}))
"""


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launcher_starts_its_own_ranks(tmp_path, capfd):
    bench = _bench_module()
    fake = tmp_path / "fake_rank.py"
    fake.write_text(_FAKE_RANK)
    envs = bench.rank_environments(4, 12345, base={})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and all(e["WORLD_SIZE"] == "4" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert bench.launch_ranks(2, ["ok"], script=str(fake), timeout=60) == 0
    out = capfd.readouterr().out
    assert json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])["n_gpus"] == 2
    t0 = time.monotonic()
    assert bench.launch_ranks(2, ["fail"], script=str(fake), timeout=60) == 1      # a dead rank ends the run, loudly
    assert time.monotonic() - t0 < 30
    assert "rank 1 exited with code 3" in capfd.readouterr().err
    assert bench.launch_ranks(2, ["wrong"], script=str(fake), timeout=60) == 1     # a line for another rank count
    assert "n_gpus = 1" in capfd.readouterr().err


def test_bench_refuses_a_rank_count_that_was_not_asked_for():
    """torch.distributed.run with one rank but --gpus 2: exit non-zero before anything is measured (and before torch)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert run.returncode != 0 and "WORLD_SIZE=1" in run.stderr and not run.stdout.strip()
