"""bench.py contract on the GPU: the one-rank line carries the required fields, and the N = 2 launch line of the driver
(torch.distributed.run, one rank per GPU) is rehearsed with two ranks sharing cuda:0 over gloo: same metric, n_gpus = 2,
and the all-reduced ELBO equals the one-rank ELBO."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_lines_one_and_two_ranks():
    common = ["--config", "cfg2", "--steps", "5", "--warmup", "1"]
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--cpu-sample-trials", "10"] + common,
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = _last_json(one.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j1, key
    assert j1["n_gpus"] == 1 and j1["steps"] == 5 and j1["dtype"] == "f64" and j1["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j1["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(j1["cpu_baseline"])
    assert "workload" in j1["config"] and "model" not in j1["config"]

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--single-device", "--no-cpu-baseline"] + common,
                         capture_output=True, text=True, timeout=900, cwd=REPO)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = _last_json(two.stdout)
    assert j2["n_gpus"] == 2 and j2["metric"] == j1["metric"] and "cpu_baseline" not in j2
    assert abs(j2["elbo"] - j1["elbo"]) <= 1e-11 * abs(j1["elbo"])


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torch.distributed.run in front: the script starts two ranks itself (here both on
    cuda:0 over gloo), rank 0's line says n_gpus = 2 and carries the three loops' values."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--single-device", "--no-cpu-baseline", "--config", "cfg2", "--steps", "5", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    j = _last_json(run.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    for key in ("value", "value_host_synced", "value_pipelined"):
        assert j[key] > 0
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--single-device", "--no-cpu-baseline", "--config", "cfg2", "--steps", "2"],
                         capture_output=True, text=True, timeout=300, cwd=REPO, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and not bad.stdout.strip()


def test_graft_entry_smoke_in_a_fresh_interpreter():
    """The driver's entry point exactly as it calls it: build() then smoke() in a new process (this is the path where
    the HIP library could be loaded before torch)."""
    run = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"],
                         capture_output=True, text=True, timeout=900, cwd=REPO)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-2000:])
    assert "smoke ok" in run.stdout
