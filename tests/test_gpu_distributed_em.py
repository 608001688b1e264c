"""Trial-sharded EM on the GPU: two ranks share cuda:0 (gloo carries the exchange in this rehearsal; on a node the
backend is nccl = RCCL), each with its own engine over its block of trials, svgpfa_amd.distributed.attach for the
all-reduce of [ELBO, dC, dd, dtheta] and ShardedLBFGS for the optimiser.  The lower-bound history must equal the
one-process run of the same EM loop on the full problem (tolerance 1e-6 relative: L-BFGS line searches amplify
rounding)."""
import io
import os
import socket
import sys

import pytest
import torch

from helpers import build_model
from svgpfa_amd.utils import synthetic

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    return synthetic.make_problem(R=6, N=8, K=2, M=[6, 7], kernels="mix", seed=31, n_quad=40, rate=12.0, jitter=1.0)


def _optim_params():
    step = dict(max_iter=6, lr=1.0, tolerance_grad=1e-7, tolerance_change=1e-9, line_search_fn="strong_wolfe")
    out = {"em_max_iter": 2}
    for name in ("estep", "mstep_embedding", "mstep_kernels", "mstep_indpointslocs"):
        out[f"{name}_estimate"] = True
        out[f"{name}_optim_params"] = dict(step)
    return out


def _run_em(model):
    from svgpfa_amd.stats.svEM import SVEM_PyTorch
    hist, _, info, _ = SVEM_PyTorch().maximize(model=model, optim_params=_optim_params(), method="ECM",
                                               out=io.StringIO(), verbose=False)
    return hist, info.message


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from svgpfa_amd import distributed
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    shard = synthetic.shard_trials(_problem(), rank, world)
    model = distributed.attach(build_model(shard))
    hist, msg = _run_em(model)
    C, d = model.getSVEmbeddingParams()
    theta = model.getKernelsParams()
    if rank == 0:
        # plain numpy (pickled by value): tensors would travel as shared-memory handles that die with this process
        q.put((hist, msg, C.cpu().numpy(), d.cpu().numpy(), [t.cpu().numpy() for t in theta]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_em_matches_single_process():
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    hist, msg, C, d, theta = q.get(timeout=500)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert "Maximum number of iterations" in msg

    model = build_model(_problem())
    href, mref = _run_em(model)
    assert len(hist) == len(href) == 3
    assert abs(hist[0] - href[0]) <= 1e-10 * abs(href[0])
    for a, b in zip(hist[1:], href[1:]):
        assert abs(a - b) <= 1e-6 * abs(b), (hist, href)
    assert hist[-1] > hist[0]
    Cr, dr = model.getSVEmbeddingParams()
    assert torch.allclose(torch.from_numpy(C), Cr.cpu(), rtol=1e-4, atol=1e-6)
    assert torch.allclose(torch.from_numpy(d), dr.cpu(), rtol=1e-4, atol=1e-6)
    for a, b in zip(theta, model.getKernelsParams()):
        assert torch.allclose(torch.from_numpy(a), b.cpu(), rtol=1e-4, atol=1e-6)


def _rank_fail(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from svgpfa_amd import distributed
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    shard = synthetic.shard_trials(_problem(), rank, world)
    model = distributed.attach(build_model(shard))
    if rank == 1:        # rank 1 only: two inducing points of one of ITS trials coincide and the jitter is negative
        Z = model.getIndPointsLocs()[0]
        Z[0, 2, 0] = Z[0, 1, 0]
        model.setPriorCovRegParam(-0.5)
    from svgpfa_amd.stats.svEM import SVEM_PyTorch
    op = _optim_params()
    for name in ("estep", "mstep_embedding"):
        op[f"{name}_estimate"] = False
    hist, _, info, _ = SVEM_PyTorch().maximize(model=model, optim_params=op, method="ECM", out=io.StringIO(),
                                               verbose=False, printIterationModelParams=False)
    q.put((rank, type(info).__name__, str(getattr(info, "error", ""))))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_local_cholesky_failure_is_raised_on_every_rank():
    """A non positive-definite Kzz on ONE rank's trials (Z is per trial) must end the EM step on EVERY rank with
    ErrorTerminationInfo (reference rollback path svEM.py:246-253), not leave the healthy rank waiting in the next
    collective: buildKernelsMatrices() agrees on the status over the ranks before raising."""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_fail, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, kind, err in got:
        assert kind == "ErrorTerminationInfo", got
        assert "positive definite" in err, got
