"""ShardedLBFGS (svgpfa_amd/stats/shardedLBFGS.py) on CPU: (1) with one rank it follows torch.optim.LBFGS, the
optimiser the reference drives (stats/svEM.py:266-294); (2) with parameters sharded over two gloo ranks it reaches the
same iterates as the one-process optimiser on the concatenated problem."""
import os
import socket
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPT = dict(lr=1.0, max_iter=25, tolerance_grad=1e-9, tolerance_change=1e-12, line_search_fn="strong_wolfe")


def _rosen(x):
    return (100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2).sum()


def _run(opt_cls, params, fn, steps=3, **kw):
    opt = opt_cls(params, **kw)
    hist = []
    for _ in range(steps):
        def closure():
            opt.zero_grad()
            loss = fn()
            loss.backward()
            return loss
        opt.step(closure)
        hist.append(float(fn()))
    st = opt.state[opt._params[0]]
    return hist, st["func_evals"], st["n_iter"]


@pytest.mark.parametrize("line_search", ["strong_wolfe", None])
def test_one_rank_follows_torch_lbfgs(line_search):
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    torch.manual_seed(0)
    x0 = torch.randn(12, dtype=torch.double) * 0.5
    kw = dict(OPT, line_search_fn=line_search)
    if line_search is None:
        kw["lr"] = 1e-3
    xa = x0.clone().requires_grad_(True)
    ha, fa, na = _run(torch.optim.LBFGS, [xa], lambda: _rosen(xa), **kw)
    xb = x0.clone().requires_grad_(True)
    hb, fb, nb = _run(ShardedLBFGS, [xb], lambda: _rosen(xb), **kw)
    assert (fa, na) == (fb, nb)
    for a, b in zip(ha, hb):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(a)), (ha, hb)
    assert torch.allclose(xa, xb, rtol=1e-6, atol=1e-8)


def test_history_limit_and_shared_params_one_rank():
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    torch.manual_seed(1)
    x0, z0 = torch.randn(9, dtype=torch.double) * 0.3, torch.randn(3, dtype=torch.double) * 0.3
    kw = dict(OPT, history_size=4, max_iter=40)

    def f(x, z):
        return _rosen(torch.cat([z, x])) + 0.5 * (z ** 2).sum()
    xa, za = x0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
    ha, fa, na = _run(torch.optim.LBFGS, [za, xa], lambda: f(xa, za), steps=2, **kw)
    xb, zb = x0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
    opt_kw = dict(kw)
    hb, fb, nb = _run(lambda p, **k: ShardedLBFGS([xb], [zb], **k), None, lambda: f(xb, zb), steps=2, **opt_kw)
    assert (fa, na) == (fb, nb)
    for a, b in zip(ha, hb):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(a))


def _problem(world):
    g = torch.Generator().manual_seed(5)
    z0 = torch.randn(4, dtype=torch.double, generator=g) * 0.2
    xs = [torch.randn(6 + r, dtype=torch.double, generator=g) * 0.2 for r in range(world)]
    As = [torch.randn(8, 6 + r, dtype=torch.double, generator=g) for r in range(world)]
    bs = [torch.randn(8, dtype=torch.double, generator=g) for r in range(world)]
    return z0, xs, As, bs


def _f_rank(x, z, A, b):
    # non-quadratic, couples the local block with the replicated parameters
    return ((A @ x - b) ** 2).sum() + torch.exp(0.3 * x[:4] * z).sum() + 0.1 * (x ** 4).sum()


def _f_shared(z):
    return 0.5 * (z ** 2).sum() + 0.05 * (z ** 4).sum()


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z0, xs, As, bs = _problem(world)
    z = z0.clone().requires_grad_(True)
    x = xs[rank].clone().requires_grad_(True)
    opt = ShardedLBFGS([x], [z], **OPT)

    def value():
        v = _f_rank(x, z, As[rank], bs[rank]) + (_f_shared(z) if rank == 0 else 0.0)
        return v

    def closure():
        opt.zero_grad()
        loss = value()
        loss.backward()
        tot = loss.detach().clone()
        dist.all_reduce(tot)            # what svgpfa_amd.distributed.attach does for [ELBO, dC, dd, dtheta]
        dist.all_reduce(z.grad)
        return tot
    hist = []
    for _ in range(2):
        opt.step(closure)
        v = value().detach().clone()
        dist.all_reduce(v)
        hist.append(float(v))
    st = opt.state[opt._params[0]]
    gathered = [None] * world
    dist.all_gather_object(gathered, x.detach().numpy())
    if rank == 0:   # plain numpy (pickled by value): tensors would travel as shared-memory handles that die with this process
        q.put((hist, st["func_evals"], st["n_iter"], z.detach().numpy(), gathered, opt.collectives))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
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
    hist, nfe, nit, z, xs_out, ncoll = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    z0, xs, As, bs = _problem(world)
    zr = z0.clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]

    def full():
        return sum(_f_rank(xr[r], zr, As[r], bs[r]) for r in range(world)) + _f_shared(zr)
    href, fref, nref = _run(torch.optim.LBFGS, [zr] + xr, full, steps=2, **OPT)
    assert (nfe, nit) == (fref, nref)
    for a, b in zip(hist, href):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(b)), (hist, href)
    assert torch.allclose(torch.from_numpy(z), zr.detach(), rtol=1e-5, atol=1e-7)
    for a, b in zip(xs_out, xr):
        assert torch.allclose(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-7)
    # one all-gather per iteration + a few per line-search evaluation: far fewer than 2 x history inner products
    assert ncoll <= 4 * nfe + 3 * nit + 4


# ---- flat mode: one contiguous vector, closure -> (value, gradient); vectors on the host or on the device ------------------
def _flat_run(x, fn, steps, **kw):
    """ShardedLBFGS(flat=x) on fn, with the closure flat mode expects (no autograd graph held by the optimiser)."""
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    maximize = kw.get("maximize", False)
    opt = ShardedLBFGS(flat=x, **kw)

    def closure():
        with torch.enable_grad():
            xv = x.detach().clone().requires_grad_(True)
            v = fn(xv)
            g, = torch.autograd.grad(v, xv)
        return (-v.detach(), -g) if maximize else (v.detach(), g)
    hist = []
    for _ in range(steps):
        opt.step(closure)
        hist.append(float(fn(x)))
    st = opt.state[opt._params[0]]
    return hist, st["func_evals"], st["n_iter"], opt


@pytest.mark.parametrize("path", ["host", "device"])
@pytest.mark.parametrize("maximize", [False, True])
@pytest.mark.parametrize("line_search", ["strong_wolfe", None])
def test_flat_mode_follows_torch_lbfgs(path, maximize, line_search):
    torch.manual_seed(0)
    x0 = torch.randn(12, dtype=torch.double) * 0.5
    kw = dict(OPT, line_search_fn=line_search)
    if line_search is None:
        kw["lr"] = 1e-3
    xa = x0.clone().requires_grad_(True)
    ha, fa, na = _run(torch.optim.LBFGS, [xa], lambda: _rosen(xa), **kw)
    xb = x0.clone()
    hb, fb, nb, opt = _flat_run(xb, _rosen, 3, maximize=maximize, host_max=(1 << 20) if path == "host" else 0, **kw)
    assert opt._host == (path == "host")
    assert (fa, na) == (fb, nb)
    for a, b in zip(ha, hb):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(a)), (ha, hb)
    assert torch.allclose(xa.detach(), xb, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("path", ["host", "device"])
def test_flat_mode_history_limit(path):
    torch.manual_seed(1)
    x0 = torch.randn(12, dtype=torch.double) * 0.3
    kw = dict(OPT, history_size=4, max_iter=40)
    xa = x0.clone().requires_grad_(True)
    ha, fa, na = _run(torch.optim.LBFGS, [xa], lambda: _rosen(xa), steps=2, **kw)
    xb = x0.clone()
    hb, fb, nb, _ = _flat_run(xb, _rosen, 2, host_max=(1 << 20) if path == "host" else 0, **kw)
    assert (fa, na) == (fb, nb)
    for a, b in zip(ha, hb):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(a)), (ha, hb)


def _flat_rank_main(rank, world, port, q):
    """The EM loop's placement: a per-trial group (flat, local: vectors on the device, inner products exchanged) and a
    replicated group (flat, shared, small: vectors on the host, NO collective), optimised in alternation."""
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z0, xs, As, bs = _problem(world)
    z, x = z0.clone(), xs[rank].clone()

    def value_and_grads():
        with torch.enable_grad():
            xv, zv = x.detach().clone().requires_grad_(True), z.detach().clone().requires_grad_(True)
            v = _f_rank(xv, zv, As[rank], bs[rank]) + (_f_shared(zv) if rank == 0 else 0.0)
            gx, gz = torch.autograd.grad(v, [xv, zv])
        tot = v.detach().clone()
        dist.all_reduce(tot)
        dist.all_reduce(gz)
        return tot, gx, gz
    hist, colls = [], []
    for _ in range(2):
        ox = ShardedLBFGS(flat=x, shared=False, host_max=0, **OPT)
        ox.step(lambda: value_and_grads()[:2])
        oz = ShardedLBFGS(flat=z, shared=True, **OPT)
        assert oz._host
        oz.step(lambda: (lambda t: (t[0], t[2]))(value_and_grads()))
        hist.append(float(value_and_grads()[0]))
        colls.append((ox.collectives, oz.collectives))
    gathered = [None] * world
    dist.all_gather_object(gathered, x.numpy())
    if rank == 0:
        q.put((hist, z.numpy(), gathered, colls))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    hist, z, xs_out, colls = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    z0, xs, As, bs = _problem(world)
    zr = z0.clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]

    def full():
        return sum(_f_rank(xr[r], zr, As[r], bs[r]) for r in range(world)) + _f_shared(zr)
    href = []
    for _ in range(2):
        for params in (xr, [zr]):
            opt = torch.optim.LBFGS(params, **OPT)

            def closure():
                opt.zero_grad()
                loss = full()
                loss.backward()
                return loss
            opt.step(closure)
        href.append(float(full()))
    for a, b in zip(hist, href):
        assert abs(a - b) <= 1e-7 * max(1.0, abs(b)), (hist, href)
    assert torch.allclose(torch.from_numpy(z), zr.detach(), rtol=1e-5, atol=1e-7)
    for a, b in zip(xs_out, xr):
        assert torch.allclose(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-7)
    for cx, cz in colls:
        assert cx > 0 and cz == 0          # the replicated group needs no exchange at all
