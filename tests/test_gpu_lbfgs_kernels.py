"""GPU tests of the optimiser's fused vector kernels (svgpfa_amd/csrc/lbfgs_api.hip, include/svgpfa.h svgpfa_lbfgs_*):
each entry point against the same primitive in plain tensor operations, and a flat-mode optimisation on the GPU against
torch.optim.LBFGS -- the optimiser the reference drives (stats/svEM.py:266-294) -- on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _pair(n, ns, w0, seed=0):
    from svgpfa_amd.stats.shardedLBFGS import _HipVec, _TorchVec
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, dtype=torch.float64, generator=g).cuda()
    return _HipVec(x, ns, w0, 100), _TorchVec(x, ns, w0, 100), x, g


def _close(a, b, scale=None):
    a, b = a.double().cpu(), b.double().cpu()
    s = float(b.abs().max()) if scale is None else scale
    assert float((a - b).abs().max()) <= TOL * max(1.0, s), (a, b)


@pytest.mark.parametrize("n,ns,w0", [(1, 0, 1.0), (2047, 0, 1.0), (2048, 2048, 0.0), (10007, 100, 0.0), (300001, 0, 1.0)])
def test_grad_stats(n, ns, w0):
    hip, ref, x, gen = _pair(n, ns, w0)
    graw = torch.randn(n, dtype=torch.float64, generator=gen).cuda()
    d = torch.randn(n, dtype=torch.float64, generator=gen).cuda()
    val = torch.tensor(3.25, dtype=torch.float64).cuda()
    carry = torch.tensor([7.5], dtype=torch.float64).cuda()
    for sign in (1.0, -1.0):
        for dd in (d, None):
            g1, o1 = hip.grad_stats(graw, sign, dd, val, carry)
            g2, o2 = ref.grad_stats(graw, sign, dd, val, carry)
            assert torch.equal(g1, g2)
            _close(o1, o2, scale=float(n))
    # without value / carry the slots read zero
    _, o = hip.grad_stats(graw, 1.0, None, None, None)
    assert float(o[4]) == 0.0 and float(o[5]) == 0.0
    # NaN in the gradient reaches max |g| (torch's abs().max() semantics)
    graw[n // 2] = float("nan")
    _, o = hip.grad_stats(graw, 1.0, d, val, carry)
    assert np.isnan(float(o[3]))


@pytest.mark.parametrize("n,ns,w0,m", [(5, 0, 1.0, 0), (2049, 0, 1.0, 3), (10007, 10007, 0.0, 5), (70001, 11, 0.0, 17)])
def test_pair_dots_and_direction(n, ns, w0, m):
    hip, ref, x, gen = _pair(n, ns, w0, seed=m)
    rnd = lambda *s: torch.randn(*s, dtype=torch.float64, generator=gen).cuda()
    g, pg, d = rnd(n), rnd(n), rnd(n)
    S1, Y1 = rnd(m + 2, n), rnd(m + 2, n)
    S2, Y2 = S1.clone(), Y1.clone()
    t = 0.37
    o1 = hip.pair_dots(g, pg, d, t, S1, Y1, m)
    o2 = ref.pair_dots(g, pg, d, t, S2, Y2, m)
    assert o1.numel() == 5 * m + 5
    _close(o1, o2, scale=float(n))
    assert torch.equal(S1, S2) and torch.equal(Y1, Y2)          # candidate pair written into row m, nothing else touched
    coef = np.concatenate([np.linspace(-1.0, 1.0, m), np.linspace(0.5, -0.5, m), [0.8]])
    d1, d2, x01, x02 = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    m1 = hip.direction(g, S1, Y1, m, coef, d1, x, x01)
    m2 = ref.direction(g, S2, Y2, m, coef, d2, x, x02)
    _close(d1, d2)
    _close(m1, m2)
    assert torch.equal(x01, x) and torch.equal(x02, x)
    # bit-reproducible
    o3 = hip.pair_dots(g, pg, d, t, S1, Y1, m)
    assert torch.equal(o1, o3)


def _chain(x):      # Rosenbrock chain: non-quadratic, couples neighbours
    return (100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2).sum()


@pytest.mark.parametrize("maximize", [False, True])
def test_flat_optimisation_on_gpu_follows_torch_lbfgs(maximize):
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS, _HipVec
    opt_kw = dict(lr=1.0, max_iter=30, tolerance_grad=1e-9, tolerance_change=1e-12, line_search_fn="strong_wolfe")
    torch.manual_seed(3)
    x0 = torch.randn(5000, dtype=torch.float64) * 0.3
    xa = x0.clone().requires_grad_(True)
    ref = torch.optim.LBFGS([xa], **opt_kw)
    hist_a = []
    for _ in range(2):
        def closure():
            ref.zero_grad()
            loss = _chain(xa)
            loss.backward()
            return loss
        ref.step(closure)
        hist_a.append(float(_chain(xa)))
    xb = x0.clone().cuda()
    opt = ShardedLBFGS(flat=xb, maximize=maximize, host_max=0, **opt_kw)

    def flat_closure():
        with torch.enable_grad():
            xv = xb.detach().clone().requires_grad_(True)
            v = _chain(xv)
            g, = torch.autograd.grad(v, xv)
        return (-v.detach(), -g) if maximize else (v.detach(), g)
    hist_b = []
    for _ in range(2):
        opt.step(flat_closure)
        hist_b.append(float(_chain(xb)))
    assert isinstance(opt._vec, _HipVec)
    sa, sb = ref.state[ref._params[0]], opt.state[opt._params[0]]
    assert (sa["func_evals"], sa["n_iter"]) == (sb["func_evals"], sb["n_iter"])
    for a, b in zip(hist_a, hist_b):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (hist_a, hist_b)
    assert torch.allclose(xa.detach(), xb.cpu(), rtol=1e-5, atol=1e-7)


def test_host_path_on_gpu_parameters():
    """A small replicated group: vectors on the host, parameters and closure on the GPU."""
    from svgpfa_amd.stats.shardedLBFGS import ShardedLBFGS
    opt_kw = dict(lr=1.0, max_iter=25, tolerance_grad=1e-9, tolerance_change=1e-12, line_search_fn="strong_wolfe")
    torch.manual_seed(4)
    x0 = torch.randn(40, dtype=torch.float64) * 0.3
    xa = x0.clone().requires_grad_(True)
    ref = torch.optim.LBFGS([xa], **opt_kw)

    def closure():
        ref.zero_grad()
        loss = _chain(xa)
        loss.backward()
        return loss
    ref.step(closure)
    xb = x0.clone().cuda()
    opt = ShardedLBFGS(flat=xb, shared=True, maximize=True, **opt_kw)
    assert opt._host

    def flat_closure():
        with torch.enable_grad():
            xv = xb.detach().clone().requires_grad_(True)
            v = _chain(xv)
            g, = torch.autograd.grad(v, xv)
        return -v.detach(), -g
    opt.step(flat_closure)
    sa, sb = ref.state[ref._params[0]], opt.state[opt._params[0]]
    assert (sa["func_evals"], sa["n_iter"]) == (sb["func_evals"], sb["n_iter"])
    assert torch.allclose(xa.detach(), xb.cpu(), rtol=1e-6, atol=1e-8)
