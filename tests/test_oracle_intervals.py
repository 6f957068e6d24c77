"""CROWN-sliced interval restatement (oracle/intervals.py): soundness by sampling (CROWN and IBP),
and agreement with the committed problem fixtures."""
import os

import numpy as np

import helpers
from oracle import intervals, nnet_io


def _load(name):
    return nnet_io.load_npz(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))


def test_crown_and_ibp_sound():
    net = _load("W10-D10")
    lo, hi = np.array([0.5, 0.5]), np.array([1.5, 1.5])
    iv = intervals.intervals_crown_sliced(net, lo, hi)
    ibp = intervals.intervals_worst_case(net, lo, hi)
    rng = np.random.default_rng(0)
    X = lo[:, None] + rng.random((2, 20000)) * (hi - lo)[:, None]
    xk = X
    for k in range(net.K - 1):
        pre = net.W(k) @ xk + net.b(k)[:, None]
        l, u = iv.acx_intvs[k]
        assert (l[:, None] - pre).max() <= 1e-6 and (pre - u[:, None]).max() <= 1e-6
        xk = np.maximum(pre, 0)
        for src in (iv, ibp):
            l, u = src.x_intvs[k + 1]
            assert (l[:, None] - xk).max() <= 1e-6 and (xk - u[:, None]).max() <= 1e-6
        l, u = iv.x_intvs[k + 1]
        assert np.all(l <= u)      # (CROWN through the last ReLU relaxation may be looser than IBP per neuron)
    y = net.W(net.K - 1) @ xk + net.b(net.K - 1)[:, None]
    l, u = iv.x_intvs[-1]
    assert (l[:, None] - y).max() <= 1e-6 and (y - u[:, None]).max() <= 1e-6


def test_fixture_inputs_reproduce():
    d = helpers.load_problem("W10-D5", 0)
    net = _load("W10-D5")
    iv = intervals.intervals_crown_sliced(net, d["x1min"], d["x1max"])
    assert np.array_equal(np.concatenate([v[0] for v in iv.x_intvs[1:-1]]), d["acymin"])
    assert np.array_equal(np.concatenate([v[1] for v in iv.x_intvs[1:-1]]), d["acymax"])
    from oracle.qc import make_sector_min_max
    smin, smax = make_sector_min_max(np.concatenate([v[0] for v in iv.acx_intvs]), np.concatenate([v[1] for v in iv.acx_intvs]))
    assert np.array_equal(smin, d["smin"]) and np.array_equal(smax, d["smax"])


def test_nnet_fixture_shapes_and_distribution():
    # scripts/make_networks.jl:22-28,43-46: entries N(0, sigma^2), sigma = 2/sqrt(W ln W)
    for name, W in (("W10-D5", 10), ("W40-D20", 40)):
        net = _load(name)
        net.check()
        vals = np.concatenate([M.ravel() for M in net.Ms[1:-1]])
        sigma = 2.0 / np.sqrt(W * np.log(W))
        assert abs(vals.std() / sigma - 1.0) < 0.05


def test_tanh_crown_restatement_is_sound():
    """BoundTanh's relaxation (exts/auto_LiRPA/operators/activation.py:843-1016, the non-optimised branch; the reference's bridge
    maps torch.nn.Tanh onto it, exts/auto_lirpa_bridge.py:31-37,86-87) restated in oracle/intervals.py: every sampled trajectory
    of a tanh network stays inside the bounds of every layer."""
    for xd, seed in (([2, 10, 10, 10, 10, 2], 0), ([3, 12, 9, 14, 4], 1), ([2, 20, 20, 20, 20, 20, 20, 2], 2)):
        net = nnet_io.random_net(xd, seed=seed)
        lo, hi = np.full(xd[0], 0.5), np.full(xd[0], 1.5)
        iv = intervals.intervals_crown_sliced(net, lo, hi, "tanh")
        rng = np.random.default_rng(seed)
        x = lo[:, None] + rng.random((xd[0], 20000)) * (hi - lo)[:, None]
        bl, bu = lo, hi
        for k in range(net.K):
            W, b = net.W(k), net.b(k)
            pre = W @ x + b[:, None]
            if k < net.K - 1:
                l, u = iv.acx_intvs[k]
                assert (l[:, None] - pre).max() <= 1e-6 and (pre - u[:, None]).max() <= 1e-6
            x = np.tanh(pre) if k < net.K - 1 else pre
            l, u = iv.x_intvs[k + 1]
            assert np.all(l <= u)
            assert (l[:, None] - x).max() <= 1e-6 and (x - u[:, None]).max() <= 1e-6, (xd, k)
            # plain interval arithmetic through the same layer
            Wp, Wn = np.maximum(W, 0), np.minimum(W, 0)
            pl, pu = Wp @ bl + Wn @ bu + b, Wp @ bu + Wn @ bl + b
            bl, bu = (np.tanh(pl), np.tanh(pu)) if k < net.K - 1 else (pl, pu)
            # (like the ReLU relaxation, CROWN through the LAST activation may be looser per neuron than interval arithmetic - the
            # reference takes the CROWN box as it is; it must stay a bounded multiple of it)
            assert (u - l).sum() <= 2.0 * (bu - bl).sum() + 1e-6


def test_tanh_relaxation_lines_bound_tanh_on_every_kind_of_interval():
    """lw x + lb <= tanh(x) <= uw x + ub on [l, u] for intervals left of, right of and across zero, narrow and wide"""
    rng = np.random.default_rng(3)
    l = np.concatenate([-rng.random(200) * 4 - 0.01, rng.random(200) * 3, -rng.random(300) * 5, np.full(20, -1e-7), [-600.0, 0.3]]).astype(np.float32)
    w = np.concatenate([rng.random(200) * 0.9 * 0 + 1e-3, rng.random(200) * 3, rng.random(300) * 9 + 1e-3, np.full(20, 2e-7), [1200.0, 1e-9]]).astype(np.float32)
    u = (l + w).astype(np.float32)
    u[:200] = np.minimum(u[:200], 0.0)
    lw, lb, uw, ub = intervals._relax(l, u, "tanh")
    for t in np.linspace(0.0, 1.0, 41):
        # (the library clips the interval to [-500, 500] before it draws the lines, activation.py:926-929: they hold there)
        x = np.clip((l + t * (u - l)).astype(np.float64), -500.0, 500.0)
        y = np.tanh(x)
        assert (lw * x + lb - y).max() <= 2e-6 and (y - (uw * x + ub)).max() <= 2e-6
