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
