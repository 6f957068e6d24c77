"""Sampled forward pass on the GPU (nnsdp_eval_network, csrc/forward.hpp) against the oracle's pointwise evaluation
(oracle/nnet_io.eval_net, restating evalFeedFwdNet, src/MyNeuralNetwork/MyNeuralNetwork.jl:40-48) and numpy.

Tolerance (floating point, stated): |y_gpu - y_ref| <= 1e-12 max|y_ref| + 1e-13 - the products are summed in a different order
(MFMA k-chains of 4 against BLAS), nothing else differs; measured 1e-15 .. 1e-14 through 40 layers."""
import os

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F
from oracle import nnet_io


def _fixture_net(name):
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    Ms = [np.array(d[f"M{k}"]) for k in range(len(xd) - 1)]
    return na.FeedFwdNet(xdims=xd, Ms=Ms), nnet_io.FeedFwdNet(xdims=xd, Ms=Ms)


def _close(Y, R):
    assert Y.shape == R.shape
    assert np.abs(Y - R).max() <= 1e-12 * np.abs(R).max() + 1e-13, np.abs(Y - R).max()


def test_batch_entry_rejects_bad_arguments_without_a_gpu():
    net, _ = _fixture_net("W10-D5")
    with pytest.raises(ValueError):
        F.evalFeedFwdNetBatch(net, np.zeros((3, 4)))          # wrong input dimension
    from nnsdp_amd import _lib
    lib = _lib.load()
    xd = np.asarray([2, 3, 2], dtype=np.int32)
    assert lib.nnsdp_eval_network(2, xd.ctypes.data_as(_lib.c_int32_p), None, 0, 4, None, None, None) < 0     # null network
    Mz = np.zeros(3 * 3 + 2 * 4)
    assert lib.nnsdp_eval_network(2, xd.ctypes.data_as(_lib.c_int32_p), Mz.ctypes.data_as(_lib.c_double_p), 7, 4, None, None, None) < 0
    assert lib.nnsdp_eval_network(2, xd.ctypes.data_as(_lib.c_int32_p), Mz.ctypes.data_as(_lib.c_double_p), 0, 0, None, None, None) == 0   # N = 0: nothing to do


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["W10-D5", "W20-D10", "W40-D20", "W40-D40"])
def test_reference_networks_match_the_oracle(name):
    net, onet = _fixture_net(name)
    rng = np.random.default_rng(3)
    X = 0.5 + rng.random((2, 4099))                           # not a multiple of the 16-sample tile
    Y = F.evalFeedFwdNetBatch(net, X)
    _close(Y, F.evalFeedFwdNet(net, X))                       # numpy (BLAS) on the host
    for j in (0, 1, 15, 16, 4098):                            # pointwise against the oracle's restatement
        _close(Y[:, j], nnet_io.eval_net(onet, X[:, j]))


@pytest.mark.gpu
@pytest.mark.parametrize("xdims,activ,N", [([3, 7, 9, 8, 6, 7, 2], "relu", 1), ([5, 50, 50, 50, 50, 50, 50, 5], "relu", 1000),
                                           ([2, 17, 33, 4], "tanh", 257), ([1, 1, 1], "relu", 16), ([4, 130, 3], "relu", 50)])
def test_ragged_widths_single_samples_and_tanh(xdims, activ, N):
    rng = np.random.default_rng(sum(xdims))
    Ms = [rng.standard_normal((xdims[k + 1], xdims[k] + 1)) / np.sqrt(xdims[k] + 1) for k in range(len(xdims) - 1)]
    net = na.FeedFwdNet(xdims=xdims, Ms=Ms, activ=na.methods.TanhActiv() if activ == "tanh" else na.methods.ReluActiv())
    X = rng.standard_normal((xdims[0], N))
    Y, ms = F.evalFeedFwdNetBatch(net, X, return_ms=True)
    _close(Y, F.evalFeedFwdNet(net, X))
    assert ms > 0.0


@pytest.mark.gpu
def test_width_limit_is_an_argument_error():
    rng = np.random.default_rng(0)
    net = na.FeedFwdNet(xdims=[2, 700, 2], Ms=[rng.standard_normal((700, 3)), rng.standard_normal((2, 701))])
    with pytest.raises(Exception, match="639"):
        F.evalFeedFwdNetBatch(net, np.zeros((2, 8)))


@pytest.mark.gpu
def test_approx_ellipsoid_matches_the_fixture_and_full_size_linearity():
    """approxEllipsoid through the GPU pass reproduces the committed (numpy-made) ellipsoid of the W40-D20 fixture, and at the full
    1e5 samples the pass is positively homogeneous the way a bias-free ReLU network is: f(c x) = c f(x) for c > 0."""
    d = helpers.load_problem("W40-D20", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    P, yc = na.approxEllipsoid(net, d["x1min"], d["x1max"])
    assert np.allclose(yc, d["yc"], rtol=1e-11, atol=1e-14)
    iP = np.linalg.inv(P)
    assert np.allclose(0.5 * (iP + iP.T), d["invP"], rtol=1e-9, atol=1e-12)
    nb = na.FeedFwdNet(xdims=net.xdims, Ms=[np.hstack([Mk[:, :-1], np.zeros((Mk.shape[0], 1))]) for Mk in net.Ms])
    rng = np.random.default_rng(5)
    X = rng.standard_normal((2, 100000))
    Y1, Y3 = F.evalFeedFwdNetBatch(nb, X), F.evalFeedFwdNetBatch(nb, 3.0 * X)
    assert np.abs(Y3 - 3.0 * Y1).max() <= 1e-12 * np.abs(Y3).max()
