"""Rows f1-f3 of SURVEY.md section 8: the product's host-side .nnet reader, CROWN-sliced intervals and QC
construction against the oracle restatement and the committed fixtures (CPU), and the findEllipsoid
front-end end to end on the GPU."""
import csv
import os

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F
from oracle import intervals as ointv, nnet_io


def _nnet_text(net, path):
    # the reference writer's layout (exts/NNet/utils/writeNNet.py:3-80, '%.9f' values)
    with open(path, "w") as f:
        f.write("// Neural Network File Format by Kyle Julian, Stanford 2016\n")
        f.write(f"{len(net.Ms)},{net.xdims[0]},{net.xdims[-1]},{max(net.xdims)},\n")
        f.write(",".join(str(v) for v in net.xdims) + ",\n0,\n")
        f.write(",".join(["-10000.0"] * net.xdims[0]) + ",\n" + ",".join(["10000.0"] * net.xdims[0]) + ",\n")
        f.write(",".join(["0.0"] * (net.xdims[0] + 1)) + ",\n" + ",".join(["1.0"] * (net.xdims[0] + 1)) + ",\n")
        for Mk in net.Ms:
            for row in Mk[:, :-1]:
                f.write(",".join("%.9f" % v for v in row) + ",\n")
            for v in Mk[:, -1]:
                f.write("%.9f,\n" % v)


def test_nnet_reader_roundtrip(tmp_path):
    ref = nnet_io.load_npz(os.path.join(helpers.GOLDEN, "nets", "scale-I2-O2-W10-D5.npz"))
    p = str(tmp_path / "net.nnet")
    _nnet_text(ref, p)
    net = na.read_nnet(p)
    assert net.xdims == ref.xdims
    for a, b in zip(net.Ms, ref.Ms):
        assert np.array_equal(a, b)                       # the fixture holds exactly the %.9f values
    o = nnet_io.read_nnet(p)
    x = np.array([0.7, 1.2])
    assert np.allclose(na.evalFeedFwdNet(net, x), nnet_io.eval_net(o, x), rtol=0, atol=1e-15)


@pytest.mark.parametrize("name,beta", [("W10-D5", 0), ("W10-D10", 2), ("W20-D10", 0)])
def test_intervals_and_qcs_match_fixture_and_oracle(name, beta):
    d = helpers.load_problem(name, beta)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    qb, qs = na.makeQcActivs(net, d["x1min"], d["x1max"], beta)
    assert np.allclose(qb.acymin, d["acymin"], rtol=0, atol=2e-6) and np.allclose(qb.acymax, d["acymax"], rtol=0, atol=2e-6)
    assert np.array_equal(qs.smin, d["smin"]) and np.array_equal(qs.smax, d["smax"])
    assert qs.vardim == len(d["smin"]) * (beta + 3) - beta * (beta + 1) // 2
    x_intvs, acx = na.makeIntervalsInfo(d["x1min"], d["x1max"], net)
    o = ointv.intervals_crown_sliced(nnet_io.FeedFwdNet(xdims=net.xdims, Ms=net.Ms), d["x1min"], d["x1max"])
    for (l, u), (lo, uo) in zip(x_intvs, o.x_intvs):
        assert np.allclose(l, lo, rtol=0, atol=2e-6) and np.allclose(u, uo, rtol=0, atol=2e-6)


def test_native_intervals_full_size_fixture():
    # BASELINE configs[1]: W40-D20, 800 hidden neurons; native C++ pre-processing against the committed fixture
    d = helpers.load_problem("W40-D20", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    qb, qs = na.makeQcActivs(net, d["x1min"], d["x1max"], 0)
    scale = np.maximum(1.0, np.abs(d["acymax"]))
    assert np.max(np.abs(qb.acymin - d["acymin"]) / scale) < 1e-5 and np.max(np.abs(qb.acymax - d["acymax"]) / scale) < 1e-5
    # sector flags may differ only where the pre-activation bound sits within float32 noise of the +-1e-4 threshold
    assert np.mean(qs.smin != d["smin"]) < 0.005 and np.mean(qs.smax != d["smax"]) < 0.005


def test_native_intervals_are_sound_on_ragged_net():
    rng = np.random.default_rng(5)
    xdims = [3, 7, 5, 9, 2]
    Ms = [rng.standard_normal((xdims[k + 1], xdims[k] + 1)) * 0.7 for k in range(4)]
    net = na.FeedFwdNet(xdims=xdims, Ms=Ms)
    lo, hi = np.array([-0.3, 0.1, -1.0]), np.array([0.4, 0.5, -0.2])
    x_intvs, acx = na.makeIntervalsInfo(lo, hi, net)
    o = ointv.intervals_crown_sliced(nnet_io.FeedFwdNet(xdims=xdims, Ms=Ms), lo, hi)
    for (l, u), (lo_, uo_) in zip(x_intvs, o.x_intvs):
        assert np.allclose(l, lo_, atol=1e-5) and np.allclose(u, uo_, atol=1e-5)
    for (l, u), (lo_, uo_) in zip(acx, o.acx_intvs):
        assert np.allclose(l, lo_, atol=1e-5) and np.allclose(u, uo_, atol=1e-5)
    X = lo[:, None] + (hi - lo)[:, None] * rng.random((3, 20000))
    xk = X
    for k in range(3):
        pre = Ms[k][:, :-1] @ xk + Ms[k][:, -1:]
        assert np.all(pre >= acx[k][0][:, None] - 1e-5) and np.all(pre <= acx[k][1][:, None] + 1e-5)
        xk = np.maximum(pre, 0)
        assert np.all(xk >= x_intvs[k + 1][0][:, None] - 1e-5) and np.all(xk <= x_intvs[k + 1][1][:, None] + 1e-5)
    y = Ms[3][:, :-1] @ xk + Ms[3][:, -1:]
    assert np.all(y >= x_intvs[4][0][:, None] - 1e-5) and np.all(y <= x_intvs[4][1][:, None] + 1e-5)


def test_native_intervals_reject_bad_box():
    net = na.FeedFwdNet(xdims=[2, 3, 2], Ms=[np.ones((3, 3)), np.ones((2, 4))])
    with pytest.raises(RuntimeError):
        na.makeIntervalsInfo([1.0, 0.0], [0.0, 1.0], net)
    with pytest.raises(ValueError):
        na.makeIntervalsInfo([0.0], [1.0], net)


def test_scale_csv_layout(tmp_path):
    s = na.QuerySolution(objective_value=1.5, values={}, termination_status="OPTIMAL", total_time=3.0, setup_time=1.0,
                         solve_time=2.0, summary={"lambda_max": 1e-9})
    p = str(tmp_path / "x.csv")
    na.write_scale_csv(p, [(0, s), (1, s)])
    rows = open(p).read().strip().split("\n")
    assert rows[0] == "beta,setup_secs,solve_secs,total_secs,obj_val,term_status,eigmax" and len(rows) == 3
    ref_hdr = open(os.path.join(helpers.GOLDEN, "dump_scale.csv")).readline().strip().split(",")[2:]
    assert rows[0].split(",") == ref_hdr                   # same columns as the reference's dump/scale files


@pytest.mark.gpu
def test_find_ellipsoid_end_to_end_gpu():
    d = helpers.load_problem("W10-D10", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    P, yc, soln = na.findEllipsoid(net, [0.5, 0.5], [1.5, 1.5], 0, na.AdmmSdpOptions(max_iters=100000, decomp_mode=na.DoubleDecomp()))
    pub = helpers.published_rho("W10-D10", 0)
    assert soln.termination_status == "OPTIMAL"
    assert min(abs(soln.objective_value - p) / p for p in pub) <= 1e-3
    assert soln.summary["lambda_max"] <= 1e-6
    # every sampled output lies inside the certified set |invP y - yc|^2 <= rho (output.jl:91-93 form)
    rng = np.random.default_rng(0)
    Y = na.evalFeedFwdNet(net, 0.5 + rng.random((2, 20000)))
    invP = np.linalg.inv(P / np.sqrt(soln.objective_value))
    assert (np.sum((invP @ Y - yc[:, None]) ** 2, axis=0)).max() <= soln.objective_value * (1 + 1e-9)


@pytest.mark.gpu
def test_find_reach_2d_poly_batched_matches_sequential():
    """the 6 hyperplane SDPs of findReach2Dpoly in lockstep through the batch handle against one-by-one solves; every
    sampled output lies inside the certified polytope (src/NnSdp.jl:73-95)."""
    d = helpers.load_problem("W10-D5", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    opts = na.AdmmSdpOptions(max_iters=100000, eps_rel=1e-6)
    hb, sb = na.findReach2Dpoly(net, d["x1min"], d["x1max"], 1, opts)
    hs, ss = na.findReach2Dpoly(net, d["x1min"], d["x1max"], 1, opts, batched=False)
    assert len(hb) == 6 and all(s.termination_status == "OPTIMAL" for s in sb + ss)
    for (nb, ob), (n1, o1) in zip(hb, hs):
        assert np.array_equal(nb, n1) and abs(ob - o1) <= 1e-5 * max(1.0, abs(o1))
    rng = np.random.default_rng(0)
    X = d["x1min"][:, None] + (d["x1max"] - d["x1min"])[:, None] * rng.random((2, 5000))
    Y = na.evalFeedFwdNet(net, X)
    for nrm, off in hb:
        assert np.all(nrm @ Y <= off + 1e-6)


@pytest.mark.gpu
def test_run_scale_beta_sweep_batched(tmp_path):
    """experiments/scale.jl on one network: the beta sweep (SDPs of different sizes) in lockstep through the batch handle
    gives the rows of one-by-one solves; the published rows of dump/scale for this network are met within 1e-3."""
    d = np.load(os.path.join(helpers.GOLDEN, "nets", "scale-I2-O2-W10-D10.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    opts = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=400000, eps_rel=1e-6)
    betas = [0, 3, 7]
    out = str(tmp_path / "scale.csv")
    rows = na.runScale(net, [0.5, 0.5], [1.5, 1.5], betas, opts, saveto=out)
    seq = na.runScale(net, [0.5, 0.5], [1.5, 1.5], betas, opts, batched=False)
    for (b, s), (b1, s1) in zip(rows, seq):
        assert b == b1 and s.termination_status == s1.termination_status == "OPTIMAL"
        assert abs(s.objective_value - s1.objective_value) <= 2e-6 * abs(s1.objective_value)
        pub = helpers.published_rho("W10-D10", b)
        assert pub and min(abs(s.objective_value - p) / abs(p) for p in pub) <= 1e-3
    got = list(csv.reader(open(out)))
    assert got[0] == ["beta", "setup_secs", "solve_secs", "total_secs", "obj_val", "term_status", "eigmax"] and [r[0] for r in got[1:]] == ["0", "3", "7"]


def test_random_network_distribution():
    net = na.randomNetwork([2, 40, 40, 40, 2], seed=3)
    ref = nnet_io.random_net([2, 40, 40, 40, 2], seed=3)            # the oracle's generator: same stream, same scale
    assert all(np.array_equal(a, b) for a, b in zip(net.Ms, ref.Ms))
    assert abs(np.std(np.concatenate([m.ravel() for m in net.Ms])) - 2 / np.sqrt(40 * np.log(40))) < 0.01
    assert np.std(na.randomNetwork([3, 8, 2], sigma=0.5, seed=1).Ms[0]) == pytest.approx(0.5, rel=0.3)
    with pytest.raises(ValueError):
        na.randomNetwork([3])


@pytest.mark.parametrize("xd,seed", [([2, 10, 10, 10, 10, 2], 0), ([3, 20, 20, 20, 4], 1), ([2, 40, 40, 40, 40, 40, 2], 2)])
def test_native_tanh_intervals_match_the_oracle(xd, seed):
    """nnsdp_make_intervals_activ with NNSDP_ACTIV_TANH (csrc/intervals.hpp, BoundTanh restated in C++) against oracle/intervals.py:
    float32 summation order is the only difference (<= 2e-6); sector slopes by makeSectorMinMax's tanh branch."""
    from oracle import intervals as oi, nnet_io
    onet = nnet_io.random_net(xd, seed=seed)
    net = na.FeedFwdNet(xdims=onet.xdims, Ms=onet.Ms, activ=na.methods.TanhActiv)
    lo, hi = np.full(xd[0], 0.5), np.full(xd[0], 1.5)
    iv = oi.intervals_crown_sliced(onet, lo, hi, "tanh")
    x_intvs, acx = F.makeIntervalsInfo(lo, hi, net)
    for (a, b), (c, d) in zip(x_intvs, iv.x_intvs):
        assert np.abs(a - c).max() <= 2e-6 and np.abs(b - d).max() <= 2e-6
    for (a, b), (c, d) in zip(acx, iv.acx_intvs):
        assert np.abs(a - c).max() <= 2e-6 and np.abs(b - d).max() <= 2e-6
    acymin, acymax, acxmin, acxmax, smin, smax, _, _ = F._intervals_native(lo, hi, net)
    s0, s1 = F.makeSectorMinMax(acxmin, acxmax, na.methods.TanhActiv)
    assert np.allclose(smin, s0, rtol=1e-13, atol=0) and np.allclose(smax, s1, rtol=1e-13, atol=0)      # (libm tanh vs numpy tanh: last bit)
    assert np.all((smin > 0) & (smin <= 1) & (smax > 0) & (smax <= 1))
