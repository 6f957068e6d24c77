"""Parity holes closed in round 2 (VERDICT r01 "missing" items 4, 5, 8 and ADVICE r01):
  scaleS                      src/Qc/output.jl:109-124 (caller experiments/vnnlib_utils.jl:46), loadFromFileScaled network_files.jl:84-114
  Tanh sector QC              src/Qc/activ_sector.jl:19 (vardim without eta / nu), :74-86 (makeSectorMinMax tanh branch)
  DoubleRelaxDecomp alias     src/Methods/chordal_sdp.jl:8,25
  generic obj_func            src/Methods/Methods.jl:41
  isSolutionGood              experiments/acas.jl:71-79 - here the certificate is always checked (status alone is not one)
CPU tests compare the product's host logic with the oracle's literal restatement; the -m gpu tests go through the C ABI."""
import os

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F, vnnlib as V
from oracle import admm as oadmm, nnet_io, operator as oop, qc


def _small_net(activ=na.methods.ReluActiv, seed=3, xdims=(2, 6, 5, 6, 2)):
    rng = np.random.default_rng(seed)
    Ms = [rng.normal(0, 0.6, size=(xdims[k + 1], xdims[k] + 1)) for k in range(len(xdims) - 1)]
    return na.FeedFwdNet(xdims=list(xdims), Ms=Ms, activ=activ), nnet_io.FeedFwdNet(xdims=list(xdims), Ms=Ms)


# ----------------------------------------------------------------------------- scaleS / loadFromFileScaled
def test_scaleS_matches_literal_restatement_and_is_the_scaled_set():
    net, onet = _small_net()
    rng = np.random.default_rng(0)
    sd = net.xdims[0] + net.xdims[-1] + 1
    S = rng.standard_normal((sd, sd))
    S = 0.5 * (S + S.T)
    al = np.array([0.7, 1.3, 0.9, 1.1])
    got = V.scaleS(S, al, net)
    want = qc.scale_S(S, al, onet)
    assert np.abs(got - want).max() <= 1e-15 * np.abs(S).max()
    # the quadratic form on (x, alpha*y, 1) under scaleS equals the original one on (x, y, 1)
    a = np.prod(al)
    for _ in range(5):
        x, y = rng.standard_normal(net.xdims[0]), rng.standard_normal(net.xdims[-1])
        v, vs = np.concatenate([x, y, [1.0]]), np.concatenate([x, a * y, [1.0]])
        assert abs(vs @ got @ vs - v @ S @ v) <= 1e-12 * max(1.0, abs(v @ S @ v))
    with pytest.raises(ValueError):
        V.scaleS(S, al[:-1], net)
    # hyperplane set through the CNF loader, as experiments/vnnlib_utils.jl:45-47 does
    Sh = V.hplaneS([1.0, -2.0], 0.3, net)
    assert np.abs(V.scaleS(Sh, al, net) - qc.scale_S(qc.hplane_S([1.0, -2.0], 0.3, onet), al, onet)).max() == 0.0


def test_loadFromFileScaled_scales_the_function(tmp_path):
    ref = nnet_io.load_npz(os.path.join(helpers.GOLDEN, "nets", "scale-I2-O2-W10-D5.npz"))
    from test_frontend import _nnet_text
    p = str(tmp_path / "net.nnet")
    _nnet_text(ref, p)
    base, al0 = F.loadFromFileScaled(p)
    assert np.all(al0 == 1.0)
    x = np.array([0.8, 1.2])
    for scaling in ("sqrtlog", ("norm", 1.5), ("const", 0.9)):
        net, al = F.loadFromFileScaled(p, scaling)
        assert len(al) == net.K
        assert np.allclose(F.evalFeedFwdNet(net, x), np.prod(al) * F.evalFeedFwdNet(base, x), rtol=1e-12, atol=1e-14)
    net, al = F.loadFromFileScaled(p, ("norm", 1.5))
    assert np.allclose([np.linalg.norm(Mk[:, :-1], 2) for Mk in net.Ms], 1.5)
    with pytest.raises(ValueError):
        F.loadFromFileScaled(p, "bogus")


# ----------------------------------------------------------------------------- tanh sector bounds
def test_makeSectorMinMax_both_branches_match_the_restatement():
    rng = np.random.default_rng(1)
    lo = rng.normal(0, 1.5, 200)
    hi = lo + rng.random(200) * 2
    for activ, name in ((na.methods.ReluActiv, "relu"), (na.methods.TanhActiv, "tanh")):
        smin, smax = F.makeSectorMinMax(lo, hi, activ)
        omin, omax = qc.make_sector_min_max(lo, hi, name)
        assert np.array_equal(smin, omin) and np.array_equal(smax, omax)
    # the tanh sector is a valid slope bound on [lo, hi].  Reference quirk kept: on a negative interval the branch at
    # activ_sector.jl:76-77 returns smin > smax (the sector generator is symmetric in the two slopes, so nothing breaks)
    smin, smax = F.makeSectorMinMax(lo, hi, na.methods.TanhActiv)
    assert np.any(smin > smax) and np.all(smin[lo > 0] <= smax[lo > 0])
    for t in np.linspace(0.0, 1.0, 7):
        x = lo + t * (hi - lo)
        s = np.tanh(x) / np.where(x == 0, 1.0, x)
        assert np.all(s >= np.minimum(smin, smax) - 1e-12) and np.all(s <= np.maximum(smin, smax) + 1e-12)


def test_tanh_sector_vardim_and_interval_path():
    net, _ = _small_net(na.methods.TanhActiv)
    qa = F.makeQcActivs(net, [-0.2, 0.1], [0.4, 0.5], 2)
    acdim = net.acdim
    lam = (2 + 1) * acdim - 3
    assert qa[1].vardim == lam                       # no eta / nu (activ_sector.jl:19)
    assert na.QcActivSector(acxdim=acdim, beta=2, smin=qa[1].smin, smax=qa[1].smax).vardim == lam + 2 * acdim
    xi, acx = F.intervalsWorstCase([-0.2, 0.1], [0.4, 0.5], net)
    rng = np.random.default_rng(0)
    X = np.array([-0.2, 0.1])[:, None] + rng.random((2, 500)) * np.array([0.6, 0.4])[:, None]
    Y = F.evalFeedFwdNet(net, X)
    assert np.all(Y >= xi[-1][0][:, None] - 1e-12) and np.all(Y <= xi[-1][1][:, None] + 1e-12)
    with pytest.raises(ValueError):
        na.FeedFwdNet(xdims=net.xdims, Ms=net.Ms, activ="sigmoid")


# ----------------------------------------------------------------------------- options / solution plumbing
def test_double_relax_alias_and_obj_func_and_solution_check():
    assert na.DoubleRelaxDecomp.code == na.DoubleDecomp.code == 2
    vals = {"γin": np.array([0.1, 0.2]), "γac1": np.array([0.0]), "γac2": np.array([0.3]), "Z": -np.eye(3)}
    mk = lambda status, lam, v=vals: na.QuerySolution(objective_value=0.0, values=v, termination_status=status, total_time=0, setup_time=0,
                                                     solve_time=0, summary={"lambda_max": lam})
    assert V.isSolutionGood(mk("OPTIMAL", -1e-9))
    assert V.isSolutionGood(mk("ITERATION_LIMIT", 5e-5))            # acas.jl:78: eigmax(Z) <= 1e-4 is enough
    assert not V.isSolutionGood(mk("OPTIMAL", 5e-3))                # OPTIMAL residuals, but Z(gamma) is not NSD: no certificate
    assert not V.isSolutionGood(mk("NUMERICAL_ERROR", -1.0))
    bad = dict(vals)
    bad["γac1"] = np.array([-1e-3])
    assert not V.isSolutionGood(mk("OPTIMAL", -1e-9, bad))         # a negative multiplier is no certificate either
    s = mk("OPTIMAL", -1e-9, dict(vals, **{"γout": np.array([2.0])}))
    q = na.ReachQuery(ffnet=None, qc_input=None, qc_reach=None, qc_activs=[], obj_func=lambda x: 3.0 * x[0] + 1.0)
    assert na.methods._apply_obj_func(q, s).objective_value == 7.0
    with pytest.raises(ValueError):
        na.methods._apply_obj_func(na.ReachQuery(ffnet=None, qc_input=None, qc_reach=None, qc_activs=[], obj_func=lambda x: -x[0]), s)


# ----------------------------------------------------------------------------- GPU: through the C ABI
def _tanh_queries(beta=2):
    lo, hi = np.array([-0.2, 0.1]), np.array([0.4, 0.5])
    for seed in range(3, 40):
        # the reference's `@assert smin <= smax` (activ_sector.jl:13) is a lexicographic vector comparison, and its tanh
        # branch returns smin > smax on negative intervals: a network passes iff its FIRST neuron is not such a one.
        # Both the Python mirror and the C ABI reproduce that; pick a network the reference itself would accept.
        net, onet = _small_net(na.methods.TanhActiv, seed=seed)
        try:
            qa = F.makeQcActivs(net, lo, hi, beta)
            break
        except AssertionError:
            continue
    normal = np.array([0.6, -0.8])
    q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=normal), qc_activs=qa)
    qo = qc.Query(net=onet, qc_input=qc.QcInputBox(lo, hi), qc_out=qc.QcReachHplane(normal=normal),
                  qc_bounded=qc.QcActivBounded(acymin=qa[0].acymin, acymax=qa[0].acymax),
                  qc_sector=qc.QcActivSector(acxdim=net.acdim, beta=beta, smin=qa[1].smin, smax=qa[1].smax, activ="tanh"))
    return q, qo


@pytest.mark.gpu
def test_tanh_makeZ_and_adjoint_match_the_literal_assembly():
    q, qo = _tanh_queries()
    assert qo.ngamma == 2 + 1 + q.ffnet.acdim + q.qc_activs[1].vardim
    rng = np.random.default_rng(5)
    for _ in range(3):
        g = rng.random(qo.ngamma)
        Zg, Zo = na.makeZ(q, g), qc.assemble_Z_literal(qo, g)
        assert np.abs(Zg - Zo).max() <= 1e-12 * max(1.0, np.abs(Zo).max())
    X = rng.standard_normal((q.ffnet.Zdim, q.ffnet.Zdim))
    X = 0.5 * (X + X.T)
    adj = na.adjoint(q, X)
    Z0 = qc.assemble_Z_literal(qo, np.zeros(qo.ngamma))
    for i in rng.choice(qo.ngamma, 12, replace=False):
        e = np.zeros(qo.ngamma)
        e[i] = 1.0
        assert abs(adj[i] - np.sum((qc.assemble_Z_literal(qo, e) - Z0) * X)) <= 1e-11 * max(1.0, np.abs(X).max())
    with pytest.raises(ValueError):
        na.makeZ(q, np.zeros(qo.ngamma + 2 * q.ffnet.acdim))      # the ReLU-length gamma is rejected for a tanh network


@pytest.mark.gpu
def test_tanh_reach_query_tracks_the_oracle():
    q, qo = _tanh_queries()
    s = na.runQuery(q, na.AdmmSdpOptions(max_iters=4000, eps_rel=1e-7, polish=False))
    L = oop.build_operator(qo, "single", normalize=True)
    r = oadmm.admm_solve(L, oadmm.AdmmOptions(max_iters=4000, eps_rel=1e-7))
    assert s.termination_status == "OPTIMAL" and r.status == "OPTIMAL"
    assert abs(s.objective_value - r.objective) <= 1e-5 * abs(r.objective) + 1e-9
    assert len(s.values["γac2"]) == q.qc_activs[1].vardim
    # the bound is sound: every sampled output stays below the certified offset
    sp = na.runQuery(q, na.AdmmSdpOptions(max_iters=4000, eps_rel=1e-7))
    rng = np.random.default_rng(0)
    X = q.qc_input.x1min[:, None] + rng.random((2, 4000)) * (q.qc_input.x1max - q.qc_input.x1min)[:, None]
    assert np.max(q.qc_reach.normal @ F.evalFeedFwdNet(q.ffnet, X)) <= sp.objective_value + 1e-9
    assert sp.summary["lambda_max"] <= 1e-7


@pytest.mark.gpu
def test_scaled_safety_query_assembles_like_the_restatement():
    net, onet = _small_net()
    al = np.array([0.8, 1.25, 0.9, 1.2])
    Ms = [np.hstack([al[k] * Mk[:, :-1], (np.prod(al[:k + 1]) * Mk[:, -1])[:, None]]) for k, Mk in enumerate(net.Ms)]
    snet, sonet = na.FeedFwdNet(xdims=net.xdims, Ms=Ms), nnet_io.FeedFwdNet(xdims=net.xdims, Ms=Ms)
    lo, hi = np.array([0.5, 0.5]), np.array([1.5, 1.5])
    S = V.scaleS(V.hplaneS([1.0, 0.5], 4.0, snet), al, snet)
    qa = F.makeQcActivs(snet, lo, hi, 1)
    q = na.SafetyQuery(ffnet=snet, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_safety=na.QcSafety(S=S), qc_activs=qa)
    qo = qc.Query(net=sonet, qc_input=qc.QcInputBox(lo, hi), qc_out=qc.QcSafety(S=qc.scale_S(qc.hplane_S([1.0, 0.5], 4.0, sonet), al, sonet)),
                  qc_bounded=qc.QcActivBounded(acymin=qa[0].acymin, acymax=qa[0].acymax),
                  qc_sector=qc.QcActivSector(acxdim=snet.acdim, beta=1, smin=qa[1].smin, smax=qa[1].smax))
    g = np.random.default_rng(2).random(qo.ngamma)
    Zg, Zo = na.makeZ(q, g), qc.assemble_Z_literal(qo, g)
    assert np.abs(Zg - Zo).max() <= 1e-12 * max(1.0, np.abs(Zo).max())
