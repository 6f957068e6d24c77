"""Randomised parity (seeded): random ReLU / tanh networks drawn like the reference's bench/rand nets (scripts/make_networks.jl:22-28),
random boxes, every output QC kind, beta in 0..4 - GPU assembly vs the literal restatement (1e-12), adjoint, and the ADMM iterate
for iterate against the oracle (fixed penalty, exact projections) in the Single, Path and Dense block structures."""
import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F
from oracle import admm as oadmm, nnet_io, operator as oop, qc

pytestmark = pytest.mark.gpu

CASES = [  # (seed, xdims, beta, out kind, activ)
    (1, [2, 7, 9, 6, 2], 0, "hplane", "relu"), (2, [3, 8, 8, 8, 8, 2], 2, "ellipsoid", "relu"), (3, [2, 12, 5, 12, 3], 1, "circle", "relu"),
    (4, [4, 10, 10, 10, 2], 3, "safety", "relu"), (5, [2, 6, 6, 6, 6, 6, 6, 2], 4, "hplane", "relu"), (6, [2, 15, 15, 2], 0, "ellipsoid", "relu"),
    (7, [2, 9, 7, 8, 2], 2, "hplane", "tanh"), (8, [3, 6, 6, 6, 2], 0, "circle", "tanh"), (9, [5, 20, 20, 20, 5], 1, "safety", "relu"),
    (10, [2, 5, 30, 5, 2], 2, "ellipsoid", "relu"),
]


def _case(seed, xdims, beta, kind, activ):
    rng = np.random.default_rng(seed)
    W = max(xdims[1:-1])
    sig = 2.0 / np.sqrt(W * np.log(W))
    Ms = [rng.normal(0.0, sig * 1.5, size=(xdims[k + 1], xdims[k] + 1)) for k in range(len(xdims) - 1)]
    act = na.methods.TanhActiv if activ == "tanh" else na.methods.ReluActiv
    net = na.FeedFwdNet(xdims=list(xdims), Ms=Ms, activ=act)
    onet = nnet_io.FeedFwdNet(xdims=list(xdims), Ms=Ms)
    c = rng.uniform(-0.5, 1.0, xdims[0])
    lo, hi = c - rng.uniform(0.05, 0.4, xdims[0]), c + rng.uniform(0.05, 0.4, xdims[0])
    if activ == "tanh":
        qa = None
        for _ in range(20):     # the reference's lexicographic smin <= smax assertion rejects some tanh nets (activ_sector.jl:13,76-77)
            try:
                qa = F.makeQcActivs(net, lo, hi, beta)
                break
            except AssertionError:
                Ms[0][0, :] = np.abs(Ms[0][0, :])          # make the first neuron's pre-activation positive on a positive box
                lo, hi = np.abs(lo) + 0.05, np.abs(lo) + 0.05 + (hi - lo)
                net = na.FeedFwdNet(xdims=list(xdims), Ms=Ms, activ=act)
                onet = nnet_io.FeedFwdNet(xdims=list(xdims), Ms=Ms)
        assert qa is not None
    else:
        qa = F.makeQcActivs(net, lo, hi, beta)
    m = xdims[-1]
    yc = F.evalFeedFwdNet(net, 0.5 * (lo + hi))
    if kind == "hplane":
        nrm = rng.standard_normal(m)
        out, oout = na.QcReachHplane(normal=nrm), qc.QcReachHplane(normal=nrm)
    elif kind == "circle":
        out, oout = na.QcReachCircle(yc=yc), qc.QcReachCircle(yc=yc)
    elif kind == "ellipsoid":
        B = rng.standard_normal((m, m))
        invP = np.linalg.inv(B @ B.T + np.eye(m))
        invP = 0.5 * (invP + invP.T)
        out, oout = na.QcReachEllipsoid(invP=invP, yc=yc), qc.QcReachEllipsoid(invP=invP, yc=yc)
    else:
        nrm = rng.standard_normal(m)
        S = qc.hplane_S(nrm, float(nrm @ yc) + 3.0, onet)
        out, oout = na.QcSafety(S=S), qc.QcSafety(S=S)
    qin = na.QcInputBox(x1min=lo, x1max=hi)
    if kind == "safety":
        q = na.SafetyQuery(ffnet=net, qc_input=qin, qc_safety=out, qc_activs=qa)
    else:
        q = na.ReachQuery(ffnet=net, qc_input=qin, qc_reach=out, qc_activs=qa)
    qo = qc.Query(net=onet, qc_input=qc.QcInputBox(lo, hi), qc_out=oout,
                  qc_bounded=qc.QcActivBounded(acymin=qa[0].acymin, acymax=qa[0].acymax),
                  qc_sector=qc.QcActivSector(acxdim=net.acdim, beta=beta, smin=qa[1].smin, smax=qa[1].smax, activ=activ))
    return q, qo


@pytest.mark.parametrize("seed,xdims,beta,kind,activ", CASES)
def test_random_problem_assembly_and_iterates(seed, xdims, beta, kind, activ):
    q, qo = _case(seed, xdims, beta, kind, activ)
    rng = np.random.default_rng(100 + seed)
    g = rng.random(qo.ngamma)
    Zg, Zo = na.makeZ(q, g), qc.assemble_Z_literal(qo, g)
    assert np.abs(Zg - Zo).max() <= 1e-12 * max(1.0, np.abs(Zo).max())
    X = rng.standard_normal(Zo.shape)
    X = 0.5 * (X + X.T)
    adj = na.adjoint(q, X)
    Z0 = qc.assemble_Z_literal(qo, np.zeros(qo.ngamma))
    for i in rng.choice(qo.ngamma, 8, replace=False):
        e = np.zeros(qo.ngamma)
        e[i] = 1.0
        assert abs(adj[i] - np.sum((qc.assemble_Z_literal(qo, e) - Z0) * X)) <= 1e-10 * max(1.0, np.abs(X).max() * np.abs(Zo).max())
    modes = [(na.SingleDecomp(), "single"), (na.DenseCone(), "dense")]
    if kind != "safety" or True:
        modes.append((na.PathDecomp(), "path"))          # hyperplane safety sets have S12 = 0 as well
    iters = 200
    for mode, oname in modes:
        s = na.runQuery(q, na.AdmmSdpOptions(max_iters=iters, decomp_mode=mode, proj_tol=1e-12, adapt_every=0, polish=False))
        r = oadmm.admm_solve(oop.build_operator(qo, oname, normalize=True), oadmm.AdmmOptions(max_iters=iters, adapt_sigma=False))
        assert s.summary["iters"] == r.iters
        assert abs(s.objective_value - r.objective) <= 1e-7 * abs(r.objective) + 1e-10, (oname, s.objective_value, r.objective)
