"""Shared test helpers: fixture loading for both sides (product package and oracle)."""
import csv
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_problem(name: str, beta: int):
    return dict(np.load(os.path.join(GOLDEN, f"problem_{name}_b{beta}.npz")))


def load_golden(name: str, beta: int):
    return dict(np.load(os.path.join(GOLDEN, f"golden_{name}_b{beta}.npz")))


def problem_Ms(d):
    K = len(d["xdims"]) - 1
    return [np.array(d[f"M{k}"], dtype=np.float64) for k in range(K)]


def product_query(d, out="ellipsoid", normal=None, S=None):
    """nnsdp_amd query from a problem fixture (inputs of the hot path only)."""
    import nnsdp_amd as na
    xdims = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xdims, Ms=problem_Ms(d))
    acdim = net.acdim
    qcs = [na.QcActivBounded(acymin=d["acymin"], acymax=d["acymax"]),
           na.QcActivSector(acxdim=acdim, beta=int(d["beta"]), smin=d["smin"], smax=d["smax"])]
    qin = na.QcInputBox(x1min=d["x1min"], x1max=d["x1max"])
    if out == "ellipsoid":
        return na.ReachQuery(ffnet=net, qc_input=qin, qc_reach=na.QcReachEllipsoid(invP=d["invP"], yc=d["yc"]), qc_activs=qcs)
    if out == "circle":
        return na.ReachQuery(ffnet=net, qc_input=qin, qc_reach=na.QcReachCircle(yc=d["yc"]), qc_activs=qcs)
    if out == "hplane":
        return na.ReachQuery(ffnet=net, qc_input=qin, qc_reach=na.QcReachHplane(normal=np.asarray(normal, dtype=float)), qc_activs=qcs)
    if out == "safety":
        return na.SafetyQuery(ffnet=net, qc_input=qin, qc_safety=na.QcSafety(S=np.asarray(S, dtype=float)), qc_activs=qcs)
    raise ValueError(out)


def oracle_query(d, out="ellipsoid", normal=None, S=None):
    from oracle import nnet_io, qc
    xdims = [int(v) for v in d["xdims"]]
    net = nnet_io.FeedFwdNet(xdims=xdims, Ms=problem_Ms(d))
    qb = qc.QcActivBounded(acymin=d["acymin"], acymax=d["acymax"])
    qs = qc.QcActivSector(acxdim=len(d["smin"]), beta=int(d["beta"]), smin=d["smin"], smax=d["smax"])
    qin = qc.QcInputBox(d["x1min"], d["x1max"])
    if out == "ellipsoid":
        qo = qc.QcReachEllipsoid(invP=d["invP"], yc=d["yc"])
    elif out == "circle":
        qo = qc.QcReachCircle(yc=d["yc"])
    elif out == "hplane":
        qo = qc.QcReachHplane(normal=np.asarray(normal, dtype=float))
    elif out == "safety":
        qo = qc.QcSafety(S=np.asarray(S, dtype=float))
    else:
        raise ValueError(out)
    return qc.Query(net=net, qc_input=qin, qc_out=qo, qc_bounded=qb, qc_sector=qs)


def dump_rows():
    with open(os.path.join(GOLDEN, "dump_scale.csv")) as fh:
        return list(csv.DictReader(fh))


def published_rho(net: str, beta: int):
    """objective values of the OPTIMAL rows the reference published for (net, beta)."""
    return [float(r["obj_val"]) for r in dump_rows()
            if r["net"] == net and int(r["beta"]) == beta and r["term_status"] == "OPTIMAL"]


def acas_shaped_query():
    """BASELINE config 5's shape without its data (no ACAS file is in the reference checkout): a 5-50x6-5 ReLU network with random
    weights, reach-hyperplane query on a box of half-width 0.05, plain interval arithmetic (no neuron stable, so the reference's Single
    cliques keep their full 106 + 4 x 151)."""
    import nnsdp_amd as na
    from nnsdp_amd import frontend as F
    net = na.randomNetwork([5] + [50] * 6 + [5], seed=1234)
    lo, hi = np.full(5, 0.25), np.full(5, 0.35)
    xi, acx = F.intervalsWorstCase(lo, hi, net)
    nrm = np.zeros(5)
    nrm[0] = 1.0
    return na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm),
                         qc_activs=F.makeQcActivsIntvs(net, xi, acx, 0))
