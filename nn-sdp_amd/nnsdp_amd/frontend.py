"""Callers and data formats either side of the hot path, host side:

  read_nnet                 .nnet text reader            exts/nnet_parser.jl:23-131, loadFromNnet network_files.jl:17-22
  makeIntervalsInfo         CROWN-sliced interval bounds  src/Intervals/intervals_auto_lirpa.jl:12-64 (+ auto_LiRPA 0.2 CROWN rules)
  makeQcActivs              QcActivBounded + QcActivSector src/Qc/activ.jl:45-72, makeSectorMinMax activ_sector.jl:63-90
  approxEllipsoid           sampled output ellipsoid      src/Utils/qc.jl:40-67
  findEllipsoid / findCircle / findReach2Dpoly            src/NnSdp.jl:35-95
  write_scale_csv           dump/scale column layout      experiments/scale.jl:60-82

These run once per query on the host (numpy); the SDP itself goes through runQuery -> libnnsdp_hip.so.
"""
from __future__ import annotations

import csv
from typing import List, Sequence, Tuple

import numpy as np

from . import methods as M


# ----------------------------------------------------------------------------- f3: .nnet reader
def read_nnet(path: str) -> M.FeedFwdNet:
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f if not ln.lstrip().startswith("//")]
    head = [v for v in lines[0].split(",") if v.strip() != ""]
    nlayers = int(head[0])
    sizes = [int(v) for v in lines[1].split(",") if v.strip() != ""][: nlayers + 1]
    pos = 7                     # flag line + mins, maxes, means, ranges are not used by the path
    Ms = []
    for k in range(nlayers):
        nin, nout = sizes[k], sizes[k + 1]
        W = np.array([[float(v) for v in lines[pos + i].split(",")[:nin]] for i in range(nout)], dtype=np.float64)
        pos += nout
        b = np.array([float(lines[pos + i].split(",")[0]) for i in range(nout)], dtype=np.float64)
        pos += nout
        Ms.append(np.hstack([W, b[:, None]]))
    return M.FeedFwdNet(xdims=sizes, Ms=Ms)


def evalFeedFwdNet(net: M.FeedFwdNet, x) -> np.ndarray:
    xk = np.asarray(x, dtype=np.float64)
    vec = xk.ndim == 1
    xk = xk[:, None] if vec else xk
    for Mk in net.Ms[:-1]:
        xk = np.maximum(Mk[:, :-1] @ xk + Mk[:, -1:], 0.0)
    xk = net.Ms[-1][:, :-1] @ xk + net.Ms[-1][:, -1:]
    return xk[:, 0] if vec else xk


# ----------------------------------------------------------------------------- f1: CROWN-sliced intervals
def _backward(Ws, bs, pre, lo, hi):
    """backward LiRPA (CROWN) bounds of the last linear layer's output, float32 like the reference's
    torch path; pre = pre-activation bounds of the earlier layers."""
    f32 = np.float32
    lA = uA = Ws[-1]
    lb = ub = bs[-1]
    for j in range(len(Ws) - 2, -1, -1):
        l, u = pre[j]
        lr = np.minimum(l, f32(0))
        ur = np.maximum(np.maximum(u, f32(0)), lr + f32(1e-8))
        du = ur / (ur - lr)                       # upper slope, intercept -lr*du
        dl = (du > f32(0.5)).astype(f32)          # 'adaptive' lower slope
        bu = -lr * du
        lb = lb + np.minimum(lA, 0) @ bu
        ub = ub + np.maximum(uA, 0) @ bu
        lA = np.maximum(lA, 0) * dl + np.minimum(lA, 0) * du
        uA = np.maximum(uA, 0) * du + np.minimum(uA, 0) * dl
        lb = lb + lA @ bs[j]
        ub = ub + uA @ bs[j]
        lA = lA @ Ws[j]
        uA = uA @ Ws[j]
    c, r = (hi + lo) / f32(2), (hi - lo) / f32(2)
    return (lA @ c - np.abs(lA) @ r + lb).astype(f32), (uA @ c + np.abs(uA) @ r + ub).astype(f32)


def makeIntervalsInfo(x1min, x1max, net: M.FeedFwdNet):
    """returns (x_intvs, acx_intvs): K+1 post-activation intervals and K-1 pre-activation intervals."""
    f32 = np.float32
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    W = [Mk[:, :-1].astype(f32) for Mk in net.Ms]
    b = [Mk[:, -1].astype(f32) for Mk in net.Ms]
    lo, hi = x1min.astype(f32), x1max.astype(f32)
    x_intvs = [(x1min, x1max)]
    pre = []
    K = net.K

    def fix(l, u):
        l, u = l.astype(np.float64), u.astype(np.float64)
        l = np.minimum(l, u)
        return l, np.maximum(l, u)
    for k in range(1, K):                 # slice k: layers 1..k followed by [I 0] (intervals_auto_lirpa.jl:12-28)
        pre.append(_backward(W[:k], b[:k], pre, lo, hi))
        n = net.xdims[k]
        x_intvs.append(fix(*_backward(W[:k] + [np.eye(n, dtype=f32)], b[:k] + [np.zeros(n, dtype=f32)], pre, lo, hi)))
    x_intvs.append(fix(*_backward(W, b, pre, lo, hi)))
    acx = []
    for k in range(K - 1):                # one float64 IBP step per layer (intervals_auto_lirpa.jl:55-62)
        Wk, bk = net.Ms[k][:, :-1], net.Ms[k][:, -1]
        l, u = x_intvs[k]
        acx.append((np.maximum(Wk, 0) @ l + np.minimum(Wk, 0) @ u + bk, np.maximum(Wk, 0) @ u + np.minimum(Wk, 0) @ l + bk))
    return x_intvs, acx


def makeQcActivs(net: M.FeedFwdNet, x1min, x1max, beta: int):
    x_intvs, acx = makeIntervalsInfo(x1min, x1max, net)
    acymin = np.concatenate([v[0] for v in x_intvs[1:-1]])
    acymax = np.concatenate([v[1] for v in x_intvs[1:-1]])
    amin = np.concatenate([v[0] for v in acx])
    amax = np.concatenate([v[1] for v in acx])
    eps = 1e-4                            # activ_sector.jl:65
    smin = (amin > eps).astype(np.float64)
    smax = 1.0 - (amax < -eps).astype(np.float64)
    return [M.QcActivBounded(acymin=acymin, acymax=acymax),
            M.QcActivSector(acxdim=len(acymin), beta=int(beta), smin=smin, smax=smax)]


# ----------------------------------------------------------------------------- f2: callers of the path
def approxEllipsoid(net: M.FeedFwdNet, x1min, x1max, N: int = 100000, seed: int = 1234):
    rng = np.random.default_rng(seed)
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    Y = evalFeedFwdNet(net, x1min[:, None] + rng.random((net.xdims[0], N)) * (x1max - x1min)[:, None])
    yc = Y.sum(axis=1) / N
    Yd = Y - yc[:, None]
    P = Yd @ Yd.T
    w, V = np.linalg.eigh(P)
    a, b = 1.0, 4.0
    if w.max() * a >= w.min() * b:
        P = V @ np.diag((w - w.min()) * ((b - a) / (w.max() - w.min())) + a) @ V.T
        P = 0.5 * (P + P.T)
    return P, yc


def findEllipsoid(net, x1min, x1max, beta: int, opts: M.AdmmSdpOptions, seed: int = 1234):
    qc_input = M.QcInputBox(x1min=x1min, x1max=x1max)
    qc_activs = makeQcActivs(net, x1min, x1max, beta)
    P, yc = approxEllipsoid(net, x1min, x1max, seed=seed)
    invP = np.linalg.inv(P)
    q = M.ReachQuery(ffnet=net, qc_input=qc_input, qc_reach=M.QcReachEllipsoid(invP=0.5 * (invP + invP.T), yc=yc), qc_activs=qc_activs)
    soln = M.runQuery(q, opts)
    rho = max(float(soln.values["γout"][0]), 0.0)
    return np.sqrt(rho) * P, yc, soln


def findCircle(net, x1min, x1max, beta: int, opts: M.AdmmSdpOptions):
    yc = evalFeedFwdNet(net, (np.asarray(x1max, float) + np.asarray(x1min, float)) / 2)
    q = M.ReachQuery(ffnet=net, qc_input=M.QcInputBox(x1min=x1min, x1max=x1max), qc_reach=M.QcReachCircle(yc=yc),
                     qc_activs=makeQcActivs(net, x1min, x1max, beta))
    return M.runQuery(q, opts)


def findReach2Dpoly(net, x1min, x1max, beta: int, opts: M.AdmmSdpOptions, num_hplanes: int = 6):
    qc_input = M.QcInputBox(x1min=x1min, x1max=x1max)
    qc_activs = makeQcActivs(net, x1min, x1max, beta)
    hplanes, solns = [], []
    for i in range(num_hplanes):
        th = (i / num_hplanes) * 2 * np.pi
        normal = np.array([np.cos(th), np.sin(th)])
        q = M.ReachQuery(ffnet=net, qc_input=qc_input, qc_reach=M.QcReachHplane(normal=normal), qc_activs=qc_activs)
        s = M.runQuery(q, opts)
        hplanes.append((normal, s.objective_value))
        solns.append(s)
    return hplanes, solns


def write_scale_csv(path: str, rows: Sequence[Tuple[int, M.QuerySolution]]):
    """beta,setup_secs,solve_secs,total_secs,obj_val,term_status,eigmax (experiments/scale.jl:60-82)."""
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["beta", "setup_secs", "solve_secs", "total_secs", "obj_val", "term_status", "eigmax"])
        for beta, s in rows:
            w.writerow([beta, s.setup_time, s.solve_time, s.total_time, s.objective_value, s.termination_status, s.summary["lambda_max"]])
