"""Callers and data formats either side of the hot path, host side:

  read_nnet                 .nnet text reader            exts/nnet_parser.jl:23-131, loadFromNnet network_files.jl:17-22
  makeIntervalsInfo         CROWN-sliced interval bounds  src/Intervals/intervals_auto_lirpa.jl:12-64 (+ auto_LiRPA 0.2 CROWN rules)
  makeQcActivs              QcActivBounded + QcActivSector src/Qc/activ.jl:45-72, makeSectorMinMax activ_sector.jl:63-90
  approxEllipsoid           sampled output ellipsoid      src/Utils/qc.jl:40-67
  findEllipsoid / findCircle / findReach2Dpoly            src/NnSdp.jl:35-95
  runScale                  beta sweep of one network      experiments/scale.jl:52-82 (batch handle)
  write_scale_csv           dump/scale column layout      experiments/scale.jl:60-82

These run once per query on the host; the interval pre-processing is native C++ inside libnnsdp_hip.so
(nnsdp_make_intervals, csrc/intervals.hpp), the SDP itself goes through runQuery -> libnnsdp_hip.so.
"""
from __future__ import annotations

import csv
from typing import List, Sequence, Tuple

import ctypes as C

import numpy as np

from . import methods as M
from . import _lib


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


def _activ_fn(net: M.FeedFwdNet):
    """makeActiv (src/MyNeuralNetwork/MyNeuralNetwork.jl:29-37)"""
    return np.tanh if M._activ_code(net.activ) == M.ACTIV_TANH else (lambda v: np.maximum(v, 0.0))


def evalFeedFwdNet(net: M.FeedFwdNet, x) -> np.ndarray:
    xk = np.asarray(x, dtype=np.float64)
    vec = xk.ndim == 1
    xk = xk[:, None] if vec else xk
    ac = _activ_fn(net)
    for Mk in net.Ms[:-1]:
        xk = ac(Mk[:, :-1] @ xk + Mk[:, -1:])
    xk = net.Ms[-1][:, :-1] @ xk + net.Ms[-1][:, -1:]
    return xk[:, 0] if vec else xk


def randomNetwork(xdims: Sequence[int], sigma: float = None, seed: int = 1234) -> M.FeedFwdNet:
    """Utils.randomNetwork (src/Utils/Utils.jl:22-26): every [W_k b_k] entry i.i.d. N(0, sigma^2).  Default sigma is the
    scaling experiments' 2 / sqrt(W ln W) with W the hidden width (scripts/make_networks.jl:43-46); numpy's generator, the
    Julia stream of the reference cannot be reproduced."""
    xdims = [int(v) for v in xdims]
    if len(xdims) < 2 or min(xdims) <= 0:
        raise ValueError("xdims needs at least two positive entries")
    if sigma is None:
        width = max(xdims[1:-1]) if len(xdims) > 2 else max(xdims)
        sigma = 2.0 / np.sqrt(width * np.log(width)) if width > 1 else 1.0
    rng = np.random.default_rng(seed)
    return M.FeedFwdNet(xdims=xdims, Ms=[rng.normal(0.0, sigma, size=(xdims[k + 1], xdims[k] + 1)) for k in range(len(xdims) - 1)])


def loadFromFileScaled(path: str, scaling=None):
    """loadFromFileScaled (src/MyNeuralNetwork/network_files.jl:84-114): W_k -> alpha_k W_k, b_k -> prod(alpha[1..k]) b_k, so
    that f'(x) = prod(alpha) f(x) for a ReLU network.  `scaling`: None / "none" (NoScaling), "sqrtlog" (SqrtLogScaling),
    ("norm", v) (FixedNormScaling(Wk_opnorm = v)), ("const", a) (FixedConstScaling(α = a)).  -> (scaled net, alphas)."""
    net = read_nnet(path)
    K = net.K
    Ws, bs = [Mk[:, :-1] for Mk in net.Ms], [Mk[:, -1] for Mk in net.Ms]
    opn = lambda W: float(np.linalg.norm(W, 2))
    if scaling is None or scaling == "none":
        al = np.ones(K)
    elif scaling == "sqrtlog":
        tgt = [np.sqrt(c * np.log(c) / K) for c in (net.xdims[k] + net.xdims[k + 1] for k in range(K))]
        al = np.array([tgt[k] / opn(Ws[k]) for k in range(K)])
    elif isinstance(scaling, tuple) and scaling[0] == "norm":
        al = np.array([float(scaling[1]) / opn(W) for W in Ws])
    elif isinstance(scaling, tuple) and scaling[0] == "const":
        al = float(scaling[1]) * np.ones(K)
    else:
        raise ValueError(f"unrecognized scaling method: {scaling}")
    Ms = [np.hstack([al[k] * Ws[k], (np.prod(al[:k + 1]) * bs[k])[:, None]]) for k in range(K)]
    return M.FeedFwdNet(xdims=list(net.xdims), Ms=Ms), al


def intervalsWorstCase(x1min, x1max, net: M.FeedFwdNet):
    """Intervals.intervalsWorstCase (src/Intervals/intervals_easy.jl:2-37): plain interval arithmetic, ReLU or tanh (both
    monotone).  -> (x_intvs, acx_intvs) like makeIntervalsInfo."""
    lo, hi = np.asarray(x1min, dtype=np.float64), np.asarray(x1max, dtype=np.float64)
    ac = _activ_fn(net)
    x_intvs, acx = [(lo, hi)], []
    for k, Mk in enumerate(net.Ms):
        W, b = Mk[:, :-1], Mk[:, -1]
        pl = np.maximum(W, 0) @ lo + np.minimum(W, 0) @ hi + b
        pu = np.maximum(W, 0) @ hi + np.minimum(W, 0) @ lo + b
        if k < net.K - 1:
            acx.append((pl, pu))
            lo, hi = ac(pl), ac(pu)
        else:
            lo, hi = pl, pu
        x_intvs.append((lo, hi))
    return x_intvs, acx


def makeSectorMinMax(acxmin, acxmax, activ=M.ReluActiv):
    """Qc.makeSectorMinMax (src/Qc/activ_sector.jl:63-90), both activations, same arithmetic (including 0/0 = NaN for a tanh
    pre-activation bound that is exactly 0, as in the reference)."""
    acxmin, acxmax = np.asarray(acxmin, dtype=np.float64), np.asarray(acxmax, dtype=np.float64)
    if len(acxmin) != len(acxmax):
        raise ValueError("acxmin / acxmax length mismatch")
    eps = 1e-4
    code = M._activ_code(activ)
    if code == M.ACTIV_RELU:
        return (acxmin > eps).astype(np.float64), 1.0 - (acxmax < -eps).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tlo, thi = np.tanh(acxmin) / acxmin, np.tanh(acxmax) / acxmax
    same = acxmin * acxmax >= 0
    return np.where(same, thi, np.minimum(tlo, thi)), np.where(same, tlo, 1.0)


def makeQcActivsIntvs(net: M.FeedFwdNet, x_intvs, acx_intvs, beta: int):
    """Qc.makeQcActivsIntvs (src/Qc/activ.jl:45-66) from given interval information."""
    acymin = np.concatenate([iv[0] for iv in x_intvs[1:-1]])
    acymax = np.concatenate([iv[1] for iv in x_intvs[1:-1]])
    smin, smax = makeSectorMinMax(np.concatenate([iv[0] for iv in acx_intvs]), np.concatenate([iv[1] for iv in acx_intvs]), net.activ)
    return [M.QcActivBounded(acymin=acymin, acymax=acymax),
            M.QcActivSector(acxdim=len(acymin), beta=int(beta), smin=smin, smax=smax, activ=net.activ)]


# ----------------------------------------------------------------------------- f1: CROWN-sliced intervals
def _intervals_native(x1min, x1max, net: M.FeedFwdNet):
    """nnsdp_make_intervals_activ (host C++ in the library, csrc/intervals.hpp): replaces the reference's per-layer
    PyCall + ONNX + auto_LiRPA round trips (src/Intervals/intervals_auto_lirpa.jl:12-64), ReLU and Tanh networks
    (BoundRelu / BoundTanh relaxations, exts/auto_lirpa_bridge.py:31-37)."""
    lib = _lib.load()
    xd = np.asarray(net.xdims, dtype=np.int32)
    K = net.K
    Mp = np.concatenate([np.asfortranarray(Mk, dtype=np.float64).ravel(order="F") for Mk in net.Ms])
    lo = np.ascontiguousarray(x1min, dtype=np.float64)
    hi = np.ascontiguousarray(x1max, dtype=np.float64)
    if lo.shape != (xd[0],) or hi.shape != (xd[0],):
        raise ValueError("x1min / x1max must have xdims[0] entries")
    acdim = int(xd[1:-1].sum())
    outs = [np.zeros(acdim) for _ in range(6)] + [np.zeros(int(xd[-1])) for _ in range(2)]
    dp = _lib.c_double_p
    _lib.check(lib.nnsdp_make_intervals_activ(K, xd.ctypes.data_as(_lib.c_int32_p), Mp.ctypes.data_as(dp), M._activ_code(net.activ),
                                              lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), *[o.ctypes.data_as(dp) for o in outs]))
    return outs


def makeIntervalsInfo(x1min, x1max, net: M.FeedFwdNet):
    """returns (x_intvs, acx_intvs): K+1 post-activation intervals and K-1 pre-activation intervals
    (Intervals.makeIntervalsInfo with the sliced auto_LiRPA method, intervals_auto_lirpa.jl:31-64)."""
    acymin, acymax, acxmin, acxmax, _, _, ymin, ymax = _intervals_native(x1min, x1max, net)
    x_intvs = [(np.asarray(x1min, dtype=np.float64), np.asarray(x1max, dtype=np.float64))]
    acx, o = [], 0
    for k in range(1, net.K):
        n = net.xdims[k]
        x_intvs.append((acymin[o:o + n], acymax[o:o + n]))
        acx.append((acxmin[o:o + n], acxmax[o:o + n]))
        o += n
    x_intvs.append((ymin, ymax))
    return x_intvs, acx


def makeQcActivs(net: M.FeedFwdNet, x1min, x1max, beta: int):
    """Qc.makeQcActivs (src/Qc/activ.jl:45-72): bounded + sector QCs from the interval pre-processing."""
    acymin, acymax, _, _, smin, smax, _, _ = _intervals_native(x1min, x1max, net)
    return [M.QcActivBounded(acymin=acymin, acymax=acymax),
            M.QcActivSector(acxdim=len(acymin), beta=int(beta), smin=smin, smax=smax, activ=net.activ)]


# ----------------------------------------------------------------------------- f2: callers of the path
def evalFeedFwdNetBatch(net: M.FeedFwdNet, X, return_ms: bool = False):
    """the network at the columns of X (xdims[0] x N) on the GPU: nnsdp_eval_network (csrc/forward.hpp, fp64 MFMA, one wave per
    16 samples).  No host fallback: raises without a GPU.  evalFeedFwdNet above is the reference's pointwise function."""
    lib = _lib.load()
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[0] != net.xdims[0]:
        raise ValueError("X must be xdims[0] x N")
    N = X.shape[1]
    xd = np.asarray(net.xdims, dtype=np.int32)
    Mp = np.concatenate([np.asfortranarray(Mk, dtype=np.float64).ravel(order="F") for Mk in net.Ms])
    Xc = np.ascontiguousarray(X.T)                      # column-major xdims[0] x N
    Y = np.empty((N, int(xd[-1])), dtype=np.float64)
    ms = C.c_double(0.0)
    dp = _lib.c_double_p
    _lib.check(lib.nnsdp_eval_network(net.K, xd.ctypes.data_as(_lib.c_int32_p), Mp.ctypes.data_as(dp), M._activ_code(net.activ), N,
                                      Xc.ctypes.data_as(dp), Y.ctypes.data_as(dp), C.byref(ms)))
    return (Y.T, ms.value) if return_ms else Y.T


def sampleTrajs(net: M.FeedFwdNet, x1min, x1max, N: int = 100000, seed: int = 1234):
    """Utils.sampleTrajs (src/Utils/qc.jl:40-47): outputs of N inputs drawn uniformly from the box, xdims[K] x N.  The draw is
    numpy's default_rng(seed) on the host (the Julia stream of the reference cannot be regenerated); the N forward passes run
    on the GPU."""
    rng = np.random.default_rng(seed)
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    return evalFeedFwdNetBatch(net, x1min[:, None] + rng.random((net.xdims[0], N)) * (x1max - x1min)[:, None])


def approxEllipsoid(net: M.FeedFwdNet, x1min, x1max, N: int = 100000, seed: int = 1234):
    """Utils.approxEllipsoid (src/Utils/qc.jl:50-67)"""
    Y = sampleTrajs(net, x1min, x1max, N, seed)
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
    q, P, yc = ellipsoidQuery(net, x1min, x1max, beta, seed=seed)
    soln = M.runQuery(q, opts)
    rho = max(float(soln.values["γout"][0]), 0.0)
    return np.sqrt(rho) * P, yc, soln


def findCircle(net, x1min, x1max, beta: int, opts: M.AdmmSdpOptions):
    yc = evalFeedFwdNet(net, (np.asarray(x1max, float) + np.asarray(x1min, float)) / 2)
    q = M.ReachQuery(ffnet=net, qc_input=M.QcInputBox(x1min=x1min, x1max=x1max), qc_reach=M.QcReachCircle(yc=yc),
                     qc_activs=makeQcActivs(net, x1min, x1max, beta))
    return M.runQuery(q, opts)


def findReach2Dpoly(net, x1min, x1max, beta: int, opts: M.AdmmSdpOptions, num_hplanes: int = 6, batched: bool = True):
    """NnSdp.findReach2Dpoly (src/NnSdp.jl:73-95): one reach-hyperplane SDP per direction.  The directions share the
    network and the interval pre-processing and are independent SDPs of identical shape: solved in lockstep through
    the batch handle (`batched=False`: one after the other, as the reference does)."""
    qc_input = M.QcInputBox(x1min=x1min, x1max=x1max)
    qc_activs = makeQcActivs(net, x1min, x1max, beta)
    normals = [np.array([np.cos(2 * np.pi * i / num_hplanes), np.sin(2 * np.pi * i / num_hplanes)]) for i in range(num_hplanes)]
    queries = [M.ReachQuery(ffnet=net, qc_input=qc_input, qc_reach=M.QcReachHplane(normal=nrm), qc_activs=qc_activs) for nrm in normals]
    solns = M.runQueries(queries, opts) if batched and len(queries) > 1 else [M.runQuery(q, opts) for q in queries]
    return [(nrm, s.objective_value) for nrm, s in zip(normals, solns)], solns


def ellipsoidQuery(net, x1min, x1max, beta: int, seed: int = 1234):
    """the ReachQuery NnSdp.findEllipsoid solves (src/NnSdp.jl:35-50) and the sampled shape matrix P."""
    qc_input = M.QcInputBox(x1min=x1min, x1max=x1max)
    qc_activs = makeQcActivs(net, x1min, x1max, beta)
    P, yc = approxEllipsoid(net, x1min, x1max, seed=seed)
    invP = np.linalg.inv(P)
    q = M.ReachQuery(ffnet=net, qc_input=qc_input, qc_reach=M.QcReachEllipsoid(invP=0.5 * (invP + invP.T), yc=yc), qc_activs=qc_activs)
    return q, P, yc


def runScale(net, x1min, x1max, betas: Sequence[int], opts: M.AdmmSdpOptions, saveto: str = None, batched: bool = True, seed: int = 1234):
    """The reference's headline experiment (experiments/scale.jl:52-82): findEllipsoid for every beta of a sweep on one
    network, one CSV row per beta.  The SDPs of a sweep are independent and of different sizes (the sector QC grows with
    beta): `batched` advances them in lockstep through the batch handle, each stopping on its own rule.
    -> list of (beta, QuerySolution)."""
    queries = [ellipsoidQuery(net, x1min, x1max, int(b), seed=seed)[0] for b in betas]
    solns = M.runQueries(queries, opts) if batched and len(queries) > 1 else [M.runQuery(q, opts) for q in queries]
    rows = list(zip([int(b) for b in betas], solns))
    if saveto:
        write_scale_csv(saveto, rows)
    return rows


def write_scale_csv(path: str, rows: Sequence[Tuple[int, M.QuerySolution]]):
    """beta,setup_secs,solve_secs,total_secs,obj_val,term_status,eigmax (experiments/scale.jl:60-82)."""
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["beta", "setup_secs", "solve_secs", "total_secs", "obj_val", "term_status", "eigmax"])
        for beta, s in rows:
            w.writerow([beta, s.setup_time, s.solve_time, s.total_time, s.objective_value, s.termination_status, s.summary["lambda_max"]])
