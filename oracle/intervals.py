"""Interval pre-processing: CROWN-sliced bounds (oracle restatement).

Follows src/Intervals/intervals_auto_lirpa.jl:12-64 (sliceFeedFwdNet,
autoLirpaBoundsOutput, intervalsAutoLirpaSliced) and the plain-CROWN rules of
the vendored auto_LiRPA v0.2 that the reference reaches through
exts/auto_lirpa_bridge.py:97-112:
  * method "CROWN" = backward LiRPA for the final node AND for every
    intermediate pre-activation (exts/auto_LiRPA/bound_general.py:1078-1079);
  * ReLU relaxation (exts/auto_LiRPA/operators/activation.py:306-323,386-388):
    lb_r=min(l,0), ub_r=max(max(u,0), lb_r+1e-8), upper slope d=ub_r/(ub_r-lb_r),
    upper intercept -lb_r*d, lower slope 1 if d>0.5 else 0, lower intercept 0;
  * weights and inputs are float32 (exts/NNet/converters/nnet2onnx.py:47,51;
    auto_lirpa_bridge.py:100), the Julia post-fix lb=min(lb,ub), ub=max(lb,ub)
    (intervals_auto_lirpa.jl:38-39), then ONE float64 IBP step per layer gives
    the pre-activation intervals (intervals_auto_lirpa.jl:55-62).

The third-party library (auto_LiRPA 0.2, torch) cannot be imported under this
image, so float32 summation order is numpy's, not
torch's; differences are at the 1e-7 relative level.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .nnet_io import FeedFwdNet


@dataclass
class IntervalsInfo:
    x_intvs: List[Tuple[np.ndarray, np.ndarray]]     # K+1 pairs (post-activation x_k, and output)
    acx_intvs: List[Tuple[np.ndarray, np.ndarray]]   # K-1 pairs (pre-activation of layer k)


def _concretize(A, bias, xl, xu, sign):
    """Linf-box concretisation: A*center -/+ |A|*radius + bias (float32)."""
    center = (xu + xl) / np.float32(2)
    diff = (xu - xl) / np.float32(2)
    base = A @ center + bias
    dev = np.abs(A) @ diff
    return base - dev if sign < 0 else base + dev


def _dtanh(x):
    """BoundTanh.dtanh (exts/auto_LiRPA/operators/activation.py:863-868), float32"""
    x = np.asarray(x, dtype=np.float32)
    mask = (np.abs(x) < np.float32(25.0)).astype(np.float32)
    cosh = np.cosh(mask * x + np.float32(1) - mask)
    return (mask * (np.float32(1) / (cosh * cosh))).astype(np.float32)


def _tanh_tangent_points(lower, upper):
    """d_lower / d_upper of BoundTanh.precompute_relaxation (activation.py:872-917), evaluated for the table entries the
    relaxation looks up (:943-953): index = max(0, int(u / 0.01)) + 1 -> the tangent point d <= 0 whose tangent passes
    below tanh at U = 0.01 index (bisection, 100 halvings, float32), and the mirror image for the lower end."""
    f32 = np.float32
    step = f32(0.01)
    iu = np.maximum(0, (upper / step).astype(np.int64)) + 1
    il = np.maximum(0, (lower / -step).astype(np.int64)) + 1
    U = (step * iu.astype(np.float32)).astype(np.float32)
    Lw = (-step * il.astype(np.float32)).astype(np.float32)

    def check_lower(up, d):
        return _dtanh(d) * (up - d) + np.tanh(d).astype(np.float32) <= np.tanh(up).astype(np.float32)

    def check_upper(lo, d):
        return _dtanh(d) * (lo - d) + np.tanh(d).astype(np.float32) >= np.tanh(lo).astype(np.float32)

    l = -np.ones_like(U)
    r = np.zeros_like(U)
    for _ in range(64):
        ok = check_lower(U, l)
        if ok.all():
            break
        l = np.where(ok, l, l * f32(2)).astype(np.float32)
    for _ in range(100):
        m = ((l + r) / f32(2)).astype(np.float32)
        ok = check_lower(U, m)
        l = np.where(ok, m, l).astype(np.float32)
        r = np.where(ok, r, m).astype(np.float32)
    d_lower = l
    l = np.zeros_like(U)
    r = np.ones_like(U)
    for _ in range(64):
        ok = check_upper(Lw, r)
        if ok.all():
            break
        r = np.where(ok, r, r * f32(2)).astype(np.float32)
    for _ in range(100):
        m = ((l + r) / f32(2)).astype(np.float32)
        ok = check_upper(Lw, m)
        l = np.where(ok, l, m).astype(np.float32)
        r = np.where(ok, m, r).astype(np.float32)
    d_upper = r
    return d_lower, d_upper


def _relax(l, u, activ):
    """per-neuron linear relaxation lw x + lb <= act(x) <= uw x + ub on [l, u] (float32).
    relu: exts/auto_LiRPA/operators/activation.py:306-323,386-388; tanh: BoundTanh.bound_relax_impl, the non-optimised branch
    (:925-956, 992-1016), masks from :11-13."""
    f32 = np.float32
    if activ == "relu":
        lb_r = np.minimum(l, f32(0))
        ub_r = np.maximum(u, f32(0))
        ub_r = np.maximum(ub_r, lb_r + f32(1e-8))
        upper_d = ub_r / (ub_r - lb_r)
        return (upper_d > f32(0.5)).astype(np.float32), np.zeros_like(upper_d), upper_d, -lb_r * upper_d
    assert activ == "tanh"
    lower = np.maximum(l, f32(-500)).astype(np.float32)
    upper = np.minimum(u, f32(500)).astype(np.float32)
    y_l, y_u = np.tanh(lower).astype(np.float32), np.tanh(upper).astype(np.float32)
    close = (upper - lower) < f32(1e-6)
    k_direct = np.where(close, _dtanh(upper), (y_u - y_l) / np.maximum(upper - lower, f32(1e-6))).astype(np.float32)
    pos = (l >= 0).astype(np.float32)
    neg = (u <= 0).astype(np.float32)
    both = f32(1) - pos - neg
    lw = np.zeros_like(lower); lb = np.zeros_like(lower); uw = np.zeros_like(lower); ub = np.zeros_like(lower)

    def add(mask, kind, k, x0, y0):
        nonlocal lw, lb, uw, ub
        if kind == "lower":
            lw = lw + mask * k; lb = lb + mask * (-x0 * k + y0)
        else:
            uw = uw + mask * k; ub = ub + mask * (-x0 * k + y0)

    add(neg, "upper", k_direct, lower, y_l)
    add(pos, "lower", k_direct, lower, y_l)
    d_lower, d_upper = _tanh_tangent_points(lower, upper)
    m = ((lower + upper) / f32(2)).astype(np.float32)
    y_m, k_m = np.tanh(m).astype(np.float32), _dtanh(m)
    add(neg, "lower", k_m, m, y_m)
    add(pos, "upper", k_m, m, y_m)
    md = both * (k_direct < _dtanh(lower)).astype(np.float32)
    add(md, "lower", k_direct, lower, y_l)
    add(both - md, "lower", _dtanh(d_lower), d_lower, np.tanh(d_lower).astype(np.float32))
    md = both * (k_direct < _dtanh(upper)).astype(np.float32)
    add(md, "upper", k_direct, lower, y_l)
    add(both - md, "upper", _dtanh(d_upper), d_upper, np.tanh(d_upper).astype(np.float32))
    return lw.astype(np.float32), lb.astype(np.float32), uw.astype(np.float32), ub.astype(np.float32)


def _crown_linear_chain(Ws, bs, xl, xu, pre_bounds, activ="relu"):
    """Backward CROWN bounds of the output of linear layer len(Ws)-1 given the
    pre-activation bounds of all earlier layers.  Ws[j], bs[j] float32.
    Returns (lb, ub) float32."""
    last = len(Ws) - 1
    lA = Ws[last].copy()
    uA = Ws[last].copy()
    lbias = bs[last].copy()
    ubias = bs[last].copy()
    for j in range(last - 1, -1, -1):
        l, u = pre_bounds[j]
        lower_d, lower_b, upper_d, upper_b = _relax(l, u, activ)
        lA_pos, lA_neg = np.maximum(lA, 0), np.minimum(lA, 0)
        uA_pos, uA_neg = np.maximum(uA, 0), np.minimum(uA, 0)
        lbias = lbias + lA_neg @ upper_b + lA_pos @ lower_b
        ubias = ubias + uA_pos @ upper_b + uA_neg @ lower_b
        lA = lA_pos * lower_d[None, :] + lA_neg * upper_d[None, :]
        uA = uA_pos * upper_d[None, :] + uA_neg * lower_d[None, :]
        # through linear layer j
        lbias = lbias + lA @ bs[j]
        ubias = ubias + uA @ bs[j]
        lA = lA @ Ws[j]
        uA = uA @ Ws[j]
    lb = _concretize(lA, lbias, xl, xu, -1)
    ub = _concretize(uA, ubias, xl, xu, +1)
    return lb.astype(np.float32), ub.astype(np.float32)


def crown_output_bounds(Ws64, bs64, x1min, x1max):
    """CROWN bounds on the output of a ReLU chain Linear-ReLU-...-Linear,
    intermediate bounds by CROWN as well.  float32 throughout."""
    Ws = [np.asarray(W, dtype=np.float32) for W in Ws64]
    bs = [np.asarray(b, dtype=np.float32) for b in bs64]
    xl = np.asarray(x1min, dtype=np.float32)
    xu = np.asarray(x1max, dtype=np.float32)
    pre_bounds = []
    for j in range(len(Ws) - 1):
        pre_bounds.append(_crown_linear_chain(Ws[: j + 1], bs[: j + 1], xl, xu, pre_bounds))
    return _crown_linear_chain(Ws, bs, xl, xu, pre_bounds), pre_bounds


def intervals_crown_sliced(net: FeedFwdNet, x1min, x1max, activ: str = "relu") -> IntervalsInfo:
    """activ = "tanh": the reference's bridge handles BoundTanh the same way (exts/auto_lirpa_bridge.py:31-37,86-87)"""
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    K = net.K
    Ws = [net.W(k) for k in range(K)]
    bs = [net.b(k) for k in range(K)]
    x_intvs = [(x1min.copy(), x1max.copy())]
    # Slices k=1..K-1: layers 1..k then an identity layer [I 0]; the k-th slice bounds x_{k+1}
    # (intervals_auto_lirpa.jl:12-28).  The pre-activation bounds of the shared prefix are
    # identical across slices, so they are computed once and reused.
    W32 = [np.asarray(W, dtype=np.float32) for W in Ws]
    b32 = [np.asarray(b, dtype=np.float32) for b in bs]
    xl32 = x1min.astype(np.float32)
    xu32 = x1max.astype(np.float32)
    pre_bounds = []
    for k in range(1, K):
        pre_bounds.append(_crown_linear_chain(W32[:k], b32[:k], xl32, xu32, pre_bounds, activ))
        n = net.xdims[k]
        Wk = W32[:k] + [np.eye(n, dtype=np.float32)]
        bk = b32[:k] + [np.zeros(n, dtype=np.float32)]
        lb, ub = _crown_linear_chain(Wk, bk, xl32, xu32, pre_bounds, activ)
        lb = lb.astype(np.float64)
        ub = ub.astype(np.float64)
        lb = np.minimum(lb, ub)
        ub = np.maximum(lb, ub)
        x_intvs.append((lb, ub))
    # last slice = the full network
    lb, ub = _crown_linear_chain(W32, b32, xl32, xu32, pre_bounds, activ)
    lb = lb.astype(np.float64)
    ub = ub.astype(np.float64)
    lb = np.minimum(lb, ub)
    ub = np.maximum(lb, ub)
    x_intvs.append((lb, ub))

    acx_intvs = []
    for k in range(K - 1):
        Wk, bk = Ws[k], bs[k]
        xkmin, xkmax = x_intvs[k]
        Wp, Wn = np.maximum(Wk, 0), np.minimum(Wk, 0)
        ykmin = Wp @ xkmin + Wn @ xkmax + bk
        ykmax = Wp @ xkmax + Wn @ xkmin + bk
        assert np.all(ykmin <= ykmax)
        acx_intvs.append((ykmin, ykmax))
    return IntervalsInfo(x_intvs=x_intvs, acx_intvs=acx_intvs)


def intervals_worst_case(net: FeedFwdNet, x1min, x1max) -> IntervalsInfo:
    """Plain IBP (src/Intervals/intervals_easy.jl:2-37), used as a soundness cross-check."""
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    x_intvs = [(x1min, x1max)]
    acx_intvs = []
    lo, hi = x1min, x1max
    for k in range(net.K):
        Wk, bk = net.W(k), net.b(k)
        Wp, Wn = np.maximum(Wk, 0), np.minimum(Wk, 0)
        ylo = Wp @ lo + Wn @ hi + bk
        yhi = Wp @ hi + Wn @ lo + bk
        if k < net.K - 1:
            acx_intvs.append((ylo, yhi))
            lo, hi = np.maximum(ylo, 0), np.maximum(yhi, 0)
        else:
            lo, hi = ylo, yhi
        x_intvs.append((lo, hi))
    return IntervalsInfo(x_intvs=x_intvs, acx_intvs=acx_intvs)
