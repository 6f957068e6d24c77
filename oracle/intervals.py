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


def _crown_linear_chain(Ws, bs, xl, xu, pre_bounds):
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
        lb_r = np.minimum(l, np.float32(0))
        ub_r = np.maximum(u, np.float32(0))
        ub_r = np.maximum(ub_r, lb_r + np.float32(1e-8))
        upper_d = ub_r / (ub_r - lb_r)
        upper_b = -lb_r * upper_d
        lower_d = (upper_d > np.float32(0.5)).astype(np.float32)
        lA_pos, lA_neg = np.maximum(lA, 0), np.minimum(lA, 0)
        uA_pos, uA_neg = np.maximum(uA, 0), np.minimum(uA, 0)
        lbias = lbias + lA_neg @ upper_b
        ubias = ubias + uA_pos @ upper_b
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


def intervals_crown_sliced(net: FeedFwdNet, x1min, x1max) -> IntervalsInfo:
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
        pre_bounds.append(_crown_linear_chain(W32[:k], b32[:k], xl32, xu32, pre_bounds))
        n = net.xdims[k]
        Wk = W32[:k] + [np.eye(n, dtype=np.float32)]
        bk = b32[:k] + [np.zeros(n, dtype=np.float32)]
        lb, ub = _crown_linear_chain(Wk, bk, xl32, xu32, pre_bounds)
        lb = lb.astype(np.float64)
        ub = ub.astype(np.float64)
        lb = np.minimum(lb, ub)
        ub = np.maximum(lb, ub)
        x_intvs.append((lb, ub))
    # last slice = the full network
    lb, ub = _crown_linear_chain(W32, b32, xl32, xu32, pre_bounds)
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
