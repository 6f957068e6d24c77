"""Literal (dense) restatement of the reference's QC / LMI assembly.

Every function mirrors one Julia function and builds the same dense matrices
the reference builds (E' * P * E, R' * Q * R), with a numeric gamma instead of
JuMP variables.  It is O(Zdim^2 * acdim) per call and meant for small nets and
for pinning the structured (per-entry) assembly used by the C oracle and the
HIP kernels.

  MyMath.e / E / Ec           src/MyMath.jl:26-52
  makeZin                     src/Qc/input.jl:19-42
  makeSide, makeZout          src/Qc/output.jl:34-49, 52-61, 64-106
  makeA, makeb, makeB, makeZac  src/Qc/activ.jl:7-42
  makeQ (bounded)             src/Qc/activ_bounded.jl:13-24
  makeQ (sector)              src/Qc/activ_sector.jl:23-60
  makeSectorMinMax            src/Qc/activ_sector.jl:63-90
  makeQcActivsIntvs           src/Qc/activ.jl:45-66
  makeCliques                 src/Methods/chordal_cliques.jl:13-59

All index arithmetic below is 0-based; comments give the 1-based Julia form.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .nnet_io import FeedFwdNet
from .intervals import IntervalsInfo, intervals_crown_sliced


# --------------------------------------------------------------------------- MyMath
def e(i: int, dim: int) -> np.ndarray:
    v = np.zeros(dim)
    v[i] = 1.0
    return v


def E(i: int, dims: List[int]) -> np.ndarray:
    """Block selector: dims[i] x sum(dims) (MyMath.jl:34-43)."""
    width = int(sum(dims))
    low = int(sum(dims[:i]))
    M = np.zeros((dims[i], width))
    M[np.arange(dims[i]), low + np.arange(dims[i])] = 1.0
    return M


def Ec(elems: List[int], N: int) -> np.ndarray:
    """Clique selector n_C x N (MyMath.jl:45-52); elems sorted, unique, 0-based."""
    elems = list(elems)
    assert elems == sorted(set(elems)) and len(elems) >= 1
    assert 0 <= elems[0] and elems[-1] < N
    M = np.zeros((len(elems), N))
    M[np.arange(len(elems)), elems] = 1.0
    return M


# --------------------------------------------------------------------------- QC descriptors
@dataclass
class QcInputBox:
    x1min: np.ndarray
    x1max: np.ndarray

    @property
    def vardim(self) -> int:
        return len(self.x1min)


@dataclass
class QcSafety:
    S: np.ndarray
    vardim: int = 0


@dataclass
class QcReachHplane:
    normal: np.ndarray
    vardim: int = 1


@dataclass
class QcReachCircle:
    yc: np.ndarray
    vardim: int = 1


@dataclass
class QcReachEllipsoid:
    invP: np.ndarray
    yc: np.ndarray
    vardim: int = 1


@dataclass
class QcActivBounded:
    acymin: np.ndarray
    acymax: np.ndarray

    @property
    def acydim(self) -> int:
        return len(self.acymin)

    @property
    def vardim(self) -> int:
        return len(self.acymin)


@dataclass
class QcActivSector:
    acxdim: int
    beta: int
    smin: np.ndarray
    smax: np.ndarray
    base_smin: float = 0.0
    base_smax: float = 1.0
    activ: str = "relu"            # "relu" | "tanh"  (ReluActiv / TanhActiv)

    @property
    def lamdim(self) -> int:
        # sum((acxdim-beta):acxdim)  (activ_sector.jl:18)
        return int(sum(range(self.acxdim - self.beta, self.acxdim + 1)))

    @property
    def vardim(self) -> int:
        # (activ isa ReluActiv) ? _λdim + 2 * acxdim : _λdim   (activ_sector.jl:19)
        return self.lamdim + (2 * self.acxdim if self.activ == "relu" else 0)

    def pairs(self) -> List[Tuple[int, int]]:
        # ijs = [(i, j) for i in 1:(acxdim-1) for j in (i+1):acxdim if j-i <= beta]
        n, b = self.acxdim, self.beta
        return [(i, j) for i in range(n - 1) for j in range(i + 1, min(n, i + b + 1))]


# --------------------------------------------------------------------------- input.jl
def make_Zin(gin, qc: QcInputBox, net: FeedFwdNet) -> np.ndarray:
    gin = np.asarray(gin, dtype=np.float64)
    assert len(gin) == qc.vardim
    G = np.diag(gin)
    P11 = -2.0 * G
    P12 = G @ (qc.x1min + qc.x1max)
    P22 = -2.0 * qc.x1min @ G @ qc.x1max
    P = np.block([[P11, P12[:, None]], [P12[None, :], np.array([[P22]])]])
    zd = net.zdims
    Ein = np.vstack([E(0, zd), E(net.K, zd)])
    return Ein.T @ P @ Ein


# --------------------------------------------------------------------------- output.jl
def make_side(net: FeedFwdNet) -> np.ndarray:
    xd, K = net.xdims, net.K
    WK, bK = net.W(K - 1), net.b(K - 1)
    d1, dK, m = xd[0], xd[K - 1], xd[K]
    R = np.zeros((d1 + m + 1, d1 + dK + 1))
    R[:d1, :d1] = np.eye(d1)
    R[d1:d1 + m, d1:d1 + dK] = WK
    R[d1:d1 + m, -1] = bK
    R[-1, -1] = 1.0
    return R


def make_S(gout, qc, net: FeedFwdNet) -> np.ndarray:
    xd, K = net.xdims, net.K
    d1, m = xd[0], xd[K]
    if isinstance(qc, QcSafety):
        return np.asarray(qc.S, dtype=np.float64)
    g = float(np.asarray(gout).reshape(-1)[0])
    S11 = np.zeros((d1, d1))
    S12 = np.zeros((d1, m))
    S13 = np.zeros(d1)
    if isinstance(qc, QcReachHplane):
        S22 = np.zeros((m, m))
        S23 = np.asarray(qc.normal, dtype=np.float64)
        S33 = -2.0 * g
    elif isinstance(qc, QcReachCircle):
        S22 = np.eye(m)
        S23 = -np.asarray(qc.yc, dtype=np.float64)
        S33 = float(qc.yc @ qc.yc) - g
    elif isinstance(qc, QcReachEllipsoid):
        S22 = qc.invP.T @ qc.invP
        S23 = -qc.invP.T @ qc.yc
        S33 = float(qc.yc @ qc.yc) - g          # reference quirk: yc'yc, not yc'invP'invP yc
    else:
        raise ValueError(f"unrecognized qc: {qc}")
    return np.block([
        [S11, S12, S13[:, None]],
        [S12.T, S22, S23[:, None]],
        [S13[None, :], S23[None, :], np.array([[S33]])],
    ])


def make_Zout(gout, qc, net: FeedFwdNet) -> np.ndarray:
    zd, K = net.zdims, net.K
    S = make_S(gout, qc, net)
    Eout = np.vstack([E(0, zd), E(K - 1, zd), E(K, zd)])
    R = make_side(net)
    return Eout.T @ R.T @ S @ R @ Eout


# --------------------------------------------------------------------------- activ.jl
def make_A(net: FeedFwdNet) -> np.ndarray:
    edims = net.zdims[:-1]
    fdims = edims[1:]
    A = np.zeros((int(sum(fdims)), int(sum(edims))))
    for k in range(net.K - 1):
        A += E(k, fdims).T @ net.W(k) @ E(k, edims)
    return A


def make_b(net: FeedFwdNet) -> np.ndarray:
    return np.concatenate([net.b(k) for k in range(net.K - 1)])


def make_B(net: FeedFwdNet) -> np.ndarray:
    edims = net.zdims[:-1]
    fdims = edims[1:]
    B = np.zeros((int(sum(fdims)), int(sum(edims))))
    for j in range(net.K - 1):
        B += E(j, fdims).T @ E(j + 1, edims)
    return B


def make_Q_bounded(gac, qc: QcActivBounded) -> np.ndarray:
    gac = np.asarray(gac, dtype=np.float64)
    n = qc.acydim
    D = np.diag(gac)
    Q = np.zeros((2 * n + 1, 2 * n + 1))
    Q22 = -2.0 * D
    Q23 = D @ (qc.acymin + qc.acymax)
    Q33 = -2.0 * qc.acymin @ D @ qc.acymax
    Q[n:2 * n, n:2 * n] = Q22
    Q[n:2 * n, -1] = Q23
    Q[-1, n:2 * n] = Q23
    Q[-1, -1] = Q33
    return Q


def make_Q_sector(gac, qc: QcActivSector) -> np.ndarray:
    gac = np.asarray(gac, dtype=np.float64)
    assert len(gac) == qc.vardim
    n, beta = qc.acxdim, qc.beta
    lam0 = gac[:n]
    if beta > 0:
        ijs = qc.pairs()
        assert n + len(ijs) == qc.lamdim
        Delta = np.zeros((len(ijs), n))
        for r, (i, j) in enumerate(ijs):
            Delta[r, i] = 1.0
            Delta[r, j] = -1.0
        v = gac[n:n + len(ijs)]
        T = Delta.T @ (v[:, None] * Delta)
    else:
        T = np.zeros((n, n))
    bmin, bmax = qc.base_smin, qc.base_smax
    smin, smax = qc.smin, qc.smax
    Q11 = -2.0 * np.diag(smin * smax * lam0) - 2.0 * (bmin * bmax * T)
    Q12 = np.diag((smin + smax) * lam0) + (bmin + bmax) * T
    Q22 = -2.0 * T
    ld = qc.lamdim
    Q13 = np.zeros(n)
    Q23 = np.zeros(n)
    if qc.activ == "relu":            # activ_sector.jl:49-57
        eta = gac[ld:ld + n]
        nu = gac[ld + n:ld + 2 * n]
        Q13 = -smin * eta - smax * nu
        Q23 = eta + nu
    Q = np.zeros((2 * n + 1, 2 * n + 1))
    Q[:n, :n] = Q11
    Q[:n, n:2 * n] = Q12
    Q[n:2 * n, :n] = Q12.T
    Q[n:2 * n, n:2 * n] = Q22
    Q[:n, -1] = Q13
    Q[-1, :n] = Q13
    Q[n:2 * n, -1] = Q23
    Q[-1, n:2 * n] = Q23
    return Q


def make_R(net: FeedFwdNet) -> np.ndarray:
    A, b, B = make_A(net), make_b(net), make_B(net)
    n = B.shape[0]
    R = np.zeros((2 * n + 1, net.Zdim))
    R[:n, :-1] = A
    R[:n, -1] = b
    R[n:2 * n, :-1] = B
    R[-1, -1] = 1.0
    return R


def make_Zac(gac, qc, net: FeedFwdNet, R: Optional[np.ndarray] = None) -> np.ndarray:
    Q = make_Q_bounded(gac, qc) if isinstance(qc, QcActivBounded) else make_Q_sector(gac, qc)
    if R is None:
        R = make_R(net)
    return R.T @ Q @ R


def make_sector_min_max(acxmin, acxmax, activ: str = "relu") -> Tuple[np.ndarray, np.ndarray]:
    """makeSectorMinMax (src/Qc/activ_sector.jl:63-90), both branches, statement by statement."""
    acxmin = np.asarray(acxmin, dtype=np.float64)
    acxmax = np.asarray(acxmax, dtype=np.float64)
    assert len(acxmin) == len(acxmax)
    eps = 1e-4
    if activ == "relu":
        smin = np.zeros(len(acxmin))
        smax = np.ones(len(acxmax))
        smin[acxmin > eps] = 1.0
        smax[acxmax < -eps] = 0.0
        return smin, smax
    if activ == "tanh":
        smin = np.zeros(len(acxmin))
        smax = np.ones(len(acxmax))
        for i in range(len(acxmin)):
            if acxmin[i] * acxmax[i] >= 0:
                smin[i] = np.tanh(acxmax[i]) / acxmax[i]
                smax[i] = np.tanh(acxmin[i]) / acxmin[i]
            else:
                smin[i] = min(np.tanh(acxmin[i]) / acxmin[i], np.tanh(acxmax[i]) / acxmax[i])
                smax[i] = 1.0
        return smin, smax
    raise ValueError(f"unsupported activation: {activ}")


def scale_S(S, alphas, net: FeedFwdNet) -> np.ndarray:
    """scaleS (src/Qc/output.jl:109-124): S for the scaled network f' = prod(alphas) f, block by block through the
    selectors F_i = E(i, [xdims[1]; xdims[end]; 1])."""
    assert len(alphas) == net.K
    alpha = float(np.prod(alphas))
    sdims = [net.xdims[0], net.xdims[-1], 1]
    F1, F2, F3 = E(0, sdims), E(1, sdims), E(2, sdims)
    S = np.asarray(S, dtype=np.float64)
    S11 = F1 @ S @ F1.T
    S12 = F1 @ S @ F2.T / alpha
    S13 = F1 @ S @ F3.T
    S22 = F2 @ S @ F2.T / alpha ** 2
    S23 = F2 @ S @ F3.T / alpha
    S33 = F3 @ S @ F3.T
    return np.block([[S11, S12, S13], [S12.T, S22, S23], [S13.T, S23.T, S33]])


def make_qc_activs(net: FeedFwdNet, x1min, x1max, beta: int,
                   intv: Optional[IntervalsInfo] = None):
    if intv is None:
        intv = intervals_crown_sliced(net, x1min, x1max)
    acdim = int(sum(net.xdims[1:-1]))
    acymin = np.concatenate([iv[0] for iv in intv.x_intvs[1:-1]])
    acymax = np.concatenate([iv[1] for iv in intv.x_intvs[1:-1]])
    qc_bounded = QcActivBounded(acymin=acymin, acymax=acymax)
    sec_min = np.concatenate([iv[0] for iv in intv.acx_intvs])
    sec_max = np.concatenate([iv[1] for iv in intv.acx_intvs])
    smin, smax = make_sector_min_max(sec_min, sec_max)
    qc_sector = QcActivSector(acxdim=acdim, beta=beta, smin=smin, smax=smax)
    assert qc_bounded.acydim == acdim
    return qc_bounded, qc_sector


# --------------------------------------------------------------------------- Utils/qc.jl
def hplane_S(normal, h, net: FeedFwdNet) -> np.ndarray:
    """Utils.hplaneS (src/Utils/qc.jl:27-37)."""
    d1, m = net.xdims[0], net.xdims[-1]
    S = np.zeros((d1 + m + 1, d1 + m + 1))
    S[d1:d1 + m, -1] = normal
    S[-1, d1:d1 + m] = normal
    S[-1, -1] = -2.0 * h
    return S


def approx_ellipsoid(net: FeedFwdNet, x1min, x1max, N: int = 100000, seed: int = 1234):
    """Utils.approxEllipsoid (src/Utils/qc.jl:50-67).  The reference draws its samples from
    Julia's RNG (seed in experiments/scale.jl:9), which cannot be reproduced; numpy's
    default_rng(seed) is used instead."""
    from .nnet_io import eval_net
    rng = np.random.default_rng(seed)
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    pts = x1min[:, None] + rng.random((net.xdims[0], N)) * (x1max - x1min)[:, None]
    Y = eval_net(net, pts)
    yc = Y.sum(axis=1) / N
    Yd = Y - yc[:, None]
    P = Yd @ Yd.T
    a, b = 1.0, 4.0
    w, V = np.linalg.eigh(P)
    if w.max() * a >= w.min() * b:
        lmin, lmax = w.min(), w.max()
        mw = (w - lmin) * ((b - a) / (lmax - lmin)) + a
        P = V @ np.diag(mw) @ V.T
        P = 0.5 * (P + P.T)
    return P, yc


# --------------------------------------------------------------------------- chordal_cliques.jl
def make_cliques(net: FeedFwdNet, beta: int):
    """Returns a list of (Ck, [Ck1, Ck2] or [Cp], [Dk1(,Dk2)]) with 0-based indices
    (src/Methods/chordal_cliques.jl:13-59).  Dk* index INTO Ck."""
    K = net.K
    xd = net.xdims

    def S(k):   # S(k) = sum(xdims[1:k]) in Julia
        return int(sum(xd[:k]))

    p = 1
    for i in range(1, K + 1):
        if S(i + 1) + beta >= S(K - 1):
            p = i
            break
    cliques = []
    zd = net.zdims
    for k in range(1, p):
        Ck1 = list(range(S(k - 1), S(k + 1) + beta))          # S(k-1)+1 : S(k+1)+beta
        Ck2 = list(range(S(K - 1), S(K) + 1))                 # S(K-1)+1 : S(K)+1
        assert Ck1[-1] <= Ck2[0]
        Ck = Ck1 + Ck2
        # NOTE: for beta>0 near the end Ck1 and Ck2 could touch; the reference asserts <=,
        # which allows one shared index; keep the list literal (duplicates would break Ec).
        n = len(Ck)
        if k == 1:
            cliques.append((Ck, [Ck1, Ck2], [list(range(n))]))
        else:
            nk, nk1 = zd[k - 1], zd[k]
            Dk1 = list(range(nk + nk1 + beta)) + [n - 1]
            Dk2 = list(range(nk + nk1, n))
            cliques.append((Ck, [Ck1, Ck2], [Dk1, Dk2]))
    Cp = list(range(S(p - 1), S(K) + 1))
    cliques.append((Cp, [Cp], [list(range(len(Cp)))]))
    return cliques


def clique_index_sets(net: FeedFwdNet, beta: int, mode: str = "single") -> List[List[int]]:
    """Index sets of the PSD blocks actually constrained by setupZs!
    (src/Methods/chordal_sdp.jl:19-57): 'single' -> one block per Ck; 'double' -> blocks
    Ck[Dk1], Ck[Dk2] for the middle cliques; 'dense' -> one block 0..Zdim-1 (DeepSDP,
    src/Methods/deep_sdp.jl:57)."""
    if mode == "dense":
        return [list(range(net.Zdim))]
    if mode == "path":
        # extension of this build (not in the reference): cliques {x_k, x_{k+1} (+beta spill), a}; exact when the
        # output QC has no x_1 -- x_K coupling (S12 = 0), see nn-sdp_amd/csrc/setup.hpp
        S = np.concatenate([[0], np.cumsum(net.xdims)]).astype(int)
        K = net.K
        return [list(range(S[k - 1], min(S[k + 1] + beta, S[K]))) + [int(S[K])] for k in range(1, K)]
    out = []
    for Ck, _, Dks in make_cliques(net, beta):
        if mode == "single" or len(Dks) == 1:
            out.append(list(Ck))
        else:
            for D in Dks:
                out.append([Ck[i] for i in D])
    return out


# --------------------------------------------------------------------------- whole Z(gamma)
@dataclass
class Query:
    """ReachQuery / SafetyQuery (src/Methods/Methods.jl:22-43) with numeric QC data."""
    net: FeedFwdNet
    qc_input: QcInputBox
    qc_out: object                      # QcSafety or QcReach*
    qc_bounded: QcActivBounded
    qc_sector: QcActivSector

    @property
    def is_reach(self) -> bool:
        return not isinstance(self.qc_out, QcSafety)

    @property
    def beta(self) -> int:
        return self.qc_sector.beta

    def gamma_dims(self):
        """(nin, nout, nac1, nac2); gamma = [gin; gout; gac1; gac2]."""
        return (self.qc_input.vardim, 1 if self.is_reach else 0,
                self.qc_bounded.vardim, self.qc_sector.vardim)

    @property
    def ngamma(self) -> int:
        return int(sum(self.gamma_dims()))

    def cost(self) -> np.ndarray:
        """Objective vector: e_{gout} for reach (NnSdp.jl:46); ones on gin and all gac for
        safety (deep_sdp.jl:25 / chordal_sdp.jl:111)."""
        nin, nout, n1, n2 = self.gamma_dims()
        c = np.zeros(nin + nout + n1 + n2)
        if self.is_reach:
            c[nin] = 1.0
        else:
            c[:] = 1.0
        return c


def split_gamma(q: Query, gamma):
    nin, nout, n1, n2 = q.gamma_dims()
    g = np.asarray(gamma, dtype=np.float64)
    o = 0
    gin = g[o:o + nin]; o += nin
    gout = g[o:o + nout]; o += nout
    g1 = g[o:o + n1]; o += n1
    g2 = g[o:o + n2]
    return gin, gout, g1, g2


def assemble_Z_literal(q: Query, gamma, R: Optional[np.ndarray] = None) -> np.ndarray:
    """Z = Zin + Zout + sum(Zacs)  (deep_sdp.jl:56 / chordal_sdp.jl:146)."""
    gin, gout, g1, g2 = split_gamma(q, gamma)
    if R is None:
        R = make_R(q.net)
    Z = make_Zin(gin, q.qc_input, q.net)
    Z = Z + make_Zout(gout if q.is_reach else None, q.qc_out, q.net)
    Z = Z + make_Zac(g1, q.qc_bounded, q.net, R)
    Z = Z + make_Zac(g2, q.qc_sector, q.net, R)
    return Z


def make_reach_ellipsoid_query(net: FeedFwdNet, x1min, x1max, beta: int,
                               seed: int = 1234, nsamples: int = 100000) -> Query:
    """NnSdp.findEllipsoid up to the runQuery call (src/NnSdp.jl:35-47)."""
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    qb, qs = make_qc_activs(net, x1min, x1max, beta)
    P, yc = approx_ellipsoid(net, x1min, x1max, nsamples, seed)
    invP = np.linalg.inv(P)
    invP = 0.5 * (invP + invP.T)
    return Query(net=net, qc_input=QcInputBox(x1min, x1max),
                 qc_out=QcReachEllipsoid(invP=invP, yc=yc), qc_bounded=qb, qc_sector=qs)


def make_reach_hplane_query(net: FeedFwdNet, x1min, x1max, beta: int, normal) -> Query:
    """One direction of NnSdp.findReach2Dpoly (src/NnSdp.jl:73-95)."""
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    qb, qs = make_qc_activs(net, x1min, x1max, beta)
    return Query(net=net, qc_input=QcInputBox(x1min, x1max),
                 qc_out=QcReachHplane(normal=np.asarray(normal, dtype=np.float64)),
                 qc_bounded=qb, qc_sector=qs)


def make_safety_query(net: FeedFwdNet, x1min, x1max, beta: int, S) -> Query:
    x1min = np.asarray(x1min, dtype=np.float64)
    x1max = np.asarray(x1max, dtype=np.float64)
    qb, qs = make_qc_activs(net, x1min, x1max, beta)
    return Query(net=net, qc_input=QcInputBox(x1min, x1max),
                 qc_out=QcSafety(S=np.asarray(S, dtype=np.float64)),
                 qc_bounded=qb, qc_sector=qs)
