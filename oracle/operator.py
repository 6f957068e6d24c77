"""Structured (per-entry) form of the affine LMI  Z(gamma) = Z0 + sum_i gamma_i G_i
on the clique sparsity pattern, as a sparse matrix in scaled-svec coordinates.

Derived entry by entry from the literal assembly in oracle/qc.py (which mirrors
src/Qc/*.jl); tests/test_oracle_assembly.py checks the two agree to 1e-12 on
random gamma.  With z = [x_1; ...; x_K; 1], a = Zdim-1 the affine index,
y_t = z-index xdims[0]+t of neuron t, and u_t = row t of [A b]
(src/Qc/activ.jl:33-39: W_k[r,:] on block k, b_k[r] on a):

  gin_i   (input.jl:24-26)          -2 e_i e_i' + (l+u)(e_i e_a' + e_a e_i') - 2 l u e_a e_a'
  gout    (output.jl:75,84,93)      -cout e_a e_a'   (cout = 2 hplane, 1 circle/ellipsoid)
  gac1_t  (activ_bounded.jl:19-21)  same three-entry pattern on y_t with (acymin, acymax)
  lam_t   (activ_sector.jl:42-43)   -2 smin smax u u' + (smin+smax)(u e_y' + e_y u')
  v_ij    (activ_sector.jl:29-35,43,45)  (du dy' + dy du') - 2 dy dy',  du=u_i-u_j, dy=e_yi-e_yj
  eta_t   (activ_sector.jl:55-56)   -smin (u e_a' + e_a u') + (e_y e_a' + e_a e_y')
  nu_t                              -smax (u e_a' + e_a u') + (e_y e_a' + e_a e_y')
  Z0      (output.jl:34-106)        Eout' R' S(gout=0) R Eout

gamma layout: [gin (xdims[0]); gout (1, reach only); gac1 (acdim); lam (acdim); v (#pairs);
eta (acdim); nu (acdim)]  = [gin; gout; gac1; gac2] of the reference's `values` dict.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import scipy.sparse as sp

from .qc import (Query, QcSafety, QcReachHplane, QcReachCircle, QcReachEllipsoid,
                 make_S, make_side, clique_index_sets)

SQRT2 = np.sqrt(2.0)


@dataclass
class Pattern:
    Zdim: int
    cliques: List[np.ndarray]          # index sets (sorted, 0-based)
    rows: np.ndarray                   # (N_E,) row index i of each pattern entry (i >= j)
    cols: np.ndarray                   # (N_E,) col index j
    index: np.ndarray                  # (Zdim, Zdim) int32 map (i,j)->entry, -1 outside pattern
    count: np.ndarray                  # (N_E,) number of cliques containing the entry
    gather: List[np.ndarray]           # per clique (n_k, n_k) int32 pattern positions

    @property
    def NE(self) -> int:
        return len(self.rows)

    def svec_scale(self) -> np.ndarray:
        s = np.full(self.NE, SQRT2)
        s[self.rows == self.cols] = 1.0
        return s


def build_pattern(Zdim: int, cliques: List[List[int]]) -> Pattern:
    cl = [np.asarray(sorted(c), dtype=np.int64) for c in cliques]
    mask = np.zeros((Zdim, Zdim), dtype=np.int32)
    for c in cl:
        mask[np.ix_(c, c)] += 1
    low = np.tril(mask)
    # entries ordered column-major over the lower triangle (j outer, i inner): deterministic
    cols, rows = np.nonzero(low.T)
    index = -np.ones((Zdim, Zdim), dtype=np.int32)
    index[rows, cols] = np.arange(len(rows), dtype=np.int32)
    index[cols, rows] = index[rows, cols]
    count = low[rows, cols].astype(np.float64)
    gather = [index[np.ix_(c, c)].astype(np.int32) for c in cl]
    return Pattern(Zdim=Zdim, cliques=cl, rows=rows, cols=cols, index=index, count=count, gather=gather)


class _Coo:
    def __init__(self, pat: Pattern):
        self.pat = pat
        self.r, self.c, self.v = [], [], []

    def add_sym(self, gen: int, pi, pv, qi, qv, alpha: float):
        """accumulate alpha * (p q' + q p') for sparse vectors p, q into generator `gen`."""
        if alpha == 0.0:
            return
        pi = np.asarray(pi); qi = np.asarray(qi)
        pv = np.asarray(pv, dtype=np.float64); qv = np.asarray(qv, dtype=np.float64)
        I = np.concatenate([np.repeat(pi, len(qi)), np.repeat(qi, len(pi))])
        J = np.concatenate([np.tile(qi, len(pi)), np.tile(pi, len(qi))])
        V = alpha * np.concatenate([np.outer(pv, qv).ravel(), np.outer(qv, pv).ravel()])
        keep = (I >= J) & (V != 0.0)
        I, J, V = I[keep], J[keep], V[keep]
        pos = self.pat.index[I, J]
        assert np.all(pos >= 0), "generator entry outside the clique pattern"
        self.r.append(pos); self.c.append(np.full(len(pos), gen)); self.v.append(V)

    def tocsc(self, ng: int):
        if not self.r:
            return sp.csc_matrix((self.pat.NE, ng))
        r = np.concatenate(self.r); c = np.concatenate(self.c); v = np.concatenate(self.v)
        M = sp.coo_matrix((v, (r, c)), shape=(self.pat.NE, ng)).tocsc()
        M.sum_duplicates()
        return M


@dataclass
class LmiOperator:
    pat: Pattern
    A: sp.csc_matrix          # (N_E, ng) columns = svec(G_i) (off-diagonals scaled by sqrt 2)
    z0: np.ndarray            # (N_E,) svec(Z0)
    c: np.ndarray             # (ng,)

    @property
    def ng(self) -> int:
        return self.A.shape[1]

    def Z_dense(self, gamma) -> np.ndarray:
        z = (self.z0 + self.A @ np.asarray(gamma, dtype=np.float64)) / self.pat.svec_scale()
        Z = np.zeros((self.pat.Zdim, self.pat.Zdim))
        Z[self.pat.rows, self.pat.cols] = z
        Z[self.pat.cols, self.pat.rows] = z
        return Z


def neuron_rows(q: Query):
    """For every neuron t: (indices, values) of u_t, its y index, and its layer."""
    net = q.net
    xd = net.xdims
    a = net.Zdim - 1
    offs = np.concatenate([[0], np.cumsum(xd[:-1])]).astype(int)
    out = []
    for k in range(net.K - 1):
        Wk, bk = net.W(k), net.b(k)
        blk = np.arange(offs[k], offs[k] + xd[k])
        for r in range(xd[k + 1]):
            ui = np.concatenate([blk, [a]])
            uv = np.concatenate([Wk[r, :], [bk[r]]])
            out.append((ui, uv, offs[k + 1] + r, k))
    return out


@dataclass
class Congruence:
    """z = T zhat with  z_i = h_i * zhat_i' + m_i * zhat_a  (i kept),  z_i = m_i * zhat_a
    (i eliminated, h_i = 0),  z_a = zhat_a.  m, h = midpoint / half-width of the interval
    known for z_i (input box for x_1, [acymin, acymax] for neurons).  Z~ = T' Z T keeps the
    NSD cone and the clique structure (a is in every clique).  This is solver-internal
    normalisation; gamma is unchanged by it and Z(gamma) is always reported in the
    reference's coordinates."""
    m: np.ndarray            # (Zdim-1,)
    h: np.ndarray            # (Zdim-1,)  0 => eliminated
    newpos: np.ndarray       # (Zdim,) reduced index or -1
    nred: int

    def tvec(self, idx, val):
        """T' applied to a sparse vector given as (indices, values) in full coordinates."""
        idx = np.asarray(idx); val = np.asarray(val, dtype=np.float64)
        a_full = len(self.newpos) - 1
        ared = self.nred - 1
        oi, ov = [], []
        acc = 0.0
        for i, v in zip(idx, val):
            if i == a_full:
                acc += v
            else:
                acc += v * self.m[i]
                if self.newpos[i] >= 0:
                    oi.append(self.newpos[i]); ov.append(v * self.h[i])
        oi.append(ared); ov.append(acc)
        return np.asarray(oi, dtype=np.int64), np.asarray(ov)


INTERVAL_GUARD = 5e-5   # relative floor on the half-width of a neuron's interval (nnsdp_options.interval_guard)


def make_congruence(q: Query, eliminate: bool = True, identity: bool = False, guard: float = INTERVAL_GUARD) -> Congruence:
    Zdim = q.net.Zdim
    if identity:
        return Congruence(m=np.zeros(Zdim - 1), h=np.ones(Zdim - 1),
                          newpos=np.arange(Zdim), nred=Zdim)
    lo = np.concatenate([q.qc_input.x1min, q.qc_bounded.acymin])
    hi = np.concatenate([q.qc_input.x1max, q.qc_bounded.acymax])
    # the last-layer block x_K is covered by acymin/acymax too (acdim = sum xdims[1:K])
    assert len(lo) == Zdim - 1
    m = 0.5 * (lo + hi)
    h = 0.5 * (hi - lo)
    # interval guard: a neuron interval narrower than `guard` x |midpoint| is widened to that (the solver then
    # works with the weaker, still valid, QC; a gamma feasible for the widened LMI is feasible for the original
    # one: Z_orig(gamma) = Z_wid(gamma) - 2 gac1 (h_wid^2 - h^2) e_a e_a').  The reference's float32 CROWN boxes
    # of collapsed deep nets (W10-D80, W20-D70, ...) are narrower than their own rounding error and exclude the
    # float64 trajectories by up to 2e-5 relative; taken literally they make the QC set empty and rho = 0 "optimal".
    nin = len(q.qc_input.x1min)
    h[nin:] = np.maximum(h[nin:], guard * np.abs(m[nin:]))
    if not eliminate:
        # no elimination (e.g. safety queries, where every multiplier has a cost): floor h
        h = np.maximum(h, 1e-6 * max(1.0, float(np.max(h))))
    keep = np.concatenate([h > 0, [True]])
    newpos = -np.ones(Zdim, dtype=np.int64)
    newpos[keep] = np.arange(int(keep.sum()))
    return Congruence(m=m, h=h, newpos=newpos, nred=int(keep.sum()))


def build_operator(q: Query, mode: str = "single", normalize: bool = False) -> LmiOperator:
    """normalize=False: the reference's coordinates (used for assembly parity and for Z(gamma)).
    normalize=True : the solver's internal coordinates (Congruence above); generators whose
    image vanishes (e.g. gac1 of an eliminated neuron) become zero columns."""
    net = q.net
    Zdim = net.Zdim
    a = Zdim - 1
    cg = make_congruence(q, eliminate=q.is_reach, identity=not normalize)
    cl_full = clique_index_sets(net, q.beta, mode)
    cl = [sorted({int(cg.newpos[i]) for i in c if cg.newpos[i] >= 0}) for c in cl_full]
    pat = build_pattern(cg.nred, cl)
    nin, nout, n1, n2 = q.gamma_dims()
    ng = nin + nout + n1 + n2
    coo = _Coo(pat)
    one = np.array([1.0])
    ea = cg.tvec([a], one)

    def ev(i):
        return cg.tvec([i], one)
    def add_box(g, i, l, u):
        """-2 (z_i - l)(z_i - u); in solver coordinates exactly -2 h^2 (zhat_i^2 - zhat_a^2) (factored: no cancellation)."""
        if not normalize:
            coo.add_sym(g, *ev(i), *ev(i), -1.0)             # -2 e_i e_i'
            coo.add_sym(g, *ev(i), *ea, (l + u))
            coo.add_sym(g, *ea, *ea, -l * u)                 # -2 l u e_a e_a'
        elif cg.newpos[i] >= 0:
            hh = cg.h[i] * cg.h[i]
            ri, ra = [int(cg.newpos[i])], [cg.nred - 1]
            coo.add_sym(g, ri, one, ri, one, -hh)
            coo.add_sym(g, ra, one, ra, one, hh)
    # --- gin
    for i in range(nin):
        add_box(i, i, q.qc_input.x1min[i], q.qc_input.x1max[i])
    # --- gout
    if nout:
        cout = 2.0 if isinstance(q.qc_out, QcReachHplane) else 1.0
        coo.add_sym(nin, *ea, *ea, -0.5 * cout)
    # --- activations
    rows = neuron_rows(q)
    acdim = len(rows)
    o1 = nin + nout                  # gac1
    ol = o1 + n1                     # lambda
    pairs = q.qc_sector.pairs()
    ov = ol + acdim                  # pair multipliers
    oe = ov + len(pairs)             # eta
    on = oe + acdim                  # nu
    relu = q.qc_sector.activ == "relu"
    assert (on + acdim if relu else oe) == ng
    smin, smax = q.qc_sector.smin, q.qc_sector.smax
    ut = [cg.tvec(ui, uv) for (ui, uv, _, _) in rows]
    yt = [ev(y) for (_, _, y, _) in rows]
    for t in range(acdim):
        add_box(o1 + t, rows[t][2], q.qc_bounded.acymin[t], q.qc_bounded.acymax[t])
        g = ol + t
        coo.add_sym(g, *ut[t], *ut[t], -smin[t] * smax[t])   # -2 smin smax u u'
        coo.add_sym(g, *ut[t], *yt[t], smin[t] + smax[t])
        if relu:
            g = oe + t
            coo.add_sym(g, *ut[t], *ea, -smin[t])
            coo.add_sym(g, *yt[t], *ea, 1.0)
            g = on + t
            coo.add_sym(g, *ut[t], *ea, -smax[t])
            coo.add_sym(g, *yt[t], *ea, 1.0)
    for r, (i, j) in enumerate(pairs):
        g = ov + r
        dyi = np.concatenate([yt[i][0], yt[j][0]]); dyv = np.concatenate([yt[i][1], -yt[j][1]])
        dui = np.concatenate([ut[i][0], ut[j][0]]); duv = np.concatenate([ut[i][1], -ut[j][1]])
        coo.add_sym(g, dui, duv, dyi, dyv, 1.0)
        coo.add_sym(g, dyi, dyv, dyi, dyv, -1.0)
    A = coo.tocsc(ng)
    # --- Z0: constant part of Zout, transformed by the congruence
    S0 = make_S(np.zeros(1), q.qc_out, net)
    R = make_side(net)                                      # rows: [x_1; W_K x_K + b_K; 1]
    xd = net.xdims
    offs = np.concatenate([[0], np.cumsum(xd[:-1])]).astype(int)
    idx = np.concatenate([np.arange(xd[0]), np.arange(offs[net.K - 1], offs[net.K - 1] + xd[net.K - 1]), [a]])
    Rt = np.zeros((R.shape[0], cg.nred))                    # R Eout T
    for r in range(R.shape[0]):
        ti, tv = cg.tvec(idx, R[r, :])
        np.add.at(Rt[r], ti, tv)
    Z0 = Rt.T @ S0 @ Rt
    Z0 = 0.5 * (Z0 + Z0.T)
    assert np.all((pat.index >= 0) | (Z0 == 0.0)), "Z0 outside pattern"
    z0 = Z0[pat.rows, pat.cols] * pat.svec_scale()
    Asc = sp.diags(pat.svec_scale()) @ A
    Asc = Asc.tocsc()
    Asc.eliminate_zeros()
    return LmiOperator(pat=pat, A=Asc, z0=z0, c=q.cost())
