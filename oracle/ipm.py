"""Independent CPU solver for the reference's DeepSDP model: a dense primal-dual interior-point method.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing under nn-sdp_amd/ may import this.

Why it exists: oracle/admm.py restates the SAME first-order method the HIP library runs, so a modelling error shared
by both would pass every HIP-vs-oracle test.  This file solves the model the reference hands to MOSEK
(src/Methods/deep_sdp.jl:36-61 reach, :10-33 safety):

        min  c' gamma      s.t.  gamma >= 0,   -Z(gamma) in PSDCone,   Z(gamma) = Zin + Zout + sum(Zacs)

with a different algorithm (Mehrotra predictor-corrector, HKM direction, infeasible start - the family MOSEK's
conic optimiser belongs to; MOSEK itself is closed source and absent), on the ONE dense cone of DeepSdpOptions, in the
reference's own coordinates, with generators taken from the LITERAL dense assembly (oracle/qc.py assemble_Z_literal,
i.e. E' P E and R' Q R exactly as src/Qc/*.jl builds them).  It shares no code with oracle/operator.py (per-entry
generator table), oracle/admm.py or the product's normalisation / clique machinery.

Standard form used below (block-diagonal cone K = S^n_+ x R^m_+, y = gamma):
    (D)  max b'y   s.t.  sum_i y_i A_i + S = C,  S in K        A_i = (G_i, -e_i),  C = (-Z0, 0),  b = -c
    (P)  min <C,X> s.t.  <A_i, X> = b_i,  X in K               X = (X_s, x_l):  <G_i, X_s> - x_l[i] = -c_i
Cost: one m x m Schur complement per iteration, O(m n^3 + m^2 n^2) dense flops - meant for W10-D5 / W10-D10 / W20-D10.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from . import qc as oqc


@dataclass
class IpmOptions:
    max_iters: int = 200
    tol_gap: float = 1e-9           # relative duality gap
    tol_feas: float = 1e-9          # relative primal / dual infeasibility
    step_frac: float = 0.98
    verbose: bool = False
    drop_zero_generators: bool = True   # multipliers whose generator is identically 0 (cost-free, effect-free): fixed at 0


@dataclass
class IpmResult:
    gamma: np.ndarray               # full length, reference layout [gin; gout; gac1; gac2]
    objective: float                # c' gamma
    dual_objective: float           # <-Z0, X_s> (a lower bound when X is feasible)
    pinf: float
    dinf: float
    gap: float
    iters: int
    status: str
    lambda_max: float               # eigmax(Z(gamma)) in the reference's coordinates
    history: list = field(default_factory=list)


def literal_generators(q: oqc.Query):
    """Z0 and the stack G_i = Z(e_i) - Z0 from the literal dense assembly (one R'QR per multiplier)."""
    R = oqc.make_R(q.net)
    ng = q.ngamma
    Z0 = oqc.assemble_Z_literal(q, np.zeros(ng), R)
    n = Z0.shape[0]
    G = np.zeros((ng, n, n))
    e = np.zeros(ng)
    for i in range(ng):
        e[i] = 1.0
        G[i] = oqc.assemble_Z_literal(q, e, R) - Z0
        e[i] = 0.0
    G = 0.5 * (G + np.transpose(G, (0, 2, 1)))
    return 0.5 * (Z0 + Z0.T), G


def _max_step(L, dM):
    """largest a with  L L' + a dM  PSD  (L = chol of the current point)."""
    T = sla.solve_triangular(L, dM, lower=True)
    T = sla.solve_triangular(L, T.T, lower=True)
    lam = np.linalg.eigvalsh(0.5 * (T + T.T))[0]
    return np.inf if lam >= 0 else -1.0 / lam


def solve_lmi(Z0: np.ndarray, G: np.ndarray, c: np.ndarray, opts: Optional[IpmOptions] = None) -> IpmResult:
    o = opts or IpmOptions()
    ng_full, n = G.shape[0], G.shape[1]
    keep = np.arange(ng_full)
    if o.drop_zero_generators:
        nz = np.abs(G).reshape(ng_full, -1).max(axis=1) > 0
        if np.any(~nz & (c != 0)):   # a zero generator with a cost stays at 0 anyway (gamma >= 0, minimisation)
            pass
        keep = np.nonzero(nz)[0]
    Gk = G[keep]
    ck = c[keep]
    m = len(keep)
    Gs = sp.csr_matrix(Gk.reshape(m, n * n))
    b = -ck
    C_s = -Z0
    normb = 1.0 + np.abs(b).max()
    normC = 1.0 + np.abs(C_s).max()

    def A_of(Xs, xl):            # <A_i, X>
        return Gs @ Xs.reshape(-1) - xl

    def At_of(y):                # sum y_i A_i  (SDP part; LP part is -y)
        return (Gs.T @ y).reshape(n, n)

    # infeasible start (SDPT3-style magnitudes)
    gnorm = np.sqrt((Gk.reshape(m, -1) ** 2).sum(axis=1))
    xi = max(10.0, np.sqrt(n), n * np.max((1.0 + np.abs(b)) / (1.0 + np.maximum(gnorm, 1.0))))
    eta = max(10.0, np.sqrt(n), np.max(np.maximum(gnorm, 1.0)), np.sqrt((C_s ** 2).sum()))
    Xs, Ss = xi * np.eye(n), eta * np.eye(n)
    xl, sl = np.full(m, xi), np.full(m, eta)
    y = np.zeros(m)
    hist: List[dict] = []
    status = "ITERATION_LIMIT"
    best = None
    for it in range(o.max_iters):
        Rp = b - A_of(Xs, xl)
        Rd_s = C_s - Ss - At_of(y)
        Rd_l = -sl + y                                  # C_l - s_l - (-y)
        mu = (np.sum(Xs * Ss) + xl @ sl) / (n + m)
        pobj = np.sum(C_s * Xs)                          # primal objective of (P)
        dobj = b @ y                                     # dual objective of (D) = -c'gamma
        pinf = np.abs(Rp).max() / normb
        dinf = max(np.abs(Rd_s).max(), np.abs(Rd_l).max() if m else 0.0) / normC
        gap = abs(pobj - dobj) / (1.0 + abs(pobj) + abs(dobj))
        hist.append(dict(it=it, obj=-dobj, lower=-pobj, pinf=pinf, dinf=dinf, gap=gap, mu=mu))
        if o.verbose:
            print(f"[ipm] {it:3d} c'g {-dobj:.10g}  lower {-pobj:.10g}  pinf {pinf:.2e} dinf {dinf:.2e} gap {gap:.2e} mu {mu:.2e}", flush=True)
        score = max(pinf, dinf, gap)
        if best is None or score < best[0]:
            best = (score, y.copy(), Xs.copy(), it)
        if pinf <= o.tol_feas and dinf <= o.tol_feas and gap <= o.tol_gap:
            status = "OPTIMAL"
            break
        try:
            Ls = np.linalg.cholesky(Ss)
            Lx = np.linalg.cholesky(Xs)
        except np.linalg.LinAlgError:
            status = "NUMERICAL_ERROR"
            break
        Sinv = sla.cho_solve((Ls, True), np.eye(n))
        Sinv = 0.5 * (Sinv + Sinv.T)
        # Schur complement  M_ij = <G_i, X G_j S^-1> + delta_ij x_l/s_l
        U = np.matmul(np.matmul(Sinv[None, :, :], Gk), Xs[None, :, :])      # S^-1 G_j X = (X G_j S^-1)'
        M = Gs @ U.reshape(m, n * n).T
        M = 0.5 * (M + M.T)
        M[np.diag_indices(m)] += xl / sl
        try:
            cf = sla.cho_factor(M, lower=True)
        except np.linalg.LinAlgError:
            M[np.diag_indices(m)] += 1e-12 * np.trace(M) / m
            try:
                cf = sla.cho_factor(M, lower=True)
            except np.linalg.LinAlgError:
                status = "NUMERICAL_ERROR"
                break
        XRdSi = Xs @ Rd_s @ Sinv
        # LP parts follow from A_i^l = -e_i:  <A_i, X Rd S^-1>_l = -x_i rd_i / s_i,  <A_i, S^-1>_l = -1 / s_i
        base = Rp + Gs @ XRdSi.reshape(-1) - xl * Rd_l / sl
        AX = A_of(Xs, xl)
        ASinv = Gs @ Sinv.reshape(-1) - 1.0 / sl

        def direction(sig_mu, corr_s=None, corr_l=None):
            rhs = base - sig_mu * ASinv + AX
            if corr_s is not None:
                rhs = rhs + (Gs @ (corr_s @ Sinv).reshape(-1) - corr_l / sl)
            dy = sla.cho_solve(cf, rhs)
            dSs = Rd_s - At_of(dy)
            dsl = Rd_l + dy
            T = sig_mu * Sinv - Xs - Xs @ dSs @ Sinv
            if corr_s is not None:
                T = T - corr_s @ Sinv
            dXs = 0.5 * (T + T.T)
            dxl = (sig_mu - xl * sl - xl * dsl - (corr_l if corr_l is not None else 0.0)) / sl
            return dy, dXs, dxl, dSs, dsl

        def steps(dXs, dxl, dSs, dsl):
            ap = _max_step(Lx, dXs)
            neg = dxl < 0
            if np.any(neg):
                ap = min(ap, np.min(-xl[neg] / dxl[neg]))
            ad = _max_step(Ls, dSs)
            neg = dsl < 0
            if np.any(neg):
                ad = min(ad, np.min(-sl[neg] / dsl[neg]))
            return ap, ad

        dy, dXs, dxl, dSs, dsl = direction(0.0)
        ap, ad = steps(dXs, dxl, dSs, dsl)
        ap, ad = min(1.0, ap), min(1.0, ad)
        mu_aff = (np.sum((Xs + ap * dXs) * (Ss + ad * dSs)) + (xl + ap * dxl) @ (sl + ad * dsl)) / (n + m)
        sigma = min(1.0, max(1e-6, (mu_aff / mu) ** 3)) if mu > 0 else 0.1
        dy, dXs, dxl, dSs, dsl = direction(sigma * mu, dXs @ dSs, dxl * dsl)
        ap, ad = steps(dXs, dxl, dSs, dsl)
        ap, ad = min(1.0, o.step_frac * ap), min(1.0, o.step_frac * ad)
        if max(ap, ad) < 1e-8:
            status = "SLOW_PROGRESS"
            break
        Xs = Xs + ap * dXs
        xl = xl + ap * dxl
        y = y + ad * dy
        Ss = Ss + ad * dSs
        sl = sl + ad * dsl
        Xs = 0.5 * (Xs + Xs.T)
        Ss = 0.5 * (Ss + Ss.T)
    if status != "OPTIMAL" and best is not None:
        _, y, Xs, _ = best
        if best[0] <= 1e-7:
            status = "NEAR_OPTIMAL"      # stalled (step length / Cholesky) inside 1e-7 of optimality: these models have no Slater point
    gamma = np.zeros(ng_full)
    gamma[keep] = np.maximum(y, 0.0)
    Z = Z0 + np.tensordot(gamma, G, axes=1)
    h = hist[-1] if status == "OPTIMAL" else hist[best[3]]
    return IpmResult(gamma=gamma, objective=float(c @ gamma), dual_objective=float(-np.sum(C_s * Xs)), pinf=h["pinf"], dinf=h["dinf"],
                     gap=h["gap"], iters=len(hist), status=status, lambda_max=float(np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1]), history=hist)


def solve_query(q: oqc.Query, opts: Optional[IpmOptions] = None) -> IpmResult:
    Z0, G = literal_generators(q)
    return solve_lmi(Z0, G, q.cost(), opts)
