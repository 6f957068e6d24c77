"""CPU (numpy) statement of the chordal ADMM the HIP library runs (oracle = checker).

Problem (src/Methods/chordal_sdp.jl:125-153 / deep_sdp.jl:36-61 with numeric data):
    (P)  min c'g   s.t.  g >= 0,  z0 + A g = sum_k H_k' z_k,  mat(z_k) <= 0 (NSD)
whose Lagrange dual over the clique pattern is
    (D)  max z0'x  s.t.  c + A'x >= 0,  mat(H_k x) >= 0 (PSD)  for every clique k.
ADMM is applied to (D) with the splitting  s = c + A'x,  X_k = H_k x  and ONE penalty sigma
for both blocks, so the x-update matrix  D + A A'  (D = diag(#cliques containing an entry))
does not depend on sigma and sigma can be adapted for free.  With the stacked operator
K = [A'; H_1; ...; H_p], q = [c; 0] and the cone C = R+^ng x PSD x ... x PSD the iteration is
the fixed-point map on nu = w + y/sigma (w = proj_C(nu), y = sigma (nu - w)):
    r   = 2 w - nu                                   (reflection)
    x   = (D + A A')^-1 ( z0/sigma + K'(r - q) )     (Woodbury, M = I + A' D^-1 A factored once)
    nu+ = nu + alpha (K x + q - w)                   (alpha = over-relaxation)
At convergence  gamma = -y_s >= 0  and  Z_k = Y_k <= 0.  The PSD projection is one symmetric
eigendecomposition per clique per iteration -- the hot kernel of the HIP library.

MOSEK (the reference's solver, call sites src/Methods/Methods.jl:61,64,83) is closed source and
absent; this ADMM is the replacement named by BASELINE.json:north_star, pinned end-to-end
against the reference's published objective values (tests/golden/dump_scale.csv).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from .operator import LmiOperator, SQRT2


@dataclass
class AdmmOptions:
    max_iters: int = 20000
    eps_rel: float = 1e-6            # on both splitting residuals, relative
    sigma: float = 0.1
    alpha: float = 1.6               # over-relaxation
    adapt_sigma: bool = True
    adapt_every: int = 50
    check_every: int = 50
    verbose: bool = False


@dataclass
class AdmmResult:
    gamma: np.ndarray                # full-length multipliers in the reference's layout
    objective: float
    iters: int
    pres: float                      # |K x + q - w| / max(|K x + q|, |w|)
    dres: float                      # |K'y - z0| / max(|K'y|, |z0|)
    status: str
    history: list = field(default_factory=list)
    x: Optional[np.ndarray] = None


def project_psd(V: np.ndarray) -> np.ndarray:
    """PSD projection by symmetric eigendecomposition (LAPACK via numpy)."""
    w, Q = np.linalg.eigh(0.5 * (V + V.T))
    return (Q * np.maximum(w, 0.0)) @ Q.T


class ScaledProblem:
    """Generator (column) normalisation + unit-norm z0 and c.  g_i = e_i * g~_i / zscale."""

    def __init__(self, L: LmiOperator, drop_tol: float = 1e-150):
        A = L.A.tocsc()
        cn = np.sqrt(np.asarray(A.multiply(A).sum(axis=0)).ravel())
        self.keep = np.nonzero(cn > drop_tol)[0]          # zero generators: gamma_i = 0
        self.ecol = 1.0 / cn[self.keep]
        self.A = (A[:, self.keep] @ sp.diags(self.ecol)).tocsc()
        c = L.c[self.keep] * self.ecol
        zn, cnrm = np.linalg.norm(L.z0), np.linalg.norm(c)
        self.zscale = 1.0 / zn if zn > 0 else 1.0
        self.cscale = 1.0 / cnrm if cnrm > 0 else 1.0
        self.z0 = L.z0 * self.zscale
        self.c = c * self.cscale
        self.pat = L.pat
        self.ng_full = L.ng

    def unscale_gamma(self, gs: np.ndarray) -> np.ndarray:
        g = np.zeros(self.ng_full)
        g[self.keep] = gs * self.ecol / self.zscale
        return g


class AdmmState:
    """The fixed-point map, written once so tests can drive single steps."""

    def __init__(self, P: ScaledProblem, sigma: float, alpha: float):
        self.P = P
        pat = P.pat
        self.A = P.A.tocsr()
        self.At = P.A.T.tocsr()
        self.ng = P.A.shape[1]
        self.Dinv = 1.0 / pat.count
        M = np.eye(self.ng) + (self.At @ sp.diags(self.Dinv) @ P.A).toarray()
        self.Mfac = sla.cho_factor(M, lower=True)
        self.nk = [len(c) for c in pat.cliques]
        self.offs = np.concatenate([[self.ng], self.ng + np.cumsum([n * n for n in self.nk])]).astype(int)
        self.N = int(self.offs[-1])
        self.G = [g.ravel() for g in pat.gather]
        self.Wm = [np.where(np.eye(n, dtype=bool), 1.0, 1.0 / SQRT2).ravel() for n in self.nk]
        self.tril = [np.tril_indices(n) for n in self.nk]
        self.trilpos = [pat.gather[k][self.tril[k]] for k in range(len(self.nk))]
        self.trilw = [np.where(self.tril[k][0] == self.tril[k][1], 1.0, SQRT2) for k in range(len(self.nk))]
        self.sigma = sigma
        self.alpha = alpha
        self.nu = np.zeros(self.N)
        self.nu[:self.ng] = np.maximum(P.c, 0.0)

    def proj(self, nu):
        w = np.empty_like(nu)
        w[:self.ng] = np.maximum(nu[:self.ng], 0.0)
        for k, n in enumerate(self.nk):
            V = nu[self.offs[k]:self.offs[k + 1]].reshape(n, n)
            w[self.offs[k]:self.offs[k + 1]] = project_psd(V).ravel()
        return w

    def Kt(self, v):
        h = self.A @ v[:self.ng]
        for k, n in enumerate(self.nk):
            Mk = v[self.offs[k]:self.offs[k + 1]].reshape(n, n)
            np.add.at(h, self.trilpos[k], Mk[self.tril[k]] * self.trilw[k])
        return h

    def step(self):
        """one ADMM iteration; returns (w, x, residual K x + q - w)."""
        P, sg = self.P, self.sigma
        nu = self.nu
        w = self.proj(nu)
        r = 2.0 * w - nu
        h = np.zeros(P.pat.NE)
        for k, n in enumerate(self.nk):
            Mk = r[self.offs[k]:self.offs[k + 1]].reshape(n, n)
            np.add.at(h, self.trilpos[k], Mk[self.tril[k]] * self.trilw[k])
        g = self.Dinv * (P.z0 / sg + h)
        p = r[:self.ng] - P.c
        qv = self.At @ g - p
        ww = sla.cho_solve(self.Mfac, qv)
        x = g - self.Dinv * (self.A @ ww)
        Kxq = np.empty(self.N)
        Kxq[:self.ng] = p + ww + P.c                  # A'x + c, with A'x = p + ww exactly
        for k, n in enumerate(self.nk):
            Kxq[self.offs[k]:self.offs[k + 1]] = x[self.G[k]] * self.Wm[k]
        res = Kxq - w
        self.nu = nu + self.alpha * res
        return w, x, res, Kxq

    def set_sigma(self, new_sigma: float):
        """keep (w, y) and re-express nu = w + y/sigma for the new penalty."""
        w = self.proj(self.nu)
        y = self.sigma * (self.nu - w)
        self.sigma = new_sigma
        self.nu = w + y / new_sigma


def admm_solve(L: LmiOperator, opts: Optional[AdmmOptions] = None) -> AdmmResult:
    """L must be the solver-coordinates operator (build_operator(..., normalize=True))."""
    opts = opts or AdmmOptions()
    P = ScaledProblem(L)
    S = AdmmState(P, opts.sigma, opts.alpha)
    status = "ITERATION_LIMIT"
    hist = []
    rp = rd = np.inf
    it = 0
    x = None
    next_adapt = opts.adapt_every
    for it in range(1, opts.max_iters + 1):
        nu_prev = S.nu
        w, x, res, Kxq = S.step()
        if it % opts.check_every == 0 or it == opts.max_iters:
            y = S.sigma * (nu_prev - w)
            Kty = S.Kt(y)
            rp = np.linalg.norm(res) / max(np.linalg.norm(Kxq), np.linalg.norm(w), 1e-300)
            rd = np.linalg.norm(Kty - P.z0) / max(np.linalg.norm(Kty), np.linalg.norm(P.z0), 1e-300)
            obj = -(P.c @ y[:S.ng]) / (P.zscale * P.cscale)
            dobj = (P.z0 @ x) / (P.zscale * P.cscale)
            hist.append((it, rp, rd, obj, dobj, S.sigma))
            if opts.verbose:
                print(f"it {it:6d} pres {rp:.3e} dres {rd:.3e} obj {obj:.8g} dobj {dobj:.8g} sigma {S.sigma:.3g}")
            if rp <= opts.eps_rel and rd <= opts.eps_rel:
                status = "OPTIMAL"
                break
            # residual balancing on a geometric schedule (a fixed period makes sigma oscillate)
            if opts.adapt_sigma and it >= next_adapt:
                next_adapt = max(it + 2 * opts.adapt_every, it * 3 // 2)
                ratio = np.sqrt(max(rp, 1e-300) / max(rd, 1e-300))
                if ratio > 1.5 or ratio < 0.67:
                    S.set_sigma(S.sigma * min(max(ratio, 0.2), 5.0))
    w = S.proj(S.nu)
    y = S.sigma * (S.nu - w)
    gs = np.maximum(-y[:S.ng], 0.0)
    gam = P.unscale_gamma(gs)
    return AdmmResult(gamma=gam, objective=float(L.c @ gam), iters=it, pres=float(rp), dres=float(rd),
                      status=status, history=hist, x=x)


def certificate(L_full: LmiOperator, gamma) -> dict:
    """Independent check of a candidate gamma in the reference's coordinates: min(gamma) and
    lambda_max of the dense Z(gamma) (src/Methods/Methods.jl:116, experiments/acas.jl:76-79)."""
    Z = L_full.Z_dense(gamma)
    w = np.linalg.eigvalsh(Z)
    return {"lambda_max": float(w[-1]), "lambda_min": float(w[0]), "gamma_min": float(np.min(gamma)),
            "objective": float(L_full.c @ np.asarray(gamma))}
