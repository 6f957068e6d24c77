"""(experiment, not a test) would a first-order sign-split correction replace Jacobi sweeps in the warm projection?
DESIGN.md section 9 item 1.  For consecutive ADMM iterates nu_k of the oracle at steady state: A' = V0' A V0 with the previous
eigenvectors; P / N = sign of diag(A'); correction X_ij = A'_ij / (d_j - d_i) on the P x N block only (skew), V1 = V0 cayley(X);
projection from the split  W = V1_P (V1_P' A V1_P)_+ V1_P'.  Reports the projection error against the exact eigendecomposition.
usage: python tests/experiments/sign_split_correction.py W40-D20 0 2000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm

name, beta, burn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
q = helpers.oracle_query(helpers.load_problem(name, beta))
P = oadmm.ScaledProblem(oop.build_operator(q, "single", normalize=True))
S = oadmm.AdmmState(P, 0.1, 1.6)
for _ in range(burn):
    S.step()


def blocks(nu):
    return [0.5 * (M + M.T) for M in (nu[S.offs[k]:S.offs[k + 1]].reshape(n, n) for k, n in enumerate(S.nk))]


prev = [np.linalg.eigh(A)[1] for A in blocks(S.nu)]
for it in range(6):
    S.step()
    stats = []
    for k, A in enumerate(blocks(S.nu)):
        n = A.shape[0]
        nrm = np.linalg.norm(A)
        w, Q = np.linalg.eigh(A)
        Wex = (Q * np.maximum(w, 0)) @ Q.T
        V0 = prev[k]
        Ap = V0.T @ A @ V0
        d = np.diag(Ap).copy()
        off0 = np.sqrt(np.sum(Ap ** 2) - np.sum(d ** 2)) / nrm
        V = V0
        errs = []
        for rnd in range(3):
            Ap = V.T @ A @ V
            d = np.diag(Ap).copy()
            pos = d > 0
            Pn = np.nonzero(pos)[0]; Nn = np.nonzero(~pos)[0]
            cpl = np.linalg.norm(Ap[np.ix_(Pn, Nn)]) / nrm
            # projection from the current split (no diagonalisation inside the blocks): exact iff the coupling is zero
            VP = V[:, Pn]
            App = VP.T @ A @ VP
            wp, Qp = np.linalg.eigh(App)
            W = VP @ ((Qp * np.maximum(wp, 0)) @ Qp.T) @ VP.T
            errs.append((np.linalg.norm(W - Wex) / nrm, cpl))
            # first-order correction on the P x N block, opposite-sign gaps |d_i| + |d_j|; Cayley transform keeps V orthogonal
            X = np.zeros((n, n))
            gap = d[Pn][:, None] - d[Nn][None, :]
            Xpn = -Ap[np.ix_(Pn, Nn)] / np.maximum(gap, 1e-300)          # X_ij = A_ij / (d_j - d_i)
            X[np.ix_(Pn, Nn)] = Xpn
            X[np.ix_(Nn, Pn)] = -Xpn.T
            V = V @ np.linalg.solve(np.eye(n) - 0.5 * X, np.eye(n) + 0.5 * X)
        stats.append((n, off0, len(Pn), errs))
        prev[k] = Q
    worst = max(stats, key=lambda t: t[3][1][0])
    print(f"it {it}: off(A')/|A| max {max(s[1] for s in stats):.1e}; |P| {min(s[2] for s in stats)}..{max(s[2] for s in stats)} of n {stats[0][0]}..{max(s[0] for s in stats)}; "
          f"projection error [split only, +1 correction, +2]: max over blocks "
          f"{max(s[3][0][0] for s in stats):.1e} {max(s[3][1][0] for s in stats):.1e} {max(s[3][2][0] for s in stats):.1e}; "
          f"P x N coupling {max(s[3][0][1] for s in stats):.1e} -> {max(s[3][1][1] for s in stats):.1e} -> {max(s[3][2][1] for s in stats):.1e}", flush=True)
