"""(experiment, not a test) how many indices of a warm-started block are NOT certifiably in the positive bulk?
A' = V0' A V0 (V0 = exact eigenvectors of the previous iterate), d = diag(A'), Gershgorin radius r_i = sum_j |A'_ij|.
C(delta) = {i : d_i - r_i <= delta |A|_2}: everything else is a block whose eigenvalues exceed delta by Gershgorin.
usage: python tests/experiments/bulk_split.py W40-D20 0 2000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm
name, beta, burn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "single"
q = helpers.oracle_query(helpers.load_problem(name, beta))
P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
S = oadmm.AdmmState(P, 0.1, 1.6)
for _ in range(burn):
    S.step()
blocks = lambda nu: [0.5 * (M + M.T) for M in (nu[S.offs[k]:S.offs[k + 1]].reshape(n, n) for k, n in enumerate(S.nk))]
prev = [np.linalg.eigh(A)[1] for A in blocks(S.nu)]
for it in range(4):
    S.step()
    out = []
    for k, A in enumerate(blocks(S.nu)):
        w, Q = np.linalg.eigh(A)
        nrm2 = max(abs(w[0]), abs(w[-1]))
        Ap = prev[k].T @ A @ prev[k]
        d = np.diag(Ap)
        R = np.abs(Ap).sum(axis=1) - np.abs(d)
        cs = [int(((d - R) <= dl * nrm2).sum()) for dl in (0.0, 1e-3, 1e-2, 1e-1)]
        out.append((A.shape[0], int((w < 0).sum()), int((np.abs(w) < 1e-3 * nrm2).sum()), int((np.abs(w) < 1e-2 * nrm2).sum()), cs, R.max() / nrm2))
        prev[k] = Q
    print(f"it {it}: per block (n, #neg, #|lam|<1e-3, #|lam|<1e-2, |C| for delta 0/1e-3/1e-2/1e-1, max radius/|A|):")
    for o in out[:6] + out[-2:]:
        print("   ", o)
