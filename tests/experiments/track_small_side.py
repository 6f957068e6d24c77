"""(experiment, not a test) projection by TRACKING the small side of the spectrum instead of a full eigendecomposition.
Per block: full eigendecomposition every R iterations (basis V, eigenvalues d, tracked set S = small-sign side + g guards);
in between: Davidson-type correction of the tracked vectors U (n x rt) with the stale basis as preconditioner,
    G = U'AU, Res = AU - UG,  U += -V_c diag(1/(d_c - theta)) V_c' Res  (V_c: untracked stale eigenvectors), re-orthonormalise,
until |Res| <= tol |A|; W = A - U (G)_- U' (small side negative) or U (G)_+ U' (small side positive).
Reports rounds, tracked sizes, failures and the projection error against the exact one, along a real ADMM run of the oracle.
usage: python tests/experiments/track_small_side.py W40-D20 0 <iters> <refresh R> <guards g>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm

name, beta, iters, R, g = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "single"
RMAX, MAXR = 16, int(os.environ.get("MAXR", "6"))
q = helpers.oracle_query(helpers.load_problem(name, beta))
P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
S = oadmm.AdmmState(P, 0.1, 1.6)
nb = len(S.nk)
state = [None] * nb
hist = dict(rounds=[], rt=[], fail=0, full=0, err=[], miss=0)


def track_project(k, A, tol):
    st = state[k]
    n = A.shape[0]
    nrm = np.linalg.norm(A)
    if st is None or st["age"] >= R:
        w, Q = np.linalg.eigh(A)
        hist["full"] += 1
        npos, nneg = int((w > 0).sum()), int((w < 0).sum())
        neg_small = nneg <= npos
        r = nneg if neg_small else npos
        rt = min(r + g, n)
        idx = np.arange(rt) if neg_small else np.arange(n - rt, n)          # algebraically smallest / largest
        comp = np.setdiff1d(np.arange(n), idx)
        state[k] = dict(V=Q, d=w, idx=idx, comp=comp, U=Q[:, idx].copy(), neg=neg_small, age=1, ok=rt <= RMAX)
        return (Q * np.maximum(w, 0)) @ Q.T
    st["age"] += 1
    if not st["ok"]:
        w, Q = np.linalg.eigh(A); hist["full"] += 1
        return (Q * np.maximum(w, 0)) @ Q.T
    U, V, d, comp = st["U"], st["V"], st["d"], st["comp"]
    Vc, dc = V[:, comp], d[comp]
    for rnd in range(MAXR + 1):
        AU = A @ U
        G = U.T @ AU
        G = 0.5 * (G + G.T)
        Res = AU - U @ G
        # only the Ritz pairs on the small side have to be accurate (the guards merely have to be present)
        th0, Y0 = np.linalg.eigh(G)
        side = (th0 < tol * nrm) if st["neg"] else (th0 > -tol * nrm)
        if np.linalg.norm((Res @ Y0)[:, side]) <= tol * nrm:
            break
        if rnd == MAXR:
            hist["fail"] += 1
            w, Q = np.linalg.eigh(A); hist["full"] += 1
            st["age"] = R          # force a refresh next time
            return (Q * np.maximum(w, 0)) @ Q.T
        th, Y = np.linalg.eigh(G)
        Ur, Rr = U @ Y, Res @ Y                                   # Ritz vectors and their residuals
        C = Vc.T @ Rr                                             # (n - rt) x rt
        den = dc[:, None] - th[None, :]
        den = np.where(np.abs(den) < 1e-3 * nrm / np.sqrt(n), np.sign(den + 1e-300) * 1e-3 * nrm / np.sqrt(n), den)
        Ur = Ur - Vc @ (C / den)
        U, _ = np.linalg.qr(Ur)
    st["U"] = U
    hist["rounds"].append(rnd); hist["rt"].append(U.shape[1])
    th, Y = np.linalg.eigh(G)
    UY = U @ Y
    if st["neg"]:
        W = A - (UY * np.minimum(th, 0)) @ UY.T
    else:
        W = (UY * np.maximum(th, 0)) @ UY.T
    return 0.5 * (W + W.T)


errs_it = []
for it in range(1, iters + 1):
    nu = S.nu
    w = np.empty_like(nu)
    w[:S.ng] = np.maximum(nu[:S.ng], 0.0)
    tol = 1e-5 if it < 1000 else 1e-6
    emax = 0.0
    for k, n in enumerate(S.nk):
        A = nu[S.offs[k]:S.offs[k + 1]].reshape(n, n)
        A = 0.5 * (A + A.T)
        W = track_project(k, A, tol)
        if it % 50 == 0:
            Wex = oadmm.project_psd(A)
            emax = max(emax, np.linalg.norm(W - Wex) / max(np.linalg.norm(A), 1e-300))
        w[S.offs[k]:S.offs[k + 1]] = W.ravel()
    # the rest of the iteration as AdmmState.step does, with the tracked projection
    S.proj = lambda nu_, w_=w: w_
    S.step()
    del S.proj
    if it % 50 == 0:
        errs_it.append(emax)
    if it % 500 == 0:
        r = np.array(hist["rounds"]) if hist["rounds"] else np.array([0])
        print(f"it {it}: tracked projections {len(hist['rounds'])}, rounds mean {r.mean():.2f} max {r.max()}, tracked size mean {np.mean(hist['rt']):.1f} max {max(hist['rt'])}, "
              f"full decompositions {hist['full']}, failures {hist['fail']}, max projection error (sampled) {max(errs_it):.1e}", flush=True)
        hist["rounds"], hist["rt"], errs_it = [], [], []
