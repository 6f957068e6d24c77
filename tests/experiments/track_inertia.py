"""(experiment, not a test) small-side tracking WITH an exact completeness check.
Per block and iteration: the number of negative eigenvalues of A + tau I is read off an LDL' factorisation (Sylvester's law of
inertia); the tracked vectors U (last small side + g guards) are corrected by Davidson rounds preconditioned with the last full
eigenbasis; the tracked projection is accepted only if (i) every Ritz pair below +tau has a residual <= tol |A| and (ii) the number
of Ritz values below -tau... equals the inertia count (no negative direction is missing).  Otherwise: full eigendecomposition
(the "fallback"), which also refreshes basis and tracked set.  Refresh is also forced every R iterations.
usage: python tests/experiments/track_inertia.py W40-D20 0 <iters> <R> <guards> [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.linalg as sla, helpers
from oracle import operator as oop, admm as oadmm

name, beta, iters, R, g = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "single"
RMAX, MAXR = 12, int(os.environ.get("MAXR", "4"))
q = helpers.oracle_query(helpers.load_problem(name, beta))
P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
S = oadmm.AdmmState(P, 0.1, 1.6)
nb = len(S.nk)
state = [None] * nb
H = dict(tracked=0, full=0, forced=0, fail_conv=0, fail_inertia=0, fail_size=0, rounds=[], err=[], launches_with_fallback=0)


def full(k, A):
    w, Q = np.linalg.eigh(A)
    n = len(w)
    nneg, npos = int((w < 0).sum()), int((w > 0).sum())
    neg = nneg <= npos
    r = nneg if neg else npos
    rt = min(r + g, n)
    idx = np.arange(rt) if neg else np.arange(n - rt, n)
    comp = np.setdiff1d(np.arange(n), idx)
    state[k] = dict(V=Q, d=w, comp=comp, U=Q[:, idx].copy(), neg=neg, age=1, ok=rt <= RMAX)
    H["full"] += 1
    return (Q * np.maximum(w, 0)) @ Q.T


def inertia_neg(M):
    """number of negative eigenvalues of the symmetric M from its LDL' factorisation (Bunch-Kaufman, 1x1 and 2x2 pivots)"""
    _, D, _ = sla.ldl(M, lower=True)
    n = D.shape[0]
    cnt, i = 0, 0
    while i < n:
        if i + 1 < n and D[i + 1, i] != 0.0:
            ev = np.linalg.eigvalsh(D[i:i + 2, i:i + 2]); cnt += int((ev < 0).sum()); i += 2
        else:
            cnt += int(D[i, i] < 0); i += 1
    return cnt


def track(k, A, tol):
    st = state[k]
    n = A.shape[0]
    nrm = np.linalg.norm(A)
    if st is None:
        return full(k, A), "init"
    if st["age"] >= R:
        H["forced"] += 1
        return full(k, A), "forced"
    st["age"] += 1
    if not st["ok"]:
        H["fail_size"] += 1
        return full(k, A), "size"
    tau = tol * nrm
    if os.environ.get("ALG") == "kernel":
        return track_kernel_like(k, A, tol, st, nrm, tau)
    U, V, d, comp = st["U"], st["V"], st["d"], st["comp"]
    Vc, dc = V[:, comp], d[comp]
    sgn = 1.0 if st["neg"] else -1.0            # work with B = sgn * A so that the small side is always the negative one
    B = sgn * A
    conv = False
    for rnd in range(MAXR + 1):
        BU = B @ U
        G = U.T @ BU
        G = 0.5 * (G + G.T)
        th, Y = np.linalg.eigh(G)
        Res = (BU - U @ G) @ Y
        side = th < tau
        if np.linalg.norm(Res[:, side]) <= tol * nrm:
            conv = True
            break
        if rnd == MAXR:
            break
        Ur = U @ Y
        C = Vc.T @ Res
        den = sgn * dc[:, None] - th[None, :]
        floor = 1e-3 * nrm / np.sqrt(n)
        den = np.where(np.abs(den) < floor, np.where(den < 0, -floor, floor), den)
        Ur = Ur - Vc @ (C / den)
        U, _ = np.linalg.qr(Ur)
    if not conv:
        H["fail_conv"] += 1
        return full(k, A), "conv"
    # completeness: inertia of B + tau I against the Ritz values strictly below -tau
    cnt = inertia_neg(B + tau * np.eye(n))
    if cnt != int((th < -tau).sum()):
        H["fail_inertia"] += 1
        return full(k, A), "inertia"
    st["U"] = U
    H["tracked"] += 1; H["rounds"].append(rnd)
    UY = U @ Y
    Wneg = (UY * np.minimum(th, 0)) @ UY.T        # negative part of B restricted to the tracked space
    W = (B - Wneg) if st["neg"] else (-Wneg)      # small side negative: A_+ = A - A_-;  small side positive (B = -A): A_+ = -(B_-)
    return 0.5 * (W + W.T), "ok"


def track_kernel_like(k, A, tol, st, nrm, tau):
    """the form the HIP kernel uses: no Ritz rotation inside the rounds (shifts = diag(G)), Cholesky-QR, and the completeness /
    positivity check by a plain Cholesky of  (B - B_-(tracked)) + tau I  instead of an inertia count"""
    n = A.shape[0]
    U, V, d, comp = st["U"], st["V"], st["d"], st["comp"]
    Vc, dc = V[:, comp], d[comp]
    sgn = 1.0 if st["neg"] else -1.0
    B = sgn * A
    conv = False
    for rnd in range(MAXR + 1):
        BU = B @ U
        G = U.T @ BU
        G = 0.5 * (G + G.T)
        Res = BU - U @ G
        gd = np.diag(G)
        side = gd < tau + 1e-3 * nrm / np.sqrt(n)
        if np.linalg.norm(Res[:, side]) <= tol * nrm:
            conv = True
            break
        if rnd == MAXR:
            break
        C = Vc.T @ Res
        den = sgn * dc[:, None] - gd[None, :]
        floor = 1e-3 * nrm / np.sqrt(n)
        den = np.where(np.abs(den) < floor, np.where(den < 0, -floor, floor), den)
        Un = U - Vc @ (C / den)
        Rq = np.linalg.cholesky(Un.T @ Un)
        U = np.linalg.solve(Rq, Un.T).T
    if not conv:
        H["fail_conv"] += 1
        return full(k, A), "conv"
    th, Y = np.linalg.eigh(G)
    UY = U @ Y
    Wneg = (UY * np.minimum(th, 0)) @ UY.T
    try:
        np.linalg.cholesky(B - Wneg + tau * np.eye(n))
    except np.linalg.LinAlgError:
        H["fail_inertia"] += 1
        return full(k, A), "inertia"
    st["U"] = UY
    H["tracked"] += 1; H["rounds"].append(rnd)
    W = (B - Wneg) if st["neg"] else (-Wneg)
    return 0.5 * (W + W.T), "ok"


for it in range(1, iters + 1):
    nu = S.nu
    w = np.empty_like(nu)
    w[:S.ng] = np.maximum(nu[:S.ng], 0.0)
    tol = 1e-5 if it < 1000 else 1e-6
    emax, anyfb = 0.0, False
    for k, n in enumerate(S.nk):
        A = nu[S.offs[k]:S.offs[k + 1]].reshape(n, n)
        A = 0.5 * (A + A.T)
        W, why = track(k, A, tol)
        anyfb = anyfb or why in ("conv", "inertia", "size")
        if it % 25 == 0:
            emax = max(emax, np.linalg.norm(W - oadmm.project_psd(A)) / max(np.linalg.norm(A), 1e-300))
        w[S.offs[k]:S.offs[k + 1]] = W.ravel()
    H["launches_with_fallback"] += int(anyfb)
    S.proj = lambda nu_, w_=w: w_
    S.step()
    del S.proj
    if it % 25 == 0:
        H["err"].append(emax)
    if it % 500 == 0:
        r = np.array(H["rounds"]) if H["rounds"] else np.array([0])
        tot = 500 * nb
        print(f"it {it}: tracked {H['tracked']/tot:.3f}, forced refresh {H['forced']/tot:.3f}, fallbacks: not converged {H['fail_conv']/tot:.4f} inertia {H['fail_inertia']/tot:.4f} "
              f"size {H['fail_size']/tot:.4f}; launches with a fallback {H['launches_with_fallback']/500:.3f}; rounds mean {r.mean():.2f}; max projection error {max(H['err']):.1e}", flush=True)
        for kk in ("tracked", "full", "forced", "fail_conv", "fail_inertia", "fail_size", "launches_with_fallback"):
            H[kk] = 0
        H["rounds"], H["err"] = [], []
