"""(experiment, not a test) multi-round form of tests/experiments/refine_proj2.py: up to R verified Ogita-Aishima rounds, the pairs first
order cannot resolve rotated exactly (Givens, disjoint subset) at EVERY round; counts rounds per block and per launch.
usage: python tests/experiments/refine_proj3.py W40-D20 0 double 4000 [acc_factor=0.1] [--rounds=3] [--load=..] [--theta=0.25]"""
import os, sys, time, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name, beta, mode, iters = args[0], int(args[1]), args[2], int(args[3])
acc_factor = float(args[4]) if len(args) > 4 else 0.1
opt = dict(rounds=3, theta=0.25, load=None, pf=1.0, blind=1, kcap=0.3, gmax=64, cluster=0.0, cmax=48)
for a_ in sys.argv:
    if a_.startswith("--") and "=" in a_:
        k_, v_ = a_[2:].split("=")
        opt[k_] = type(opt[k_])(v_) if opt[k_] is not None else v_
q = helpers.oracle_query(helpers.load_problem(name, beta))
P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
theta = opt["theta"]


class Refiner:
    def __init__(self, nk):
        self.V = [None] * len(nk)
        self.tol = 1e-4
        self.tol_acc = 1e-4
        self.reset()

    def reset(self):
        self.hist = np.zeros(12, dtype=int)      # per block: rounds used (0 .. R), 10 = Jacobi, 11 = blind
        self.lh = np.zeros(12, dtype=int)        # per launch: max over blocks
        self.giv = 0
        self.csize = []
        self.maxerr = 0.0
        self.cur = 0
        self.calls = 0

    def project(self, k, A):
        self.calls += 1
        n = A.shape[0]
        fro = np.linalg.norm(A)
        I = np.eye(n)

        def exact():
            w, Q = np.linalg.eigh(A)
            self.V[k] = Q
            self.hist[10] += 1
            self.cur = max(self.cur, 10)
            return (Q * np.maximum(w, 0)) @ Q.T

        if self.V[k] is None or fro == 0.0:
            return exact()
        V = self.V[k]
        for rnd in range(opt["rounds"] + 1):
            B = V.T @ A @ V
            B = 0.5 * (B + B.T)
            R = I - V.T @ V
            d = np.diag(B).copy()
            E = B - np.diag(d)
            off = np.linalg.norm(E) / fro
            orth = np.linalg.norm(R)
            lim = self.tol if rnd == 0 else self.tol_acc
            if off <= lim and orth <= lim:
                self.hist[rnd] += 1
                self.cur = max(self.cur, rnd)
                self.V[k] = V
                W = (V * np.maximum(d / np.sum(V * V, axis=0), 0.0)) @ V.T
                return W
            if rnd == opt["rounds"]:
                break
            G = d[None, :] - d[:, None]
            if opt["cluster"] > 0:
                # near-zero cluster + the indices of pairs first order cannot resolve: exact eigendecomposition of that sub-block
                small = 0.1 * self.tol * fro / n
                with np.errstate(divide="ignore", invalid="ignore"):
                    K0 = np.abs(np.where(np.eye(n, dtype=bool), 0.0, E / np.where(G == 0, 1e-300, G)))
                bad = ((np.abs(E) > theta * np.abs(G)) | (K0 > opt["kcap"])) & (np.abs(E) > small)
                np.fill_diagonal(bad, False)
                inC = (np.abs(d) <= opt["cluster"] * off * fro) | bad.any(axis=0)
                C = np.nonzero(inC)[0]
                self.csize.append(len(C))
                if len(C) > opt["cmax"]:
                    break
                if len(C) >= 2:
                    wC, QC = np.linalg.eigh(B[np.ix_(C, C)])
                    J = np.eye(n)
                    J[np.ix_(C, C)] = QC
                    V = V @ J
                    B = J.T @ B @ J
                    R = J.T @ R @ J
                    d = np.diag(B).copy()
                    E = B - np.diag(d)
                    G = d[None, :] - d[:, None]
            # exact Givens rotations on the pairs first order cannot resolve (disjoint subset, largest coupling first)
            small = 0.1 * self.tol * fro / n
            with np.errstate(divide="ignore", invalid="ignore"):
                K0 = np.abs(np.where(np.eye(n, dtype=bool), 0.0, E / np.where(G == 0, 1e-300, G)))
            bad = ((np.abs(E) > theta * np.abs(G)) | (K0 > opt["kcap"])) & (np.abs(E) > small)
            np.fill_diagonal(bad, False)
            if bad.any():
                ii, jj = np.nonzero(np.tril(bad, -1))
                order = np.argsort(-np.abs(E[ii, jj]))
                used = np.zeros(n, dtype=bool)
                J = np.eye(n)
                ng = 0
                for t in order:
                    i, j = ii[t], jj[t]
                    if used[i] or used[j] or ng >= opt["gmax"]:
                        continue
                    used[i] = used[j] = True
                    th = 0.5 * np.arctan2(2 * B[i, j], B[j, j] - B[i, i])
                    cs, sn = np.cos(th), np.sin(th)
                    J[i, i] = cs; J[j, j] = cs; J[i, j] = sn; J[j, i] = -sn
                    ng += 1
                self.giv += ng
                V = V @ J
                B = J.T @ B @ J
                R = J.T @ R @ J
                d = np.diag(B).copy()
                E = B - np.diag(d)
                G = d[None, :] - d[:, None]
                with np.errstate(divide="ignore", invalid="ignore"):
                    K0 = np.abs(np.where(np.eye(n, dtype=bool), 0.0, E / np.where(G == 0, 1e-300, G)))
                bad = ((np.abs(E) > theta * np.abs(G)) | (K0 > opt["kcap"]))
                np.fill_diagonal(bad, False)
            else:
                bad = ((np.abs(E) > theta * np.abs(G)) | (K0 > opt["kcap"]))
                np.fill_diagonal(bad, False)
            ok = ~bad
            np.fill_diagonal(ok, False)
            lam = d / (1.0 - np.diag(R))
            Et = np.where(ok, (B + lam[None, :] * R) / np.where(ok, lam[None, :] - lam[:, None], 1.0), 0.5 * R)
            Eo = Et - np.diag(np.diag(Et))
            unres = np.linalg.norm(np.where(bad, E, 0.0)) / fro
            pred = opt["pf"] * (np.linalg.norm(E) / fro) * np.linalg.norm(Eo) + unres
            Vn = V + V @ Et
            if rnd == 0 and opt["blind"] and pred <= self.tol_acc and np.abs(Eo).max() <= 0.5:
                self.hist[11] += 1
                self.cur = max(self.cur, 1)
                dn = d * (1.0 + 2.0 * np.diag(Et)) + np.sum(Eo * Eo * (2.0 * d[None, :] - d[:, None]), axis=0)   # kernel's second-order diagonal
                nrm2 = np.sum(Vn * Vn, axis=0)
                self.V[k] = Vn
                W = (Vn * np.maximum(dn / nrm2, 0.0)) @ Vn.T
                if self.calls % 53 == 0:
                    w, Q = np.linalg.eigh(A)
                    self.maxerr = max(self.maxerr, np.linalg.norm(W - (Q * np.maximum(w, 0)) @ Q.T) / fro / max(self.tol_acc, 1e-300))
                return W
            V = Vn
        return exact()


def run(iters, eps=1e-6):
    S = oadmm.AdmmState(P, 0.1, 1.6)
    rf = Refiner(S.nk)

    def proj(nu):
        w = np.empty_like(nu)
        w[:S.ng] = np.maximum(nu[:S.ng], 0.0)
        rf.cur = 0
        for k, n in enumerate(S.nk):
            A = nu[S.offs[k]:S.offs[k + 1]].reshape(n, n)
            w[S.offs[k]:S.offs[k + 1]] = rf.project(k, 0.5 * (A + A.T)).ravel()
        rf.lh[rf.cur] += 1
        return w
    S.proj = proj
    next_adapt = 50
    t0 = time.time()
    obj = 0.0
    it0 = 1
    if opt["load"]:
        st = pickle.load(open(opt["load"], "rb"))
        S.nu, S.sigma, it0, next_adapt = st["nu"], st["sigma"], st["it"] + 1, st["next_adapt"]
    every = 500 if iters - it0 <= 6000 else 1000
    for it in range(it0, iters + 1):
        nu_prev = S.nu
        w, x, res, Kxq = S.step()
        if it % 50 == 0:
            y = S.sigma * (nu_prev - w)
            Kty = S.Kt(y)
            rp = np.linalg.norm(res) / max(np.linalg.norm(Kxq), np.linalg.norm(w), 1e-300)
            rd = np.linalg.norm(Kty - P.z0) / max(np.linalg.norm(Kty), np.linalg.norm(P.z0), 1e-300)
            obj = -(P.c @ y[:S.ng]) / (P.zscale * P.cscale)
            rf.tol = min(1e-4, max(1e-9, 0.01 * max(rp, rd)))
            rf.tol_acc = min(1e-3, max(1e-9, acc_factor * max(rp, rd)))
            if it % every == 0:
                h, lh = rf.hist, rf.lh
                tb, tl = max(h.sum(), 1), max(lh.sum(), 1)
                print(f"  it {it:6d} pres {rp:.2e} dres {rd:.2e} obj {obj:.8g} | blocks %: r0 {100*h[0]/tb:.0f} blind {100*h[11]/tb:.0f} r1 {100*h[1]/tb:.0f} r2 {100*h[2]/tb:.0f} r3 {100*h[3]/tb:.0f} J {100*h[10]/tb:.1f}; "
                      f"launch max %: <=blind {100*(lh[0]+lh[1])/tl:.0f} r2 {100*lh[2]/tl:.0f} r3 {100*lh[3]/tl:.0f} J {100*lh[10]/tl:.0f}; Givens/call {rf.giv/max(rf.calls,1):.2f} cluster med {np.median(rf.csize) if rf.csize else 0:.0f} p90 {np.quantile(rf.csize, .9) if rf.csize else 0:.0f} max {max(rf.csize) if rf.csize else 0} err/tol_acc {rf.maxerr:.2f} t {time.time()-t0:.0f}s", flush=True)
                rf.reset()
            if rp <= eps and rd <= eps:
                return it, obj
            if it >= next_adapt:
                next_adapt = max(it + 100, it * 3 // 2)
                ratio = np.sqrt(max(rp, 1e-300) / max(rd, 1e-300))
                if ratio > 1.5 or ratio < 0.67:
                    S.set_sigma(S.sigma * min(max(ratio, 0.2), 5.0))
    return iters, obj


print(f"{name} beta={beta} {mode}: blocks {[len(c) for c in P.pat.cliques]} acc {acc_factor} {opt}")
print(run(iters))
