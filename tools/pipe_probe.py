"""(diagnostic) the tile-parallel refinement pipeline (refine = 4 of the warm test entry) beside the one-CU stage (refine = 1) and LAPACK:
one step, converged as given, a rejected block (falls back to the kernel behind the pipeline), a carried state, launch times.
usage: python tools/pipe_probe.py [blocks=19]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, nnsdp_amd as na

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 19
rng = np.random.default_rng(0)

def sym(n, frac_neg=1 / 3):
    m = max(1, int(n * frac_neg))
    spec = np.concatenate([np.linspace(0.2, 2.0, n - m), -np.linspace(0.1, 1.5, m)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q

def perturb(A, eta):
    D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D

def proj(A):
    lam, Q = np.linalg.eigh(0.5 * (A + A.T))
    return (Q * np.maximum(lam, 0)) @ Q.T

bad = 0
for n in (27, 41, 57, 68, 85, 96, 101, 121, 151, 160):
    for frac in (1 / 3, 0.7):
        base = [sym(n, frac) for _ in range(nb if n <= 96 else 5)]
        for name, eta, tol in (("step", 1e-6, 3e-7), ("converged", 0.0, 1e-7), ("far", 3e-2, 1e-6)):
            mats = [perturb(A, eta) if eta > 0 else A for A, _ in base]
            Wx = [proj(A) for A in mats]
            res = {}
            for refine in (1, 4):
                best = 1e9
                for rep in range(3):
                    W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=refine)
                    best = min(best, ms)
                err = max(np.linalg.norm(W[i] - Wx[i]) / np.linalg.norm(mats[i]) for i in range(len(mats)))
                orth = max(np.linalg.norm(V[i].T @ V[i] - np.eye(n)) for i in range(len(mats)))
                asym = max(np.abs(W[i] - W[i].T).max() for i in range(len(mats)))
                res[refine] = (err, orth, oc, best, asym)
            e1, e4 = res[1], res[4]
            lim = 30 * tol if name != "far" else 10 * tol
            flag = "" if e4[0] <= lim and e4[1] <= 1e-3 else "   <-- BAD"
            bad += flag != ""
            print(f"n={n:3d} neg={frac:.2f} {name:9s} one-CU: err {e1[0]:.2e} orth {e1[1]:.1e} oc {e1[2]} {1e3 * e1[3]:6.1f} us | "
                  f"pipe: err {e4[0]:.2e} orth {e4[1]:.1e} oc {e4[2]} {1e3 * e4[3]:6.1f} us asym {e4[4]:.1e}{flag}", flush=True)

# carried state: 30 small moves from the basis the previous call returned
for n in (68, 85, 151):
    A, Q = sym(n)
    mats, bases = [A] * 3, [Q] * 3
    st = np.zeros(12, dtype=np.int32)
    worst, steps = 0.0, 0
    for it in range(30):
        mats = [perturb(M, 3e-5) for M in mats]
        W, bases, oc, ms = na.project_psd_warm(mats, bases, 1e-5, refine=4, state=st)
        steps += oc[1]
        worst = max(worst, max(np.linalg.norm(W[i] - proj(mats[i])) / np.linalg.norm(mats[i]) for i in range(3)))
    ok = worst <= 30 * 1e-5
    bad += not ok
    print(f"n={n} carried over 30 moves: worst err {worst:.2e} steps taken {steps}/90 {'ok' if ok else '<-- BAD'}", flush=True)
print("BAD" if bad else "ALL OK", bad)
