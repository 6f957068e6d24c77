"""(diagnostic) the projection kernel with two workgroups per block for the warm-start congruence (ProjArgs::split) against the
one-workgroup form through the warm test entry: identical bits expected (same instruction sequence per tile), launch times.
usage: python tools/split_probe.py [blocks=19]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, nnsdp_amd as na
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 19
rng = np.random.default_rng(0)
def sym(n):
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
def perturb(A, eta):
    D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D
bad = 0
for n in (41, 48, 57, 64, 68, 80, 85, 91, 96):
    base = [sym(n) for _ in range(nb)]
    for name, eta, tol, refine in (("step", 1e-6, 3e-7, True), ("converged", 0.0, 1e-7, True), ("far (sweeps)", 3e-2, 1e-6, True), ("sweeps only", 1e-6, 3e-7, False)):
        mats = [perturb(A, eta) if eta > 0 else A for A, _ in base]
        out = {}
        for split in ("0", "1"):
            os.environ["NNSDP_SPLIT_WARM"] = split
            best = 1e9
            for rep in range(3):
                W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=refine)
                best = min(best, ms)
            out[split] = (W, V, oc, best)
        same = all(np.array_equal(a, b) for a, b in zip(out["0"][0], out["1"][0])) and all(np.array_equal(a, b) for a, b in zip(out["0"][1], out["1"][1]))
        bad += not same
        print(f"n={n:3d} {name:13s}: one workgroup {1e3 * out['0'][3]:7.1f} us, two {1e3 * out['1'][3]:7.1f} us, outcome {out['1'][2]}, identical bits: {same}", flush=True)
print("ALL IDENTICAL" if bad == 0 else f"{bad} CASES DIFFER")
