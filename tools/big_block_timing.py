"""us per projection of a 151-wide PSD block (the reference's cliques of width-50 networks, chordal_cliques.jl:33-36) through the
library path (rocSOLVER dsyevd + rocBLAS dgemm, one block at a time), beside the LDS-resident kernel on the 101-wide blocks the
PathDecomp extension gives the same networks.  usage: python tools/big_block_timing.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, nnsdp_amd as na
rng = np.random.default_rng(0)
def sym(n):
    A = rng.standard_normal((n, n)); return 0.5 * (A + A.T)
for ns in ([151], [151] * 4, [106, 151, 151, 151, 151], [101] * 5, [128] * 4, [160] * 4):
    mats = [sym(n) for n in ns]
    best = 1e9
    for rep in range(4):
        P, ev, ms = na.project_psd_batched(mats)
        best = min(best, ms)
    err = max(np.abs(Pk - (lambda w, Q: (Q * np.maximum(w, 0)) @ Q.T)(*np.linalg.eigh(A))).max() for Pk, A in zip(P, mats))
    print(f"blocks {ns}: {1e3 * best:9.1f} us per call, {1e3 * best / len(ns):8.1f} us per block (cold start), max |P - LAPACK| {err:.1e}", flush=True)

# warm form (as inside a solve: the basis of the previous projection, a slowly moving matrix) through the packed LDS variant
def sym2(n):
    spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
for ns in ([151] * 4, [106, 151, 151, 151, 151], [160] * 4, [129] * 4):
    base = [sym2(n) for n in ns]
    for eta, tol in ((1e-4, 1e-6), (1e-6, 1e-8)):
        mats = []
        for A, _ in base:
            D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
            mats.append(A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D)
        best = 1e9
        for rep in range(3):
            W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=False)
            best = min(best, ms)
        err = max(np.linalg.norm(Wk - (lambda w, Q: (Q * np.maximum(w, 0)) @ Q.T)(*np.linalg.eigh(A))) / np.linalg.norm(A) for Wk, A in zip(W, mats))
        print(f"warm, blocks {ns}, move {eta:g}, tol {tol:g}: {1e3 * best:9.1f} us per launch (all blocks side by side), |W - LAPACK| / |A| {err:.1e}", flush=True)
        best = 1e9
        for rep in range(3):
            W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=True)
            best = min(best, ms)
        err = max(np.linalg.norm(Wk - (lambda w, Q: (Q * np.maximum(w, 0)) @ Q.T)(*np.linalg.eigh(A))) / np.linalg.norm(A) for Wk, A in zip(W, mats))
        print(f"      with the refinement stage (outcome {oc}, Gram product taken): {1e3 * best:9.1f} us per launch, |W - LAPACK| / |A| {err:.1e}", flush=True)
