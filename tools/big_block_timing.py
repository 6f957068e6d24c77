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
