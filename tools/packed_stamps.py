"""(diagnostic) cycle stamps of the packed variant (blocks 129 .. 160) of the projection kernel in its warm form, from the -DNNSDP_STAMPS
build of the library (see tools/refine_stamps.py).  usage: python tools/packed_stamps.py [n=151] [blocks=4] [refine=0]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, nnsdp_amd as na
from nnsdp_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libnnsdp_hip_stamps.so")
_lib._lib = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 151
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
refine = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(0)
def sym2(n):
    spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
base = [sym2(n) for _ in range(nb)]
for eta, tol in ((1e-4, 1e-6), (1e-6, 1e-8)):
    mats = []
    for A, _ in base:
        D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
        mats.append(A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D)
    for rep in range(2):
        print(f"--- move {eta:g} tol {tol:g} launch {rep}", flush=True)
        W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=refine)
        print(f"    kernel {1e3 * ms:.1f} us outcome {oc}", flush=True)
