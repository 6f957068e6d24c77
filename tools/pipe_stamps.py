"""(diagnostic) wall-clock stamps inside the five kernels of the tile-parallel refinement pipeline, from a -DNNSDP_STAMPS build of the
library that is never shipped:
    hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -pthread -DNNSDP_STAMPS nn-sdp_amd/csrc/api.hip -o nn-sdp_amd/nnsdp_amd/libnnsdp_hip_stamps.so -lrocsolver -lrocblas -ldl
usage: python tools/pipe_stamps.py [n=85] [blocks=19]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, nnsdp_amd as na
from nnsdp_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libnnsdp_hip_stamps.so")
_lib._lib = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 85
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 19
rng = np.random.default_rng(0)
def sym(n):
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
def perturb(A, eta):
    D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D
base = [sym(n) for _ in range(nb)]
mats = [perturb(A, 1e-6) for A, _ in base]
for rep in range(3):
    print(f"--- launch {rep}", flush=True)
    W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], 3e-7, refine=4)
    print(f"    outcome {oc} pipeline + kernel {1e3 * ms:.1f} us (eager launches)", flush=True)
