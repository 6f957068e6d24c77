import sys, os
sys.path.insert(0, "/root/repo/nn-sdp_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, nnsdp_amd as na
from oracle import admm as oadmm
rng = np.random.default_rng(3)
def sym(n):
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
def perturb(A, eta):
    D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D
for n in (57, 68, 73, 80, 81, 85, 90, 96):
    for eta, tol in ((3e-4, 3e-6), (1e-3, 1e-5), (3e-3, 1e-5), (1e-2, 1e-4)):
        A0, Q0 = sym(n)
        A1 = perturb(A0, eta)
        for rf in (1, 2):
            W, V, oc, ms = na.project_psd_warm([A1], [Q0], tol, refine=rf)
            err = np.linalg.norm(W[0] - oadmm.project_psd(A1)) / np.linalg.norm(A1)
            print(f"n {n} eta {eta:g} tol {tol:g} refine {rf}: outcome {oc} err/tol {err / tol:9.3f} orth {np.linalg.norm(V[0].T @ V[0] - np.eye(n)):.2e} finite {np.isfinite(W[0]).all()} {1e3 * ms:.0f} us", flush=True)
