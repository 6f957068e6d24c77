"""kernel time of the projection launch by path: synthetic warm launches (every block on one path) and solver windows at several depths
usage: python tools/refine_timing.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, helpers, nnsdp_amd as na

rng = np.random.default_rng(0)
def sym(n):
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spec) @ Q.T, Q
def perturb(A, eta):
    D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D

for ns in ([85] * 19, [68] * 20, [57, 85, 82, 79, 77, 73, 76, 77, 73, 64, 65, 76, 79, 70, 65, 68, 71, 72, 68]):
    base = [sym(n) for n in ns]
    for name, eta, tol, refine in (("one refinement step", 1e-6, 3e-7, True), ("converged as given", 0.0, 1e-7, True), ("sweeps only (1 sweep)", 1e-6, 3e-7, False),
                                   ("refine rejected -> sweeps", 3e-2, 1e-4, True), ("sweeps only, same input", 3e-2, 1e-4, False)):
        mats = [perturb(A, eta) if eta > 0 else A for A, _ in base]
        best = 1e9
        for rep in range(5):
            W, V, oc, ms = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=refine)
            best = min(best, ms)
        print(f"blocks {len(ns)} x n<={max(ns)}: {name:28s} outcome {oc} kernel {1e3 * best:7.1f} us", flush=True)

q = helpers.product_query(helpers.load_problem("W40-D20", 0))
for mode, nm in ((na.SingleDecomp(), "single"), (na.DoubleDecomp(), "double")):
    for rf in (2, 1, 0):
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=10 ** 9, proj_refine=rf))
        done = 0
        for upto in (2000, 5000, 10000, 15000, 30000):
            if nm == "double" and upto > 15000:
                break
            s.advance(upto - done); done = upto
            s.iterate(50, time_eig=True)
            t0 = time.perf_counter()
            ms = s.iterate(400, time_eig=True)
            dt = time.perf_counter() - t0
            done += 450
            print(f"{nm} refine {int(rf)} after {upto:6d} iterations: k_proj {1e3 * ms / 400:6.1f} us/launch, step {1e6 * dt / 400:6.1f} us (eager, events)", flush=True)
        s.close()
