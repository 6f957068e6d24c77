#!/usr/bin/env python3
"""Quick end-to-end sanity run on a GPU box (not a test): projection, assembly, small solves."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na

rng = np.random.default_rng(0)
mats = []
for n in (1, 2, 5, 31, 64, 85, 97, 121, 128):
    A = rng.standard_normal((n, n)); mats.append(0.5 * (A + A.T))
res, evs, ms = na.project_psd_batched(mats)
for A, P, ev in zip(mats, res, evs):
    w, Q = np.linalg.eigh(A); ref = (Q * np.maximum(w, 0)) @ Q.T
    print(f"n={A.shape[0]:4d} proj err {np.abs(P-ref).max():.2e} eig err {np.abs(np.sort(ev)-w).max():.2e}")
print("proj kernel ms", ms)
big = [mats[5]] * 256
t = time.time(); _, _, ms = na.project_psd_batched(big); print("256 x n=85 cold: kernel ms", ms, "wall", time.time() - t)
big = [mats[7]] * 256
t = time.time(); _, _, ms = na.project_psd_batched(big); print("256 x n=121 cold: kernel ms", ms, "wall", time.time() - t)

d = helpers.load_problem("W10-D5", 3); g = helpers.load_golden("W10-D5", 3)
q = helpers.product_query(d)
for i in range(2):
    Z = na.makeZ(q, g["gammas"][i]); print("makeZ err", np.abs(Z - g["Zs"][i]).max())
print("adjoint err", np.abs(na.adjoint(q, g["Xs"][0]) - g["adj"][0]).max())
for name, beta in (("W10-D5", 0), ("W10-D10", 0), ("W10-D20", 0), ("W20-D10", 0)):
    d = helpers.load_problem(name, beta)
    q = helpers.product_query(d)
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(max_iters=int(os.environ.get("ITERS", 6000)), verbose=(name=="W20-D10")))
    print(name, beta, "rho", s.objective_value, s.termination_status, "iters", s.summary["iters"], "pres %.2e dres %.2e lmax %.2e" % (s.summary["pres"], s.summary["dres"], s.summary["lambda_max"]),
          "setup %.2f solve %.2f total %.2f wall %.2f" % (s.setup_time, s.solve_time, s.total_time, time.time() - t), "published", helpers.published_rho(name, beta))
