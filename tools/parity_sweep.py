#!/usr/bin/env python3
"""End-to-end parity sweep on the GPU against the reference's published results (dump/scale/*.csv ->
tests/golden/dump_scale.csv): NnSdp.findEllipsoid on [0.5,1.5]^2 for the bench/rand nets that have OPTIMAL
published rows, through the product path only (nnsdp_amd front-end + libnnsdp_hip.so).
Writes a CSV: net,beta,rho,rho_admm,lambda_max,status,iters,total_s,published_min,published_max,rel_to_nearest."""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na

cases = [("W10-D10", 0), ("W10-D10", 3), ("W10-D10", 7), ("W10-D20", 0), ("W10-D30", 0), ("W10-D50", 0), ("W10-D60", 0),
         ("W20-D10", 0), ("W20-D10", 7), ("W20-D20", 0), ("W20-D30", 0), ("W20-D50", 0)]
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_sweep.csv")
rows = []
for name, beta in cases:
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    t = time.time()
    _, _, s = na.findEllipsoid(net, [0.5, 0.5], [1.5, 1.5], beta,
                               na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=400000, max_time=90, eps_rel=1e-6))
    pub = helpers.published_rho(name, beta)
    rel = min(abs(s.objective_value - p) / abs(p) for p in pub) if pub else float("nan")
    rows.append([name, beta, s.objective_value, s.summary["objective_admm"], s.summary["lambda_max"], s.termination_status, s.summary["iters"],
                 round(time.time() - t, 2), min(pub) if pub else "", max(pub) if pub else "", rel])
    print(rows[-1], flush=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["net", "beta", "rho_certified", "rho_admm_iterate", "eigmax_Z", "status", "iters", "total_s", "published_min", "published_max", "rel_diff_to_nearest_published"])
    w.writerows(rows)
