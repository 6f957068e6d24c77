#!/usr/bin/env python3
"""Wall-clock to certificate on the published rows (dump/scale) beside the reference's own published MOSEK totals
(unknown hardware - context, not a same-node comparison).  findEllipsoid through the product path, Double decomposition,
two stopping rules:
  residual_1e-6       ADMM residuals <= 1e-6 relative (the tolerance the reference ASKED MOSEK for, scale.jl:40-42)
  certified_gap_1e-3  stop once the polished, exactly feasible objective is within 1e-3 of the ADMM primal/dual estimates -
                      the accuracy the published rows actually HAVE (they sit 4e-4 .. 1.3e-2 above the optimum)
Every returned certificate is checked (gamma >= 0, eigmax(Z) in the reference's coordinates).  CSV -> gpurun_out/time_to_cert_sweep.csv"""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na

cases = [("W10-D10", 0), ("W10-D10", 3), ("W10-D10", 7), ("W10-D20", 0), ("W10-D30", 0), ("W10-D50", 0), ("W10-D60", 0),
         ("W20-D10", 0), ("W20-D10", 7), ("W20-D20", 0), ("W20-D30", 0), ("W20-D50", 0), ("W20-D100", 0), ("W20-D100", 7)]
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "time_to_cert_sweep.csv")
rows_pub = helpers.dump_rows()
hdr = ["net", "beta", "rule", "iters", "wall_s", "setup_s", "solve_s", "rho", "status", "eigmax_Z", "gamma_min", "rel_to_pub_min",
       "mosek_best_total_s", "mosek_best_method", "mosek_chordal2_total_s", "ratio_wall_over_mosek_best"]
res = []
# warm the library (first call pays module load / rocSOLVER initialisation)
d0 = np.load(os.path.join(helpers.GOLDEN, "nets", "scale-I2-O2-W10-D10.npz"))
for name, beta in cases:
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    q, P, yc = na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], beta)
    pub = [(r["method"], float(r["total_secs"]), float(r["obj_val"])) for r in rows_pub
           if r["net"] == name and int(r["beta"]) == beta and r["term_status"] == "OPTIMAL"]
    best = min(pub, key=lambda t: t[1])
    c2 = [t for t in pub if t[0] == "chordalsdp2"]
    pmin = min(t[2] for t in pub)
    for rule, kw in (("certified_gap_1e-3", dict(cert_tol=1e-3)), ("residual_1e-6", dict())):
        o = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=800000, max_time=200, eps_rel=1e-6, **kw)
        t = time.time()
        s = na.runQuery(q, o)
        wall = time.time() - t
        gmin = min(float(np.min(s.values[k])) for k in ("γin", "γout", "γac1", "γac2"))
        row = [name, beta, rule, s.summary["iters"], round(wall, 3), round(s.setup_time, 3), round(s.solve_time, 3), s.objective_value,
               s.termination_status, s.summary["lambda_max"], gmin, (pmin - s.objective_value) / pmin, best[1], best[0],
               c2[0][1] if c2 else "", wall / best[1]]
        res.append(row)
        print(dict(zip(hdr, row)), flush=True)
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(hdr)
            w.writerows(res)
for rule in ("certified_gap_1e-3", "residual_1e-6"):
    rr = [r for r in res if r[2] == rule]
    slower2 = [f"{r[0]} b{r[1]} ({r[15]:.2f}x)" for r in rr if r[15] > 2.0]
    print(f"{rule}: {len(rr)} rows, wall / published MOSEK best total: median {np.median([r[15] for r in rr]):.3f}, max {max(r[15] for r in rr):.2f}; "
          f"slower than 2x MOSEK: {slower2 or 'none'}", flush=True)
