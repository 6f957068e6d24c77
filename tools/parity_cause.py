#!/usr/bin/env python3
"""Where does the one-signed gap to the published dump/scale objectives come from?  

For every (net, beta) with three OPTIMAL published rows this script solves findEllipsoid through the product path twice
in one batch: with the CROWN-sliced intervals as restated, and with every post-activation interval widened by 0.5 % of
its width about its centre.  From the two it derives
  gap      = (published_min - rho) / published_min            (> 0: we are below every published value)
  spread   = (published_max - published_min) / published_min  (disagreement among the reference's own three methods,
                                                               which share one optimum in exact arithmetic)
  w_star   = the uniform relative widening of the intervals that would move rho onto published_min
A rule difference in the interval pre-processing would show as ONE common w_star; solver (MOSEK) termination accuracy
shows as w_star scattered over orders of magnitude and a gap that tracks the reference's own spread.
Writes a CSV (default gpurun_out/parity_cause.csv) and prints the rank correlation of gap and spread."""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na

W = 0.005
nets = ["W10-D10", "W10-D20", "W10-D30", "W10-D50", "W10-D60", "W10-D70", "W10-D80",
        "W20-D10", "W20-D20", "W20-D30", "W20-D40", "W20-D50", "W20-D60", "W20-D70", "W20-D80"]
extra = [("W10-D10", 3), ("W10-D10", 7), ("W20-D10", 3), ("W20-D10", 7), ("W10-D30", 7), ("W20-D30", 7)]
cases = [(n, 0) for n in nets] + extra
if len(sys.argv) > 2:
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[2].split(",")]
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_cause.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
hdr = ["net", "beta", "rho_certified", "rho_admm", "rho_admm_widened", "status", "iters", "secs", "pub_deepsdp", "pub_chordal", "pub_chordal2",
       "gap_to_pub_min", "ref_spread", "drho_per_unit_w", "w_star"]
rows = []


def pub_by_method(name, beta):
    d = {}
    for r in helpers.dump_rows():
        if r["net"] == name and int(r["beta"]) == beta and r["term_status"] == "OPTIMAL":
            d[r["method"]] = float(r["obj_val"])
    return d


for name, beta in cases:
    pub = pub_by_method(name, beta)
    if len(pub) < 3:
        print("skip (not three OPTIMAL rows)", name, beta, flush=True)
        continue
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    q0, P, yc = na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], beta)
    qb, qs = q0.qc_activs
    c, r = 0.5 * (qb.acymin + qb.acymax), 0.5 * (qb.acymax - qb.acymin)
    qw = na.ReachQuery(ffnet=net, qc_input=q0.qc_input, qc_reach=q0.qc_reach,
                       qc_activs=[na.QcActivBounded(acymin=c - (1 + W) * r, acymax=c + (1 + W) * r), qs])
    opts = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=600000, max_time=150, eps_rel=1e-6)
    t = time.time()
    s0, s1 = na.runQueries([q0, qw], opts)
    secs = time.time() - t
    r0, r1 = s0.summary["objective_admm"], s1.summary["objective_admm"]
    pmin, pmax = min(pub.values()), max(pub.values())
    sens = (r1 - r0) / W
    row = [name, beta, s0.objective_value, r0, r1, s0.termination_status, s0.summary["iters"], round(secs, 1),
           pub["deepsdp"], pub["chordalsdp"], pub["chordalsdp2"], (pmin - s0.objective_value) / pmin, (pmax - pmin) / pmin,
           sens, (pmin - r0) / sens if sens != 0 else float("nan")]
    rows.append(row)
    print(dict(zip(hdr, row)), flush=True)
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(hdr)
        w.writerows(rows)

if len(rows) >= 3:
    g = np.array([r[11] for r in rows]); s = np.array([r[12] for r in rows]); ws = np.array([r[14] for r in rows])
    rk = lambda v: np.argsort(np.argsort(v)).astype(float)
    rho_s = np.corrcoef(rk(g), rk(s))[0, 1]
    print(f"rows {len(rows)}: gap in [{g.min():.2e}, {g.max():.2e}], all positive: {bool((g > 0).all())}; "
          f"Spearman(gap, ref spread) = {rho_s:.3f}; Pearson(log gap, log spread) = {np.corrcoef(np.log(np.abs(g)), np.log(s))[0, 1]:.3f}; "
          f"w_star in [{ws.min():.2e}, {ws.max():.2e}] (max/min = {ws.max() / max(ws.min(), 1e-300):.1f})", flush=True)
