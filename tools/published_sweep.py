#!/usr/bin/env python3
"""Every (network, beta) pair of the reference's published scale experiment that has OPTIMAL rows (dump/scale/*.csv ->
tests/golden/dump_scale.csv: 17 bench/rand networks x beta = 0..7 = 136 pairs, three published objectives each - DeepSDP, Chordal,
Chordal-2, experiments/scale.jl:60-82) through the product path only: native CROWN intervals, sampled ellipsoid, the batch handle
(the eight betas of a network side by side), Double decomposition, certificate polish.  Two stopping rules: the certified gap
(cert_tol = 1e-3: the accuracy the published values actually have) for every pair, and residuals 1e-6 (what the reference asked MOSEK
for) for the pairs that reach it inside --tight-seconds per network.

Writes a CSV (one row per pair and rule): certified rho, ADMM iterate, eigmax(Z) in the reference's coordinates, min gamma, the
largest |invP y - yc|^2 over 20 000 sampled forward passes, the three published values' min / nearest / max, the signed relative
distances, status, iterations, wall time of the batch.
usage: python tools/published_sweep.py [out.csv] [--tight-seconds S] [--nets W10-D10,W20-D50,...]"""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F

args = sys.argv[1:]
out = os.path.join(ROOT, "gpurun_out", "published_sweep.csv")
tight_s, only = 0.0, None
while args:
    a = args.pop(0)
    if a == "--tight-seconds": tight_s = float(args.pop(0))
    elif a == "--nets": only = args.pop(0).split(",")
    else: out = a
nets = sorted((f[len("scale-I2-O2-"):-4] for f in os.listdir(os.path.join(helpers.GOLDEN, "nets"))),
              key=lambda s: (int(s.split("-")[0][1:]), int(s.split("-")[1][1:])))
nets = [n for n in nets if all(len(helpers.published_rho(n, b)) == 3 for b in range(8)) and (only is None or n in only)]
rows = []
t_all = time.time()
for name in nets:
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    qs = [na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], b)[0] for b in range(8)]
    X = 0.5 + np.random.default_rng(7).random((2, 20000))
    Y = F.evalFeedFwdNet(net, X)
    out_rel_std = float(np.max(Y.std(axis=1) / np.maximum(np.abs(Y.mean(axis=1)), 1e-300)))      # how far the output moves over the box at all
    for rule, opts in (("certified_gap_1e-3", na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=2000000, max_time=600, eps_rel=1e-6, cert_tol=1e-3)),
                       ("residual_1e-6", na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=2000000, max_time=tight_s, eps_rel=1e-6))):
        if rule == "residual_1e-6" and tight_s <= 0:
            continue
        t = time.time()
        sols = na.runQueries(qs, opts)
        wall = time.time() - t
        for b, (q, s) in enumerate(zip(qs, sols)):
            pub = helpers.published_rho(name, b)
            rho = s.objective_value
            near = min(pub, key=lambda p: abs(p - rho))
            samp = float(np.sum((q.qc_reach.invP @ Y - q.qc_reach.yc[:, None]) ** 2, axis=0).max())
            gmin = min(float(np.min(s.values[k])) for k in ("γin", "γout", "γac1", "γac2"))
            rows.append([name, b, rule, s.termination_status, s.summary["iters"], f"{rho:.12g}", f"{s.summary['objective_admm']:.12g}", f"{s.summary['lambda_max']:.3e}",
                         f"{gmin:.3e}", f"{samp:.12g}", f"{min(pub):.12g}", f"{near:.12g}", f"{max(pub):.12g}", f"{(rho - min(pub)) / min(pub):.4e}",
                         f"{(rho - near) / near:.4e}", f"{(max(pub) - min(pub)) / min(pub):.4e}", s.summary["n_cliques"], s.summary["max_clique"], f"{wall:.2f}", f"{out_rel_std:.3e}"])
        done = [r for r in rows if r[0] == name and r[2] == rule]
        print(f"{name} {rule}: batch of 8 in {wall:.1f} s; rel to min published " + " ".join(r[13] for r in done) + " status " + ",".join(sorted(set(r[3] for r in done))), flush=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["net", "beta", "rule", "status", "iters", "rho_certified", "rho_admm_iterate", "eigmax_Z", "gamma_min", "sampled_max", "published_min", "published_nearest",
                "published_max", "rel_to_published_min", "rel_to_published_nearest", "published_spread", "blocks", "max_block", "batch_wall_s", "output_rel_std"])
    w.writerows(rows)
# the two-sided distance histogram of the certified-gap rows
rel = np.array([float(r[14]) for r in rows if r[2] == "certified_gap_1e-3"])
if len(rel):
    edges = [-1, -1e-2, -3e-3, -1e-3, -3e-4, 0, 3e-4, 1e-3, 3e-3, 1]
    hist, _ = np.histogram(rel, bins=edges)
    print("signed relative distance to the NEAREST published value, certified-gap rule (ours - published) / published:")
    for lo, hi, c in zip(edges[:-1], edges[1:], hist):
        print(f"  [{lo:+.0e}, {hi:+.0e}): {c}")
    print(f"pairs {len(rel)}, inside 1e-3: {int(np.sum(np.abs(rel) <= 1e-3))}, above every published value by more than 1e-3: "
          f"{sum(1 for r in rows if r[2] == 'certified_gap_1e-3' and float(r[5]) > float(r[12]) * (1 + 1e-3))}; total {time.time() - t_all:.0f} s")
