#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists; the GPU box never sees the reference).

Fixtures are DATA only:
  nets/<name>.npz            weights of the reference's bench/rand/<name>.nnet (binary copy)
  dump_scale.csv             the 480 published rows of dump/scale/*.csv (the only golden values)
  problem_<name>_b<beta>.npz inputs of the hot path for findEllipsoid on the box [0.5,1.5]^2:
                             network, CROWN-sliced intervals, sector bounds, sampled ellipsoid
  golden_<name>_b<beta>.npz  oracle outputs on those inputs: clique lists, Z(gamma) probes,
                             adjoint probes
Usage: python tests/golden/make_fixtures.py [--ref /root/reference]
(test infrastructure: the only users of oracle/ are tests/, smoke() and bench.py's cpu_baseline leg)
"""
import argparse
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nnet_io, qc, operator as op  # noqa: E402

NETS = ["W10-D5", "W10-D10", "W10-D20", "W10-D30", "W10-D50", "W10-D60", "W10-D70", "W10-D80",
        "W20-D10", "W20-D20", "W20-D30", "W20-D40", "W20-D50", "W20-D60", "W20-D70", "W20-D80", "W20-D90", "W20-D100",
        "W40-D20", "W40-D40"]
PROBLEMS = [("W10-D5", 0), ("W10-D5", 3), ("W10-D10", 0), ("W10-D10", 2), ("W10-D20", 0), ("W20-D10", 0),
            ("W40-D20", 0), ("W40-D20", 2), ("W40-D40", 0)]
GOLDEN = [("W10-D5", 0), ("W10-D5", 3), ("W10-D10", 2)]
X1MIN, X1MAX = [0.5, 0.5], [1.5, 1.5]          # experiments/scale.jl:26-27


def save_problem(path, q: qc.Query):
    net = q.net
    arrs = {"xdims": np.asarray(net.xdims, dtype=np.int32), "beta": np.int32(q.beta),
            "x1min": q.qc_input.x1min, "x1max": q.qc_input.x1max,
            "acymin": q.qc_bounded.acymin, "acymax": q.qc_bounded.acymax,
            "smin": q.qc_sector.smin, "smax": q.qc_sector.smax,
            "invP": q.qc_out.invP, "yc": q.qc_out.yc}
    for k, M in enumerate(net.Ms):
        arrs[f"M{k}"] = M
    np.savez_compressed(path, **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--missing-nets-only", action="store_true", help="only add nets/<name>.npz files that do not exist yet")
    args = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(os.path.join(gold, "nets"), exist_ok=True)
    nets = {}
    for name in NETS:
        net = nnet_io.read_nnet(os.path.join(args.ref, "bench", "rand", f"scale-I2-O2-{name}.nnet"))
        dst = os.path.join(gold, "nets", f"scale-I2-O2-{name}.npz")
        if not (args.missing_nets_only and os.path.exists(dst)):
            nnet_io.save_npz(net, dst)
        nets[name] = net
    if args.missing_nets_only:
        return
    # published results
    rows = []
    for f in sorted(glob.glob(os.path.join(args.ref, "dump", "scale", "*.csv"))):
        base = os.path.basename(f)[:-len(".nnet.csv")]
        method, netname = base.split("-scale-I2-O2-")
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append([method, netname, r["beta"], r["setup_secs"], r["solve_secs"], r["total_secs"],
                             r["obj_val"], r["term_status"], r["eigmax"]])
    with open(os.path.join(gold, "dump_scale.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["method", "net", "beta", "setup_secs", "solve_secs", "total_secs", "obj_val", "term_status", "eigmax"])
        wr.writerows(rows)
    print("dump rows", len(rows))
    for name, beta in PROBLEMS:
        q = qc.make_reach_ellipsoid_query(nets[name], X1MIN, X1MAX, beta, seed=1234)
        save_problem(os.path.join(gold, f"problem_{name}_b{beta}.npz"), q)
        print("problem", name, beta, "ngamma", q.ngamma)
        if (name, beta) in GOLDEN:
            rng = np.random.default_rng(7)
            gammas = rng.random((3, q.ngamma))
            R = qc.make_R(q.net)
            Zs = np.stack([qc.assemble_Z_literal(q, g, R) for g in gammas])
            Xs = rng.standard_normal((2, q.net.Zdim, q.net.Zdim))
            Xs = 0.5 * (Xs + np.transpose(Xs, (0, 2, 1)))
            Z0 = qc.assemble_Z_literal(q, np.zeros(q.ngamma), R)
            adj = np.zeros((2, q.ngamma))
            for i in range(q.ngamma):
                e = np.zeros(q.ngamma)
                e[i] = 1.0
                Gi = qc.assemble_Z_literal(q, e, R) - Z0
                for j in range(2):
                    adj[j, i] = np.sum(Gi * Xs[j])
            out = {"gammas": gammas, "Zs": Zs, "Xs": Xs, "adj": adj}
            for mode in ("single", "double"):
                cl = qc.clique_index_sets(q.net, beta, mode)
                out[f"cl_{mode}_ptr"] = np.cumsum([0] + [len(c) for c in cl]).astype(np.int32)
                out[f"cl_{mode}_idx"] = np.concatenate(cl).astype(np.int32)
            np.savez_compressed(os.path.join(gold, f"golden_{name}_b{beta}.npz"), **out)


if __name__ == "__main__":
    main()
