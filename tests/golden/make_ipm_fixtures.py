#!/usr/bin/env python3
"""Independent optimum of the restated DeepSDP model (oracle/ipm.py: dense primal-dual interior point on the literal
assembly, reference coordinates, no normalisation, no cliques) for the small published nets, and the trace of the
method - how far above the optimum the primal objective c'gamma sits at each relative gap.

Outputs (committed):
  tests/golden/ipm_optimum.json          net, beta -> rho (c'gamma), lower bound <Z0, X>, residuals, eigmax(Z(gamma))
  profiles/r02_ipm_trace_<net>_b<beta>.csv  it, c'gamma, lower, pinf, dinf, gap, mu

Run in the build container:  python tests/golden/make_ipm_fixtures.py [W10-D5:0 W10-D10:0 W20-D10:0 ...]
(test infrastructure; nothing of this runs on the GPU box)."""
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import helpers  # noqa: E402
from oracle import ipm, nnet_io, qc  # noqa: E402

DEFAULT = ["W10-D5:0", "W10-D5:3", "W10-D10:0", "W20-D10:0"]


def query_for(name, beta):
    path = os.path.join(helpers.GOLDEN, f"problem_{name}_b{beta}.npz")
    if os.path.exists(path):
        return helpers.oracle_query(helpers.load_problem(name, beta))
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = nnet_io.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    return qc.make_reach_ellipsoid_query(net, [0.5, 0.5], [1.5, 1.5], beta, seed=1234)


def main():
    cases = sys.argv[1:] or DEFAULT
    out_json = os.path.join(helpers.GOLDEN, "ipm_optimum.json")
    res = json.load(open(out_json)) if os.path.exists(out_json) else {}
    for c in cases:
        name, beta = c.split(":")
        beta = int(beta)
        q = query_for(name, beta)
        t = time.time()
        r = ipm.solve_query(q, ipm.IpmOptions(max_iters=int(os.environ.get("IPM_MAX_ITERS", "500")), tol_gap=1e-9, tol_feas=1e-9,
                                              verbose="-v" in os.environ.get("IPM_FLAGS", "")))
        secs = time.time() - t
        pub = sorted(helpers.published_rho(name, beta))
        # only converged runs are pins (tests/test_oracle_ipm.py iterates over ipm_optimum.json); a run that hit the iteration cap is
        # kept apart with the bracket it reached
        converged = r.status in ("OPTIMAL", "NEAR_OPTIMAL")
        if not converged:
            out_unc = os.path.join(helpers.GOLDEN, "ipm_unconverged.json")
            unc = json.load(open(out_unc)) if os.path.exists(out_unc) else {}
            unc[f"{name}_b{beta}"] = dict(net=name, beta=beta, rho=r.objective, lower=r.dual_objective, gap=float(r.gap), pinf=float(r.pinf), dinf=float(r.dinf),
                                           lambda_max=r.lambda_max, status=r.status, iters=r.iters, secs=round(time.time() - t, 1), published=sorted(helpers.published_rho(name, beta)))
            json.dump(unc, open(out_unc, "w"), indent=1, sort_keys=True)
            print(c, "NOT CONVERGED", unc[f"{name}_b{beta}"], flush=True)
        res[f"{name}_b{beta}"] = dict(net=name, beta=beta, rho=r.objective, lower=r.dual_objective, gap=r.gap, pinf=r.pinf, dinf=r.dinf,
                                     lambda_max=r.lambda_max, status=r.status, iters=r.iters, secs=round(secs, 1), published=pub)
        print(c, res[f"{name}_b{beta}"], flush=True)
        if not converged:
            del res[f"{name}_b{beta}"]
        with open(os.path.join(ROOT, "profiles", f"{os.environ.get('IPM_ROUND', 'r03')}_ipm_trace_{name}_b{beta}.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["it", "c_gamma", "lower_bound", "pinf", "dinf", "rel_gap", "mu"])
            for h in r.history:
                w.writerow([h["it"], repr(float(h["obj"])), repr(float(h["lower"])), h["pinf"], h["dinf"], h["gap"], h["mu"]])
        with open(out_json, "w") as fh:
            json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
