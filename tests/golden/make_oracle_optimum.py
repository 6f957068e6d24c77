#!/usr/bin/env python3
"""Converged optimum of the CPU oracle (oracle/admm.py, numpy + LAPACK eigh per clique) on BASELINE.json's own configs, for
which the reference publishes nothing: bench/rand W40-D20 beta = 0 / 2 and W40-D40 beta = 0, findEllipsoid on [0.5,1.5]^2,
inputs = the committed problem fixtures.  Stored under tests/golden/oracle_optimum.json; the -m gpu tests compare the HIP
solver's converged rho (Single and Double decomposition) with it.

Run in the build container (minutes to tens of minutes per case):
    python tests/golden/make_oracle_optimum.py [W40-D20:0 W40-D20:2 W40-D40:0]
(test infrastructure; the GPU box only reads the JSON)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import helpers  # noqa: E402
from oracle import admm as oadmm, operator as oop  # noqa: E402


def main():
    cases = sys.argv[1:] or ["W40-D20:0", "W40-D20:2", "W40-D40:0"]
    out = os.path.join(helpers.GOLDEN, "oracle_optimum.json")
    res = json.load(open(out)) if os.path.exists(out) else {}
    for c in cases:
        name, beta = c.split(":")
        beta = int(beta)
        q = helpers.oracle_query(helpers.load_problem(name, beta))
        L = oop.build_operator(q, "double", normalize=True)
        t = time.time()
        r = oadmm.admm_solve(L, oadmm.AdmmOptions(max_iters=200000, eps_rel=1e-6))
        secs = time.time() - t
        Lf = oop.build_operator(q, "dense", normalize=False) if q.net.Zdim <= 900 else None
        lam = float(np.linalg.eigvalsh(Lf.Z_dense(r.gamma))[-1]) if Lf is not None else None
        g = r.gamma
        res[f"{name}_b{beta}"] = dict(net=name, beta=beta, decomp="double", rho=r.objective, iters=r.iters, pres=r.pres, dres=r.dres,
                                     status=r.status, secs=round(secs, 1), gamma_norm=float(np.linalg.norm(g)), gamma_min=float(g.min()),
                                     lambda_max_raw_iterate=lam)
        print(c, res[f"{name}_b{beta}"], flush=True)
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
