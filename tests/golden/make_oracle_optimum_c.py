#!/usr/bin/env python3
"""Same fixture as make_oracle_optimum.py (tests/golden/oracle_optimum.json), but the iteration loop runs in the C++/OpenMP port of
the oracle's ADMM (oracle/c/admm_cpu.cpp, iterates equal to oracle/admm.py to 1e-11) so that W40-D40 reaches a real OPTIMAL at 1e-6
in hours instead of the 2.3 h ITERATION_LIMIT run of the numpy loop.  Check / penalty schedule: that of oracle.admm.admm_solve.
    python tests/golden/make_oracle_optimum_c.py W40-D40:0 [threads=4] [max_iters=1000000]
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
from oracle import admm as oadmm, admm_c, operator as oop  # noqa: E402


def main():
    name, beta = sys.argv[1].split(":")
    beta = int(beta)
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    max_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
    q = helpers.oracle_query(helpers.load_problem(name, beta))
    L = oop.build_operator(q, "double", normalize=True)
    P = oadmm.ScaledProblem(L)
    cpu = admm_c.CpuAdmm(P, 0.1, 1.6, threads=threads)
    S = cpu.S
    t0 = time.time()
    it, next_adapt, status = 0, 50, "ITERATION_LIMIT"
    r = None
    while it < max_iters:
        r = cpu.step(50)            # 49 plain iterations + the residuals of the 50th, as admm_solve checks every 50
        it += 50
        rp, rd = r["pres"], r["dres"]
        if it % 5000 == 0:
            print(f"it {it} pres {rp:.3e} dres {rd:.3e} obj {r['objective']:.10g} sigma {cpu.sigma:.3g} t {time.time() - t0:.0f}s", flush=True)
        if rp <= 1e-6 and rd <= 1e-6:
            status = "OPTIMAL"
            break
        if it >= next_adapt:
            next_adapt = max(it + 100, it * 3 // 2)
            ratio = np.sqrt(max(rp, 1e-300) / max(rd, 1e-300))
            if ratio > 1.5 or ratio < 0.67:
                S.nu, S.sigma = cpu.nu.copy(), cpu.sigma
                S.set_sigma(cpu.sigma * min(max(ratio, 0.2), 5.0))
                cpu.nu[:] = S.nu
                cpu.sigma = S.sigma
    S.nu, S.sigma = cpu.nu.copy(), cpu.sigma
    w = S.proj(S.nu)
    y = S.sigma * (S.nu - w)
    gam = P.unscale_gamma(np.maximum(-y[:S.ng], 0.0))
    out = os.path.join(helpers.GOLDEN, "oracle_optimum.json")
    res = json.load(open(out)) if os.path.exists(out) else {}
    res[f"{name}_b{beta}"] = dict(net=name, beta=beta, decomp="double", rho=float(L.c @ gam), iters=it, pres=float(r["pres"]), dres=float(r["dres"]),
                                 status=status, secs=round(time.time() - t0, 1), gamma_norm=float(np.linalg.norm(gam)), gamma_min=float(gam.min()),
                                 lambda_max_raw_iterate=None, loop="oracle/c/admm_cpu.cpp (C++/OpenMP port of oracle/admm.py)")
    print(res[f"{name}_b{beta}"], flush=True)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
