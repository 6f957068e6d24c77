#!/usr/bin/env python3
"""How fast does the POLISHED (exactly feasible) objective converge?  Diagnostic, not a test."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
name, beta, step, total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
d = helpers.load_problem(name, beta)
sv = na.Solver(helpers.product_query(d), na.AdmmSdpOptions(max_iters=10**9))
t0 = time.time(); it = 0
# emulate run_loop's sigma adaptation through run() in slices is not exposed; use residuals() + iterate
while it < total:
    sv.iterate(step - 1); it += step
    pres, dres, pobj, dobj = sv.residuals()
    tf = time.time(); s = sv.finish(); tf = time.time() - tf
    print(f"it {it:6d} t {time.time()-t0:6.2f}s pres {pres:.2e} dres {dres:.2e} rho_admm {s.summary['objective_admm']:.7g} dobj {dobj:.7g} rho_cert {s.objective_value:.7g} "
          f"shift {s.summary['polish_shift']:.2e} lmax {s.summary['lambda_max']:.1e} finish {tf*1e3:.0f} ms", flush=True)
