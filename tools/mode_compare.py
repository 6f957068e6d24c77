#!/usr/bin/env python3
"""Single vs Double decomposition on a full-size problem: iterations and wall-clock to certificate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
name, beta = sys.argv[1], int(sys.argv[2])
d = helpers.load_problem(name, beta)
q = helpers.product_query(d)
for mode in (na.SingleDecomp(), na.DoubleDecomp()):
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=150000, max_time=60))
    sm = s.summary
    print(type(mode).__name__, f"wall {time.time()-t:.2f}s setup {s.setup_time:.2f} solve {s.solve_time:.2f} status {s.termination_status} iters {sm['iters']} "
          f"rho {s.objective_value:.7g} rho_admm {sm['objective_admm']:.7g} lmax {sm['lambda_max']:.1e} blocks {sm['n_cliques']} nmax {sm['max_clique']} sweeps {sm['avg_sweeps']:.2f} "
          f"us/iter {1e6*s.solve_time/sm['iters']:.0f}", flush=True)
