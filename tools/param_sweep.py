#!/usr/bin/env python3
"""ADMM parameter sweep on one fixture problem (diagnostic): iterations and solve time to eps_rel = 1e-6."""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
name, mode = sys.argv[1], sys.argv[2]
m = {"single": na.SingleDecomp(), "double": na.DoubleDecomp(), "path": na.PathDecomp()}[mode]
q = helpers.product_query(helpers.load_problem(name, 0))
na.runQuery(q, na.AdmmSdpOptions(decomp_mode=m, max_iters=200))   # warm up the process
for sigma, alpha, adapt in itertools.product((0.03, 0.1, 0.3), (1.5, 1.6, 1.75), (25, 50, 100)):
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=m, max_iters=300000, eps_rel=1e-6, sigma=sigma, alpha=alpha, adapt_every=adapt, max_time=30))
    print(f"sigma {sigma} alpha {alpha} adapt {adapt}: {s.termination_status} iters {s.summary['iters']} solve {s.solve_time:.2f}s rho {s.objective_value:.8f}", flush=True)
