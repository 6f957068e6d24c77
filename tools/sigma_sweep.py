#!/usr/bin/env python3
"""(diagnostic) iterations to 1e-6 for one setting of the penalty adaptation (NNSDP_SIGMA_GEOM / NNSDP_SIGMA_POW) over several problems."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
tot = 0
out = []
for name, beta, mode in (("W40-D20", 0, na.DoubleDecomp()), ("W40-D20", 2, na.DoubleDecomp()), ("W40-D20", 0, na.SingleDecomp()), ("W20-D10", 0, na.DoubleDecomp()),
                         ("W10-D20", 0, na.DoubleDecomp()), ("W40-D40", 0, na.DoubleDecomp())):
    q = helpers.product_query(helpers.load_problem(name, beta))
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=400000, eps_rel=1e-6, max_time=40))
    out.append(f"{name}b{beta}{type(mode).__name__[0]}:{s.summary['iters']}{'' if s.termination_status == 'OPTIMAL' else '!'}")
    tot += s.summary["iters"]
print(os.environ.get("NNSDP_SIGMA_GEOM", "1.5"), os.environ.get("NNSDP_SIGMA_POW", "1.0"), "total", tot, " ".join(out), flush=True)
