#!/usr/bin/env python3
"""(diagnostic) penalty / residual trajectory of one solve, from the verbose log"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
name, beta, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
m = {"single": na.SingleDecomp(), "double": na.DoubleDecomp()}[mode]
q = helpers.product_query(helpers.load_problem(name, beta))
s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=m, max_iters=400000, eps_rel=1e-6, max_time=60, verbose=True))
print("done", s.termination_status, s.summary["iters"], file=sys.stderr)
