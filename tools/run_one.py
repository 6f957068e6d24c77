#!/usr/bin/env python3
"""Solve one fixture problem and print the result summary (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
name, beta, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kw = dict(a.split("=") for a in sys.argv[4:])
kw = {k: (float(v) if "." in v or "e" in v else int(v)) for k, v in kw.items()}
d = helpers.load_problem(name, beta)
m = {"single": na.SingleDecomp(), "double": na.DoubleDecomp(), "dense": na.DenseCone(), "path": na.PathDecomp()}[mode]
t = time.time()
s = na.runQuery(helpers.product_query(d), na.AdmmSdpOptions(decomp_mode=m, **kw))
print(name, beta, mode, f"wall {time.time()-t:.2f}s", s.termination_status, "rho", s.objective_value, s.summary)
