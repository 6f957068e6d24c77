#!/usr/bin/env python3
"""(diagnostic) trace of the certified (polished) objective and of the ADMM primal / dual estimates along a solve, for tuning the
certified-gap stopping rule:  NNSDP_TRACE_POLISH=<period> python tools/polish_trace.py [fixture beta iters]  (stderr of the library)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, nnsdp_amd as na
name = sys.argv[1] if len(sys.argv) > 1 else "W40-D20"
beta = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
q = helpers.product_query(helpers.load_problem(name, beta))
for mode in (na.DoubleDecomp(), na.PathDecomp()):
    print("==", name, beta, type(mode).__name__, flush=True)
    sys.stderr.write(f"== {name} {beta} {type(mode).__name__}\n"); sys.stderr.flush()
    na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=iters, eps_rel=1e-9))
