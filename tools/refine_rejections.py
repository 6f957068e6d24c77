import sys, os
os.environ["NNSDP_REFINE_STATS"]="1"
sys.path.insert(0, "/root/repo/nn-sdp_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, helpers, nnsdp_amd as na
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
for mode in (na.SingleDecomp(), na.DoubleDecomp()):
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=10**9))
    done = 0
    for upto in (2000, 6000, 12000, 16000):
        s.advance(upto - done); done = upto
        sol = s.finish()
        print(type(mode).__name__, "after", upto, sol.summary["refine_blocks"], flush=True)
    s.close()
