import sys, time
sys.path.insert(0, "nn-sdp_amd"); sys.path.insert(0, "tests")
import numpy as np, helpers, nnsdp_amd as na
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
for guard in (5e-5, 0.0):
    for rep in range(2):
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), interval_guard=guard))
        s.iterate(5, time_eig=True)
        out = []
        for w in range(6):
            t = time.perf_counter(); ms = s.iterate(20, time_eig=True); dt = time.perf_counter() - t
            out.append((round(1e3 * ms / 20, 1), round(20 / dt)))
        print("guard", guard, "K3 us / it/s per 20-iteration window:", out, flush=True)
        s.close()
