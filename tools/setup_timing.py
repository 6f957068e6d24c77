"""(diagnostic) where the wall-clock of a solve goes outside the iterations: NNSDP_SETUP_TIMING=1 python tools/setup_timing.py"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/nn-sdp_amd"); sys.path.insert(0, "/root/repo/tests")
import helpers, nnsdp_amd as na
wl = sys.argv[1] if len(sys.argv) > 1 else "W40-D20"
q = helpers.product_query(helpers.load_problem(wl, 0))
for rep in range(3):
    for mode in ((na.DoubleDecomp(),) if wl != "W40-D20" else (na.DoubleDecomp(), na.SingleDecomp())):
        print("---", type(mode).__name__, rep, flush=True)
        t = time.time()
        s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=400000, cert_tol=1e-3))
        print(f"wall {time.time()-t:.3f} iters {s.summary['iters']} library: setup {s.setup_time:.3f} solve {s.solve_time:.3f} total (create .. finish) {s.total_time:.3f}", flush=True)
