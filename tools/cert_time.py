"""(diagnostic) time to the certified-gap certificate (cert_tol = 1e-3) and to residuals 1e-6 on a fixture, a warm second solve each.
usage: python tools/cert_time.py [workload=W40-D20] [mode=double]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers, nnsdp_amd as na
wl = sys.argv[1] if len(sys.argv) > 1 else "W40-D20"
mode = {"single": na.SingleDecomp(), "double": na.DoubleDecomp(), "path": na.PathDecomp()}[sys.argv[2] if len(sys.argv) > 2 else "double"]
q = helpers.product_query(helpers.load_problem(wl, 0))
for rep in range(2):
    for rule, kw in (("cert 1e-3", dict(eps_rel=1e-6, cert_tol=1e-3)), ("res 1e-6", dict(eps_rel=1e-6))):
        t = time.perf_counter()
        s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=500000, max_time=60, **kw))
        if rep: print(f"{wl} {type(mode).__name__} {rule}: wall {time.perf_counter() - t:.3f} s solve {s.solve_time:.3f} s iters {s.summary['iters']} {s.termination_status} rho {s.objective_value:.9g} "
                      f"sweeps/visit {s.summary['avg_sweeps']:.2f} refine {s.summary['refine_blocks']}", flush=True)
