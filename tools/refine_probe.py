"""whole solves with / without the refinement stage of the projection kernel, at several acceptance settings (diagnostic env overrides)
usage: python tools/refine_probe.py [single|double|both]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, helpers, nnsdp_amd as na
which = sys.argv[1] if len(sys.argv) > 1 else "both"
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
modes = [(na.DoubleDecomp(), "double"), (na.SingleDecomp(), "single")]
modes = [m for m in modes if which in ("both", m[1])]
for mode, nm in modes:
    for rf, acc, kcap in ((0, 0, 0), (1, 30, 0.05)):
        if rf:
            os.environ["NNSDP_REFINE_ACC"] = str(acc); os.environ["NNSDP_REFINE_KCAP"] = str(kcap)
        for cert in (0.0, 1e-3):
            s = na.runQuery(q, na.AdmmSdpOptions(max_iters=100000, eps_rel=1e-6, decomp_mode=mode, proj_refine=rf, max_time=60, cert_tol=cert))
            rb = s.summary["refine_blocks"]
            tot = max(sum(rb), 1)
            print(f"{nm:6s} refine {int(rf)} acc {acc:3d} kcap {kcap:4.2f} cert {cert:g}: {s.termination_status} iters {s.summary['iters']:6d} solve {s.solve_time:6.3f} s "
                  f"({1e6 * s.solve_time / s.summary['iters']:6.1f} us/it) rho {s.objective_value:.9f} admm {s.summary['objective_admm']:.9f} sweeps/block {s.summary['avg_sweeps']:.3f} "
                  f"refine [conv {rb[0]} step {rb[1]} sweeps {rb[2]} ({100.0 * rb[2] / tot:.2f} %) skipped {rb[3]} ({100.0 * rb[3] / tot:.2f} %) checked {rb[4]}]", flush=True)
