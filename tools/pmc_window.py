"""The bench workload's handle sequence for the rocprofv3 --pmc passes of tools/collect_profiles.sh: W40-D20 beta = 0, Single
decomposition, 2 000 iterations of the regular solve loop, the warm-up + timed window, 18 000 more, the late window - the same
launches bench.py issues for `value`, without its process around them (no torch, no per-launch events, no certificate): since round 4
the counter tool segfaults in one of its own threads when bench.py itself runs under --pmc on this pool (5 of 5 runs with the
collection's arguments; tools/pmc_probe.sh), and collects fine on this.  Launches are eager (NNSDP_NO_GRAPH=1 is set by the caller).
usage: rocprofv3 --pmc <counters> --kernel-trace ... -- python3 tools/pmc_window.py [burn=2000] [warm=50] [steps=200] [late=18000]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers, nnsdp_amd as na
burn, warm, steps, late = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 2000), (2, 50), (3, 200), (4, 18000)))
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
s.advance(burn)
s.iterate(warm)
s.iterate(steps)
if late > 0:
    s.advance(late)
    s.iterate(warm)
    s.iterate(steps)
print("pres dres pobj dobj", s.residuals())
s.close()
