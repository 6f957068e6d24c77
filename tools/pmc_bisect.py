"""(diagnostic) which call of a solve does rocprofv3 --pmc die in?  progress lines go to stderr unbuffered.
usage: rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/pmc_bisect.py <stage: create|iterate|advance|graph>"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers, nnsdp_amd as na
stage = sys.argv[1]
t0 = time.time()
def say(m): print(f"[{time.time() - t0:6.1f}] {m}", file=sys.stderr, flush=True)
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
say("query")
s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
say("created")
if stage in ("iterate", "advance", "graph"):
    s.iterate(20, time_eig=True); say("iterate 20 timed")
    s.iterate(20); say("iterate 20")
if stage in ("advance", "graph"):
    s.advance(200); say("advance 200")
if stage == "graph":
    s.iterate(512); say("iterate 512")
    s.advance(2000); say("advance 2000")
if stage == "long":
    s.iterate(20, time_eig=True); say("iterate 20 timed")
    for k in range(10):
        s.advance(200); say(f"advance {200 * (k + 1)}")
    s.iterate(200, time_eig=True); say("iterate 200 timed")
    s.iterate(2048); say("iterate 2048")
    s.advance(6000); say("advance 6000")
s.close(); say("closed")
