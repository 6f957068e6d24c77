"""(diagnostic) 13 W40-D20 SDPs as ONE lockstep batch against TWO lockstep batches (7 + 6) on two streams driven by two host threads:
the projection launch is compute-bound on the CUs its blocks occupy and the M^-1 product is HBM-bound, so two batches out of phase
could overlap the two.  usage: python tools/batch_two_streams.py [sdps=13] [advance=2000] [timed=2000]"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers, nnsdp_amd as na

B = int(sys.argv[1]) if len(sys.argv) > 1 else 13
adv = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
timed = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
opts = na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9)

def rate(groups):
    bs = [na.SolverBatch([q] * g, opts) for g in groups]
    for b in bs:
        b.advance(adv); b.iterate(64)
    def run(b): b.iterate(timed)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(b,)) for b in bs]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    for b in bs: b.close()
    return sum(groups) * timed / dt, 1e6 * dt / timed

def split(n, g):
    return [n // g + (1 if i < n % g else 0) for i in range(g)]
for groups in [split(B, g) for g in (1, 2, 3, 4, 5, 6, 13)] + [[7, 6], [B]]:
    r, us = rate(groups)
    print(f"groups {groups}: {r:.0f} aggregate it/s ({us:.1f} us per lockstep iteration of the slowest group)", flush=True)
