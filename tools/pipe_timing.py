"""(diagnostic) iterations/s late in a solve with and without the tile-parallel refinement pipeline: advance a handle deep into the
solve, then time graph-replayed iterations.  NNSDP_PIPE=0/1/2/3 selects the mode (0 never, 1 default, 2 auto for every size, 3 always) (read at solver creation).
usage: [NNSDP_PIPE=..] python tools/pipe_timing.py [workload=W40-D20] [mode=single] [advance=8000] [timed=4000]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers, nnsdp_amd as na

wl = sys.argv[1] if len(sys.argv) > 1 else "W40-D20"
mode = sys.argv[2] if len(sys.argv) > 2 else "single"
adv = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
m = {"single": na.SingleDecomp(), "double": na.DoubleDecomp(), "path": na.PathDecomp()}[mode]
q = helpers.product_query(helpers.load_problem(wl, 0))
s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=m, max_iters=10 ** 9))
s.advance(adv)
s.iterate(64)
t0 = time.perf_counter()
s.iterate(timed)
dt = time.perf_counter() - t0
ms = s.iterate(200, time_eig=True)
pres, dres, pobj, dobj = s.residuals()
soln = s.finish()
print(f"pipe={os.environ.get('NNSDP_PIPE', 'default')} {wl} {mode}: {timed / dt:.0f} it/s graph replay ({1e6 * dt / timed:.1f} us/step), "
      f"K3 eager {1e3 * ms / 200:.1f} us/launch; pres {pres:.2e} dres {dres:.2e} obj {pobj:.8g}; refine_blocks {soln.summary.get('refine_blocks')}")
s.close()
