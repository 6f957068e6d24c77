"""(diagnostic) cost of the clique-sharded iteration's exchange on ONE card with a one-rank communicator (no peer: what the transport
itself adds to an iteration): unsharded, RCCL (ncclAllReduce inside the hipGraph), the hipIpc device-side transport.
usage: python tools/shard_overhead.py [workload=W40-D20]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, helpers, nnsdp_amd as na
wl = sys.argv[1] if len(sys.argv) > 1 else "W40-D20"
q = helpers.product_query(helpers.load_problem(wl, 0))
def run(kind):
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
    if kind == "rccl": s.set_comm(1, 0, na.comm_unique_id())
    if kind == "ipc": s.set_comm_ipc(1, 0, lambda a: None)
    s.advance(6000)
    s.iterate(70)
    t0 = time.perf_counter(); s.iterate(2100); dt = time.perf_counter() - t0
    r = s.residuals()
    s.close()
    return 1e6 * dt / 2100, r
base = None
for kind in ("unsharded", "ipc", "rccl"):
    us, r = run(kind)
    base = base or us
    print(f"{wl} {kind:9s}: {us:6.1f} us per iteration (+{us - base:5.1f} over unsharded), pres {r[0]:.3e} obj {r[2]:.9g}", flush=True)
