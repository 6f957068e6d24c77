#!/usr/bin/env python3
"""Batched regime only (diagnostic / profiling): B copies of a fixture problem advanced in lockstep by the batch handle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import helpers
import nnsdp_amd as na
B = int(sys.argv[1]) if len(sys.argv) > 1 else 13
name = sys.argv[2] if len(sys.argv) > 2 else "W40-D20"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
q = helpers.product_query(helpers.load_problem(name, 0))
sb = na.SolverBatch([q] * B, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
sb.advance(1000)
sb.iterate(64)
t = time.time()
sb.iterate(iters)
dt = time.time() - t
print(f"B={B} {name}: {B * iters / dt:.0f} it/s aggregate, {1e6 * dt / iters:.1f} us per lockstep iteration")
sb.close()
