#!/usr/bin/env python3
"""(diagnostic) an ACAS-Xu shaped reach-hyperplane query (5-50x6-5, the reference's own Single cliques: 106 + 4 x 151) with and without the
packed variant's refinement stage: iterations, solve time per iteration, certified bound.  usage: python tools/acas_shaped_refine.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nnsdp_amd as na
from oracle import nnet_io                  # (only the random network of tests/test_gpu_parity.py's 151-block case: weights, nothing else)
if len(sys.argv) > 1 and sys.argv[1] == "bench":      # the network of bench.py's wide_blocks leg
    net = na.randomNetwork([5] + [50] * 6 + [5], seed=1234)
else:
    onet = nnet_io.random_net([5] + [50] * 6 + [5], seed=1)
    net = na.FeedFwdNet(xdims=onet.xdims, Ms=onet.Ms)
x0 = np.full(5, 0.3)
lo, hi = x0 - 0.05, x0 + 0.05
from nnsdp_amd import frontend as F
xi, acx = F.intervalsWorstCase(lo, hi, net)       # (plain interval arithmetic: no neuron is stable, the cliques keep their full 106 / 151)
qa = F.makeQcActivsIntvs(net, xi, acx, 0)
nrm = np.zeros(5); nrm[0] = 1.0
q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=qa)
for rf in ((1,) if len(sys.argv) > 1 else (0, 1)):
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=200000, eps_rel=1e-5, max_time=60, proj_refine=rf))
    print(f"5-50x6-5 reach hyperplane, Single (max block {s.summary['max_clique']}), proj_refine {rf}: {s.termination_status} bound {s.objective_value:.7f} iters {s.summary['iters']} "
          f"solve {s.solve_time:.3f} s = {1e6 * s.solve_time / max(s.summary['iters'], 1):.0f} us/iter, lambda_max {s.summary['lambda_max']:.2e} sweeps/visit {s.summary['avg_sweeps']:.2f} "
          f"refine {s.summary['refine_blocks']} wall {time.time() - t:.2f} s", flush=True)

# where in the solve the stage is taken: block visits per window of 2 000 iterations
slv = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9, proj_refine=1))
prev = [0] * 5
done = 0
for upto in range(2000, 20001, 2000):
    t = time.time()
    slv.advance(upto - done); done = upto
    dt = time.time() - t
    cur = slv.finish().summary["refine_blocks"]
    d = [c - p_ for c, p_ in zip(cur, prev)]; prev = cur
    print(f"iterations {upto - 2000:6d} .. {upto:6d}: {1e6 * dt / 2000:7.0f} us/iter; block visits: converged {d[0]}, step {d[1]}, sweeps {d[2]}, not attempted {d[3]}", flush=True)
slv.close()
