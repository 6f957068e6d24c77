#!/usr/bin/env python3
"""(diagnostic) ACAS-Xu-shaped safety verification (BASELINE configs[4]; no ACAS file is in the reference checkout): synthetic
5-50x6-5 ReLU net, box of half-width 0.05, property 'Y_0 stays below y_0(x0) + margin' written as VNNLIB text."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nnsdp_amd as na
from nnsdp_amd import vnnlib as vl
W = int(sys.argv[1]) if len(sys.argv) > 1 else 50
margin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
net = na.randomNetwork([5] + [W] * 6 + [5], seed=1234)
x0 = np.full(5, 0.3)
y0 = na.evalFeedFwdNet(net, x0)
spec = "".join(f"(assert (>= X_{i} {x0[i] - 0.05}))(assert (<= X_{i} {x0[i] + 0.05}))" for i in range(5)) + f"(assert (>= Y_0 {y0[0] + margin}))"
xi, acx = na.makeIntervalsInfo(x0 - 0.05, x0 + 0.05, net)
dead = sum(int(np.sum(u <= 0)) for l, u in xi[1:-1])
print(f"W={W}: y0={y0[0]:.4f}, output interval [{xi[-1][0][0]:.4f}, {xi[-1][1][0]:.4f}], fixed (dead) hidden neurons {dead} of {6 * W}")
for mode, via in ((na.PathDecomp(), False), (na.PathDecomp(), True)):
    opts = na.AdmmSdpOptions(decomp_mode=mode, max_iters=60000, eps_rel=1e-5, max_time=120, cert_tol=1e-3 if via else 0.0)
    t = time.time()
    try:
        solns, nq, status = vl.verifyAcasSpec(net, spec, 1, opts, via_reach=via)
        s = solns[0]
        print(f"via_reach={via} {type(mode).__name__}: margin {s.summary.get('margin')} {status} {s.termination_status} iters {s.summary['iters']} blocks {s.summary['n_cliques']} max {s.summary['max_clique']} "
              f"lambda_max {s.summary['lambda_max']:.2e} solve {s.solve_time:.2f}s wall {time.time() - t:.2f}s sweeps {s.summary['avg_sweeps']:.2f}", flush=True)
    except Exception as e:
        print(f"via_reach={via} {type(mode).__name__}: {type(e).__name__}: {e}", flush=True)
