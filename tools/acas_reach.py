#!/usr/bin/env python3
"""(diagnostic) the same ACAS-shaped property through a reach-hyperplane query: certified bound on normal' y"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nnsdp_amd as na
W = int(sys.argv[1]) if len(sys.argv) > 1 else 50
net = na.randomNetwork([5] + [W] * 6 + [5], seed=1234)
x0 = np.full(5, 0.3)
lo, hi = x0 - 0.05, x0 + 0.05
xi, _ = na.makeIntervalsInfo(lo, hi, net)
X = lo[:, None] + (hi - lo)[:, None] * np.random.default_rng(0).random((5, 20000))
Y = na.evalFeedFwdNet(net, X)
print(f"CROWN output interval of y_0: [{xi[-1][0][0]:.5f}, {xi[-1][1][0]:.5f}]; sampled max {Y[0].max():.5f}")
qa = na.makeQcActivs(net, lo, hi, 1)
nrm = np.zeros(5); nrm[0] = 1.0
q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=qa)
for mode, kw in ((na.PathDecomp(), dict(cert_tol=1e-3)), (na.PathDecomp(), {}), (na.SingleDecomp(), dict(cert_tol=1e-3))):
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=200000, eps_rel=1e-6, max_time=120, **kw))
    print(f"{type(mode).__name__} {kw}: {s.termination_status} bound {s.objective_value:.6f} iters {s.summary['iters']} max block {s.summary['max_clique']} "
          f"lambda_max {s.summary['lambda_max']:.2e} solve {s.solve_time:.2f}s wall {time.time() - t:.2f}s", flush=True)
