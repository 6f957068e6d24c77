import os, sys, time
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nnsdp_amd as na
from nnsdp_amd import frontend as F
from oracle import nnet_io
onet = nnet_io.random_net([5] + [50] * 6 + [5], seed=1)
net = na.FeedFwdNet(xdims=onet.xdims, Ms=onet.Ms)
x0 = np.full(5, 0.3); lo, hi = x0 - 0.05, x0 + 0.05
xi, acx = F.intervalsWorstCase(lo, hi, net)
qa = F.makeQcActivsIntvs(net, xi, acx, 0)
nrm = np.zeros(5); nrm[0] = 1.0
q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=qa)
for mode in (na.PathDecomp(), na.DoubleDecomp()):
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=200000, eps_rel=1e-5, max_time=100))
    print(f"alg {os.environ.get('NNSDP_PROJ_ALG','default')} {type(mode).__name__} (max block {s.summary['max_clique']}, {s.summary['n_cliques']} blocks): {s.termination_status} bound {s.objective_value:.7f} iters {s.summary['iters']} "
          f"solve {s.solve_time:.3f} s = {1e6 * s.solve_time / max(s.summary['iters'], 1):.0f} us/iter, sweeps/visit {s.summary['avg_sweeps']:.2f} refine {s.summary['refine_blocks']}", flush=True)
