import os, sys
ROOT="/root/repo"
for p in (ROOT, ROOT+"/nn-sdp_amd", ROOT+"/tests"): sys.path.insert(0,p)
import numpy as np, nnsdp_amd as na
from nnsdp_amd import frontend as F
net = na.randomNetwork([5] + [50] * 6 + [5], seed=1234)
x0 = np.full(5, 0.3); lo, hi = x0 - 0.05, x0 + 0.05
xi, acx = F.intervalsWorstCase(lo, hi, net)
nrm = np.zeros(5); nrm[0] = 1.0
q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=F.makeQcActivsIntvs(net, xi, acx, 0))
s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10**9))
for upto in list(range(9000, 9120)):
    s.advance(upto - int(s.info(3)))
    print("== iteration", int(s.info(3)), flush=True)
    s.finish()
