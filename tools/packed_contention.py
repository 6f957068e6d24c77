#!/usr/bin/env python3
"""(diagnostic) three processes on one card, each advancing the same ACAS-Xu shaped solve (packed variant, 106 + 4 x 151 blocks, refinement
stage on) by 3 000 iterations, twice: the multiplier blocks must come out bit-identical in every process and round (a barrier mismatch
or a race in the stage shows up under exactly this contention).  usage: python tools/packed_contention.py"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, hashlib
ROOT = sys.argv[1]
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, nnsdp_amd as na
from nnsdp_amd import frontend as F
net = na.randomNetwork([5] + [50] * 6 + [5], seed=1234)
x0 = np.full(5, 0.3); lo, hi = x0 - 0.05, x0 + 0.05
xi, acx = F.intervalsWorstCase(lo, hi, net)
nrm = np.zeros(5); nrm[0] = 1.0
q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=F.makeQcActivsIntvs(net, xi, acx, 0))
for rep in range(2):
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
    s.advance(3000)
    m = s.raw_multipliers()
    sol = s.finish()
    print("DIGEST", hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest()[:16], sol.summary["refine_blocks"], f"pres {sol.summary['pres']:.3e}", flush=True)
    s.close()
'''
procs = [subprocess.Popen([sys.executable, "-c", child, ROOT], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(3)]
outs = [p.communicate(timeout=500)[0] for p in procs]
digests = [l.split()[1] for o in outs for l in o.splitlines() if l.startswith("DIGEST")]
for o in outs:
    print(o.strip())
print("processes", len(procs), "digests", len(digests), "distinct", len(set(digests)))
sys.exit(0 if len(digests) == 6 and len(set(digests)) == 1 else 1)
