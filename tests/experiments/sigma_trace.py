"""(experiment, not a test) tail of the oracle ADMM: see DESIGN.md section 4, negative result (12).  usage: python tests/experiments/<this>.py W10-D20 0 ..."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm
name, beta = sys.argv[1], int(sys.argv[2])
sig0 = float(sys.argv[3]) if len(sys.argv)>3 else 0.1
adapt = int(sys.argv[4]) if len(sys.argv)>4 else 1
iters = int(sys.argv[5]) if len(sys.argv)>5 else 150000
q = helpers.oracle_query(helpers.load_problem(name, beta))
L = oop.build_operator(q, "double", normalize=True)
t=time.time()
r = oadmm.admm_solve(L, oadmm.AdmmOptions(max_iters=iters, eps_rel=1e-6, sigma=sig0, adapt_sigma=bool(adapt)))
print(name, r.status, "%.8f"%r.objective, r.iters, "%.1fs"%(time.time()-t))
last=None
for h in r.history:
    if last is None or h[5]!=last or h[0]%5000==0:
        print("  it %6d rp %.2e rd %.2e obj %.8f sigma %.4g"%(h[0],h[1],h[2],h[3],h[5]))
        last=h[5]
