"""(experiment, not a test) tail of the oracle ADMM: see DESIGN.md section 4, negative result (12).  usage: python tests/experiments/<this>.py W10-D20 0 ..."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm
name, beta = sys.argv[1], int(sys.argv[2])
N = int(sys.argv[3]); iters=int(sys.argv[4]); use_ray=int(sys.argv[5])
q = helpers.oracle_query(helpers.load_problem(name, beta))
L = oop.build_operator(q, "double", normalize=True)
P = oadmm.ScaledProblem(L); S = oadmm.AdmmState(P, 0.1, 1.6)
def resid(S, nu_prev, w, x, res, Kxq):
    y = S.sigma*(nu_prev - w); Kty = S.Kt(y)
    rp = np.linalg.norm(res)/max(np.linalg.norm(Kxq), np.linalg.norm(w),1e-300)
    rd = np.linalg.norm(Kty-P.z0)/max(np.linalg.norm(Kty), np.linalg.norm(P.z0),1e-300)
    obj = -(P.c@y[:S.ng])/(P.zscale*P.cscale)
    return rp, rd, obj
def fpres(S, nu):
    old = S.nu; S.nu = nu.copy(); S.step(); f = np.linalg.norm(S.nu - nu); S.nu = old; return f
next_adapt=50; t=time.time(); snap=None; snap_it=0; njump=0
for it in range(1, iters+1):
    nu_prev = S.nu.copy()
    w,x,res,Kxq = S.step()
    if it % 50 == 0:
        rp,rd,obj = resid(S, nu_prev, w,x,res,Kxq)
        if it % 2000 == 0 or (rp<=1e-6 and rd<=1e-6): print("it %6d rp %.2e rd %.2e obj %.8f sigma %.4g |nu| %.4g jumps %d %.0fs"%(it,rp,rd,obj,S.sigma,np.linalg.norm(S.nu),njump,time.time()-t), flush=True)
        if rp<=1e-6 and rd<=1e-6: break
        if it>=next_adapt:
            next_adapt=max(it+100, it*3//2)
            ratio=np.sqrt(max(rp,1e-300)/max(rd,1e-300))
            if ratio>1.5 or ratio<0.67: S.set_sigma(S.sigma*min(max(ratio,0.2),5.0)); snap=None
    if use_ray and it % N == 0:
        if snap is not None and max(rp,rd) < 1e-4:
            d = S.nu - snap
            f0 = fpres(S, S.nu); best=(f0,0.0)
            for tt in (1,2,4,8,16,32,64,128):
                f = fpres(S, S.nu + tt*d)
                if f < best[0]: best=(f,tt)
                else: break
            if best[1]>0:
                S.nu = S.nu + best[1]*d; njump+=1
                if it % 2000 == 0: print("   jump t=%g f %.3e -> %.3e"%(best[1], f0, best[0]))
        snap = S.nu.copy()
