"""(experiment, not a test) tail of the oracle ADMM: see DESIGN.md section 4, negative result (12).  usage: python tests/experiments/<this>.py W10-D20 0 ..."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers
from oracle import operator as oop, admm as oadmm
name, beta = sys.argv[1], int(sys.argv[2])
mem = int(sys.argv[3]); interval=int(sys.argv[4]); iters=int(sys.argv[5])
sig0 = 0.1
q = helpers.oracle_query(helpers.load_problem(name, beta))
L = oop.build_operator(q, "double", normalize=True)
P = oadmm.ScaledProblem(L); S = oadmm.AdmmState(P, sig0, 1.6)
def resid(S, nu_prev, w, x, res, Kxq):
    y = S.sigma*(nu_prev - w); Kty = S.Kt(y)
    rp = np.linalg.norm(res)/max(np.linalg.norm(Kxq), np.linalg.norm(w),1e-300)
    rd = np.linalg.norm(Kty-P.z0)/max(np.linalg.norm(Kty), np.linalg.norm(P.z0),1e-300)
    obj = -(P.c@y[:S.ng])/(P.zscale*P.cscale)
    return rp, rd, obj
G=[]; F=[]   # history of T(nu) and f = T(nu)-nu
next_adapt=50; t=time.time(); naa=0; nrej=0
for it in range(1, iters+1):
    nu_prev = S.nu.copy()
    w,x,res,Kxq = S.step()
    Tnu = S.nu.copy(); f = Tnu - nu_prev
    if mem>0:
        G.append(Tnu); F.append(f)
        if len(G)>mem: G.pop(0); F.pop(0)
        if it % interval == 0 and len(G)>=3:
            Fm = np.array(F).T; Gm=np.array(G).T
            # type-II: min || F a ||, sum a = 1  -> solve via differences
            dF = Fm[:,1:]-Fm[:,:-1]; dG = Gm[:,1:]-Gm[:,:-1]
            gam,_,_,_ = np.linalg.lstsq(dF, Fm[:,-1], rcond=1e-10)
            nu_aa = Gm[:,-1] - dG@gam
            # safeguard: one trial step from nu_aa, accept if residual norm not larger than current
            S2nu = S.nu; S.nu = nu_aa.copy(); 
            w2,x2,res2,K2 = S.step(); f2 = S.nu - nu_aa
            if np.linalg.norm(f2) <= np.linalg.norm(f):
                naa+=1; G=[S.nu.copy()]; F=[f2]   # accepted: continue from T(nu_aa); restart memory
                nu_prev=nu_aa; w,x,res,Kxq = w2,x2,res2,K2
            else:
                nrej+=1; S.nu = S2nu; G=[];F=[]
    if it % 50 == 0:
        rp,rd,obj = resid(S, nu_prev, w,x,res,Kxq)
        if it % 2000 == 0 or (rp<=1e-6 and rd<=1e-6): print("it %6d rp %.2e rd %.2e obj %.8f sigma %.4g aa %d rej %d  %.0fs"%(it,rp,rd,obj,S.sigma,naa,nrej,time.time()-t), flush=True)
        if rp<=1e-6 and rd<=1e-6: break
        if it>=next_adapt:
            next_adapt=max(it+100, it*3//2)
            ratio=np.sqrt(max(rp,1e-300)/max(rd,1e-300))
            if ratio>1.5 or ratio<0.67: S.set_sigma(S.sigma*min(max(ratio,0.2),5.0)); G=[];F=[]
