#!/usr/bin/env python3
"""(diagnostic) sensitivity of the certified rho of findEllipsoid to the tightness of the interval pre-processing: our CROWN-sliced
bounds, the same bounds widened by a factor, and plain interval arithmetic.  Brackets the published values (dump/scale)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F
name, beta = sys.argv[1], int(sys.argv[2])
d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
xd = [int(v) for v in d["xdims"]]
net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
lo, hi = np.array([0.5, 0.5]), np.array([1.5, 1.5])
pub = helpers.published_rho(name, beta)
opts = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=400000, eps_rel=1e-6, max_time=60)
q0, P, yc = na.ellipsoidQuery(net, lo, hi, beta)


def ibp():
    l, u = lo, hi
    ymin, ymax, amin, amax = [], [], [], []
    for k, Mk in enumerate(net.Ms[:-1]):
        W, b = Mk[:, :-1], Mk[:, -1]
        pl = np.maximum(W, 0) @ l + np.minimum(W, 0) @ u + b
        pu = np.maximum(W, 0) @ u + np.minimum(W, 0) @ l + b
        amin.append(pl); amax.append(pu)
        l, u = np.maximum(pl, 0), np.maximum(pu, 0)
        ymin.append(l); ymax.append(u)
    return map(np.concatenate, (ymin, ymax, amin, amax))


def solve(acymin, acymax, smin, smax, tag):
    qa = [na.QcActivBounded(acymin=acymin, acymax=acymax), na.QcActivSector(acxdim=len(acymin), beta=beta, smin=smin, smax=smax)]
    q = na.ReachQuery(ffnet=net, qc_input=q0.qc_input, qc_reach=q0.qc_reach, qc_activs=qa)
    s = na.runQuery(q, opts)
    rel = (s.objective_value - min(pub)) / min(pub)
    print(f"{tag}: rho {s.objective_value:.8f} {s.termination_status}  (rho - published_min)/published_min = {rel:+.2e}", flush=True)


qb, qs = q0.qc_activs
print(f"{name} beta={beta}: published {min(pub):.8f} .. {max(pub):.8f}")
solve(qb.acymin, qb.acymax, qs.smin, qs.smax, "CROWN-sliced (ours)")
for f in (1.001, 1.01, 1.05, 1.2):
    c, r = 0.5 * (qb.acymin + qb.acymax), 0.5 * (qb.acymax - qb.acymin)
    solve(np.maximum(c - f * r, 0.0), c + f * r, qs.smin, qs.smax, f"post-activation widths x {f}")
ymin, ymax, amin, amax = ibp()
solve(ymin, ymax, (amin > 1e-4).astype(float), 1.0 - (amax < -1e-4).astype(float), "interval arithmetic (IBP)")
solve(qb.acymin, qb.acymax, np.zeros_like(qs.smin), np.ones_like(qs.smax), "CROWN bounds, sector flags all [0,1]")
