#!/usr/bin/env python3
"""(diagnostic) iterations to residuals 1e-6 / to the certified gap 1e-3 over the published rows + BASELINE's W40 configs, for the
penalty-adaptation rule selected by NNSDP_SIGMA_RULE (0 = round 1, 1 = smoothed ratio + back-off only after a change)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na
cases = [("W10-D10", 0), ("W10-D10", 7), ("W10-D20", 0), ("W10-D30", 0), ("W10-D50", 0), ("W10-D60", 0), ("W20-D10", 0), ("W20-D10", 7),
         ("W20-D20", 0), ("W20-D30", 0), ("W20-D50", 0), ("W20-D100", 0), ("W40-D20", 0), ("W40-D20", 2), ("W40-D40", 0)]
tot = {"residual_1e-6": 0, "certified_gap_1e-3": 0}
for name, beta in cases:
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    q, P, yc = na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], beta)
    out = []
    for rule, kw in (("residual_1e-6", dict()), ("certified_gap_1e-3", dict(cert_tol=1e-3))):
        s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=600000, max_time=120, eps_rel=1e-6, **kw))
        tot[rule] += s.summary["iters"]
        out.append(f"{rule} {s.summary['iters']:7d} {s.termination_status:8s} rho {s.objective_value:.8g}")
    print(f"{name:9s} b{beta}: " + " | ".join(out), flush=True)
print("rule", os.environ.get("NNSDP_SIGMA_RULE", "1"), "total iterations", tot, flush=True)
