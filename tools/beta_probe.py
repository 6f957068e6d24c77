#!/usr/bin/env python3
"""How the dense Woodbury core M^-1 (ng x ng, 8 ng^2 bytes read per iteration) scales over the reference's beta sweep
(experiments/scale.jl:28): set-up time, bytes of M^-1, iterations/s for W40-D20 / W40-D40 at beta = 0 and 7, Double decomposition.
(diagnostic; numbers quoted in DESIGN.md section 4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import helpers
import nnsdp_amd as na

cases = [(c.split(":")[0], int(c.split(":")[1])) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["W40-D20:0", "W40-D20:7", "W40-D40:0", "W40-D40:7"])]
for name, beta in cases:
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
    q, P, yc = na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], beta)
    t = time.time()
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=10 ** 9))
    t_setup = time.time() - t
    s.advance(500)
    s.iterate(64)
    t = time.time()
    n = 400
    s.iterate(n)
    dt = time.time() - t
    r = s.finish()
    ngam = len(r.values["γin"]) + 1 + len(r.values["γac1"]) + len(r.values["γac2"])
    print(f"{name} beta={beta}: ngamma {ngam}, blocks {r.summary['n_cliques']} (max {r.summary['max_clique']}), setup {t_setup:.2f} s, "
          f"{n / dt:.0f} it/s ({1e6 * dt / n:.0f} us/iteration, hipGraph replay)", flush=True)
    s.close()
