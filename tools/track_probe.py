#!/usr/bin/env python3
"""Diagnostic for the small-side tracking projection (NNSDP_TRACK=1): solves one fixture with and without it and prints
iterations, wall time, objective and the tracked share (stderr of the library, NNSDP_TRACK_STATS=1).
usage: python tools/track_probe.py [W40-D20 0 single|double]   (run in two processes: the switch is read once per process)"""
import os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, helpers, nnsdp_amd as na
    name, beta, mode = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    q = helpers.product_query(helpers.load_problem(name, beta))
    dm = na.DoubleDecomp() if mode == "double" else na.SingleDecomp()
    for iters in (2000, 6000):
        t = time.time()
        r = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=dm, polish=False, max_iters=iters, eps_rel=1e-12))
        dt = time.time() - t
        print(f"track={os.environ.get('NNSDP_TRACK','0')} {name} b{beta} {mode}: {iters} iterations, solve {r.solve_time:.3f} s "
              f"= {iters / r.solve_time:.0f} it/s, admm objective {r.summary['objective_admm']!r}", flush=True)
    t = time.time()
    s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=dm, max_iters=400000, eps_rel=1e-6, max_time=120))
    print(f"track={os.environ.get('NNSDP_TRACK','0')} full solve: {s.termination_status} rho {s.objective_value!r} admm {s.summary['objective_admm']!r} iters {s.summary['iters']} in {time.time()-t:.2f} s", flush=True)
    sys.exit(0)
args = sys.argv[1:] or ["W40-D20", "0", "single"]
for trk in ("0", "1"):
    env = dict(os.environ, NNSDP_TRACK=trk, NNSDP_TRACK_STATS="1")
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + args, env=env, check=False)
