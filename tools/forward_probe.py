#!/usr/bin/env python3
"""(diagnostic) the sampled forward pass alone, for rocprofv3: 5 launches of k_forward_mfma on 1e5 samples of W40-D20 / W40-D40."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers, nnsdp_amd as na
from nnsdp_amd import frontend as F
for name in ("W40-D20", "W40-D40"):
    d = helpers.load_problem(name, 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    X = 0.5 + np.random.default_rng(1).random((2, 100000))
    ms = [F.evalFeedFwdNetBatch(net, X, return_ms=True)[1] for _ in range(5)]
    fl = 2.0 * 1e5 * sum(net.xdims[k + 1] * (net.xdims[k] + 1) for k in range(net.K))
    print(name, "kernel ms", [round(m, 3) for m in ms], f"best {fl / min(ms) / 1e9:.2f} TFLOP/s fp64", flush=True)
