#!/usr/bin/env python3
"""(diagnostic) randomized stress of the projection kernel against LAPACK: random sizes 41..128, random spectra (clustered,
rank-deficient, wide dynamic range), cold starts.  Prints the worst relative error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import nnsdp_amd as na
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    nmax = int(rng.choice([60, 74, 75, 90, 91, 96, 110, 128]))
    mats = []
    for _ in range(64):
        n = int(rng.integers(41, nmax + 1))
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        kind = rng.integers(0, 5)
        if kind == 0: lam = rng.standard_normal(n)
        elif kind == 1: lam = np.repeat(rng.standard_normal(4), [n // 4] * 3 + [n - 3 * (n // 4)]) + 1e-9 * rng.standard_normal(n)
        elif kind == 2: lam = np.where(rng.random(n) < 0.5, 0.0, rng.standard_normal(n))
        elif kind == 3: lam = rng.standard_normal(n) * 10.0 ** rng.uniform(-8, 8, n)
        else: lam = np.sign(rng.standard_normal(n)) * (1.0 + 1e-6 * rng.standard_normal(n))
        A = (Q * lam) @ Q.T * 10.0 ** rng.uniform(-3, 3)
        mats.append(0.5 * (A + A.T))
    res, evs, _ = na.project_psd_batched(mats)
    for A, P, ev in zip(mats, res, evs):
        w, U = np.linalg.eigh(A)
        ref = (U * np.maximum(w, 0)) @ U.T
        nrm = np.abs(A).max()
        e = max(np.abs(P - ref).max(), np.abs(np.sort(ev) - w).max()) / nrm
        if not np.isfinite(e): e = np.inf
        worst = max(worst, e)
print("worst relative error", worst)
