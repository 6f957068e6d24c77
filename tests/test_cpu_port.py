"""The C++/OpenMP CPU port of the ADMM iteration (oracle/c/admm_cpu.cpp, driven by oracle/admm_c.py; bench.py's
cpu_baseline leg) against the numpy oracle it restates: same iterates, same residuals.  Built by __graft_entry__.build()."""
import os

import numpy as np
import pytest

import helpers
from oracle import admm as oadmm, admm_c, operator as oop

pytestmark = pytest.mark.skipif(not os.path.exists(admm_c.LIB), reason="oracle/c/libadmm_cpu.so not built (run __graft_entry__.build())")


@pytest.mark.parametrize("name,beta,mode,iters", [("W10-D5", 0, "single", 120), ("W10-D5", 3, "double", 80), ("W10-D10", 0, "double", 60)])
def test_cpp_port_reproduces_the_numpy_iteration(name, beta, mode, iters):
    q = helpers.oracle_query(helpers.load_problem(name, beta))
    P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
    S = oadmm.AdmmState(P, 0.1, 1.6)
    Cc = admm_c.CpuAdmm(P, 0.1, 1.6, threads=2)
    for _ in range(iters):
        nu_prev = S.nu
        w, x, res, Kxq = S.step()
    st = Cc.step(iters)
    assert np.abs(Cc.nu - S.nu).max() <= 1e-11 * max(1.0, np.abs(S.nu).max())
    y = S.sigma * (nu_prev - w)
    rp = np.linalg.norm(res) / max(np.linalg.norm(Kxq), np.linalg.norm(w))
    Kty = S.Kt(y)
    rd = np.linalg.norm(Kty - P.z0) / max(np.linalg.norm(Kty), np.linalg.norm(P.z0))
    assert abs(st["pres"] - rp) <= 1e-9 * rp and abs(st["dres"] - rd) <= 1e-9 * rd
    obj = -(P.c @ y[:S.ng]) / (P.zscale * P.cscale)
    assert abs(st["objective"] - obj) <= 1e-9 * max(1.0, abs(obj))
