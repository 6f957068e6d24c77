"""ctypes driver of oracle/c/libadmm_cpu.so: the oracle's ADMM iteration (admm.py AdmmState.step) in C++ / OpenMP with
LAPACK dsyevd per clique - the same-box CPU baseline SURVEY.md section 8d(2) asks for.  TEST / MEASUREMENT
INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline leg); set-up stays in numpy (operator.py, admm.py), only the
iteration loop is compiled code."""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np
import scipy
import scipy.sparse as sp

from . import admm as oadmm

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "c", "libadmm_cpu.so")
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_longlong)
_bp = C.POINTER(C.c_ubyte)


class _Problem(C.Structure):
    _fields_ = [("NE", C.c_int), ("ng", C.c_int), ("ncl", C.c_int),
                ("csr_ptr", _ip), ("csr_col", _ip), ("csr_val", _dp), ("csc_ptr", _ip), ("csc_row", _ip), ("csc_val", _dp),
                ("z0", _dp), ("c", _dp), ("Dinv", _dp), ("Mchol", _dp), ("nk", _ip), ("off", _lp), ("gidx", _ip),
                ("sptr", _ip), ("soff", _lp), ("isdiag", _bp)]


def lapack_path() -> str:
    c = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so")))
    if not c:
        raise FileNotFoundError("no OpenBLAS found inside the scipy wheel (the image has no system LAPACK)")
    return os.path.realpath(c[0])


class CpuAdmm:
    """same state layout as oracle.admm.AdmmState (nu = [multipliers | clique matrices]); step(n) runs n iterations."""

    def __init__(self, P: oadmm.ScaledProblem, sigma: float = 0.1, alpha: float = 1.6, threads: int = 0):
        if not os.path.exists(LIB):
            raise FileNotFoundError(f"{LIB} not built (make -C oracle/c, or __graft_entry__.build())")
        self.lib = C.CDLL(LIB)
        self.lib.admm_cpu_init.argtypes = [C.c_char_p]
        self.lib.admm_cpu_run.argtypes = [C.POINTER(_Problem), C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp]
        if self.lib.admm_cpu_init(lapack_path().encode()) != 0:
            raise RuntimeError("admm_cpu_init failed")
        S = oadmm.AdmmState(P, sigma, alpha)          # numpy set-up: M factor, offsets, gather tables
        self.S, self.P, self.sigma, self.alpha, self.threads = S, P, sigma, alpha, threads
        pat = P.pat
        A = sp.csr_matrix(P.A)
        A.sort_indices()
        Ac = sp.csc_matrix(P.A)
        Ac.sort_indices()
        k = self.keep = {}
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        k["csr_ptr"], k["csr_col"], k["csr_val"] = i32(A.indptr), i32(A.indices), f64(A.data)
        k["csc_ptr"], k["csc_row"], k["csc_val"] = i32(Ac.indptr), i32(Ac.indices), f64(Ac.data)
        k["z0"], k["c"], k["Dinv"] = f64(P.z0), f64(P.c), f64(S.Dinv)
        import scipy.linalg as sla
        Minv = sla.cho_solve(S.Mfac, np.eye(S.ng))
        k["Mchol"] = np.ascontiguousarray(0.5 * (Minv + Minv.T))      # explicit symmetric inverse, as the HIP library holds it
        k["nk"] = i32(S.nk)
        k["off"] = np.ascontiguousarray(S.offs, dtype=np.int64)
        k["gidx"] = i32(np.concatenate([g for g in S.G]))
        # lower-triangle sources of every pattern entry
        src_e = np.concatenate([S.trilpos[q] for q in range(len(S.nk))])
        src_o = np.concatenate([S.offs[q] + S.tril[q][0] * S.nk[q] + S.tril[q][1] for q in range(len(S.nk))])
        order = np.argsort(src_e, kind="stable")
        k["soff"] = np.ascontiguousarray(src_o[order], dtype=np.int64)
        k["sptr"] = i32(np.concatenate([[0], np.cumsum(np.bincount(src_e, minlength=pat.NE))]))
        k["isdiag"] = np.ascontiguousarray(pat.rows == pat.cols, dtype=np.uint8)
        pr = _Problem()
        pr.NE, pr.ng, pr.ncl = pat.NE, S.ng, len(S.nk)
        for name, typ in (("csr_ptr", _ip), ("csr_col", _ip), ("csr_val", _dp), ("csc_ptr", _ip), ("csc_row", _ip), ("csc_val", _dp),
                          ("z0", _dp), ("c", _dp), ("Dinv", _dp), ("Mchol", _dp), ("nk", _ip), ("off", _lp), ("gidx", _ip),
                          ("sptr", _ip), ("soff", _lp), ("isdiag", _bp)):
            setattr(pr, name, k[name].ctypes.data_as(typ))
        self.pr = pr
        self.nu = S.nu.copy()
        self.stats = np.zeros(5)

    def step(self, iters: int = 1):
        rc = self.lib.admm_cpu_run(C.byref(self.pr), self.sigma, self.alpha, int(iters), int(self.threads),
                                   self.nu.ctypes.data_as(_dp), self.stats.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError(f"admm_cpu_run failed (LAPACK info {rc})")
        sc = self.P.zscale * self.P.cscale
        return dict(pres=self.stats[0], dres=self.stats[1], objective=self.stats[2] / sc, dual_objective=self.stats[3] / sc, seconds=self.stats[4])
