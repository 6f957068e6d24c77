"""ctypes binding of libnnsdp_hip.so (include/nnsdp.h).  No torch types cross this boundary.

The library is built in-tree by __graft_entry__.build() (hipcc --offload-arch=gfx950).  There is
no CPU fallback: if the shared object is missing, or no HIP device is present when a compute
entry point is called, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnnsdp_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class NnsdpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libnnsdp_hip error {code}: {msg}")
        self.code = code


class Problem(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("xdims", c_int32_p), ("M", c_double_p),
        ("x1min", c_double_p), ("x1max", c_double_p),
        ("acymin", c_double_p), ("acymax", c_double_p),
        ("smin", c_double_p), ("smax", c_double_p),
        ("beta", C.c_int32), ("query_kind", C.c_int32), ("out_kind", C.c_int32),
        ("normal", c_double_p), ("yc", c_double_p), ("invP", c_double_p), ("S", c_double_p),
        ("activ", C.c_int32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("decomp_mode", C.c_int32), ("max_iters", C.c_int32), ("eps_rel", C.c_double),
        ("max_time", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
        ("adapt_every", C.c_int32), ("check_every", C.c_int32), ("normalize", C.c_int32),
        ("warm_start", C.c_int32), ("proj_tol", C.c_double), ("polish", C.c_int32), ("cert_tol", C.c_double), ("verbose", C.c_int32), ("device", C.c_int32),
        ("interval_guard", C.c_double), ("minv_mode", C.c_int32), ("proj_refine", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("gamma_in", c_double_p), ("gamma_out", c_double_p), ("gamma_ac1", c_double_p),
        ("gamma_ac2", c_double_p), ("Z", c_double_p),
        ("objective", C.c_double), ("status", C.c_int32), ("iters", C.c_int32),
        ("pres", C.c_double), ("dres", C.c_double), ("lambda_max", C.c_double),
        ("t_setup", C.c_double), ("t_solve", C.c_double), ("t_total", C.c_double), ("t_eig", C.c_double),
        ("n_cliques", C.c_int32), ("max_clique", C.c_int32),
        ("eig_flops_per_iter", C.c_int64), ("eig_bytes_per_iter", C.c_int64), ("avg_sweeps", C.c_double), ("objective_admm", C.c_double), ("polish_shift", C.c_double),
        ("refine_blocks", C.c_int64 * 5),
    ]


# every symbol include/nnsdp.h declares: (name, restype, argtypes)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)   # nnsdp_allreduce_fn

SYMBOLS = [
    ("nnsdp_version", C.c_int, []),
    ("nnsdp_last_error", C.c_char_p, []),
    ("nnsdp_status_string", C.c_char_p, [C.c_int32]),
    ("nnsdp_default_options", None, [C.POINTER(Options)]),
    ("nnsdp_problem_dims", C.c_int, [C.POINTER(Problem), c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ("nnsdp_solve", C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.POINTER(Result)]),
    ("nnsdp_solver_create", C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    ("nnsdp_solver_iterate", C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    ("nnsdp_solver_advance", C.c_int, [C.c_void_p, C.c_int32]),
    ("nnsdp_solver_iterate_async", C.c_int, [C.c_void_p, C.c_int32]),
    ("nnsdp_solver_sync", C.c_int, [C.c_void_p]),
    ("nnsdp_solver_residuals", C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("nnsdp_solver_apply_minv", C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int32_p, C.POINTER(C.c_int64)]),
    ("nnsdp_solver_raw_multipliers", C.c_int, [C.c_void_p, c_double_p]),
    ("nnsdp_solver_info", C.c_int, [C.c_void_p, C.c_int32, c_double_p]),
    ("nnsdp_solver_run", C.c_int, [C.c_void_p, C.POINTER(Result)]),
    ("nnsdp_solver_finish", C.c_int, [C.c_void_p, C.POINTER(Result)]),
    ("nnsdp_solver_destroy", C.c_int, [C.c_void_p]),
    ("nnsdp_assemble_Z", C.c_int, [C.POINTER(Problem), c_double_p, c_double_p]),
    ("nnsdp_adjoint", C.c_int, [C.POINTER(Problem), c_double_p, c_double_p]),
    ("nnsdp_make_cliques", C.c_int, [C.c_int32, c_int32_p, C.c_int32, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ("nnsdp_batch_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    ("nnsdp_batch_iterate", C.c_int, [C.c_void_p, C.c_int32]),
    ("nnsdp_batch_run", C.c_int, [C.c_void_p, c_int32_p]),
    ("nnsdp_batch_destroy", C.c_int, [C.c_void_p]),
    ("nnsdp_batch_resync", C.c_int, [C.c_void_p]),
    ("nnsdp_solver_finish_status", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Result)]),
    ("nnsdp_eval_network", C.c_int, [C.c_int32, c_int32_p, c_double_p, C.c_int32, C.c_int64, c_double_p, c_double_p, c_double_p]),
    ("nnsdp_make_intervals", C.c_int, [C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p] + [c_double_p] * 8),
    ("nnsdp_make_intervals_activ", C.c_int, [C.c_int32, c_int32_p, c_double_p, C.c_int32, c_double_p, c_double_p] + [c_double_p] * 8),
    ("nnsdp_project_psd_batched", C.c_int, [C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("nnsdp_project_psd_warm", C.c_int, [C.c_int32, c_int32_p, c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p, c_int32_p, c_double_p]),
    ("nnsdp_project_psd_warm_state", C.c_int, [C.c_int32, c_int32_p, c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p, c_int32_p, c_double_p, c_int32_p]),
    ("nnsdp_comm_unique_id", C.c_int, [C.c_char_p]),
    ("nnsdp_shard_plan", C.c_int, [C.POINTER(Problem), C.POINTER(Options), C.c_int32, c_int32_p, c_int32_p, c_int32_p]),
    ("nnsdp_solver_set_comm", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    ("nnsdp_solver_set_comm_callback", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, ALLREDUCE_FN, C.c_void_p]),
    ("nnsdp_solver_set_comm_ipc", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, ALLREDUCE_FN, C.c_void_p]),
]

_lib = None


def load():
    """Load the shared library (once).  Raises if it was not built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nnsdp_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int):
    if code != 0:
        raise NnsdpError(code, load().nnsdp_last_error().decode("utf-8", "replace"))
