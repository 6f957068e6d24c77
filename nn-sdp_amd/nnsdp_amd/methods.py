"""Host-side mirror of the reference's Methods / Qc interface for the hot path.

Same names, argument meaning and error behaviour as the Julia modules so that parity tests read
like the reference's own call sites:

  FeedFwdNet                     src/MyNeuralNetwork/MyNeuralNetwork.jl:12-27
  QcInputBox                     src/Qc/input.jl:3-8
  QcSafety / QcReachHplane / QcReachCircle / QcReachEllipsoid   src/Qc/output.jl:3-31
  QcActivBounded / QcActivSector src/Qc/activ_bounded.jl:3-10, src/Qc/activ_sector.jl:2-20
  SafetyQuery / ReachQuery / QuerySolution   src/Methods/Methods.jl:19-55
  AdmmSdpOptions <: QueryOptions replaces ChordalSdpOptions (src/Methods/chordal_sdp.jl:10-16)
  runQuery(query, opts)          src/Methods/Methods.jl:91-131   <- the drop-in boundary
  makeCliques                    src/Methods/chordal_cliques.jl:13-59
  makeZ(gamma...)                Zin + Zout + sum(Zacs), src/Qc/{input,output,activ}.jl

Everything numeric happens in libnnsdp_hip.so through ctypes (nnsdp_amd/_lib.py); nothing here
imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib

QUERY_SAFETY, QUERY_REACH = 0, 1
ACTIV_RELU, ACTIV_TANH = 0, 1


class ReluActiv:
    """MyNeuralNetwork.ReluActiv (src/MyNeuralNetwork/MyNeuralNetwork.jl:8)"""
    code = ACTIV_RELU


class TanhActiv:
    """MyNeuralNetwork.TanhActiv (src/MyNeuralNetwork/MyNeuralNetwork.jl:9)"""
    code = ACTIV_TANH


def _activ_code(a) -> int:
    code = getattr(a, "code", a)
    if code not in (ACTIV_RELU, ACTIV_TANH):
        raise ValueError(f"unsupported activation: {a}")
    return int(code)
OUT_SAFETY_S, OUT_HPLANE, OUT_CIRCLE, OUT_ELLIPSOID = 0, 1, 2, 3


class SingleDecomp:
    code = 1


class DoubleDecomp:
    code = 2


class PathDecomp:
    """extension: cliques {x_k, x_{k+1}, affine}; exact when the output QC has no x_1 -- x_K coupling."""
    code = 3


class AutoDecomp:
    """extension: PathDecomp when the query allows it (no x_1 -- x_K coupling in the output QC: every reach query, hyperplane safety
    sets), DoubleDecomp otherwise.  The default stays the reference's own (`decomp_mode::DecompMode = SingleDecomp()`,
    src/Methods/chordal_sdp.jl:13)."""
    code = 4


DoubleRelaxDecomp = DoubleDecomp   # the reference treats the two identically (src/Methods/chordal_sdp.jl:8,25)


class DenseCone:
    """DeepSdpOptions' single dense cone (src/Methods/deep_sdp.jl:57)."""
    code = 0


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


@dataclass
class FeedFwdNet:
    xdims: List[int]
    Ms: List[np.ndarray]
    activ: Any = ReluActiv

    def __post_init__(self):
        _activ_code(self.activ)
        self.xdims = [int(v) for v in self.xdims]
        self.Ms = [_f64(M) for M in self.Ms]
        assert len(self.xdims) >= 3
        assert len(self.xdims) == self.K + 1
        for k in range(self.K):
            assert self.Ms[k].shape == (self.xdims[k + 1], self.xdims[k] + 1)

    @property
    def K(self) -> int:
        return len(self.Ms)

    @property
    def zdims(self) -> List[int]:
        return self.xdims[:-1] + [1]

    @property
    def Zdim(self) -> int:
        return sum(self.xdims[:-1]) + 1

    @property
    def acdim(self) -> int:
        return sum(self.xdims[1:-1])


@dataclass
class QcInputBox:
    x1min: np.ndarray
    x1max: np.ndarray

    def __post_init__(self):
        self.x1min, self.x1max = _f64(self.x1min), _f64(self.x1max)
        assert len(self.x1min) == len(self.x1max)

    @property
    def vardim(self) -> int:
        return len(self.x1min)


@dataclass
class QcSafety:
    S: np.ndarray
    vardim: int = 0


@dataclass
class QcReachHplane:
    normal: np.ndarray
    vardim: int = 1


@dataclass
class QcReachCircle:
    yc: np.ndarray
    vardim: int = 1


@dataclass
class QcReachEllipsoid:
    invP: np.ndarray
    yc: np.ndarray
    vardim: int = 1


@dataclass
class QcActivBounded:
    acymin: np.ndarray
    acymax: np.ndarray

    def __post_init__(self):
        self.acymin, self.acymax = _f64(self.acymin), _f64(self.acymax)
        assert len(self.acymin) == len(self.acymax)
        assert np.all(self.acymin <= self.acymax)

    @property
    def vardim(self) -> int:
        return len(self.acymin)


@dataclass
class QcActivSector:
    acxdim: int
    beta: int
    smin: np.ndarray
    smax: np.ndarray
    activ: Any = ReluActiv

    def __post_init__(self):
        _activ_code(self.activ)
        self.smin, self.smax = _f64(self.smin), _f64(self.smax)
        assert self.acxdim == len(self.smin) == len(self.smax)
        assert 0 <= self.beta
        # `@assert smin <= smax` (activ_sector.jl:13) is Julia's lexicographic vector comparison; a NaN anywhere (the reference's
        # tanh branch produces 0/0 for a pre-activation bound that is exactly 0, activ_sector.jl:74-86) is rejected outright:
        # the lexicographic test stops at the first differing entry and would let it through into the operator
        assert not (np.isnan(self.smin).any() or np.isnan(self.smax).any()), "smin / smax contain NaN"
        assert self.smin.tolist() <= self.smax.tolist()

    @property
    def lamdim(self) -> int:
        return (self.beta + 1) * self.acxdim - self.beta * (self.beta + 1) // 2

    @property
    def vardim(self) -> int:
        # (activ isa ReluActiv) ? _λdim + 2 * acxdim : _λdim   (activ_sector.jl:19)
        return self.lamdim + (2 * self.acxdim if _activ_code(self.activ) == ACTIV_RELU else 0)


@dataclass
class SafetyQuery:
    ffnet: FeedFwdNet
    qc_input: QcInputBox
    qc_safety: QcSafety
    qc_activs: Sequence[Any]


@dataclass
class ReachQuery:
    ffnet: FeedFwdNet
    qc_input: QcInputBox
    qc_reach: Any
    qc_activs: Sequence[Any]
    obj_func: Any = None          # x -> x[1] in every reference call site (NnSdp.jl:46,66,87)


@dataclass
class AdmmSdpOptions:
    """`AdmmSdpOptions <: QueryOptions`; `verbose` is read by runQuery (Methods.jl:110)."""
    decomp_mode: Any = field(default_factory=SingleDecomp)
    max_iters: int = 20000
    eps_rel: float = 1e-6
    max_time: float = 0.0
    sigma: float = 0.1
    alpha: float = 1.6
    adapt_every: int = 50
    check_every: int = 50
    normalize: bool = True
    warm_start: bool = True
    proj_tol: float = 0.0          # 0 = adaptive (see include/nnsdp.h)
    polish: bool = True            # exact-feasibility polish of the returned certificate
    cert_tol: float = 0.0          # > 0: early stop on the certified objective (see include/nnsdp.h)
    verbose: bool = False
    device: int = -1
    interval_guard: float = 5e-5   # relative floor on neuron interval half-widths inside the solver (see include/nnsdp.h)
    minv_mode: int = 0             # 0 auto, 1 dense M^-1, 2 structured M^-1 (block-banded by layer + low rank; see include/nnsdp.h)
    proj_refine: int = 1           # GEMM-only refinement of the persistent eigenbasis in front of the Jacobi sweeps: 0 off, 1, 2 (see include/nnsdp.h)

    def to_c(self) -> _lib.Options:
        o = _lib.Options()
        _lib.load().nnsdp_default_options(C.byref(o))
        o.decomp_mode = int(getattr(self.decomp_mode, "code", self.decomp_mode))
        o.max_iters = int(self.max_iters)
        o.eps_rel = float(self.eps_rel)
        o.max_time = float(self.max_time)
        o.sigma = float(self.sigma)
        o.alpha = float(self.alpha)
        o.adapt_every = int(self.adapt_every)
        o.check_every = int(self.check_every)
        o.normalize = int(bool(self.normalize))
        o.warm_start = int(bool(self.warm_start))
        o.proj_tol = float(self.proj_tol)
        o.polish = int(bool(self.polish))
        o.cert_tol = float(self.cert_tol)
        o.verbose = int(bool(self.verbose))
        o.device = int(self.device)
        o.interval_guard = float(self.interval_guard)
        o.minv_mode = int(self.minv_mode)
        o.proj_refine = int(self.proj_refine)
        return o


@dataclass
class QuerySolution:
    objective_value: float
    values: Dict[str, np.ndarray]
    termination_status: str
    total_time: float
    setup_time: float
    solve_time: float
    summary: Dict[str, Any]
    model: Any = None


class _CProblem:
    """Owns the numpy buffers a nnsdp_problem points into."""

    def __init__(self, query):
        net = query.ffnet
        self.keep = []
        p = _lib.Problem()
        p.K = net.K
        self.xdims = np.asarray(net.xdims, dtype=np.int32)
        p.xdims = self.xdims.ctypes.data_as(_lib.c_int32_p)
        # Julia layout: each M_k column-major, back to back
        self.M = np.concatenate([np.asfortranarray(M).ravel(order="F") for M in net.Ms]).astype(np.float64)
        p.M = self._ptr(self.M)
        qb = [q for q in query.qc_activs if isinstance(q, QcActivBounded)]
        qs = [q for q in query.qc_activs if isinstance(q, QcActivSector)]
        if len(qb) != 1 or len(qs) != 1:
            raise ValueError("qc_activs must hold one QcActivBounded and one QcActivSector (Qc/activ.jl:45-66)")
        qb, qs = qb[0], qs[0]
        if qb.vardim != net.acdim or qs.acxdim != net.acdim:
            raise ValueError("activation QC dimension does not match the network")
        p.x1min = self._ptr(_f64(query.qc_input.x1min))
        p.x1max = self._ptr(_f64(query.qc_input.x1max))
        if query.qc_input.vardim != net.xdims[0]:
            raise ValueError("input box dimension does not match the network")
        p.acymin, p.acymax = self._ptr(qb.acymin), self._ptr(qb.acymax)
        p.smin, p.smax = self._ptr(qs.smin), self._ptr(qs.smax)
        p.beta = int(qs.beta)
        if _activ_code(qs.activ) != _activ_code(net.activ):
            raise ValueError("QcActivSector.activ does not match ffnet.activ")
        p.activ = _activ_code(net.activ)
        m = net.xdims[-1]
        if isinstance(query, ReachQuery):
            p.query_kind = QUERY_REACH
            qo = query.qc_reach
            if isinstance(qo, QcReachHplane):
                p.out_kind = OUT_HPLANE
                assert len(qo.normal) == m
                p.normal = self._ptr(_f64(qo.normal))
            elif isinstance(qo, QcReachCircle):
                p.out_kind = OUT_CIRCLE
                assert len(qo.yc) == m
                p.yc = self._ptr(_f64(qo.yc))
            elif isinstance(qo, QcReachEllipsoid):
                p.out_kind = OUT_ELLIPSOID
                assert len(qo.yc) == m
                p.yc = self._ptr(_f64(qo.yc))
                p.invP = self._ptr(np.asfortranarray(_f64(qo.invP)).ravel(order="F").copy())
            else:
                raise ValueError(f"unrecognized qc: {qo}")
        elif isinstance(query, SafetyQuery):
            p.query_kind = QUERY_SAFETY
            p.out_kind = OUT_SAFETY_S
            sd = net.xdims[0] + m + 1
            S = _f64(query.qc_safety.S)
            assert S.shape == (sd, sd)
            p.S = self._ptr(np.asfortranarray(S).ravel(order="F").copy())
        else:
            raise ValueError(f"unrecognized query: {query}")
        self.p = p
        self.net = net
        self.nin = net.xdims[0]
        self.nout = 1 if p.query_kind == QUERY_REACH else 0
        self.n1 = net.acdim
        self.n2 = qs.vardim

    def _ptr(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.keep.append(a)
        return a.ctypes.data_as(_lib.c_double_p)

    @property
    def ngamma(self) -> int:
        return self.nin + self.nout + self.n1 + self.n2


def _alloc_result(cp: _CProblem, want_Z: bool = True):
    r = _lib.Result()
    Zdim = cp.net.Zdim
    bufs = {
        "gin": np.zeros(cp.nin), "gout": np.zeros(max(cp.nout, 1)),
        "gac1": np.zeros(cp.n1), "gac2": np.zeros(cp.n2),
        "Z": np.zeros((Zdim, Zdim), order="F") if want_Z else None,
    }
    r.gamma_in = bufs["gin"].ctypes.data_as(_lib.c_double_p)
    r.gamma_out = bufs["gout"].ctypes.data_as(_lib.c_double_p)
    r.gamma_ac1 = bufs["gac1"].ctypes.data_as(_lib.c_double_p)
    r.gamma_ac2 = bufs["gac2"].ctypes.data_as(_lib.c_double_p)
    if want_Z:
        r.Z = bufs["Z"].ctypes.data_as(_lib.c_double_p)
    return r, bufs


def _solution(cp: _CProblem, r, bufs) -> QuerySolution:
    lib = _lib.load()
    values = {"γin": bufs["gin"], "γac1": bufs["gac1"], "γac2": bufs["gac2"], "Z": bufs["Z"]}
    if cp.nout:
        values["γout"] = bufs["gout"][:1]
    summary = dict(iters=r.iters, pres=r.pres, dres=r.dres, lambda_max=r.lambda_max, t_eig=r.t_eig,
                   n_cliques=r.n_cliques, max_clique=r.max_clique,
                   eig_flops_per_iter=r.eig_flops_per_iter, eig_bytes_per_iter=r.eig_bytes_per_iter,
                   avg_sweeps=r.avg_sweeps, objective_admm=r.objective_admm, polish_shift=r.polish_shift,
                   refine_blocks=[int(v) for v in r.refine_blocks])
    return QuerySolution(objective_value=r.objective, values=values,
                         termination_status=lib.nnsdp_status_string(r.status).decode(),
                         total_time=r.t_total, setup_time=r.t_setup, solve_time=r.t_solve, summary=summary)


def _apply_obj_func(query, soln: QuerySolution) -> QuerySolution:
    """ReachQuery.obj_func (Methods.jl:41): affine and increasing in γout[1] at every reference call site (x -> x[1]); any
    such function has the same minimiser, so it is evaluated on the solution rather than handed to the solver."""
    f = getattr(query, "obj_func", None)
    if f is not None and isinstance(query, ReachQuery):
        if not f(np.array([1.0])) > f(np.array([0.0])):
            raise ValueError("obj_func must be increasing in γout[1]")
        soln.objective_value = float(f(soln.values["γout"]))
    return soln


def runQuery(query, opts: AdmmSdpOptions) -> QuerySolution:
    """Methods.runQuery(query, opts) with opts::AdmmSdpOptions (src/Methods/Methods.jl:91-131)."""
    lib = _lib.load()
    cp = _CProblem(query)
    o = opts.to_c()
    r, bufs = _alloc_result(cp)
    _lib.check(lib.nnsdp_solve(C.byref(cp.p), C.byref(o), C.byref(r)))
    soln = _apply_obj_func(query, _solution(cp, r, bufs))
    if opts.verbose:
        print(f"setup: {soln.setup_time:.3f} \tsolve: {soln.solve_time:.3f} \ttotal: {soln.total_time:.3f} \t"
              f"obj: {soln.objective_value:.5f} ({soln.termination_status}) \tλmax: {r.lambda_max:.7f}")
    return soln


solveQuery = runQuery   # NnSdp.solveQuery (src/NnSdp.jl:25-29)


class Solver:
    """Handle form (nnsdp_solver_*): lets a caller time exactly K iterations (bench.py)."""

    def __init__(self, query, opts: AdmmSdpOptions):
        self.lib = _lib.load()
        self.query = query
        self.cp = _CProblem(query)
        self.o = opts.to_c()
        self.h = C.c_void_p()
        _lib.check(self.lib.nnsdp_solver_create(C.byref(self.cp.p), C.byref(self.o), C.byref(self.h)))

    def iterate(self, iters: int, time_eig: bool = False) -> float:
        ms = C.c_double(0.0)
        _lib.check(self.lib.nnsdp_solver_iterate(self.h, int(iters), C.byref(ms) if time_eig else None))
        return ms.value

    def advance(self, iters: int) -> None:
        """`iters` iterations with the solve loop's checks and sigma / tolerance adaptation, without stopping."""
        _lib.check(self.lib.nnsdp_solver_advance(self.h, int(iters)))

    def set_comm(self, nranks: int, rank: int, unique_id: bytes) -> None:
        """clique-sharded mode: call on every rank with rank 0's id (comm_unique_id) before iterating."""
        assert len(unique_id) == 128
        _lib.check(self.lib.nnsdp_solver_set_comm(self.h, int(nranks), int(rank), unique_id))

    def set_comm_callback(self, nranks: int, rank: int, allreduce) -> None:
        """clique-sharded mode over the caller's own collective: allreduce(a) sums the float64 numpy array `a` over all ranks
        IN PLACE (e.g. torch.distributed.all_reduce(torch.from_numpy(a)) on a gloo group)."""
        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
                return 0
            except Exception:      # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._ar_cb = _lib.ALLREDUCE_FN(_cb)       # keep the trampoline alive as long as the handle
        _lib.check(self.lib.nnsdp_solver_set_comm_callback(self.h, int(nranks), int(rank), self._ar_cb, None))

    def set_comm_ipc(self, nranks: int, rank: int, allreduce) -> None:
        """clique-sharded mode with the device-side exchange over hipIpc-mapped peer buffers (ranks = processes of one node);
        `allreduce` as in set_comm_callback: used for the set-up and the check iterations' control decisions only."""
        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
                return 0
            except Exception:      # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._ar_cb = _lib.ALLREDUCE_FN(_cb)
        _lib.check(self.lib.nnsdp_solver_set_comm_ipc(self.h, int(nranks), int(rank), self._ar_cb, None))

    def iterate_async(self, iters: int) -> None:
        _lib.check(self.lib.nnsdp_solver_iterate_async(self.h, int(iters)))

    def sync(self) -> None:
        _lib.check(self.lib.nnsdp_solver_sync(self.h))

    def apply_minv(self, q):
        """(M^-1 q, structured?, operand bytes): the Woodbury core applied to a full-length multiplier vector (test entry)."""
        q = _f64(q)
        if len(q) != self.cp.ngamma:
            raise ValueError("q must have one entry per multiplier")
        out = np.zeros_like(q)
        st, nb = C.c_int32(), C.c_int64()
        _lib.check(self.lib.nnsdp_solver_apply_minv(self.h, q.ctypes.data_as(_lib.c_double_p), out.ctypes.data_as(_lib.c_double_p),
                                                    C.byref(st), C.byref(nb)))
        return out, bool(st.value), int(nb.value)

    def raw_multipliers(self) -> np.ndarray:
        """the multiplier block of the solver's fixed-point variable (solver coordinates; diagnostic / test entry)"""
        out = np.zeros(self.cp.ngamma)
        _lib.check(self.lib.nnsdp_solver_raw_multipliers(self.h, out.ctypes.data_as(_lib.c_double_p)))
        return out

    def info(self, what: int) -> float:
        """nnsdp_solver_info: 0 hipGraph launches, 1 RCCL all-reduce capturable into a hipGraph, 2 sharded, 3 iterations, 4 blocks, 5 largest block, 6 hipIpc transport (2 = fine-grained exchange buffers)"""
        v = C.c_double()
        _lib.check(self.lib.nnsdp_solver_info(self.h, int(what), C.byref(v)))
        return v.value

    def residuals(self):
        a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.nnsdp_solver_residuals(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def run(self) -> QuerySolution:
        r, bufs = _alloc_result(self.cp)
        _lib.check(self.lib.nnsdp_solver_run(self.h, C.byref(r)))
        return _apply_obj_func(self.query, _solution(self.cp, r, bufs))

    def finish(self) -> QuerySolution:
        r, bufs = _alloc_result(self.cp)
        _lib.check(self.lib.nnsdp_solver_finish(self.h, C.byref(r)))
        return _apply_obj_func(self.query, _solution(self.cp, r, bufs))

    def close(self):
        if self.h:
            self.lib.nnsdp_solver_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shardPlan(query, opts: AdmmSdpOptions, nranks: int):
    """(block dimensions, start) - the PSD blocks the solver works on and their contiguous partition over `nranks` ranks
    (rank r owns blocks start[r] .. start[r+1]-1), as Solver.set_comm uses it.  Host only: needs no GPU."""
    lib = _lib.load()
    cp = _CProblem(query)
    o = opts.to_c()
    n = C.c_int32()
    _lib.check(lib.nnsdp_shard_plan(C.byref(cp.p), C.byref(o), int(nranks), C.byref(n), None, None))
    bn = np.zeros(n.value, dtype=np.int32)
    st = np.zeros(int(nranks) + 1, dtype=np.int32)
    _lib.check(lib.nnsdp_shard_plan(C.byref(cp.p), C.byref(o), int(nranks), C.byref(n), bn.ctypes.data_as(_lib.c_int32_p),
                                    st.ctypes.data_as(_lib.c_int32_p)))
    return bn.tolist(), st.tolist()


def comm_unique_id() -> bytes:
    """RCCL unique id (128 bytes) for the clique-sharded mode; generate on rank 0, broadcast to all ranks."""
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().nnsdp_comm_unique_id(buf))
    return buf.raw


class SolverBatch:
    """Independent SDPs advanced in lockstep on one GPU: the beta sweep of experiments/scale.jl:28, the hyperplane
    directions of NnSdp.findReach2Dpoly (src/NnSdp.jl:73-95), the sub-queries of an ACAS clause.  A single W40-D20
    SDP occupies 19 of the 256 CUs; the batch handle (nnsdp_batch_*) issues ONE launch per stage for all SDPs.
    `iterate_streams` is the older form, one HIP stream per SDP (set GPU_MAX_HW_QUEUES, e.g. 16, before HIP
    initialises: with the default 4 hardware queues at most 4 streams make progress at a time)."""

    def __init__(self, queries, opts):
        optl = list(opts) if isinstance(opts, (list, tuple)) else [opts] * len(queries)
        if len(optl) != len(queries):
            raise ValueError("one options object per query (or a single one for all)")
        self.solvers = [Solver(q, o) for q, o in zip(queries, optl)]
        self.lib = _lib.load()
        self.h = C.c_void_p()
        hs = (C.c_void_p * len(self.solvers))(*[s.h for s in self.solvers])
        _lib.check(self.lib.nnsdp_batch_create(hs, len(self.solvers), C.byref(self.h)))

    def _resync(self) -> None:
        _lib.check(self.lib.nnsdp_batch_resync(self.h))

    def advance(self, iters: int) -> None:
        for s in self.solvers:
            s.advance(iters)
        self._resync()

    def iterate(self, iters: int) -> None:
        """exactly `iters` plain iterations of every SDP, one launch per stage for the whole batch."""
        _lib.check(self.lib.nnsdp_batch_iterate(self.h, int(iters)))

    def iterate_streams(self, iters: int, chunk: int = 64) -> None:
        done = 0
        while done < iters:
            n = min(chunk, iters - done)
            for s in self.solvers:
                s.iterate_async(n)
            done += n
        for s in self.solvers:
            s.sync()
        self._resync()

    def run(self) -> List[QuerySolution]:
        """full solves (stopping rules of runQuery, each SDP on its own) -> one QuerySolution per query."""
        st = (C.c_int32 * len(self.solvers))()
        _lib.check(self.lib.nnsdp_batch_run(self.h, st))
        out = []
        for s, code in zip(self.solvers, st):
            r, bufs = _alloc_result(s.cp)
            _lib.check(self.lib.nnsdp_solver_finish_status(s.h, int(code), C.byref(r)))
            out.append(_apply_obj_func(s.query, _solution(s.cp, r, bufs)))
        return out

    def residuals(self):
        """one check iteration of every member (each advances by one iteration), then lockstep is re-established"""
        out = [s.residuals() for s in self.solvers]
        self._resync()
        return out

    def finish(self):
        return [s.finish() for s in self.solvers]

    def close(self):
        if self.h:
            self.lib.nnsdp_batch_destroy(self.h)
            self.h = C.c_void_p()
        for s in self.solvers:
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def runQueries(queries, opts) -> List[QuerySolution]:
    """runQuery for several independent queries at once (batch handle); results in the order of the queries."""
    sb = SolverBatch(queries, opts)
    try:
        return sb.run()
    finally:
        sb.close()


def makeZ(query, gamma) -> np.ndarray:
    """Z = Zin + Zout + sum(Zacs) for numeric gamma = [γin; γout; γac1; γac2] (GPU)."""
    lib = _lib.load()
    cp = _CProblem(query)
    g = _f64(gamma)
    if len(g) != cp.ngamma:
        raise ValueError(f"gamma has length {len(g)}, expected {cp.ngamma}")
    Z = np.zeros((cp.net.Zdim, cp.net.Zdim), order="F")
    _lib.check(lib.nnsdp_assemble_Z(C.byref(cp.p), g.ctypes.data_as(_lib.c_double_p), Z.ctypes.data_as(_lib.c_double_p)))
    return Z


def adjoint(query, X) -> np.ndarray:
    """[<G_i, X>]_i for every multiplier (GPU)."""
    lib = _lib.load()
    cp = _CProblem(query)
    Xf = np.asfortranarray(_f64(X))
    out = np.zeros(cp.ngamma)
    _lib.check(lib.nnsdp_adjoint(C.byref(cp.p), Xf.ctypes.data_as(_lib.c_double_p), out.ctypes.data_as(_lib.c_double_p)))
    return out


def makeCliques(xdims: Sequence[int], beta: int, decomp_mode=SingleDecomp) -> List[List[int]]:
    """Index sets of the PSD blocks (0-based), chordal_cliques.jl:13-59 + chordal_sdp.jl:19-57.
    Pure host index arithmetic: works without a GPU."""
    lib = _lib.load()
    xd = np.asarray(xdims, dtype=np.int32)
    K = len(xd) - 1
    n, tot = C.c_int32(), C.c_int32()
    mode = int(getattr(decomp_mode, "code", decomp_mode))
    _lib.check(lib.nnsdp_make_cliques(K, xd.ctypes.data_as(_lib.c_int32_p), int(beta), mode, C.byref(n), C.byref(tot), None, None))
    ptr = np.zeros(n.value + 1, dtype=np.int32)
    idx = np.zeros(tot.value, dtype=np.int32)
    _lib.check(lib.nnsdp_make_cliques(K, xd.ctypes.data_as(_lib.c_int32_p), int(beta), mode, C.byref(n), C.byref(tot),
                                      ptr.ctypes.data_as(_lib.c_int32_p), idx.ctypes.data_as(_lib.c_int32_p)))
    return [idx[ptr[k]:ptr[k + 1]].tolist() for k in range(n.value)]


def project_psd_batched(mats: Sequence[np.ndarray]):
    """Batched PSD projection (the hot kernel) of symmetric matrices with n <= 128.
    Returns (projections, eigenvalue arrays, kernel milliseconds)."""
    lib = _lib.load()
    if len(mats) == 0:
        return [], [], 0.0
    ns = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
    for m in mats:
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("matrices must be square")
    flat = np.concatenate([np.asfortranarray(_f64(m)).ravel(order="F") for m in mats])
    out = np.zeros_like(flat)
    ev = np.zeros(int(ns.sum()))
    ms = C.c_double()
    _lib.check(lib.nnsdp_project_psd_batched(len(mats), ns.ctypes.data_as(_lib.c_int32_p), flat.ctypes.data_as(_lib.c_double_p),
                                             out.ctypes.data_as(_lib.c_double_p), ev.ctypes.data_as(_lib.c_double_p), C.byref(ms)))
    res, evs, o, eo = [], [], 0, 0
    for n in ns:
        res.append(out[o:o + n * n].reshape(n, n, order="F").copy())
        evs.append(ev[eo:eo + n].copy())
        o += n * n
        eo += n
    return res, evs, ms.value


def project_psd_warm(mats: Sequence[np.ndarray], bases: Sequence[np.ndarray], tol: float, refine: bool = True, state: Optional[np.ndarray] = None):
    """The projection kernel in its warm form (test entry): bases[b] holds the eigenbasis kept from the previous projection of
    block b (columns = eigenvectors); refine as AdmmSdpOptions.proj_refine (True = 1).  Returns (projections, updated bases, outcome
    counts [converged, one refinement step, sent on to the sweeps, not attempted, accepted after a checked step], kernel milliseconds).
    state: an int32 array of 4 x len(mats) zeros before the first call, updated in place - the refinement stage's per-block state as
    a solver carries it from one iteration to the next (nnsdp_project_psd_warm_state)."""
    lib = _lib.load()
    ns = np.asarray([m.shape[0] for m in mats], dtype=np.int32)
    flat = np.concatenate([np.asfortranarray(_f64(m)).ravel(order="F") for m in mats])
    vb = np.concatenate([np.asfortranarray(_f64(v)).ravel(order="F") for v in bases])
    if vb.shape != flat.shape:
        raise ValueError("one basis per matrix, same shapes")
    out = np.zeros_like(flat)
    oc = np.zeros(5, dtype=np.int32)
    ms = C.c_double()
    if state is not None and (state.dtype != np.int32 or state.shape != (4 * len(mats),) or not state.flags.c_contiguous):
        raise ValueError("state must be a contiguous int32 array of 4 x len(mats)")
    _lib.check(lib.nnsdp_project_psd_warm_state(len(mats), ns.ctypes.data_as(_lib.c_int32_p), flat.ctypes.data_as(_lib.c_double_p),
                                                vb.ctypes.data_as(_lib.c_double_p), float(tol), int(refine), out.ctypes.data_as(_lib.c_double_p),
                                                oc.ctypes.data_as(_lib.c_int32_p), C.byref(ms),
                                                state.ctypes.data_as(_lib.c_int32_p) if state is not None else None))
    res, vs, o = [], [], 0
    for n in ns:
        res.append(out[o:o + n * n].reshape(n, n, order="F").copy())
        vs.append(vb[o:o + n * n].reshape(n, n, order="F").copy())
        o += n * n
    return res, vs, oc.tolist(), ms.value
