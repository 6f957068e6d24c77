/*
 * nnsdp.h -- C ABI of the MI355X-native Chordal-DeepSDP assembler + ADMM solver.
 *
 * This is the drop-in boundary for the hot path of AntonXue/nn-sdp.  The reference has no FFI
 * on this path; the seam is Julia multiple dispatch on the options type,
 *     Methods.runQuery(query::Query, opts::QueryOptions)      src/Methods/Methods.jl:91-131
 * which today builds a JuMP model (setupSafety!/setupReach!, src/Methods/deep_sdp.jl:10-61,
 * src/Methods/chordal_sdp.jl:96-153) and hands it to MOSEK (Methods.jl:61,64,83).  A Julia
 * method runQuery(query, opts::AdmmSdpOptions) ccall's nnsdp_solve() below instead
 * (INTEGRATION.md shows the stub); the Python mirror in nn-sdp_amd/nnsdp_amd binds the same
 * symbols through ctypes.
 *
 * Conventions: plain pointers and sizes only, caller owns every buffer, the library copies in,
 * writes results into caller memory and retains nothing after a call returns (handles excepted,
 * until destroyed).  All matrices are COLUMN-MAJOR Float64 (Julia layout).  Every function
 * returns 0 on success, <0 for an invalid argument, >0 for a runtime (HIP/rocSOLVER/RCCL)
 * failure; nnsdp_last_error() returns a thread-local message.  Nothing throws across the ABI.
 * Blocking calls; distinct handles may be used from distinct threads, one handle from one
 * thread at a time.
 *
 * WHAT "SAME RESULT AS THE REFERENCE" MEANS HERE (the parity contract; DESIGN.md section 7, tests/test_published_sweep.py on all 136
 * published (network, beta) pairs of dump/scale, tests/test_published_parity.py on 21 of them two-sided).  The returned (gamma, Z) is
 *   - FEASIBLE for the reference's own LMI with the caller's own interval bounds: gamma >= 0 exactly, eigmax(Z(gamma)) <= 1e-6 in the
 *     reference's coordinates (its own OPTIMAL rows: +1e-7 .. +5e-6), so `objective` is a valid bound whatever else holds;
 *   - NEVER LOOSER than what the reference published, up to the stopping rule: objective <= (1 + 2e-3) x the median of the three
 *     published values (DeepSDP, Chordal, Chordal-2) with cert_tol = 1e-3, <= (1 + 1e-3) x with eps_rel = 1e-6;
 *   - and NOT two-sided equal to them: the published objectives are MOSEK iterates accepted before optimality, one-signed, up to
 *     2.2 % ABOVE the optimum of their own LMI (an independent interior point encloses that optimum on the pinned rows and this library
 *     lands inside the enclosure: tests/test_oracle_ipm.py).  7 of 21 rows agree to 1e-3 two-sided, the others are strict xfails with
 *     their measured distance.  A caller that needs the reference's number reproduced to 1e-3 gets a tighter one instead.
 * Bit-exact parity holds where the path is integer / index work (cliques, selectors, gather lists) and 1e-12 relative for the
 * assembled LMI against the literal restatement of src/Qc (tests/test_gpu_parity.py).
 */
#ifndef NNSDP_H
#define NNSDP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNSDP_VERSION 200 /* 0.2.0 */

/* query_kind: Methods.SafetyQuery / Methods.ReachQuery (src/Methods/Methods.jl:22-43) */
enum { NNSDP_QUERY_SAFETY = 0, NNSDP_QUERY_REACH = 1 };
/* out_kind: Qc.QcSafety / QcReachHplane / QcReachCircle / QcReachEllipsoid (src/Qc/output.jl:3-31) */
enum { NNSDP_OUT_SAFETY_S = 0, NNSDP_OUT_HPLANE = 1, NNSDP_OUT_CIRCLE = 2, NNSDP_OUT_ELLIPSOID = 3 };
/* decomp_mode: DeepSdpOptions (one dense cone, src/Methods/deep_sdp.jl:2-7) or
 * ChordalSdpOptions.decomp_mode = SingleDecomp / DoubleDecomp (src/Methods/chordal_sdp.jl:4-16) */
enum { NNSDP_DECOMP_DENSE = 0, NNSDP_DECOMP_SINGLE = 1, NNSDP_DECOMP_DOUBLE = 2,
       NNSDP_DECOMP_PATH = 3 /* extension: cliques {x_k, x_k+1, affine}, exact when the output QC has S12 = 0 */,
       NNSDP_DECOMP_AUTO = 4 /* PATH when the query allows it (every reach query, hyperplane safety sets), DOUBLE otherwise: the
                                fastest exact decomposition (width-50 networks: 3.6 s instead of 13 s for the reference's cliques) */ };
/* ffnet.activ: ReluActiv / TanhActiv (src/MyNeuralNetwork/MyNeuralNetwork.jl:7-9) */
enum { NNSDP_ACTIV_RELU = 0, NNSDP_ACTIV_TANH = 1 };
/* termination status; strings as consumed by experiments/acas.jl:77 via nnsdp_status_string() */
enum { NNSDP_STATUS_OPTIMAL = 0, NNSDP_STATUS_ITERATION_LIMIT = 1, NNSDP_STATUS_TIME_LIMIT = 2,
       NNSDP_STATUS_SLOW_PROGRESS = 3, NNSDP_STATUS_NUMERICAL_ERROR = 4 };

/*
 * The numeric content of a Methods.Query: FeedFwdNet (src/MyNeuralNetwork/MyNeuralNetwork.jl:12-27),
 * QcInputBox (src/Qc/input.jl:3-8), QcActivBounded (src/Qc/activ_bounded.jl:3-10),
 * QcActivSector for ReLU (src/Qc/activ_sector.jl:2-20) and one QcOutput (src/Qc/output.jl:3-31).
 */
typedef struct nnsdp_problem {
  int32_t K;              /* number of affine layers = length(ffnet.Ms) */
  const int32_t* xdims;   /* K+1 layer sizes */
  const double* M;        /* Ms[1..K] back to back, each xdims[k+1] x (xdims[k]+1) column-major = [W_k b_k] */
  const double* x1min;    /* xdims[0] */
  const double* x1max;    /* xdims[0] */
  const double* acymin;   /* acdim = sum(xdims[1..K-1]) : QcActivBounded.acymin */
  const double* acymax;   /* acdim */
  const double* smin;     /* acdim : QcActivSector.smin (base_smin = 0) */
  const double* smax;     /* acdim : QcActivSector.smax (base_smax = 1) */
  int32_t beta;           /* QcActivSector.beta */
  int32_t query_kind;     /* NNSDP_QUERY_* */
  int32_t out_kind;       /* NNSDP_OUT_* */
  const double* normal;   /* HPLANE: xdims[K] */
  const double* yc;       /* CIRCLE / ELLIPSOID: xdims[K] */
  const double* invP;     /* ELLIPSOID: xdims[K] x xdims[K] column-major */
  const double* S;        /* SAFETY_S: (xdims[0]+xdims[K]+1)^2 column-major */
  int32_t activ;          /* NNSDP_ACTIV_*: ffnet.activ.  TANH: QcActivSector.vardim = lambda_dim, no eta / nu multipliers
                             (src/Qc/activ_sector.jl:19,49-57); smin/smax are then real numbers in [0,1] (:74-86) */
} nnsdp_problem;

/* The fields of `AdmmSdpOptions <: QueryOptions` (the replacement of ChordalSdpOptions). */
typedef struct nnsdp_options {
  int32_t decomp_mode;    /* NNSDP_DECOMP_* */
  int32_t max_iters;      /* ADMM iteration cap (MOSEK analogue: MSK_IPAR_INTPNT_MAX_ITERATIONS) */
  double eps_rel;         /* relative tolerance on both splitting residuals (INTPNT_CO_TOL_PFEAS/DFEAS) */
  double max_time;        /* seconds; <= 0: none (MSK_DPAR_OPTIMIZER_MAX_TIME, experiments/scale.jl:32) */
  double sigma;           /* initial ADMM penalty */
  double alpha;           /* over-relaxation in (0,2) */
  int32_t adapt_every;    /* residual-balancing period, iterations; 0 = fixed sigma */
  int32_t check_every;    /* convergence-check period, iterations (= iterations per hipGraph launch) */
  int32_t normalize;      /* 1: solver-internal interval congruence + fixed-neuron elimination (reach queries) (default) */
  int32_t warm_start;     /* 1: warm-start each eigendecomposition from the previous eigenvectors */
  double proj_tol;        /* Jacobi stops at off(A) <= proj_tol |A|_F; 0 = adaptive: 0.01 x the current residual,
                             clamped to [1e-9, 1e-4] (inexact projections well below the residual level) */
  int32_t polish;         /* 1: make the returned (gamma, Z) exactly feasible (diagonal shift + Schur complement for gout) */
  double cert_tol;        /* > 0 (reach queries): also stop once the polished, exactly feasible objective is within
                             cert_tol (relative) of the ADMM primal/dual objective estimates; 0 = residual test only.
                             A TARGET, not a proven bound: the polished objective is a rigorous UPPER bound of the optimum, but what it
                             is compared with are estimates of the optimum, trusted once both residuals are below cert_tol / 10 (no
                             dual-feasible point is constructed: DESIGN.md section 9).  Measured on the traced solves: 4.9e-4 .. 9.4e-4
                             above the converged optimum for cert_tol = 1e-3.  The stopping iteration is reproducible (no atomics). */
  int32_t verbose;        /* QueryOptions.verbose (src/Methods/Methods.jl:110) */
  int32_t device;         /* HIP device ordinal, -1 = current */
  double interval_guard;  /* (normalize = 1) a neuron interval [acymin, acymax] narrower than interval_guard x |midpoint| is
                             widened to that inside the solver (default 5e-5; 0 = take the bounds literally).  The returned
                             gamma stays feasible for the LMI with the caller's bounds.  The reference's float32 CROWN
                             boxes of collapsed deep nets are narrower than their own rounding error and, taken
                             literally, make the QC set empty (rho = 0 would be "optimal"); MOSEK at 1e-6 never
                             resolves that, an exact solver does. */
  int32_t minv_mode;      /* the Woodbury core M^-1, M = I + A'D^-1A over the kept multipliers: 1 = dense inverse (8 ng^2 bytes read
                             per iteration), 2 = structured (block-banded by network layer + low rank: two-level domain
                             decomposition, three launches, O(ng b) bytes), 0 = auto (structured from 3500 kept multipliers on,
                             when the generator table has that structure) */
  int32_t proj_refine;    /* 1 (default): warm PSD blocks of 41 .. 160 first try the GEMM-only refinement of the eigenbasis kept from the
                             previous iteration (one rotation of all pairs to second order on the matrix cores, accepted when its predicted
                             off(A) is below 30 x the projection tolerance) and fall back to the exact Jacobi sweeps - up to 96 with the
                             basis in LDS, 97 .. 160 in the packed-triangle variant with the basis in HBM; 2: blocks up to 96 whose
                             prediction misses by less than 30 x also take the step and are checked (B rebuilt, off(A) measured) before
                             the sweeps - measured: no gain on W40-D20; 0: sweeps only.
                             A launch that holds a block above 96 runs the same stage as FIVE short launches over the whole chip
                             (tile-parallel pipeline, csrc/refine_pipe.hpp: workgroup = (block, tile column, row group)) in front of the
                             one-CU kernel, which then only sees the blocks the pipeline did not carry; it switches itself on once 70 % of
                             a check window's block visits took the step and off below 40 % (width-50 networks: 2.1x per solve); a block
                             whose prediction misses the accepted level by less than 10x takes the step anyway and is analysed afresh by a
                             second pass of the five launches instead of going to the sweeps (another 1.5x on such networks).  Launches
                             of blocks up to 96 keep the one-CU form (the five launches tie with it there: DESIGN.md section 4). */
} nnsdp_options;

/* Contents of Methods.QuerySolution (src/Methods/Methods.jl:46-55) plus solver diagnostics.
 * gamma_* and Z are caller-allocated (any may be NULL to skip).  Sizes:
 *   gamma_in xdims[0]; gamma_out 1 (reach only); gamma_ac1 acdim;
 *   gamma_ac2 lambda_dim + 2*acdim (ReLU) or lambda_dim (Tanh), lambda_dim = (beta+1)*acdim - beta*(beta+1)/2;
 *   Z Zdim x Zdim column-major, Zdim = sum(xdims[0..K-1]) + 1  (values[:Z], Methods.jl:86). */
typedef struct nnsdp_result {
  double* gamma_in;
  double* gamma_out;
  double* gamma_ac1;
  double* gamma_ac2;
  double* Z;
  double objective;       /* objective_value */
  int32_t status;         /* NNSDP_STATUS_* */
  int32_t iters;
  double pres;            /* |K x + q - w| / max(|K x + q|, |w|)      (dual feasibility of (P)) */
  double dres;            /* |K'y - z0| / max(|K'y|, |z0|)           (LMI equality of (P)) */
  double lambda_max;      /* eigmax(Z(gamma)) in the reference's coordinates (Methods.jl:116) */
  double t_setup;         /* seconds: pattern + generators + factorisation (setup_time) */
  double t_solve;         /* seconds: ADMM loop (solve_time) */
  double t_total;         /* seconds (total_time) */
  double t_eig;           /* seconds of t_solve inside the PSD-projection kernel (HIP events) */
  int32_t n_cliques;      /* PSD blocks solved (after normalisation) */
  int32_t max_clique;     /* largest block dimension solved */
  int64_t eig_flops_per_iter; /* 10 * sum n_k^3 over the blocks solved */
  int64_t eig_bytes_per_iter; /* 2 * 8 * sum n_k^2 */
  double avg_sweeps;      /* Jacobi sweeps per block per iteration, averaged over the solve */
  double objective_admm;  /* objective of the raw ADMM iterate (before the polish) */
  double polish_shift;    /* diagonal shift applied by the polish in solver coordinates; -1: polish not applied */
  int64_t refine_blocks[5]; /* projection refinement stage, block visits over the solve: already converged / one GEMM step / sent on to
                             the Jacobi sweeps / attempt skipped (back-off after failures) / accepted after a checked step */
} nnsdp_result;

int nnsdp_version(void);
const char* nnsdp_last_error(void);
const char* nnsdp_status_string(int32_t status);
void nnsdp_default_options(nnsdp_options* opts);

/* Sizes derived from a problem: Zdim, acdim, length of gamma_ac2, total length of gamma. */
int nnsdp_problem_dims(const nnsdp_problem* p, int32_t* Zdim, int32_t* acdim, int32_t* nac2, int32_t* ngamma);

/* Replaces Methods.runQuery's setup + solve (src/Methods/Methods.jl:91-131). */
int nnsdp_solve(const nnsdp_problem* p, const nnsdp_options* o, nnsdp_result* r);

/* Handle form of the same solve, used by bench.py to time exactly K iterations. */
typedef struct nnsdp_solver nnsdp_solver;
int nnsdp_solver_create(const nnsdp_problem* p, const nnsdp_options* o, nnsdp_solver** out);
/* run `iters` ADMM iterations (no convergence test); eig_ms (may be NULL) receives the HIP-event
 * time of the projection kernel summed over these iterations. */
int nnsdp_solver_iterate(nnsdp_solver* s, int32_t iters, double* eig_ms);
/* advance exactly `iters` iterations with the solve loop's convergence checks and sigma / projection-tolerance
 * adaptation, but without stopping (brings a handle to the solver's steady state before timing) */
int nnsdp_solver_advance(nnsdp_solver* s, int32_t iters);
/* enqueue `iters` iterations on the solver's own HIP stream without waiting: several handles (independent
 * SDPs: the beta sweep of experiments/scale.jl:28, the hyperplanes of NnSdp.findReach2Dpoly) then run
 * concurrently on one GPU; nnsdp_solver_sync waits for one handle. */
int nnsdp_solver_iterate_async(nnsdp_solver* s, int32_t iters);
int nnsdp_solver_sync(nnsdp_solver* s);
/* ADVANCES the handle by ONE check iteration (an ordinary ADMM iteration that also accumulates the residual sums) and returns
 * that iteration's relative residuals and primal / dual objective estimates; it does not stop, adapt the penalty or polish.
 * After calling it on members of a batch handle, call nnsdp_batch_resync (the batch advances its members in lockstep). */
int nnsdp_solver_residuals(nnsdp_solver* s, double* pres, double* dres, double* pobj, double* dobj);
/* test / diagnostic entry: out = M^-1 q for a full-length multiplier vector q (entries of dropped multipliers are ignored and
 * returned as 0), through whichever form the handle uses; *structured (may be NULL) tells which, *operand_bytes its size */
int nnsdp_solver_apply_minv(nnsdp_solver* s, const double* q, double* out, int32_t* structured, int64_t* operand_bytes);
/* test / diagnostic entry: the multiplier block of the solver's fixed-point variable nu (solver coordinates and scaling), one
 * entry per multiplier of the problem (dropped multipliers 0).  In clique-sharded mode this block is replicated; the two-rank
 * test compares it bit for bit between ranks. */
int nnsdp_solver_raw_multipliers(nnsdp_solver* s, double* out);
/* diagnostic: what = 0 hipGraph launches so far, 1 whether an ncclAllReduce could be captured into a hipGraph (sharded mode over RCCL),
 * 2 clique-sharded mode on, 3 iterations done, 4 PSD blocks, 5 largest block, 6 the hipIpc transport (0 off, 1 on with ordinary device
 * memory behind the exchange buffers, 2 on with fine-grained device memory - the default) */
int nnsdp_solver_info(nnsdp_solver* s, int32_t what, double* out);
/* iterate until converged / limits; fills r like nnsdp_solve */
int nnsdp_solver_run(nnsdp_solver* s, nnsdp_result* r);
int nnsdp_solver_finish(nnsdp_solver* s, nnsdp_result* r);
int nnsdp_solver_destroy(nnsdp_solver* s);

/* Batch handle (no reference analogue): several independent SDPs - the beta sweep of experiments/scale.jl:28, the
 * hyperplane directions of NnSdp.findReach2Dpoly (src/NnSdp.jl:73-95), the sub-queries of an ACAS clause
 * (experiments/acas.jl:96-114) - advanced in lockstep with ONE kernel launch per stage for all of them.  The solvers
 * stay owned by the caller and must outlive the batch; all on one device, not clique-sharded, same check_every, proj_refine on or
 * off for all of them.
 *   nnsdp_batch_iterate  exactly `iters` plain iterations of every SDP (no checks, fixed penalty), synchronous
 *   nnsdp_batch_run      full solves with the stopping rules of nnsdp_solve, each SDP on its own; status[count]
 *                        receives the NNSDP_STATUS_* of every solver (collect results with nnsdp_solver_finish_status) */
typedef struct nnsdp_batch nnsdp_batch;
int nnsdp_batch_create(nnsdp_solver** solvers, int32_t count, nnsdp_batch** out);
int nnsdp_batch_iterate(nnsdp_batch* b, int32_t iters);
int nnsdp_batch_run(nnsdp_batch* b, int32_t* status);
int nnsdp_batch_destroy(nnsdp_batch* b);
/* re-establish lockstep after members of the batch were advanced individually (nnsdp_solver_residuals / _iterate / _advance):
 * the next batched iteration is a cold one for every member and the launch tables are rebuilt */
int nnsdp_batch_resync(nnsdp_batch* b);
/* result of a solver that stopped with `status` (as returned by nnsdp_batch_run): certificate polish, gamma, Z */
int nnsdp_solver_finish_status(nnsdp_solver* s, int32_t status, nnsdp_result* r);

/* Replaces Z = Zin + Zout + sum(Zacs) with numeric gamma: Qc.makeZin (src/Qc/input.jl:19-42),
 * makeZout (src/Qc/output.jl:52-106), makeZac (src/Qc/activ.jl:30-42).  gamma = [gin; gout; gac1; gac2]
 * (ngamma doubles); Z is Zdim x Zdim column-major.  Runs on the GPU. */
int nnsdp_assemble_Z(const nnsdp_problem* p, const double* gamma, double* Z);

/* Adjoint of the generator part: out[i] = <G_i, X> for every multiplier i (X symmetric Zdim x Zdim). */
int nnsdp_adjoint(const nnsdp_problem* p, const double* X, double* out);

/* Replaces Methods.makeCliques + the index sets used by setupZs!
 * (src/Methods/chordal_cliques.jl:13-59, src/Methods/chordal_sdp.jl:19-57).  Two-pass:
 * call with ptr == NULL to get n_cliques and total; then with ptr[n_cliques+1], idx[total]
 * (0-based z-indices, CSR). */
int nnsdp_make_cliques(int32_t K, const int32_t* xdims, int32_t beta, int32_t decomp_mode,
                       int32_t* n_cliques, int32_t* total, int32_t* ptr, int32_t* idx);

/* Interval pre-processing on the host: CROWN-sliced bounds of every hidden layer and
 * the sector flags, the direct inputs of the assembler.  Replaces Intervals.intervalsAutoLirpaSliced
 * (src/Intervals/intervals_auto_lirpa.jl:12-64; one PyCall + ONNX round trip per layer through
 * exts/auto_lirpa_bridge.py:97-112) and makeSectorMinMax's interval test (src/Qc/activ_sector.jl:63-72, eps = 1e-4).
 * K, xdims, M as in nnsdp_problem; acdim = xdims[1] + ... + xdims[K-1].  Outputs (caller-allocated, any may be NULL):
 * acymin/acymax[acdim] post-activation bounds, acxmin/acxmax[acdim] pre-activation bounds, smin/smax[acdim] sector
 * flags in {0,1}, ymin/ymax[xdims[K]] bounds of the network output.  No GPU needed. */
int nnsdp_make_intervals(int32_t K, const int32_t* xdims, const double* M, const double* x1min, const double* x1max,
                         double* acymin, double* acymax, double* acxmin, double* acxmax, double* smin, double* smax,
                         double* ymin, double* ymax);

/* The same for either activation of the reference (ffnet.activ): activ = NNSDP_ACTIV_TANH restates auto_LiRPA's BoundTanh relaxation
 * (exts/auto_LiRPA/operators/activation.py:843-1016, reached through exts/auto_lirpa_bridge.py:31-37,86-87) and fills smin / smax by
 * makeSectorMinMax's tanh branch (src/Qc/activ_sector.jl:74-86: real slopes in [0, 1]).  activ = NNSDP_ACTIV_RELU is nnsdp_make_intervals. */
int nnsdp_make_intervals_activ(int32_t K, const int32_t* xdims, const double* M, int32_t activ, const double* x1min, const double* x1max,
                               double* acymin, double* acymax, double* acxmin, double* acxmax, double* smin, double* smax,
                               double* ymin, double* ymax);

/* Sampled forward pass on the GPU: Y[:, s] = ffnet(X[:, s]) for N points in fp64.  Replaces the
 * N = 1e5 calls of evalFeedFwdNet (src/MyNeuralNetwork/MyNeuralNetwork.jl:40-48) inside Utils.sampleTrajs (src/Utils/qc.jl:40-47),
 * whose outputs shape the ellipsoid of NnSdp.findEllipsoid (approxEllipsoid, src/Utils/qc.jl:50-67).  K, xdims, M as in
 * nnsdp_problem; activ = NNSDP_ACTIV_*; X is xdims[0] x N, Y is xdims[K] x N, both column-major and caller-owned.
 * kernel_ms (may be NULL) receives the HIP-event time of the launch.  Layer widths up to 639. */
int nnsdp_eval_network(int32_t K, const int32_t* xdims, const double* M, int32_t activ, int64_t N, const double* X, double* Y,
                       double* kernel_ms);

/* Batched projection onto the PSD cone, the hot kernel (replaces the cone handling inside MOSEK;
 * reference of the arithmetic: LinearAlgebra.eigen on Symmetric).  mats: `batch` symmetric
 * matrices back to back, matrix b is n[b] x n[b] column-major.  Matrices up to 160 go through the LDS-resident Jacobi kernel
 * (one launch for all of them; 129 .. 160 in its packed-triangle variant), larger ones (up to 4096) one at a time through rocSOLVER
 * dsyevd + rocBLAS dgemm.  out receives
 * the projections, eigvals (may be NULL) sum(n) eigenvalues.  in/out are HOST pointers. */
int nnsdp_project_psd_batched(int32_t batch, const int32_t* n, const double* mats, double* out,
                              double* eigvals, double* kernel_ms);

/* Test / diagnostic entry of the same kernel in its WARM form, as the solver runs it from the second iteration on: basis (in / out,
 * matrices back to back like mats, column-major, columns = eigenvectors) is the eigenbasis kept from the previous projection; tol
 * the relative stopping level off(V'AV) <= tol |A|_F; refine != 0 puts the GEMM-only refinement stage (nnsdp_options.proj_refine)
 * in front of the Jacobi sweeps (values as proj_refine).  outcome[5] (may be NULL) counts the blocks: converged as given / one
 * refinement step / sent on to the sweeps / not attempted / accepted after a checked step.  Matrices up to 160.  Host pointers. */
int nnsdp_project_psd_warm(int32_t batch, const int32_t* n, const double* mats, double* basis, double tol, int32_t refine, double* out,
                           int32_t* outcome, double* kernel_ms);
/* The same with the refinement stage's per-block state carried between calls, as a solver carries it between iterations: state (in /
 * out, 4 x batch int32, all zero before the first call; NULL = fresh state every call) holds the back-off word - bits 24..27 of word
 * 0 count the visits that may still run without the Gram product V'V - and the running estimate of |I - V'V|_F (words 2..3, a double). */
int nnsdp_project_psd_warm_state(int32_t batch, const int32_t* n, const double* mats, double* basis, double tol, int32_t refine, double* out,
                                 int32_t* outcome, double* kernel_ms, int32_t* state);

/* Multi-GPU clique-sharded mode (one SDP over several GPUs of one node): every rank creates the same solver,
 * then nnsdp_solver_set_comm() before the first iteration.  Rank r projects a contiguous range of cliques
 * (balanced by n_k^3); the consensus sum sum_k H_k'(2 w_k - nu_k) is exchanged with ONE ncclAllReduce (RCCL
 * over xGMI) per iteration, everything else is replicated.  The 128-byte unique id is produced on rank 0
 * and distributed by the host launcher (torch.distributed in bench.py --mode shard).  RCCL is dlopen'ed on
 * first use.  Independent SDPs need none of this (nnsdp_amd/parallel.py).  Every stopping / penalty / tolerance decision of
 * a sharded solve is taken from all-reduced numbers, so all ranks take it identically (including the time limit).
 * The replicated multiplier block is re-synchronised from rank 0 at every check iteration and the certificate (polish, eigmax) and
 * the cert_tol stop are computed by rank 0 alone and broadcast, so all ranks return the same bits; nnsdp_solver_run / _finish are
 * therefore COLLECTIVE calls in this mode.  Over RCCL the iterations between checks replay a hipGraph that contains the all-reduce
 * (probed at set_comm; eager otherwise).  Covered by a two-process run on one GPU through nnsdp_solver_set_comm_callback (below;
 * RCCL refuses two ranks on one device) and a one-rank RCCL run; RCCL with more than one rank has not run yet (the build pool
 * offers one GPU). */
int nnsdp_comm_unique_id(char* id128);
/* Host-only (no GPU, no RCCL): the PSD blocks the solver works on for (problem, options) and their partition over `nranks`
 * ranks - exactly what nnsdp_solver_set_comm uses.  Two-pass: n_blocks first (block_n = start = NULL), then
 * block_n[n_blocks] (block dimensions after the normalisation) and start[nranks+1] (rank r owns blocks start[r] .. start[r+1]-1). */
int nnsdp_shard_plan(const nnsdp_problem* p, const nnsdp_options* o, int32_t nranks, int32_t* n_blocks, int32_t* block_n,
                     int32_t* start);
int nnsdp_solver_set_comm(nnsdp_solver* s, int32_t nranks, int32_t rank, const char* id128);
/* The same sharded mode over the CALLER's collective instead of RCCL (MPI.Allreduce! from the Julia side; gloo in the
 * two-process GPU test): fn(user, buf, count) replaces the HOST buffer buf[count] by its element-wise sum over all ranks and
 * returns 0.  The library stages the per-iteration exchange (and the 8 control numbers of a check iteration) through host
 * memory for it; partition, kernels and the collective control decisions are those of nnsdp_solver_set_comm. */
typedef int (*nnsdp_allreduce_fn)(void* user, double* buf, int64_t count);
int nnsdp_solver_set_comm_callback(nnsdp_solver* s, int32_t nranks, int32_t rank, nnsdp_allreduce_fn fn, void* user);
/* The sharded mode with a DEVICE-SIDE exchange for ranks that are processes of one node (one card, or peers over xGMI; at most 8):
 * every rank's exchange buffer is mapped into the others' address space with hipIpc, and the per-iteration exchange is a one-shot
 * all-gather + local reduce in rank order by the library's own kernels (slot of the rank's buffer, exchange number published behind a
 * system-scope release, bounded spin on the peers' numbers) - no library collective in the iteration, the same bits on every rank,
 * capturable into the iteration's hipGraph.  fn is used as in nnsdp_solver_set_comm_callback for the set-up (the 64-byte handles) and
 * for the control decisions of the check iterations.  A peer that does not publish within a few seconds fails the next check
 * iteration with an error instead of hanging the device.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only supports dmabuf. */
int nnsdp_solver_set_comm_ipc(nnsdp_solver* s, int32_t nranks, int32_t rank, nnsdp_allreduce_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* NNSDP_H */
