// K3, tile-parallel form of the refinement stage (round 4): the GEMM-only update of the persistent eigenbasis of every PSD block
// (kernels.hip, proj_body: B = V'AV, E~ from B, X = E~ + E~^2 / 2, V <- V (I + X), second-order eigenvalues, W = sum lambda v v')
// spread over the WHOLE chip instead of one CU per block.  Every 16 x 16 output tile of every product is ONE WAVE running one chain of
// v_mfma_f64_16x16x4_f64.  What crosses tiles - the diagonal of B, the decision sums, E~ for E~^2, V' for the reconstruction - crosses
// at a KERNEL BOUNDARY: five short launches on the solver's stream (replayed inside its hipGraph), no in-kernel spin, nothing to
// dead-lock when other processes share the card, deterministic (fixed-order reductions), any block size up to 160 in one code path:
//
//   P1 k_pipe_T   T = A V            all tiles      + partial d = diag(V'T), R_ii = 1 - |v_i|^2, |A|_F^2, Vt = V', snapshot of the state
//   P2 k_pipe_B   B = V'T (G = V'V)  lower tiles    + pair analysis -> E~ and E~' (both triangles), 7 partial sums per tile
//   P3 k_pipe_X   X = E~ + E~^2 / 2  lower tiles    + the block's accept / reject decision from the sums (recorded for P4, P5)
//   P4 k_pipe_V   V' = V + V X       all tiles      + partial column norms / eigenvalue sums
//   P5 k_pipe_W   W = sum mu v'v''   lower tiles    + w, nu rescale, V' -> Vg, state word, pmode[block] = handled
//
// A block whose step is not accepted (early in a solve, penalty changes, pairs first order cannot resolve) is left untouched with
// pmode = 0 and the one-CU kernel launched behind the pipeline runs it exactly as before (stage attempt with exact pair rotations,
// Newton-Schulz repair, sweeps); handled blocks return from that kernel at once.  The decision rule, its thresholds and the state word
// are those of proj_body's stage.
//
// Workgroup = (block, tile column tj, group of up to `rows` tile rows), one wave per tile row; the map is built on the host so that
// the workgroups of one block land on ONE XCD (workgroup id mod 8: MI355X_MICROARCH.md, dispatch) and share that XCD's L2.
//
// How the operands reach the matrix cores - three forms were built and measured (profiles/r04_pipe_stamps_*.log,
// tools/load_issue_probe.hip):
//  1. every wave pulls its two 16 x n strips from L2 straight into registers, a lane walking its own run of a strided operand: 64
//     cache lines per wave instruction, 9-15 us per launch;
//  2. every matrix kept in the orientation in which the TILE's index is the fast one (the producing kernel writes the transposed
//     copy the consumer needs: Vt by P1, Et by P2), so that a load is four 128-byte segments: 7-10 us per launch, and the stamps put
//     2.2-4 us of a wave's life into ISSUING its 50-90 loads.  The micro-benchmark says why: a CU issues ONE vector-memory wave
//     instruction per ~13 cycles whatever its width (48 loads: 0.28 us for a wave alone, 1.5 us with six waves on the CU, the same
//     for 8 and 16 bytes per lane) - the instruction COUNT per CU is the cost, and six waves each loading their own strips issue
//     300-500 of them;
//  3. (this file) each strip is staged ONCE per workgroup through LDS with 16-byte loads - (rows + 2) strips of 1 KB pieces, ~80
//     instructions per CU - and the chains read their operands from LDS (strip[k][16]: a wave's ds_read_b64 covers two contiguous
//     128-byte rows per half, conflict-free).  No register arrays, no size templates.
// nu is taken as it is stored (symmetric to rounding by construction of the update; sym(nu) is formed where it matters, in W).
#pragma once

namespace nnsdp {

struct PipeWg { int b, tj, r0, rw; };     // block, tile column, first tile row of the group, tile rows (waves) of the group

static constexpr int kMaxGroups = 2;      // row groups per tile column (10 tile rows at n = 160 in two groups of 5)
static constexpr int kPipeWaves = 6;      // waves (tile rows) per workgroup at most

struct PipeArgs {
  const int* cn;            // block sizes (blocks of this launch)
  const long long* coff;    // element offsets of the blocks in the packed clique storage
  const PipeWg* wgmap;      // workgroup -> (block, tile column, row group); b < 0: idle
  double* nu;               // packed matrices (rescaled in place when kappa != 1, as the one-CU kernel does)
  double* w;                // packed projections (out)
  double* Vg;               // packed eigenbases (in / out)
  double* T;                // n^2 per block: T = A V (row-major), then X (row-major)
  double* E;                // n^2 per block: E~ (row-major)
  double* U;                // n^2 per block: the new basis (column-major)
  double* Vt;               // n^2 per block: V' (row-major copy of the basis, written by P1)
  double* Et;               // n^2 per block: E~' (row-major)
  double* drec;             // [block] PipeRec: the decision, recorded by P3
  double* dpart;            // [block][kMaxGroups][vs] partial diag(V'AV) per row group
  double* rdg;              // [block][vs] 1 - |v_i|^2 (0 on visits without the Gram product)
  double* lpart;            // [block][kMaxGroups][vs][4] partial column sums per row group: |v'_j|^2, sum E~^2, sum E~^2 d
  double* frop;             // [block][16] |A|_F^2 per tile row
  double* psum;             // [block][pt][8] partial sums of the lower tiles: off2, k2, unpp, unnn, unx, kd2, r2
  int* vrec;                // [block][4] snapshot of rstate taken by P1: word, do_gram, rdef (double)
  int* pmode;               // [block] 0: not handled (the one-CU kernel runs it), 1: stepped, 2: converged as it arrived
  int* again;               // [block] written by pass 0's P5: 1 = the block took a BLIND step (below) and pass 1 analyses it afresh
  int* rstate;              // refinement state per block (ProjArgs::rstate)
  int* stats;               // ProjArgs::stats
  const double* kappa;      // device scalar, may be null
  const double* tol_dev;    // device scalar, may be null
  double tol;
  double refine_acc, refine_kcap, refine_loose, refine_k2cap;
  int gram_credit;
  // Second pass (near misses).  A block whose prediction misses the accepted level by less than `refine_near` x takes the step anyway -
  // the basis is updated, nothing else - and the five launches run once more for exactly those blocks (pass = 1): second-order
  // convergence makes the second analysis pass by a wide margin, at the price of one more pipeline pass (~60 us) instead of the
  // packed sweeps (~1 ms for a 151-block).  The second pass decides by its own measured numbers (Gram product forced), so
  // nothing rests on the first prediction; a block it rejects goes to the one-CU kernel as before.
  double refine_near;       // <= 1: off
  int pass;                 // 0 / 1
  int vs;                   // stride of the per-block vectors (multiple of 16, >= largest block)
  int pt;                   // stride of psum in tiles (>= lower tiles of the largest block)
  int rows;                 // tile rows per row group (<= kPipeWaves; fewer when the strips of the largest block would not fit LDS)
  double* eig;              // optional eigenvalue output (test entry), with eoff
  const long long* eoff;
  long long* dbg;           // (-DNNSDP_STAMPS builds only) [5 kernels][workgroup][8] wall-clock stamps of thread 0
};

#ifdef NNSDP_STAMPS
#define PST(kern, slot) { if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)(kern) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); }
// finer stamps: after a wave-uniform value is available (the scalar loads behind it have returned) / after all vector loads landed
#define PSTS(kern, slot, sval) { if (a.dbg) { long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "s"(sval) : "memory"); if (threadIdx.x == 0) a.dbg[((size_t)(kern) * gridDim.x + blockIdx.x) * 8 + (slot)] = t_; } }
#define PSTL(kern, slot) { if (a.dbg) { long long t_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (threadIdx.x == 0) a.dbg[((size_t)(kern) * gridDim.x + blockIdx.x) * 8 + (slot)] = t_; } }
#else
#define PST(kern, slot)
#define PSTS(kern, slot, sval)
#define PSTL(kern, slot)
#endif

struct PipeDecision {
  int mode;                 // 0 not handled, 1 step, 2 converged as it arrived, 3 (pass 0 only) blind step: basis updated, analysed again in pass 1
  bool up;                  // rebuild W from the positive side
  bool loose, do_gram;
  double r2, k2, rdef;
  int credit, gcred;
  double off2, unpp, unnn, unx, kd2, accT;      // (diagnostics: the terms of the prediction)
};
// decision record written by P3's first wave for P4 / P5 (the tail is diagnostic: NNSDP_PIPE_DEBUG prints it at finish)
struct PipeRec { int mode, up, loose, do_gram, credit, gcred; double r2, k2, rdef; double off2, unpp, unnn, unx, kd2, accT; };

// diag(V'AV)_j from the row groups' partial sums (fixed order).  Both slots are read whatever the block's group count (the unused
// one stays zero): a loop over a run-time count makes every call a load - wait - add sequence, and the kernels' prologues call this
// a dozen times (stamps: 1.8 us of a 3.5 us staging phase were such serialised round trips).
__device__ __forceinline__ double pipe_dvec(const PipeArgs& a, const int b, const int ng, const int j) {
  static_assert(kMaxGroups == 2, "two partial slots");
  (void)ng;
  const double* p = a.dpart + (size_t)b * kMaxGroups * a.vs + j;
  return p[0] + p[a.vs];
}

// The block's accept / reject decision from the partial sums of P2: executed by one full wave, every lane returns the same bits
// (fixed-order sums), and every wave of P3 that calls it gets the same answer.
__device__ __forceinline__ PipeDecision pipe_decide(const PipeArgs& a, const int b, const int n) {
  const int lane = threadIdx.x & 63;
  const int nt = (n + 15) >> 4, ntl = nt * (nt + 1) / 2, ng = (nt + a.rows - 1) / a.rows;
  PipeDecision D;
  const int rs = a.vrec[4 * b];
  D.do_gram = a.vrec[4 * b + 1] != 0;
  D.rdef = *reinterpret_cast<const double*>(a.vrec + 4 * b + 2);
  D.credit = (rs >> 16) & 255; D.gcred = (rs >> 24) & 15;
  D.mode = 0; D.up = true; D.loose = false; D.r2 = 0.0; D.k2 = 0.0;
  D.off2 = D.unpp = D.unnn = D.unx = D.kd2 = D.accT = 0.0;
  if ((rs & 255) != 0) return D;                       // back-off: not attempted
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = lane; t < ntl; t += 64) {
    const double* p = a.psum + ((size_t)b * a.pt + t) * 8;
#pragma unroll
    for (int q = 0; q < 7; ++q) s[q] += p[q];
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) s[q] = wave_sum(s[q]);
  // (all loads of the decision are requested together: three lanes-worth of diag(B), the 16 row norms)
  int cpos = 0, cneg = 0;
  double dv3[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) dv3[u] = pipe_dvec(a, b, ng, min(lane + 64 * u, a.vs - 1));
  const double fr = a.frop[(size_t)b * 16 + (lane & 15)];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const double dv = (lane + 64 * u < n) ? dv3[u] : 0.0;
    cpos += __popcll(__ballot(dv > 0.0)); cneg += __popcll(__ballot(dv < 0.0));
  }
  const double fro2 = wave_sum((lane < nt) ? fr : 0.0);
  const double tolv = a.tol_dev ? *a.tol_dev : a.tol;
  const double Tl = tolv * sqrt(fro2), accT = a.refine_acc * Tl;
  const double off2 = s[0], k2 = s[1], unpp = s[2], unnn = s[3], unx = s[4], kd2 = s[5];
  const double r2 = D.do_gram ? s[6] : D.rdef * D.rdef;
  D.r2 = r2; D.k2 = k2;
  D.off2 = off2; D.unpp = unpp; D.unnn = unnn; D.unx = unx; D.kd2 = kd2; D.accT = accT;
  D.up = cpos <= cneg;
  if (off2 <= Tl * Tl && r2 <= tolv * tolv) { D.mode = 2; return D; }
  if (r2 <= 1e-4) {
    const double pred0 = 1.5 * sqrt(off2) * sqrt(k2) + k2 * sqrt(kd2) * (1.0 / 3.0);
    const double pred_pos = pred0 + sqrt(unpp + unx), pred_neg = pred0 + sqrt(unnn + unx);
    const bool prefer_pos = cpos <= cneg;
    const bool kok = k2 <= a.refine_k2cap;
    int side = 0;
    if (kok && (prefer_pos ? pred_pos : pred_neg) <= accT) side = prefer_pos ? 1 : -1;
    else if (kok && (prefer_pos ? pred_neg : pred_pos) <= accT) side = prefer_pos ? -1 : 1;
    else if (kok && D.credit >= 16 && fmin(pred_pos, pred_neg) <= a.refine_loose * accT) { side = pred_pos <= pred_neg ? 1 : -1; D.loose = true; }
    if (side != 0) { D.mode = 1; D.up = side > 0; }
    else if (a.pass == 0 && kok && fmin(pred_pos, pred_neg) <= a.refine_near * accT) { D.mode = 3; D.up = pred_pos <= pred_neg; }      // blind step, second pass
  }
  return D;
}

// ---- LDS strips -----------------------------------------------------------------------------------------------------------------
// A strip is the 16 columns [i0, i0 + 16) of a matrix M stored with that index fastest (M[k n + i]), as S[k * 16 + r], k < kp
// (kp = n rounded up to 4; rows past n and columns past n are zero).  Staging is split in two so that the requests of ALL strips of a
// workgroup are in flight before the first LDS store: pipe_strip_issue puts the thread's pieces (16 bytes: two consecutive columns
// of one row k) into registers - every load unconditional, from a clamped address: a guarded load (`in range ? M[..] : 0`) makes
// the compiler wait for each value behind its load, one L2 round trip at a time - and pipe_strip_store masks and writes them.
static constexpr int kStripPieces = 4;     // 16-byte pieces per thread and strip: 384 threads x 4 >= 8 x 160 rows
static constexpr int kStripMin = 2 * 16 * 17;      // a strip's slot also serves its wave as 16 x 16 transposition scratch (two of them in P2)
struct PipePiece { double2 v[kStripPieces]; };
// what a thread needs to know about ITS pieces is the same for every strip of the block and is computed once (the first form rebuilt
// row, column pair, clamps and 64-bit addresses per piece and strip: ~50 vector instructions per piece on the critical path of a
// kernel whose whole life is a few microseconds): byte offset of the piece relative to the strip's first column, LDS byte offset,
// row-in-range mask.  A strip whose 16 columns all exist (every tile but the last of a block whose size is not a multiple of 16)
// takes the fast path: one load per piece, no column clamp.
struct PipeStage {
  unsigned goff[kStripPieces];     // (min(k, n - 1) n + 2 h) 8
  int np;                          // pieces of this thread's waves that exist at all (uniform over the workgroup: ceil(8 kp / threads))
  unsigned rowok;                  // bit q: piece q lies inside the strip and its row k < n
  unsigned instrip;                // bit q: piece q lies inside the strip (p < 8 kp)
};
__device__ __forceinline__ PipeStage pipe_stage_init(const int n, const int kp) {
  PipeStage G;
  const int nthr = blockDim.x;
  G.np = (8 * kp + nthr - 1) / nthr;
  G.rowok = 0; G.instrip = 0;
#pragma unroll
  for (int q = 0; q < kStripPieces; ++q) {
    const int p = threadIdx.x + q * nthr, k = p >> 3, h = p & 7;
    G.goff[q] = (unsigned)(min(k, n - 1) * n + 2 * h) * 8u;
    if (p < 8 * kp) { G.instrip |= 1u << q; if (k < n) G.rowok |= 1u << q; }
  }
  return G;
}
__device__ __forceinline__ void pipe_strip_issue(PipePiece& P, const PipeStage& G, const double* __restrict__ M, const int n, const int i0) {
  const char* base = reinterpret_cast<const char*>(M + i0);          // (wave-uniform)
  if (i0 + 16 <= n) {
#pragma unroll
    for (int q = 0; q < kStripPieces; ++q)
      if (q < G.np) P.v[q] = *reinterpret_cast<const double2*>(base + G.goff[q]);
  } else {
    // last, partial tile: the pair of columns clamped to (n - 2, n - 1); pipe_strip_store sorts the values out
#pragma unroll
    for (int q = 0; q < kStripPieces; ++q) {
      const int h = (threadIdx.x + q * blockDim.x) & 7;
      const int back = max(i0 + 2 * h - (n - 2), 0);                 // columns the piece is moved back by
      if (q < G.np) P.v[q] = *reinterpret_cast<const double2*>(base + G.goff[q] - 8u * (unsigned)back);
    }
  }
}
__device__ __forceinline__ void pipe_strip_store(const PipePiece& P, const PipeStage& G, double* __restrict__ S, const int n, const int i0) {
  const int nthr = blockDim.x;
  const bool partial = i0 + 16 > n;
#pragma unroll
  for (int q = 0; q < kStripPieces; ++q) {
    if (q < G.np && ((G.instrip >> q) & 1u)) {
      double2 v = P.v[q];
      if (partial) {
        const int c = i0 + 2 * ((threadIdx.x + q * nthr) & 7);
        if (c > n - 2) { v.x = (c == n - 1) ? v.y : 0.0; v.y = 0.0; }     // (the clamped piece holds columns n - 2, n - 1)
      }
      if (!((G.rowok >> q) & 1u)) { v.x = 0.0; v.y = 0.0; }
      *reinterpret_cast<double2*>(S + 2 * (size_t)(threadIdx.x + q * nthr)) = v;               // S[k * 16 + 2 h]
    }
  }
}
// the product's chain: C = sum_k A[r][k] B[k][c] with A[r][k] = SA[k * 16 + r], B[k][c] = SB[k * 16 + c]; even and odd steps on two
// accumulators (a wave is nearly alone on its SIMD here and one dependent chain advances at ~150 cycles per instruction)
// (the LDS reads of the NEXT pair of steps are requested before this pair's MFMAs are issued)
__device__ __forceinline__ d4_t pipe_chain(const double* __restrict__ SA, const double* __restrict__ SB, const int ksq, const int lane) {
  d4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
  const double* pa = SA + lane;            // (4 kk + lc) 16 + lr = 64 kk + lane
  const double* pb = SB + lane;
  const int last = ksq - 1;
  double a0 = pa[0], b0 = pb[0], a1 = pa[64 * min(1, last)], b1 = pb[64 * min(1, last)];
  for (int kk = 0; kk < ksq; kk += 2) {
    const int k2 = min(kk + 2, last), k3 = min(kk + 3, last);
    const double na0 = pa[64 * k2], nb0 = pb[64 * k2], na1 = pa[64 * k3], nb1 = pb[64 * k3];
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
    if (kk + 1 < ksq) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
    a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
  }
  return c0 + c1;
}
__device__ __forceinline__ d4_t pipe_chain_scaled(const double* __restrict__ SA, const double* __restrict__ SB, const double* __restrict__ mu, const int ksq, const int lane) {
  d4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
  const double* pa = SA + lane;
  const double* pb = SB + lane;
  const double* pm = mu + (lane >> 4);     // mu[4 kk + lc]
  const int last = ksq - 1;
  double a0 = pa[0] * pm[0], b0 = pb[0], a1 = pa[64 * min(1, last)] * pm[4 * min(1, last)], b1 = pb[64 * min(1, last)];
  for (int kk = 0; kk < ksq; kk += 2) {
    const int k2 = min(kk + 2, last), k3 = min(kk + 3, last);
    const double na0 = pa[64 * k2] * pm[4 * k2], nb0 = pb[64 * k2], na1 = pa[64 * k3] * pm[4 * k3], nb1 = pb[64 * k3];
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
    if (kk + 1 < ksq) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
    a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
  }
  return c0 + c1;
}

#define PIPE_PROLOGUE                                                                                       \
  extern __shared__ __align__(16) double pipe_lds[];                                                       \
  const PipeWg m = a.wgmap[blockIdx.x];                                                                    \
  const int b = m.b, tj = m.tj;                                                                            \
  if (b < 0) return;                                                                                       \
  if (a.pass == 1 && a.again[b] == 0) return;                                                              \
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2, kp = 4 * ksq, ng = (nt + a.rows - 1) / a.rows; \
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;                \
  const int ti = m.r0 + wv;                                                                                \
  const size_t ssz = (size_t)max(kp * 16, kStripMin);                                                      \
  const PipeStage G = pipe_stage_init(n, kp);                                                              \
  (void)nt; (void)ng; (void)lr; (void)lc; (void)ti; (void)ssz;

// ---- P1: T = A V (all tiles), partial d = diag(V'T), R_ii, |A|_F^2 per tile row, state snapshot, Vt = V' ----------------------------
__global__ __launch_bounds__(64 * kPipeWaves) void k_pipe_T(PipeArgs a) {
  PIPE_PROLOGUE
  PST(0, 0)
  const int group = m.r0 / a.rows;
  const int rs = a.rstate[4 * b];
  const double rdef = *reinterpret_cast<const double*>(a.rstate + 4 * b + 2);
  const double tolv = a.tol_dev ? *a.tol_dev : a.tol;
  const int gcred = (rs >> 24) & 15;
  const bool do_gram = gcred == 0 || !(rdef <= 0.03 * a.refine_acc * tolv);
  if (tj == 0 && m.r0 == 0 && threadIdx.x == 0) {      // (nobody writes rstate while the pipeline's first four launches run)
    a.vrec[4 * b] = rs; a.vrec[4 * b + 1] = do_gram ? 1 : 0;
    *reinterpret_cast<double*>(a.vrec + 4 * b + 2) = rdef;
  }
  if ((rs & 255) != 0) return;            // back-off: the one-CU kernel counts it down (uniform over the workgroup)
  const bool active = wv < m.rw;
  const double* nuk = a.nu + a.coff[b];
  const double* vk = a.Vg + a.coff[b];
  // LDS: Vs = the 16 columns J of V, k-major with stride 17 (filled from the column-major basis: a thread stores single doubles 17
  // apart), then one standard strip of nu per tile row of the group
  double* Vs = pipe_lds;
  double* SA = pipe_lds + (((size_t)kp * 17 + 1) & ~(size_t)1);
  PipePiece pa[kPipeWaves];
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw) pipe_strip_issue(pa[r], G, nuk, n, 16 * (m.r0 + r));          // A[i][k] = nu[k n + i]
  {
    // the columns of V are contiguous: thread -> (column jl, 16-byte chunk of rows)
    const int nch = kp >> 1;                 // chunks of 2 rows per column
    double2 vv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = threadIdx.x + q * blockDim.x, jl = min(p / nch, 15), ch = p - (p / nch) * nch;
      const int col = min(16 * tj + jl, n - 1), k = min(2 * ch, n - 2);
      if (q * (int)blockDim.x < 16 * nch) vv[q] = *reinterpret_cast<const double2*>(vk + (unsigned)(col * n + k));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = threadIdx.x + q * blockDim.x, jl = p / nch, ch = p - jl * nch;
      if (q * (int)blockDim.x < 16 * nch && jl < 16) {
        const int col = 16 * tj + jl, k = 2 * ch;
        double2 v = vv[q];
        if (k > n - 2) { v.x = (k == n - 1) ? v.y : 0.0; v.y = 0.0; }
        if (col >= n) { v.x = 0.0; v.y = 0.0; }
        Vs[k * 17 + jl] = v.x; Vs[(k + 1) * 17 + jl] = v.y;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw) pipe_strip_store(pa[r], G, SA + r * ssz, n, 16 * (m.r0 + r));
  PST(0, 1)
  __syncthreads();
  PST(0, 2)
  if (group == 0) {
    // Vt[k][16 tj .. 16 tj + 15] = the strip's rows (what P2 reads as its row operand); R_jj = 1 - |v_j|^2
    double* vt = a.Vt + a.coff[b];
    const int j = 16 * tj + lr;
    for (int k = 4 * wv + lc; k < n; k += 4 * kPipeWaves)
      if (j < n) vt[(size_t)k * n + j] = Vs[k * 17 + lr];
    if (wv == 0) {
      double s = 0.0;
      for (int k = lc; k < kp; k += 4) { const double v = Vs[k * 17 + lr]; s += v * v; }
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (lane < 16 && 16 * tj + lane < a.vs) a.rdg[(size_t)b * a.vs + 16 * tj + lane] = (16 * tj + lane < n && do_gram) ? 1.0 - s : 0.0;
    }
  }
  double* red = SA + (size_t)a.rows * ssz;      // [waves][64] partial d
  double pd = 0.0;
  if (active) {
    // chain with the B operand from the stride-17 strip
    d4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
    const double* pa_ = SA + wv * ssz + lane;
    const double* pb_ = Vs + lc * 17 + lr;
    const int last = ksq - 1;
    double a0 = pa_[0], b0 = pb_[0], a1 = pa_[64 * min(1, last)], b1 = pb_[68 * min(1, last)];
    for (int kk = 0; kk < ksq; kk += 2) {
      const int k2 = min(kk + 2, last), k3 = min(kk + 3, last);
      const double na0 = pa_[64 * k2], nb0 = pb_[68 * k2], na1 = pa_[64 * k3], nb1 = pb_[68 * k3];
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
      if (kk + 1 < ksq) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
      a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
    }
    const d4_t c = c0 + c1;
    PST(0, 3)
    double* Tk = a.T + a.coff[b];
    const int j = 16 * tj + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ti + lc + 4 * r;
      if (row < n && j < n) { Tk[(size_t)row * n + j] = c[r]; pd += Vs[row * 17 + lr] * c[r]; }
    }
    if (tj == 0) {
      // |A|_F^2 of this tile row (the strip holds rows ti of nu, all columns)
      double f = 0.0;
      const double* sa = SA + wv * ssz;
      for (int e = lane; e < kp * 16; e += 64) f += sa[e] * sa[e];
      f = wave_sum(f);
      if (lane == 0) a.frop[(size_t)b * 16 + ti] = f;
    }
  }
  red[wv * 64 + lane] = pd;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int t = threadIdx.x, col = 16 * tj + t;
    double d = 0.0;
    for (int w_ = 0; w_ < m.rw; ++w_)
#pragma unroll
      for (int q = 0; q < 4; ++q) d += red[w_ * 64 + q * 16 + t];
    if (col < a.vs) a.dpart[((size_t)b * kMaxGroups + group) * a.vs + col] = col < n ? d : 0.0;
  }
  PST(0, 4)
}

// ---- P2: B = V'T (and G = V'V on Gram visits), pair analysis, E~ and its transpose ------------------------------------------------
__global__ __launch_bounds__(64 * kPipeWaves) void k_pipe_B(PipeArgs a) {
  PIPE_PROLOGUE
  PST(1, 0)
  const int rs = a.vrec[4 * b];
  if ((rs & 255) != 0) return;
  if (m.r0 + m.rw <= tj) return;           // the whole group lies above the diagonal (uniform)
  const bool do_gram = a.vrec[4 * b + 1] != 0;
  const double* vt = a.Vt + a.coff[b];
  const double* Tk = a.T + a.coff[b];
  // LDS: ST = T[:, J], SG = Vt[:, J] (= V[:, J]: the Gram product's column operand, and the row operand of the diagonal tile),
  // then Vt[:, I] per tile row
  double* ST = pipe_lds;
  double* SG = pipe_lds + ssz;
  double* SV = pipe_lds + 2 * ssz;
  PipePiece pt_, pg, pv[kPipeWaves];
  pipe_strip_issue(pt_, G, Tk, n, 16 * tj);
  pipe_strip_issue(pg, G, vt, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_issue(pv[r], G, vt, n, 16 * (m.r0 + r));
  const bool active = wv < m.rw && ti >= tj;
  double di[4], ri[4], dj = 0.0, rj = 0.0;
  {
    const int jc = min(16 * tj + lr, a.vs - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = min(16 * ti + lc + 4 * r, a.vs - 1); di[r] = pipe_dvec(a, b, ng, row); ri[r] = a.rdg[(size_t)b * a.vs + row]; }
    dj = pipe_dvec(a, b, ng, jc); rj = a.rdg[(size_t)b * a.vs + jc];
  }
  pipe_strip_store(pt_, G, ST, n, 16 * tj);
  pipe_strip_store(pg, G, SG, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_store(pv[r], G, SV + r * ssz, n, 16 * (m.r0 + r));
  PST(1, 1)
  __syncthreads();
  PST(1, 2)
  if (!active) return;                     // (no further workgroup barrier)
  const double* SA = ti == tj ? SG : SV + wv * ssz;       // V'[i][k] = Vt[k n + i]
  const d4_t c = pipe_chain(SA, ST, ksq, lane);
  d4_t gc = {0.0, 0.0, 0.0, 0.0};
  if (do_gram) gc = pipe_chain(SA, SG, ksq, lane);
  PST(1, 3)
  const int j = 16 * tj + lr;
  const double kcap = a.refine_kcap;
  double o2 = 0.0, q2 = 0.0, upp = 0.0, unn = 0.0, ux = 0.0, qd2 = 0.0, g2 = 0.0;
  double eo[4], fo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    eo[r] = 0.0; fo[r] = 0.0;
    if (row < n && j < n) {
      if (row > j) {
        const double bb = c[r], rr = -gc[r];
        const double li = di[r] * (1.0 + ri[r]), lj = dj * (1.0 + rj);
        const double gap = lj - li;
        o2 += 2.0 * bb * bb;
        g2 += 2.0 * rr * rr;
        if (fabs(bb) <= kcap * fabs(gap) && gap != 0.0) {
          const double e = (bb + lj * rr) * rcp_nr2(gap), f = rr - e;
          q2 += e * e + f * f;
          qd2 += e * e * lj * lj + f * f * li * li;
          eo[r] = e; fo[r] = f;
        } else {
          eo[r] = 0.5 * rr; fo[r] = 0.5 * rr;
          q2 += 0.5 * rr * rr;
          const double dd = di[r] * dj;
          if (bb * bb < dd) { if (di[r] > 0.0) upp += 2.0 * bb * bb; else unn += 2.0 * bb * bb; }
          else ux += 2.0 * bb * bb;
        }
      } else if (row == j) {
        eo[r] = 0.5 * ri[r]; fo[r] = eo[r];
        g2 += ri[r] * ri[r];
      }
    }
  }
  o2 = wave_sum(o2); q2 = wave_sum(q2); upp = wave_sum(upp); unn = wave_sum(unn); ux = wave_sum(ux); qd2 = wave_sum(qd2); g2 = wave_sum(g2);
  if (lane == 0) {
    double* p = a.psum + ((size_t)b * a.pt + (ti * (ti + 1) / 2 + tj)) * 8;
    p[0] = o2; p[1] = q2; p[2] = upp; p[3] = unn; p[4] = ux; p[5] = qd2; p[6] = g2; p[7] = 0.0;
  }
  // E~ (row-major) and Et = E~' (row-major): the pair (row > j) gives E~[row][j] = e, E~[j][row] = f.  At address row n + j (this
  // lane layout): E <- e, Et <- f; at address j n + row (through a 16 x 16 transpose in this wave's own, now dead, strip slot): E <- f,
  // Et <- e.
  double* Ek = a.E + a.coff[b];
  double* Etk = a.Et + a.coff[b];
  double* tre = SV + wv * ssz;             // 2 x 16 x 17 doubles (a strip slot holds at least kStripMin)
  double* trf = tre + 16 * 17;
  wave_lds_sync();                         // (this wave's chain reads of its strip are done)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (row < n && j < n && row >= j) { Ek[(size_t)row * n + j] = eo[r]; Etk[(size_t)row * n + j] = fo[r]; }
    tre[(lc + 4 * r) * 17 + lr] = eo[r]; trf[(lc + 4 * r) * 17 + lr] = fo[r];
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ii = 16 * ti + lr, jj = 16 * tj + lc + 4 * r;     // pair (ii, jj), ii > jj
    if (ii < n && jj < n && ii > jj) { Ek[(size_t)jj * n + ii] = trf[lr * 17 + lc + 4 * r]; Etk[(size_t)jj * n + ii] = tre[lr * 17 + lc + 4 * r]; }
  }
  PST(1, 4)
}

// ---- P3: the decision (every wave, the first one records it), X = E~ + E~^2 / 2 (lower tiles, both triangles written) ---------------
__global__ __launch_bounds__(64 * kPipeWaves) void k_pipe_X(PipeArgs a) {
  PIPE_PROLOGUE
  PST(2, 0)
  const bool writer = tj == 0 && m.r0 == 0 && wv == 0;
  PipeRec* rec = reinterpret_cast<PipeRec*>(a.drec) + b;
  if ((a.vrec[4 * b] & 255) != 0) { if (writer && lane == 0) rec->mode = 0; return; }
  if (m.r0 + m.rw <= tj) return;
  const double* Ek = a.E + a.coff[b];
  const double* Etk = a.Et + a.coff[b];
  // LDS: SE = E~[:, J] (column operand), SEt = Et[:, J] (row operand of the diagonal tile), Et[:, I] per tile row.
  // (The strips are requested before the decision is known: plain reads of scratch that always exists; the decision's own
  // dependent loads then run in their shadow.)
  double* SE = pipe_lds;
  double* SEt = pipe_lds + ssz;
  double* SV = pipe_lds + 2 * ssz;
  PipePiece pe, pet, pv[kPipeWaves];
  pipe_strip_issue(pe, G, Ek, n, 16 * tj);
  pipe_strip_issue(pet, G, Etk, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_issue(pv[r], G, Etk, n, 16 * (m.r0 + r));
  const bool active = wv < m.rw && ti >= tj;
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  double et[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rowt = min(16 * tj + lc + 4 * r, n - 1);
    et[r] = Ek[(size_t)rowt * n + min(i, n - 1)];
  }
  const PipeDecision D = pipe_decide(a, b, n);
  if (writer && lane == 0) {
    rec->mode = D.mode; rec->up = D.up ? 1 : 0; rec->loose = D.loose ? 1 : 0; rec->do_gram = D.do_gram ? 1 : 0;
    rec->credit = D.credit; rec->gcred = D.gcred; rec->r2 = D.r2; rec->k2 = D.k2; rec->rdef = D.rdef;
    rec->off2 = D.off2; rec->unpp = D.unpp; rec->unnn = D.unnn; rec->unx = D.unx; rec->kd2 = D.kd2; rec->accT = D.accT;
  }
  if (D.mode != 1 && D.mode != 3) return;  // (uniform over the workgroup)
  pipe_strip_store(pe, G, SE, n, 16 * tj);
  pipe_strip_store(pet, G, SEt, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_store(pv[r], G, SV + r * ssz, n, 16 * (m.r0 + r));
  PST(2, 1)
  __syncthreads();
  PST(2, 2)
  if (!active) return;
  const double* SA = ti == tj ? SEt : SV + wv * ssz;      // E~[i][k] = Et[k n + i]
  const d4_t c = pipe_chain(SA, SE, ksq, lane);
  PST(2, 3)
  double* Xk = a.T + a.coff[b];
  double* tr = SV + wv * ssz;              // (this wave's own strip slot: dead after its chain)
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (row < n && j < n) Xk[(size_t)row * n + j] = SE[(size_t)row * 16 + lr] + 0.5 * c[r];      // E~[row][j] sits in the column strip
    tr[(lc + 4 * r) * 17 + lr] = c[r];
  }
  if (ti != tj) {
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowt = 16 * tj + lc + 4 * r;
      if (rowt < n && i < n) Xk[(size_t)rowt * n + i] = et[r] + 0.5 * tr[lr * 17 + lc + 4 * r];      // E~^2 is symmetric
    }
  }
  PST(2, 4)
}

// ---- P4: V' = V + V X (all tiles, computed transposed so that the stores are coalesced), partial column sums -----------------------
__global__ __launch_bounds__(64 * kPipeWaves) void k_pipe_V(PipeArgs a) {
  PIPE_PROLOGUE
  PST(3, 0)
  PSTS(3, 5, n)
  const int group = m.r0 / a.rows;
  const PipeRec rec = reinterpret_cast<const PipeRec*>(a.drec)[b];
  const double* vk = a.Vg + a.coff[b];
  const double* Xk = a.T + a.coff[b];
  const double* Ek = a.E + a.coff[b];
  double* Uk = a.U + a.coff[b];
  const bool active = wv < m.rw;
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  // LDS: SX = X[:, J] (row operand: X'[j][k] = X[k n + j]), Vg[:, I] per tile row (V'[k][i] = V[i][k] = Vg[k n + i])
  double* SX = pipe_lds;
  double* SV = pipe_lds + ssz;
  PipePiece px, pv[kPipeWaves];
  pipe_strip_issue(px, G, Xk, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw) pipe_strip_issue(pv[r], G, vk, n, 16 * (m.r0 + r));
  double v0[4], ee[4], dk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = min(16 * tj + lc + 4 * r, n - 1), row = min(16 * ti + lc + 4 * r, n - 1);
    v0[r] = vk[(size_t)col * n + min(i, n - 1)];
    ee[r] = Ek[(size_t)row * n + min(j, n - 1)];     // E~[k = row][j]
    dk[r] = pipe_dvec(a, b, ng, row);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (!(active && row < n && j < n && row != j)) ee[r] = 0.0;
  }
  PST(3, 6)
  PSTL(3, 7)
  if (rec.mode == 0) return;               // (uniform over the workgroup)
  double* lp = a.lpart + (((size_t)b * kMaxGroups + group) * a.vs + 16 * tj) * 4;
  if (rec.mode == 2) {
    // converged as it arrived: the basis and diag(B) are the result; P5 reads U whatever the mode
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * tj + lc + 4 * r;
      if (active && col < n && i < n) Uk[(size_t)col * n + i] = v0[r];
    }
    return;
  }
  pipe_strip_store(px, G, SX, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw) pipe_strip_store(pv[r], G, SV + r * ssz, n, 16 * (m.r0 + r));
  PST(3, 1)
  __syncthreads();
  PST(3, 2)
  double* red = SV + (size_t)a.rows * ssz;       // [waves][16] norms, [waves][64] x 2 column sums
  double* red2 = red + kPipeWaves * 16;
  double* red3 = red2 + kPipeWaves * 64;
  double nr[4] = {0.0, 0.0, 0.0, 0.0};
  double s1 = 0.0, s2 = 0.0;
  if (active) {
    const d4_t c = pipe_chain(SX, SV + wv * ssz, ksq, lane);
    PST(3, 3)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * tj + lc + 4 * r;
      const double vn = v0[r] + c[r];
      if (col < n && i < n) Uk[(size_t)col * n + i] = vn;
      nr[r] = row_sum16((col < n && i < n) ? vn * vn : 0.0);
      s1 += ee[r] * ee[r]; s2 += ee[r] * ee[r] * dk[r];
    }
  }
  if (lr == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv * 16 + lc + 4 * r] = nr[r];
  }
  red2[wv * 64 + lane] = s1; red3[wv * 64 + lane] = s2;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int t = threadIdx.x;
    double nrm = 0.0, c1 = 0.0, c2 = 0.0;
    for (int w_ = 0; w_ < m.rw; ++w_) {
      nrm += red[w_ * 16 + t];
#pragma unroll
      for (int q = 0; q < 4; ++q) { c1 += red2[w_ * 64 + q * 16 + t]; c2 += red3[w_ * 64 + q * 16 + t]; }
    }
    if (16 * tj + t < a.vs) { double* q = lp + 4 * t; q[0] = nrm; q[1] = c1; q[2] = c2; q[3] = 0.0; }
  }
  PST(3, 4)
}

// ---- P5: W = sum over the chosen side (lower tiles, mirrored), nu rescale, V' -> Vg, state ---------------------------------------
__global__ __launch_bounds__(64 * kPipeWaves) void k_pipe_W(PipeArgs a) {
  PIPE_PROLOGUE
  PST(4, 0)
  const PipeRec rec = reinterpret_cast<const PipeRec*>(a.drec)[b];
  const bool writer = tj == 0 && m.r0 == 0 && wv == 0;
  if (rec.mode == 0) { if (writer && lane == 0) { a.pmode[b] = 0; if (a.pass == 0) a.again[b] = 0; } return; }      // (uniform over the workgroup)
  if (rec.mode == 3) {
    // blind step: the new basis becomes the persistent one (U and Vg share their layout: the workgroups of the first row group copy
    // their tile column's 16 columns), the state asks the second pass for a Gram visit, nothing is projected
    if (m.r0 == 0) {
      const double* Uc = a.U + a.coff[b];
      double* vg = a.Vg + a.coff[b];
      const int e0 = 16 * tj * n, e1 = min(16 * (tj + 1), n) * n;
      for (int e = e0 + (int)threadIdx.x; e < e1; e += (int)blockDim.x) vg[e] = Uc[e];
    }
    if (writer && lane == 0) {
      a.rstate[4 * b] = (rec.credit << 16);                  // no back-off, Gram credit 0: the second pass measures V'V
      *reinterpret_cast<double*>(a.rstate + 4 * b + 2) =
          (rec.do_gram ? rec.r2 + 2.0 * sqrt(rec.r2 * rec.k2) : rec.rdef * (1.0 + 2.2 * sqrt(rec.k2))) + 0.25 * rec.k2 * rec.k2;
      a.pmode[b] = 0; a.again[b] = 1;
    }
    return;
  }
  if (m.r0 + m.rw <= tj) return;
  const double* Uk = a.U + a.coff[b];
  double* nuk = a.nu + a.coff[b];
  double* wk = a.w + a.coff[b];
  const bool active = wv < m.rw && ti >= tj;
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  // LDS: SJ = U[:, J] (column operand V'[j][k] = U[k n + j]; row operand of the diagonal tile), U[:, I] per tile row, the
  // eigenvalue weights mu[k]
  double* SJ = pipe_lds;
  double* SV = pipe_lds + ssz;
  double* mu = SV + (size_t)a.rows * ssz;
  PipePiece pj, pv[kPipeWaves];
  pipe_strip_issue(pj, G, Uk, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_issue(pv[r], G, Uk, n, 16 * (m.r0 + r));
  // nu at the two places this wave writes: nd at [(16 ti + lc + 4r) n + 16 tj + lr] - the MIRROR of the entry the lane computes -
  // and nt_ at [(16 tj + lc + 4r) n + 16 ti + lr], the entry itself in the transposed lane layout
  const double kap = a.kappa ? *a.kappa : 1.0;
  double nd[4], nt_[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r, rowt = 16 * tj + lc + 4 * r;
    nd[r] = nuk[(size_t)min(row, n - 1) * n + min(j, n - 1)];
    nt_[r] = nuk[(size_t)min(rowt, n - 1) * n + min(i, n - 1)];
  }
  const bool up = rec.up != 0;
  // eigenvalues from the row groups' partial column sums (P4), the weights of the chosen side
  for (int k = threadIdx.x; k < kp; k += blockDim.x) {
    double l = 0.0;
    if (k < n) {
      const double d = pipe_dvec(a, b, ng, k);
      if (rec.mode == 2) l = d;
      else {
        const double* q0 = a.lpart + ((size_t)b * kMaxGroups * a.vs + k) * 4;
        const double* q1 = q0 + (size_t)a.vs * 4;
        const double nrm = q0[0] + q1[0], c1 = q0[1] + q1[1], c2 = q0[2] + q1[2];
        l = (d * (1.0 + a.rdg[(size_t)b * a.vs + k] + c1) - c2) / nrm;
      }
      if (a.eig && tj == 0 && m.r0 == 0) a.eig[a.eoff[b] + k] = l;
    }
    mu[k] = up ? fmax(l, 0.0) : fmin(l, 0.0);
  }
  pipe_strip_store(pj, G, SJ, n, 16 * tj);
#pragma unroll
  for (int r = 0; r < kPipeWaves; ++r)
    if (r < m.rw && m.r0 + r > tj) pipe_strip_store(pv[r], G, SV + r * ssz, n, 16 * (m.r0 + r));
  PST(4, 1)
  __syncthreads();
  PST(4, 2)
  if (!active) return;
  const double* SI = ti == tj ? SJ : SV + wv * ssz;
  const d4_t c = pipe_chain_scaled(SI, SJ, mu, ksq, lane);     // V'[i][k] mu_k V'[j][k]
  PST(4, 3)
  double* tr = SV + wv * ssz;              // (this wave's own strip slot; the diagonal wave's slot is used by nobody else)
  // the basis' tile (rows ti, columns tj) and its mirror tile, taken from the strips before the slot is reused:
  //   u1: U[(16 tj + lc + 4r) n + 16 ti + lr] = row k = 16 tj + lc + 4r of the strip of columns I
  //   u2: U[(16 ti + lc + 4r) n + 16 tj + lr] = row k = 16 ti + lc + 4r of the strip of columns J
  double u1[4], u2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    u1[r] = SI[(size_t)min(16 * tj + lc + 4 * r, kp - 1) * 16 + lr];
    u2[r] = SJ[(size_t)min(16 * ti + lc + 4 * r, kp - 1) * 16 + lr];
  }
  wave_lds_sync();
  // value of entry (i' = 16 ti + lc + 4r, j' = 16 tj + lr), i' >= j'
  double vd[4];
  if (!up) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[lr * 17 + lc + 4 * r] = nt_[r];      // nu(i' = 16 ti + lr, j' = 16 tj + lc + 4r) -> transposed
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) vd[r] = 0.5 * (nd[r] + tr[(lc + 4 * r) * 17 + lr]) - c[r];
    wave_lds_sync();
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) vd[r] = c[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tr[(lc + 4 * r) * 17 + lr] = vd[r];
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r, rowt = 16 * tj + lc + 4 * r;
    // direct: element at [row n + j] = W(j, row) (mirror), valid for row >= j
    if (row < n && j < n && row >= j) {
      const size_t ix = (size_t)row * n + j;
      wk[ix] = vd[r];
      if (kap != 1.0) nuk[ix] = vd[r] + kap * (nd[r] - vd[r]);
    }
    // transposed: element at [rowt n + i] = W(i, rowt), the entry itself, for i > rowt (the diagonal went above)
    if (rowt < n && i < n && i > rowt) {
      const size_t ix = (size_t)rowt * n + i;
      const double v = tr[lr * 17 + lc + 4 * r];
      wk[ix] = v;
      if (kap != 1.0) nuk[ix] = v + kap * (nt_[r] - v);
    }
  }
  if (rec.mode == 1) {
    // the new basis becomes the persistent one: this wave's tile and its mirror tile (U is only read in this launch, Vg only written)
    double* vg = a.Vg + a.coff[b];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * tj + lc + 4 * r, colm = 16 * ti + lc + 4 * r;
      if (col < n && i < n) vg[(size_t)col * n + i] = u1[r];
      if (ti != tj && colm < n && j < n) vg[(size_t)colm * n + j] = u2[r];
    }
  }
  if (writer && lane == 0) {
    const int outcome = rec.mode == 1 ? 1 : 0;
    if (a.stats) atomicAdd(&a.stats[rec.loose ? 8 : 4 + outcome], 1);
    const int word = rec.loose ? 0 : (min(rec.credit + 1, 255) << 16);
    double rnew = rec.do_gram ? sqrt(rec.r2) : rec.rdef;
    if (outcome == 1) rnew = (rec.do_gram ? rec.r2 + 2.0 * sqrt(rec.r2 * rec.k2) : rec.rdef * (1.0 + 2.2 * sqrt(rec.k2))) + 0.25 * rec.k2 * rec.k2;
    a.rstate[4 * b] = word | ((rec.do_gram ? a.gram_credit : rec.gcred - 1) << 24);
    *reinterpret_cast<double*>(a.rstate + 4 * b + 2) = rnew;
    a.pmode[b] = rec.mode;
    if (a.pass == 0) a.again[b] = 0;
  }
  PST(4, 4)
}

// largest block the pipeline takes
static constexpr int kPipeMaxBlock = 160;

// host side: scratch and the workgroup map of one set of blocks; the five launches
struct RefinePipe {
  int nblocks = 0, nmax = 0, nwg = 0, vs = 16, pt = 1, rows = kPipeWaves;
  size_t lds = 0;
  PipeWg* wgmap = nullptr;
  double *T = nullptr, *E = nullptr, *U = nullptr, *Vt = nullptr, *Et = nullptr, *drec = nullptr, *dpart = nullptr, *rdg = nullptr, *lpart = nullptr, *frop = nullptr,
         *psum = nullptr;
  int *vrec = nullptr, *pmode = nullptr, *again = nullptr;
  double near = 10.0;        // PipeArgs::refine_near (NNSDP_PIPE_NEAR; <= 1: one pass only)
  bool ready = false;
  RefinePipe() = default;
  RefinePipe(const RefinePipe&) = delete;            // (owns device memory)
  RefinePipe& operator=(const RefinePipe&) = delete;
  ~RefinePipe() { release(); }
  void release() {
    void* ps[] = {wgmap, T, E, U, Vt, Et, drec, dpart, rdg, lpart, frop, psum, vrec, pmode, again};
    for (void* p : ps) if (p) (void)hipFree(p);
    wgmap = nullptr; T = E = U = Vt = Et = drec = dpart = rdg = lpart = frop = psum = nullptr; vrec = pmode = again = nullptr; ready = false;
  }
  // LDS bytes of a launch whose largest block is nmax with `r` tile rows per group: P2 / P3 hold (r + 2) strips, P1 the stride-17
  // strip + r strips, P4 / P5 (r + 1) strips; reduction scratch and the eigenvalue weights behind them
  static size_t lds_bytes(int nmax, int r) {
    const size_t kp = (size_t)((nmax + 3) & ~3);
    const size_t strip = std::max(kp * 16, (size_t)kStripMin);
    const size_t tail = (size_t)kPipeWaves * (16 + 128) + kp + 64;
    const size_t a1 = ((kp * 17 + 1) & ~(size_t)1) + (size_t)r * strip + tail;
    const size_t a2 = (size_t)(r + 2) * strip + tail;
    return std::max(a1, a2) * sizeof(double);
  }
  // cn[nb]: sizes of the blocks of the launch in launch order; nmat: elements of their packed storage
  hipError_t build(const int* cn, int nb, long long nmat) {
    release();
    nblocks = nb; nmax = 0;
    for (int k = 0; k < nb; ++k) nmax = nmax > cn[k] ? nmax : cn[k];
    if (nb <= 0 || nmax > kPipeMaxBlock || nmax < 2) return hipSuccess;
    vs = (nmax + 15) & ~15;
    const int ntm = vs >> 4;
    pt = ntm * (ntm + 1) / 2;
    rows = kPipeWaves;
    if (const char* e = std::getenv("NNSDP_PIPE_ROWS")) rows = std::min(std::max(std::atoi(e), 3), kPipeWaves);      // (diagnostic)
    while (rows > 1 && lds_bytes(nmax, rows) > 160 * 1024) --rows;      // (blocks of 129 .. 160: five tile rows per group)
    lds = lds_bytes(nmax, rows);
    if (lds > 160 * 1024) return hipSuccess;
    // workgroups of one block on one XCD: bucket = block mod 8, workgroup id = 8 x position + bucket
    std::vector<std::vector<PipeWg>> bucket(8);
    for (int k = 0; k < nb; ++k) {
      const int nt = (cn[k] + 15) / 16, ng = (nt + rows - 1) / rows;
      if (ng > kMaxGroups || cn[k] < 2) return hipSuccess;
      for (int tj = 0; tj < nt; ++tj)
        for (int g = 0; g < ng; ++g) {
          const int r0 = g * rows, rw = std::min(rows, nt - r0);
          bucket[k & 7].push_back(PipeWg{k, tj, r0, rw});
        }
    }
    size_t depth = 0;
    for (auto& bk : bucket) depth = depth > bk.size() ? depth : bk.size();
    std::vector<PipeWg> map(8 * depth, PipeWg{-1, 0, 0, 0});
    for (int x = 0; x < 8; ++x)
      for (size_t p = 0; p < bucket[x].size(); ++p) map[8 * p + x] = bucket[x][p];
    nwg = (int)map.size();
    hipError_t e = hipSuccess;
    auto al = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 8); } };
    al((void**)&wgmap, map.size() * sizeof(PipeWg));
    al((void**)&T, ((size_t)nmat + 16) * 8); al((void**)&E, ((size_t)nmat + 16) * 8); al((void**)&U, ((size_t)nmat + 16) * 8);
    al((void**)&Vt, ((size_t)nmat + 16) * 8); al((void**)&Et, ((size_t)nmat + 16) * 8); al((void**)&drec, (size_t)nb * sizeof(PipeRec));
    al((void**)&dpart, (size_t)nb * kMaxGroups * vs * 8); al((void**)&rdg, (size_t)nb * vs * 8); al((void**)&lpart, (size_t)nb * kMaxGroups * vs * 4 * 8);
    al((void**)&frop, (size_t)nb * 16 * 8); al((void**)&psum, (size_t)nb * pt * 8 * 8);
    al((void**)&vrec, (size_t)nb * 4 * sizeof(int)); al((void**)&pmode, (size_t)nb * sizeof(int)); al((void**)&again, (size_t)nb * sizeof(int));
    if (const char* e = std::getenv("NNSDP_PIPE_NEAR")) near = std::atof(e);                                        // (diagnostic)
    if (e == hipSuccess) e = hipMemcpy(wgmap, map.data(), map.size() * sizeof(PipeWg), hipMemcpyHostToDevice);
    if (lds > 64 * 1024) {
      const void* ks[] = {(const void*)&k_pipe_T, (const void*)&k_pipe_B, (const void*)&k_pipe_X, (const void*)&k_pipe_V, (const void*)&k_pipe_W};
      for (const void* kf : ks) if (e == hipSuccess) e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    ready = e == hipSuccess;
    return e;
  }
  // the pipeline's arguments from the one-CU kernel's (same block list, same state)
  PipeArgs args(const ProjArgs& p) const {
    PipeArgs a{};
    a.cn = p.cn; a.coff = p.coff; a.wgmap = wgmap; a.nu = p.nu; a.w = p.w; a.Vg = p.Vg;
    a.T = T; a.E = E; a.U = U; a.Vt = Vt; a.Et = Et; a.drec = drec; a.dpart = dpart; a.rdg = rdg; a.lpart = lpart; a.frop = frop; a.psum = psum;
    a.vrec = vrec; a.pmode = pmode; a.again = again; a.refine_near = near; a.pass = 0;
    a.rstate = p.rstate; a.stats = p.stats; a.kappa = p.kappa; a.tol_dev = p.tol_dev; a.tol = p.tol;
    a.refine_acc = p.refine_acc; a.refine_kcap = p.refine_kcap; a.refine_loose = p.refine_loose; a.refine_k2cap = p.refine_k2cap; a.gram_credit = p.gram_credit;
    a.vs = vs; a.pt = pt; a.rows = rows; a.eig = p.eig; a.eoff = p.eoff;
    return a;
  }
  void launch(const PipeArgs& a, hipStream_t st) const {
    const dim3 g(nwg), t(64 * kPipeWaves);
    PipeArgs b = a;
    for (int pass = 0; pass < (a.refine_near > 1.0 ? 2 : 1); ++pass) {
      b.pass = pass;
      hipLaunchKernelGGL(k_pipe_T, g, t, lds, st, b);
      hipLaunchKernelGGL(k_pipe_B, g, t, lds, st, b);
      hipLaunchKernelGGL(k_pipe_X, g, t, lds, st, b);
      hipLaunchKernelGGL(k_pipe_V, g, t, lds, st, b);
      hipLaunchKernelGGL(k_pipe_W, g, t, lds, st, b);
    }
  }
};

}  // namespace nnsdp
