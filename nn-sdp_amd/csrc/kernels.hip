// HIP kernels for gfx950 (MI355X): batched PSD projection by LDS-resident parallel Jacobi, and the
// operator / consensus kernels of one ADMM iteration.  fp64 throughout (1e-6 certificates need it).
//
// Data layout in HBM (all fp64 unless noted):
//   nu   [ng | sum_k n_k^2]   fixed-point variable: multiplier block then the clique matrices,
//                              matrix k column-major n_k x n_k at offset ng + coff[k]
//   w    same shape            projection of nu onto R+^ng x PSD^p
//   Vg   [sum_k n_k^2]         eigenvectors of the last projection (warm start), column-major
//   x,g  [NE]                  pattern vectors (scaled svec over the union of clique blocks)
//   gidx int32 [sum_k n_k^2]   pattern entry of every clique-matrix element, bit 31 = diagonal
//   CSR (rows = pattern entries) and CSC (columns = multipliers) of the generator table A
//   Minv [ng x ng]             inverse of M = I + A' D^-1 A (symmetric), the Woodbury core
//   rstate int32 [4 x blocks]  refinement stage per block: back-off word (incl. the credit of visits without a Gram product) and the
//                              running estimate of the eigenbasis' defect |I - V'V|_F
//   Tg, Ug [sum_k n_k^2]       scratch of the packed variant (blocks 97 .. 160): warm-start product / rotation log / X, and the new basis
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace nnsdp {

static constexpr int kThreads = 256;
static constexpr double kSqrt2 = 1.41421356237309504880;
static constexpr double kInvSqrt2 = 0.70710678118654752440;

typedef double d4_t __attribute__((ext_vector_type(4)));

// lane permutation inside a row of 16 lanes on the VALU (DPP): no LDS traffic, unlike __shfl_* (ds_bpermute)
template <int CTRL>
__device__ __forceinline__ double dpp_row(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// sums over aligned groups of 8 / 16 lanes (every lane of the group gets the group's sum)
__device__ __forceinline__ double row_sum8(double v) { v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v); v += dpp_row<0x141>(v); return v; }
__device__ __forceinline__ double row_sum16(double v) { v = row_sum8(v); v += dpp_row<0x140>(v); return v; }

// sum over the 64 lanes of a (fully active) wave, the same bits in every lane.  Four DPP butterflies give every lane its row's sum
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: the two operands of every addition are swapped between partner
// lanes, and addition commutes), the four row sums are read as scalars.  The __shfl_down form took 12 ds_bpermute per value: with
// 16 waves reducing half a dozen values each, the CU's LDS pipe was the bottleneck of the refinement stage's analysis.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_row<0xB1>(v);
  v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v);
  v += dpp_row<0x140>(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}

// block-wide sum, result valid in every thread; scratch >= blockDim/64 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wv] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  if ((nw & (nw - 1)) == 0) {
    // one LDS read and log2(nw) butterfly steps (every lane ends with the same bits: the additions commute) instead of nw dependent
    // LDS reads - the serial form cost ~3 000 cycles per call with 16 waves, and the projection kernel makes a dozen of them
    const double s = scratch[lane & (nw - 1)];
    if (nw == 16) return row_sum16(s);
    if (nw == 8) return row_sum8(s);
    double t = s;
    for (int o = nw >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  }
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += scratch[i];
  return s;
}

// ---------------------------------------------------------------------------------------------
// K3: projection of every clique matrix onto the PSD cone.
// One workgroup per clique.  The symmetric matrix lives in LDS for the whole kernel; eigenvectors
// live in LDS too when both fit in the CU's 160 KB (V_LDS), otherwise in HBM/L2 (large blocks).
// Two-sided cyclic Jacobi with the round-robin (chess tournament) ordering: each round applies
// np/2 disjoint rotations, A <- J'AJ by 2x2 blocks (every block owned by one thread, in place),
// V <- V J by (row, pair) items.  Warm start: A <- V0' A V0 with the previous eigenvectors makes
// the matrix nearly diagonal, so one or two sweeps suffice in the ADMM steady state.
// ---------------------------------------------------------------------------------------------
struct ProjArgs {
  const int* cn;          // block sizes n_k
  const long long* coff;  // element offset of block k inside the packed clique storage
  const long long* eoff;  // offset of block k in the packed eigenvalue array (may be null)
  double* nu;             // packed matrices (in; rescaled in place when kappa != 1)
  double* w;              // packed projections (out)
  double* Vg;             // packed eigenvectors (in for warm start, out)
  double* eig;            // packed eigenvalues (out, may be null)
  double* Tg;             // scratch of the size of Vg for the packed variant's warm start (blocks 129 .. 160; may be null otherwise)
  double* Ug;             // second scratch of that size: the packed variant's refinement stage writes the new basis there (null: no stage)
  const double* kappa;    // device scalar: nu <- w + kappa (nu - w) (penalty change), may be null
  const double* tol_dev;  // device scalar overriding tol (lets the host adapt it between graph launches), may be null
  int* stats;             // [0] += sweeps used (atomic), [1] = max sweeps seen, [2..3] rotation counts, [4..8] refinement stage: blocks
                          // accepted without a step / after one unchecked step / sent on to the sweeps / not attempted (back-off) /
                          // accepted after a checked step
  int warm;               // 1: use Vg as the starting basis
  int refine;             // 1: warm blocks first try the GEMM-only refinement of the persistent basis (ping-pong variant only);
                          // 2: also the checked form (step, rebuild B, measure) for blocks whose prediction fails by less than 30 x
  int* rstate;            // refinement state per block, 4 ints: [0] = (Gram credit << 24) | (success credit << 16) | (back-off level << 8) |
                          // iterations still to skip, [2..3] = estimate of |I - V'V|_F as a double; zero-initialised (may be null)
  double refine_acc;      // a refinement step is accepted without a check when its PREDICTED off(A) is below refine_acc x tol |A|
  double refine_kcap;     // pairs whose first-order rotation angle B_ij / (d_j - d_i) exceeds this are left to the sweeps
  int gram_credit;        // visits a block may run without the Gram product after a visit that measured it (0 .. 15)
  int refine_pivots;      // exact rotations of the dominant pair first order cannot resolve, per visit (0 = off)
  double refine_loose;    // >= 1: the one-off acceptance level of an isolated near miss, as a multiple of refine_acc (1 = off)
  double refine_k2cap;    // a step is only taken when |K|_F^2 is below this (0.09: |K|_F <= 0.3; see api.hip, refine_k2cap)
  int max_sweeps;
  double tol;             // stop after a sweep that started with off(A)/|A|_F <= tol (quadratic convergence: it ends near tol^2)
  const int* pmode;       // per block, may be null: != 0 = the tile-parallel pipeline launched in front (refine_pipe.hpp) has already
                          // projected this block in this iteration; the workgroup returns at once
  // two workgroups per block for the warm-start congruence (ping-pong variant, warm launches of one SDP; launch_proj): the grid is
  // 2 x nblk - workgroups [0, nblk) are HELPERS (they never wait: dispatched first, they always finish), [nblk, 2 nblk) LEADERS.  Both
  // load the block; each computes the tile columns of T = A V and of B = V'T it owns (same instruction sequence per tile as the
  // one-workgroup form: identical bits); the helper leaves its tiles of B in `sB` behind a release of `sack[k]` and exits, the leader
  // takes them after its own and goes on alone.  sack / sseen count the exchanges per block (device-resident: replayed graphs need no
  // per-launch argument); a wait that runs out sets *serr (the solve ends with NUMERICAL_ERROR at its next check).
  int split, nblk;
  double* sB;             // nblk x kSplitTileDoubles
  unsigned* sack;         // per block: exchanges completed by the helper
  unsigned* sseen;        // per block: exchanges consumed by the leader
  unsigned* sxcc;         // per block: the XCD the helper ran on (diagnostic)
  int* serr;
  long long spin_limit;
};
static constexpr int kHwRegXccId = (3 << 11) | 20;     // s_getreg: 4 bits at offset 0 of HW_REG_XCC_ID (20): the XCD a wave runs on
static constexpr int kSplitTileDoubles = 96 * 96;      // tile-ordered storage of one block's B (6 x 6 tiles of 256 doubles)

// element (i,j) of the symmetric LDS matrix, lower triangle is the only copy that is kept current
__device__ __forceinline__ int sym_at(int i, int j, int lda) { return i >= j ? i * lda + j : j * lda + i; }

// 1/sqrt(x) to full fp64 accuracy from the hardware estimate (v_rsq_f64) + 3 Newton steps; avoids the
// long software sqrt/divide sequences on the critical path of every Jacobi round
__device__ __forceinline__ double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 3; ++i) r = r * (1.5 - 0.5 * x * r * r);
  return r;
}

// rotation (c, s) that annihilates a_pq:  J = [c s; -s c],  A <- J' A J.
// With d = aqq - app, b = 2 apq, h = hypot(d, b):  cos(2 theta) = |d|/h,  c = sqrt((1 + |d|/h)/2),
// s = sign(d) b / (2 h c)  (the small-angle root, |theta| <= pi/4).  Two rsqrt, no divide.
__device__ __forceinline__ void jacobi_cs(double app, double aqq, double apq, double thr, double& c, double& s) {
  c = 1.0; s = 0.0;
  if (fabs(apq) <= thr) return;   // below the per-element share of the tolerance: identity
  double d = aqq - app, b = 2.0 * apq;
  double h2 = d * d + b * b;
  if (apq != 0.0 && h2 > 0.0 && h2 < 1e300) {
    double rh = rsqrt_nr(h2);
    double u = 0.5 + 0.5 * fabs(d) * rh;     // in [0.5, 1]
    double rc = rsqrt_nr(u);
    c = u * rc;
    s = 0.5 * b * rh * rc;
    if (d < 0.0) s = -s;
    // one renormalisation makes c^2 + s^2 = 1 to rounding irrespective of the estimate accuracy
    double nrm = rsqrt_nr(c * c + s * s);
    c *= nrm; s *= nrm;
  } else if (apq != 0.0) {
    // extreme magnitudes (over/underflow of d^2 + b^2): scaled evaluation
    double sc = fmax(fabs(d), fabs(b));
    double dd = d / sc, bb = b / sc;
    double rh = 1.0 / sqrt(dd * dd + bb * bb);
    double u = 0.5 + 0.5 * fabs(dd) * rh;
    c = sqrt(u);
    s = 0.5 * bb * rh / c;
    if (d < 0.0) s = -s;
  }
}

// same rotation plus its tangent t = s / c, for the scaled ("fast") rotations of the ping-pong sweeps.  Two Newton
// steps on the hardware estimate (relative error <= 2^-23 -> 2^-45 -> below rounding), first-order renormalisation.
__device__ __forceinline__ double rsqrt_nr2(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}
__device__ __forceinline__ double rcp_nr2(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
__device__ __forceinline__ void jacobi_cst(double app, double aqq, double apq, double thr, double& c, double& s, double& t) {
  c = 1.0; s = 0.0; t = 0.0;
  if (fabs(apq) <= thr) return;
  double d = aqq - app, b = 2.0 * apq;
  double h2 = d * d + b * b;
  if (apq != 0.0 && h2 > 1e-280 && h2 < 1e280) {
    double rh = rsqrt_nr2(h2);
    double u = 0.5 + 0.5 * fabs(d) * rh;     // in [0.5, 1]
    double rc = rsqrt_nr2(u);                // 1/sqrt(u); rc^2 = 1/u
    double sg = d < 0.0 ? -0.5 : 0.5;
    double w = sg * b * rh * rc;
    c = u * rc;
    s = w;
    t = w * rc * rc * rc * u;                // s / c = w / (u rc), with 1/u = rc^2
    double nrm = 1.5 - 0.5 * (c * c + s * s);
    c *= nrm; s *= nrm;
  } else if (apq != 0.0) {
    double sc = fmax(fabs(d), fabs(b));
    double dd = d / sc, bb = b / sc;
    double rh = 1.0 / sqrt(dd * dd + bb * bb);
    double u = 0.5 + 0.5 * fabs(dd) * rh;
    c = sqrt(u);
    s = 0.5 * bb * rh / c;
    if (d < 0.0) s = -s;
    t = s / c;
  }
}

// round-robin (chess tournament) schedule in closed form: index 0 stays in top slot 0, the other
// np-1 indices sit on a cycle  t1 -> t2 -> ... -> t(h-1) -> b(h-1) -> ... -> b0 -> t1  and advance one
// position per round.  Pair slot ia holds (top, bottom) = (pair_top, pair_bot) at round r.
__device__ __forceinline__ int cyc_content(int j, int r, int M, int half) {
  int j0 = j - r;
  if (j0 < 0) j0 += M;
  return (j0 <= half - 2) ? j0 + 1 : 3 * half - 2 - j0;
}
__device__ __forceinline__ int pair_top(int ia, int r, int M, int half) { return ia == 0 ? 0 : cyc_content(ia - 1, r, M, half); }
__device__ __forceinline__ int pair_bot(int ia, int r, int M, int half) { return cyc_content(2 * half - 2 - ia, r, M, half); }

// NT = 1024 (16 waves hide the LDS latency of the rotation passes) for large blocks, 256 for small.
// The LAST wave of the workgroup computes the next round's rotation parameters while the other
// waves apply the current round's rotations to the eigenvectors.
// LDS-ordered hand-off between lanes of ONE wave (DS operations of a wave execute in order; this only
// stops the compiler from reordering them and drains the counter before dependent reads)
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// neighbour-lane moves of an fp64 value on the DPP path (no LDS): lane i receives lane i+1 / lane i-1 across the
// whole 64-lane wave (wave_shl / wave_shr; verified on gfx950), 0 beyond the wave's ends
__device__ __forceinline__ double lane_next(double x) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_prev(double x) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// value of lane l (wave-uniform l) in every lane
__device__ __forceinline__ double lane_bcast(double x, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// LDS scratch of the register-resident (systolic) sweeps, in doubles: rotation inputs prm[2][65][4] and the
// boundary rows halo[NW][2][64][2] exchanged between neighbouring waves
static constexpr int kSysPrm = 2 * 65 * 4;
__host__ __device__ constexpr int sys_scratch_doubles(int nw) { return kSysPrm + nw * 256; }
static constexpr int kPpLda = 99;          // ping-pong variant: fixed stride of the MFMA-phase matrices (blocks up to 96; odd)
static constexpr int kPpLdp = 100;         // ping-pong variant: stride of the sweep layout (even: aligned 16-byte row pairs)
static constexpr int kSysHead = 16 + 66;   // red[16], sel[npg + 2 <= 130 ints] in front of A (fixed offsets)

// BLOCK = true: block Jacobi.  Indices are grouped in blocks of 8; a round pairs the blocks (round robin),
// one wave diagonalises each 16x16 diagonal sub-problem in place (one cyclic sweep, rotations accumulated
// in a 16x16 J), then A <- J'AJ and V <- VJ are applied as 16x16x16 products on v_mfma_f64_16x16x4_f64.
// 3 workgroup barriers per block round (nb-1 block rounds per sweep) instead of 2 per element round.
// ALG = 2: register-resident systolic sweeps (below).  SPW = pair slots (2 matrix rows each) per wave, RPW = eigenvector
// rows per wave; NT/64 * SPW >= ceil(n/2) and NT/64 * RPW >= n + 1 for every block of the launch.
template <bool V_LDS, int NT, int ALG = 0, int SPW = 1, int RPW = 1>
__device__ __forceinline__ void proj_body(const ProjArgs& a, const int k, const int role = 0) {      // role: 0 alone, 1 leader, 2 helper (ProjArgs::split)
  constexpr bool BLOCK = ALG == 1;
  constexpr bool SYS = ALG == 2;
  constexpr bool PP = ALG == 3;
  // ALG = 4: blocks 129 .. 160 (the reference's 151-wide cliques of width-50 networks, chordal_cliques.jl:33-36).  The full matrix does
  // not fit a CU's LDS beside anything else (160 x 161 doubles = 206 KB), its LOWER TRIANGLE does (103 KB), and the round-robin sweeps
  // only ever touch that: same sweeps as ALG = 0 on a packed triangle, eigenvectors in HBM / L2, the warm-start congruence through a
  // scratch matrix in HBM.
  constexpr bool PK = ALG == 4;
  static_assert(!PK || (!V_LDS && NT == 1024), "packed variant: V in HBM, 1024 threads");
  static_assert(!PP || (V_LDS && NT == 1024 && (RPW == 5 || RPW == 6 || RPW == 7)), "ping-pong sweeps: V in LDS, 1024 threads");
  extern __shared__ double lds[];
  if (a.pmode && a.pmode[k] != 0) return;      // (uniform over the workgroup, in front of every barrier)
  const int n = a.cn[k];
  const int np = (n + 1) & ~1;   // Jacobi dimension (even)
  const int npg = (n + 15) & ~15;  // storage / MFMA dimension (multiple of 16; zero rows, identity in V)
  const int half = np >> 1;
  constexpr bool PPL = ALG == 3;   // ping-pong sweeps keep a one-cell border around the matrix: 2 more rows / columns of storage
  const int lda = PPL ? kPpLda : npg + 1;    // odd stride (in doubles): column walks hit distinct banks; a constant for the ping-pong variant (immediate LDS offsets)
  const int nrow = PPL ? npg + 2 : npg;      // rows of LDS storage per matrix
  const int tid = threadIdx.x;
  // systolic variant: reduction scratch and the selection list sit in FRONT of A at fixed offsets, because the sweeps
  // reuse the (then dead) A / V storage as exchange scratch whatever this block's own size is
  double* A = SYS ? lds + kSysHead : lds;
  // 2 buffers x half pair descriptors {c, s, (p, q), pad} = 4 doubles each, 16-byte aligned
  double* desc = A + (PPL ? (size_t)nrow * kPpLdp : PK ? ((((size_t)npg * (npg + 1)) / 2 + 1) & ~(size_t)1) : (((size_t)nrow * lda + 1) & ~(size_t)1));
  // element (i, j) of the lower triangle (i >= j) / of the symmetric matrix in the variant's storage
  auto ixl = [&](int i, int j) { return PK ? ((i * (i + 1)) >> 1) + j : i * lda + j; };
  auto ixs = [&](int i, int j) { return i >= j ? ixl(i, j) : ixl(j, i); };
  double* red = SYS ? lds : desc + (BLOCK ? 0 : 4 * npg);   // 16 doubles of reduction scratch (block mode: no pair descriptors)
  int* sel = reinterpret_cast<int*>(red + 16);  // npg + 2 ints: eigen-indices on the chosen side, counters
  double* V;
  int ldv;
  if (V_LDS) { V = SYS ? desc : red + 16 + (npg >> 1) + 2; ldv = lda; }
  else { V = a.Vg + a.coff[k]; ldv = n; }
  const double* nuk = a.nu + a.coff[k];
  // block mode scratch: per block pair a 16x16 J (row-major) and 16 doubles of rotation parameters
  double* Jall = V + (size_t)npg * ldv;

#ifdef NNSDP_STAMPS
  long long sec_t[6]; sec_t[0] = clock64();
  long long rst[20];
  for (int i_ = 0; i_ < 20; ++i_) rst[i_] = 0;
#define RST(i) { rst[i] = clock64(); }
#else
#define RST(i)
#endif
  // ---- load: A = sym(nu_k) (full, both triangles, for the warm-start products), padded row/col zero; starting basis
  double fro2 = 0.0;
  const bool warm = a.warm != 0;
  if constexpr (PP) {
    // every global load of the block (matrix, read ONCE, and the warm basis) is issued before the first LDS store; the matrix is
    // symmetrised in LDS afterwards.  The block is n^2 contiguous doubles: thread t takes the 16-byte pieces t, t + 1024, ... of that
    // LINEAR range (8 + 8 wave instructions per thread instead of 12 + 12 guarded 8-byte ones - a CU issues one vector-memory wave
    // instruction per ~13 cycles whatever its width, tools/load_issue_probe.hip - every load unconditional from a clamped address:
    // no exec-mask branch around it) and finds (row, column) of its two elements afterwards.
    constexpr int MR = 5;            // ceil(96 * 96 / 2 / 1024) pieces per thread
    const int ln = tid & 63, wvi = tid >> 6;
    const double* vgk = a.Vg + a.coff[k];
    const int n2 = n * n, npc = (n2 + 1) >> 1;             // elements, pieces (the last one of an odd block is moved back by one)
    double2 ta[MR], tv[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int e = min(2 * (tid + NT * m), n2 - 2);
      if (NT * m < npc) {
        ta[m] = *reinterpret_cast<const double2*>(nuk + e);
        if (warm) tv[m] = *reinterpret_cast<const double2*>(vgk + e);
      }
    }
    // padding first (rows / columns n .. npg - 1: zero in A, identity in V), disjoint from what the pieces write
    const float rn = 1.0f / (float)n, rg = 1.0f / (float)npg;
    const int pad = npg - n;
    for (int q = tid; q < pad * npg; q += NT) {            // rows n .. npg - 1, every column
      const int r = (int)(((float)q + 0.5f) * rg), j = q - r * npg, i = n + r;
      A[i * lda + j] = 0.0; V[i + j * ldv] = (i == j) ? 1.0 : 0.0;
    }
    for (int q = tid; q < pad * n; q += NT) {              // columns n .. npg - 1, rows below n
      const int r = (int)(((float)q + 0.5f) * rn), i = q - r * n, j = n + r;
      A[i * lda + j] = 0.0; V[i + j * ldv] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int pc = tid + NT * m;
      if (pc < npc) {
        int e = 2 * pc;
        double a0 = ta[m].x, a1 = ta[m].y, v0 = warm ? tv[m].x : 0.0, v1 = warm ? tv[m].y : 0.0;
        const bool moved = e > n2 - 2;                     // (odd n^2: the last piece was read from n2 - 2 and holds (n2 - 2, n2 - 1))
        if (moved) { a0 = a1; v0 = v1; }
        int j = (int)(((float)e + 0.5f) * rn), i = e - j * n;          // nu_k[j n + i]: column j, row i (e < 9216: the float quotient is exact)
        if (!warm) v0 = (i == j) ? 1.0 : 0.0;
        A[i * lda + j] = a0; V[i + j * ldv] = v0;
        if (!moved) {
          ++i; if (i == n) { i = 0; ++j; }
          if (!warm) v1 = (i == j) ? 1.0 : 0.0;
          A[i * lda + j] = a1; V[i + j * ldv] = v1;
        }
      }
    }
    RST(12)
    __syncthreads();
    RST(13)
    for (int j = wvi; j < n; j += NT >> 6)
      for (int i = ln + j + 1; i < n; i += 64) {
        const double v = 0.5 * (A[i * lda + j] + A[j * lda + i]);
        A[i * lda + j] = v; A[j * lda + i] = v;
        fro2 += 2.0 * v * v;
      }
    if (tid < n) { const double v = A[tid * lda + tid]; fro2 += v * v; }
    RST(14)
    fro2 = block_sum(fro2, red);
  } else {
  if constexpr (PK) {
    for (int j = tid >> 6; j < npg; j += NT >> 6)
      for (int i = (tid & 63) + j; i < npg; i += 64) {
        double v = 0.0;
        if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
        A[ixl(i, j)] = v;
        fro2 += (i == j) ? v * v : 2.0 * v * v;
      }
  } else {
  for (int j = tid >> 6; j < npg; j += NT >> 6)
    for (int i = tid & 63; i < npg; i += 64) {
      double v = 0.0;
      if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
      A[i * lda + j] = v;
      fro2 += v * v;
    }
  }
  fro2 = block_sum(fro2, red);
  // ---- starting basis
  if (V_LDS) {
    for (int j = tid >> 6; j < npg; j += NT >> 6)
      for (int i = tid & 63; i < npg; i += 64) {
        double v = (i == j) ? 1.0 : 0.0;
        if (warm && i < n && j < n) v = a.Vg[a.coff[k] + (size_t)j * n + i];
        V[i + j * ldv] = v;
      }
  } else if (!warm) {
    for (int j = tid >> 6; j < n; j += NT >> 6)
      for (int i = tid & 63; i < n; i += 64) V[i + (size_t)j * ldv] = (i == j) ? 1.0 : 0.0;
  }
  }
  __syncthreads();
#ifdef NNSDP_STAMPS
  sec_t[1] = clock64();
#endif
  const int nv = V_LDS ? np : n;  // rows/cols of V that exist
  // A <- V' A V on the matrix cores (V in LDS); a lambda because the refinement stage's fall-back re-runs it
  auto congruence = [&](const int part) {
    // A <- V' A V with v_mfma_f64_16x16x4_f64: T = A V (all 16x16 tiles), then A' = V' T (lower tiles; the
    // Jacobi sweeps read the lower triangle only).  Operand maps (verified on gfx950): lane l holds
    // A[l&15][l>>4], B[l>>4][l&15]; result reg r of lane l is C[(l>>4) + 4r][l&15].
    // part: 0 = the whole product; 1 / 2 = the leader's / the helper's tile columns of a split block (helper: columns 1 .. hc,
    // leader: 0 and hc + 1 .. nt - 1; nt = 6: helper 12 tiles of T and 9 of B, leader 24 and 12)
    constexpr int NW = NT / 64;
    constexpr int MAXT = (NT == 512) ? 5 : 3;  // ceil(36 tiles / 16 waves), ceil(36 / 8), ceil(9 / 4)
    const int nt = npg >> 4, ks = (np + 3) >> 2;   // rows / columns past np are zero (identity in V): the K loop stops at np
    const int lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, lc = lane >> 4;
    const int hc = nt >= 6 ? nt / 3 : 1;        // (the helper gets LESS than half: its hand-over - stores, count, the leader's reads - runs while the leader still computes)
    const int ncol = part == 0 ? nt : part == 2 ? hc : nt - hc;
    auto colof = [&](int c) { return part == 0 ? c : part == 2 ? c + 1 : (c == 0 ? 0 : hc + c); };
    // lower tile number t of this part -> (ti, tj); false past the end
    auto lower_tile = [&](int t, int& ti, int& tj) {
      if (part == 0) {
        if (t >= nt * (nt + 1) / 2) return false;
        ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        tj = t - ti * (ti + 1) / 2;
        return true;
      }
      for (int c = 0; c < ncol; ++c) {
        const int cj = colof(c), cnt = nt - cj;
        if (t < cnt) { tj = cj; ti = cj + t; return true; }
        t -= cnt;
      }
      return false;
    };
    d4_t acc[MAXT];
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      int t = wv + m * NW;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
      if (t < nt * ncol) {
        int ti = t / ncol, tj = colof(t - ti * ncol);
        const double* ap = A + (16 * ti + lr) * lda + lc;
        const double* bp = V + lc + (size_t)(16 * tj + lr) * ldv;
        for (int kk = 0; kk < ks; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk], bp[4 * kk], c, 0, 0, 0);
      }
      acc[m] = c;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      int t = wv + m * NW;
      if (t < nt * ncol) {
        int ti = t / ncol, tj = colof(t - ti * ncol);
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(16 * ti + lc + 4 * r) * lda + 16 * tj + lr] = acc[m][r];
      }
    }
    __syncthreads();
    int tis[MAXT], tjs[MAXT];
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      int t = wv + m * NW;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
      tis[m] = -1; tjs[m] = 0;
      int ti, tj;
      if (lower_tile(t, ti, tj)) {
        tis[m] = ti; tjs[m] = tj;
        const double* ap = V + lc + (size_t)(16 * ti + lr) * ldv;   // V'[i][k] = V[k][i]
        const double* bp = A + lc * lda + 16 * tj + lr;             // T[k][j]
        for (int kk = 0; kk < ks; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk], bp[4 * kk * lda], c, 0, 0, 0);
      }
      acc[m] = c;
    }
    if (part == 2) {
      // helper: its tiles of B go to the exchange buffer in tile order (tile (ti, tj) at (ti nt + tj) x 256, element (lc + 4 r, lr) at
      // 64 r + lane), then one release of the block's exchange count; nothing else of this workgroup is needed
      double* const out = a.sB + (size_t)k * kSplitTileDoubles;
#pragma unroll
      for (int m = 0; m < MAXT; ++m)
        if (tis[m] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(out + (size_t)(tis[m] * nt + tjs[m]) * 256 + 64 * r + lane, acc[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      // (every element is an agent-scope store of its own - coherent at the device's memory side without an L2 write-back - so the
      // hand-over only has to ORDER them in front of the count: wait for this wave's stores, meet the other waves, then count)
      if (tid == 0) __hip_atomic_store(a.sxcc + k, (unsigned)__builtin_amdgcn_s_getreg(kHwRegXccId), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      // ONE agent-scope release for the whole workgroup (the barrier orders every wave's stores in front of it): the L2 write-back it
      // implies is what a leader on ANOTHER XCD needs - under multi-process contention the pair does not always stay on one XCD
      if (tid == 0) __hip_atomic_store(a.sack + k, a.sack[k] + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      if (tis[m] >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) A[(16 * tis[m] + lc + 4 * r) * lda + 16 * tjs[m] + lr] = acc[m][r];
      }
    }
    if (part == 1) {
      // leader: the helper's tiles (columns 1 .. hc), one per wave and round, behind an acquire of the exchange count
      const unsigned want = a.sseen[k] + 1u;
      if (tid == 0) {
        long long spins = 0;
        while (__hip_atomic_load(a.sack + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > a.spin_limit) { atomicExch(a.serr, 1); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // (one invalidate, after the count has arrived - not one per poll)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __syncthreads();
      const double* const in = a.sB + (size_t)k * kSplitTileDoubles;
      int nth = 0;
      for (int c = 1; c <= hc; ++c) nth += nt - c;
      for (int t = wv; t < nth; t += NW) {
        int tt = t, ti = 0, tj = 0;
        for (int c = 1; c <= hc; ++c) { const int cnt = nt - c; if (tt < cnt) { tj = c; ti = c + tt; break; } tt -= cnt; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          A[(16 * ti + lc + 4 * r) * lda + 16 * tj + lr] = __hip_atomic_load(in + (size_t)(ti * nt + tj) * 256 + 64 * r + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (tid == 0) a.sseen[k] = want;
    }
    __syncthreads();
    if (BLOCK) {   // block mode reads both triangles: mirror the lower one
      for (int j = tid >> 6; j < npg; j += NT >> 6)
        for (int i = (tid & 63) + j + 1; i < npg; i += 64) A[j * lda + i] = A[i * lda + j];
      __syncthreads();
    }
  };
  auto pk_congruence = [&]() {
    // A <- V' A V for the packed variant, both products on the matrix cores with the streamed operand staged through LDS ONCE:
    //   pass 1  T[:, J] = A V[:, J] for 32-column panels J of V (panel in LDS, A read through the packed triangle), T row-major into
    //           the HBM scratch;
    //   pass 2  B = T'V (= V'T: B is symmetric) accumulated over chunks of 16 rows l of T and V (both chunks in LDS: rows of the
    //           row-major T are contiguous, 16 consecutive l of a column of V are one 128-byte line), lower tiles, four per wave,
    //           written into the packed storage at the end (A is dead once pass 1 is through).
    // (The first form - 4 x 2 register tiles on the vector pipe, every thread streaming its operands from L2 - took 1.05 M cycles of
    // a 151-wide block's 4.9 M per warm launch.)
    constexpr int NW = NT / 64, kPanelLd = 161;
    double* const Pn = red + 16 + (npg >> 1) + 2;        // 32 x kPanelLd doubles
    double* const Tk = a.Tg + a.coff[k];
    const int nt = (n + 15) >> 4, ks = (n + 3) >> 2;
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lc = lane >> 4;
    for (int j0 = 0; j0 < n; j0 += 32) {
      {
        const int jl = tid >> 5;                           // 32 threads per column, 32 columns
        for (int kx = tid & 31; kx < 4 * ks; kx += 32) Pn[jl * kPanelLd + kx] = (j0 + jl < n && kx < n) ? V[kx + (size_t)(j0 + jl) * ldv] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int t = wv + m * NW, hh = t / nt, ti = t - hh * nt;      // (row tile, 16-column half of the panel)
        if (hh < 2 && j0 + 16 * hh < n) {
          d4_t c = {0.0, 0.0, 0.0, 0.0};
          const int i = 16 * ti + lr;
          const double* bp = Pn + (16 * hh + lr) * kPanelLd + lc;
          for (int kk = 0; kk < ks; ++kk) {
            const int kx = 4 * kk + lc;
            const double av = (i < n && kx < n) ? A[ixs(i, kx)] : 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[4 * kk], c, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + lc + 4 * r, col = j0 + 16 * hh + lr;
            if (row < n && col < n) Tk[(size_t)row * n + col] = c[r];
          }
        }
      }
      __syncthreads();
    }
    // (every read of A is done, and the block's writes of T are visible to the block)
    {
      constexpr int kTilesPerWave = 4;
      const int ntl = nt * (nt + 1) / 2;
      double* const Tc = Pn;
      double* const Vc = Pn + 16 * kPanelLd;
      int tti[kTilesPerWave], ttj[kTilesPerWave];
      d4_t acc[kTilesPerWave];
#pragma unroll
      for (int m = 0; m < kTilesPerWave; ++m) {
        const int t = wv + m * NW;
        int ti = -1, tj = -1;
        if (t < ntl) { ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; tj = t - ti * (ti + 1) / 2; }
        tti[m] = ti; ttj[m] = tj;
        acc[m] = d4_t{0.0, 0.0, 0.0, 0.0};
      }
      for (int l0 = 0; l0 < n; l0 += 16) {
        {
          const int ll = tid >> 6, l = l0 + ll;              // T: one wave per row l, lanes along i
          for (int i = tid & 63; i < 16 * nt; i += 64) Tc[ll * kPanelLd + i] = (l < n && i < n) ? Tk[(size_t)l * n + i] : 0.0;
          const int lv = tid & 15, l2 = l0 + lv;             // V: 16 lanes along l (one line), columns j
          for (int j = tid >> 4; j < 16 * nt; j += NT >> 4) Vc[lv * kPanelLd + j] = (l2 < n && j < n) ? V[l2 + (size_t)j * ldv] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < kTilesPerWave; ++m)
          if (tti[m] >= 0) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const int kx = 4 * k4 + lc;
              acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(Tc[kx * kPanelLd + 16 * tti[m] + lr], Vc[kx * kPanelLd + 16 * ttj[m] + lr], acc[m], 0, 0, 0);
            }
          }
        __syncthreads();
      }
#pragma unroll
      for (int m = 0; m < kTilesPerWave; ++m)
        if (tti[m] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * tti[m] + lc + 4 * r, col = 16 * ttj[m] + lr;
            if (row < n && col <= row) A[ixl(row, col)] = acc[m][r];
          }
        }
    }
    __syncthreads();
  };
  if (warm && V_LDS) {
    congruence(PP ? role : 0);
    if (PP && role == 2) return;
  } else if (warm && PK) {
    pk_congruence();
  } else if (warm) {
    // (blocks too large to keep V in LDS) A <- V' A V as two register-tiled products (4 x 2 tiles, accumulators in VGPRs):
    //   T = A V   (written over A),   A' = V' T   (written over T)
    constexpr int kTilesMax = 2048 / NT;               // (128/4) * (128/2) tiles at most
    const int ti_n = (np + 3) >> 2, tj_n = np >> 1;    // tile grid
    double acc[kTilesMax][8];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int m = 0; m < kTilesMax; ++m) {
        int t = tid + m * NT;
        int tj = t / ti_n, ti = t - tj * ti_n;         // consecutive threads -> consecutive row tiles
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        if (tj < tj_n) {
          int i0 = ti * 4, j0 = tj * 2;
          int i1 = min(i0 + 1, np - 1), i2 = min(i0 + 2, np - 1), i3 = min(i0 + 3, np - 1);
          if (pass == 0) {
            // T[i][j] = sum_l A[i][l] V[l][j]
            const double* v0 = V + (size_t)min(j0, nv - 1) * ldv;
            const double* v1 = V + (size_t)min(j0 + 1, nv - 1) * ldv;
            const double *a0 = A + i0 * lda, *a1 = A + i1 * lda, *a2 = A + i2 * lda, *a3 = A + i3 * lda;
            for (int l = 0; l < nv; ++l) {
              double x0 = v0[l], x1 = v1[l];
              double y0 = a0[l], y1 = a1[l], y2 = a2[l], y3 = a3[l];
              c0 += y0 * x0; c1 += y1 * x0; c2 += y2 * x0; c3 += y3 * x0;
              d0 += y0 * x1; d1 += y1 * x1; d2 += y2 * x1; d3 += y3 * x1;
            }
            if (j0 >= nv) { c0 = a0[j0]; c1 = a1[j0]; c2 = a2[j0]; c3 = a3[j0]; }
            if (j0 + 1 >= nv) { d0 = a0[j0 + 1]; d1 = a1[j0 + 1]; d2 = a2[j0 + 1]; d3 = a3[j0 + 1]; }
          } else {
            // A'[i][j] = sum_l V[l][i] T[l][j]
            const double *u0 = V + (size_t)min(i0, nv - 1) * ldv, *u1 = V + (size_t)min(i1, nv - 1) * ldv;
            const double *u2 = V + (size_t)min(i2, nv - 1) * ldv, *u3 = V + (size_t)min(i3, nv - 1) * ldv;
            for (int l = 0; l < nv; ++l) {
              double x0 = A[l * lda + j0], x1 = A[l * lda + j0 + 1];
              double y0 = u0[l], y1 = u1[l], y2 = u2[l], y3 = u3[l];
              c0 += y0 * x0; c1 += y1 * x0; c2 += y2 * x0; c3 += y3 * x0;
              d0 += y0 * x1; d1 += y1 * x1; d2 += y2 * x1; d3 += y3 * x1;
            }
            if (i0 >= nv) { c0 = A[i0 * lda + j0]; d0 = A[i0 * lda + j0 + 1]; }
            if (i1 >= nv) { c1 = A[i1 * lda + j0]; d1 = A[i1 * lda + j0 + 1]; }
            if (i2 >= nv) { c2 = A[i2 * lda + j0]; d2 = A[i2 * lda + j0 + 1]; }
            if (i3 >= nv) { c3 = A[i3 * lda + j0]; d3 = A[i3 * lda + j0 + 1]; }
          }
        }
        acc[m][0] = c0; acc[m][1] = c1; acc[m][2] = c2; acc[m][3] = c3;
        acc[m][4] = d0; acc[m][5] = d1; acc[m][6] = d2; acc[m][7] = d3;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kTilesMax; ++m) {
        int t = tid + m * NT;
        int tj = t / ti_n, ti = t - tj * ti_n;
        if (tj < tj_n) {
          int i0 = ti * 4, j0 = tj * 2;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (i0 + r < np) { A[(i0 + r) * lda + j0] = acc[m][r]; A[(i0 + r) * lda + j0 + 1] = acc[m][4 + r]; }
        }
      }
      __syncthreads();
    }
    // exact symmetry into the lower triangle (the two passes round differently)
    for (int j = tid >> 6; j < np; j += NT >> 6)
      for (int i = tid & 63; i < np; i += 64)
        if (i > j) A[i * lda + j] = 0.5 * (A[i * lda + j] + A[j * lda + i]);
    __syncthreads();
  }

#ifdef NNSDP_STAMPS
  sec_t[2] = clock64();
#endif
  // ---- Jacobi sweeps (only the lower triangle of A is read and written from here on)
  // (wave-uniform values are moved to scalar registers: the 1024-thread variants run at the 128-VGPR limit)
  auto uniform = [](double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
  };
  const double tolv = a.tol_dev ? *a.tol_dev : a.tol;
  fro2 = uniform(fro2);
  const double thresh2 = uniform(tolv * tolv * fro2);  // converged when off(A) <= tol |A|_F, measured directly before each sweep
  const double rot_thr = uniform(0.5 * tolv * sqrt(fro2) / np);   // skipped elements together stay below tol/2
  int nrot = 0;
  constexpr int kParamLanes = PK ? 128 : 64;    // pair slots the parameter wave(s) can serve: half <= 64 up to n = 128, <= 80 in the packed variant (two waves)
  const int plane = tid - (NT - kParamLanes);   // lane index inside the parameter wave(s) (>= 0 there)
  const int M = np - 1;                         // rounds per sweep
  // static work assignment (round independent, so the integer divisions happen once):
  //  * 2x2 blocks of the lower block triangle; block rows ia and half-1-ia are paired into rows of equal
  //    length half+1 and the rectangle ceil(half/2) x (half+1) is dealt out thread by thread
  //  * eigenvector units (pair slot, chunk of 32 rows) dealt out to 32-lane groups of all waves but the last
  constexpr int MAXB = PK ? 4 : (NT == 1024) ? (V_LDS ? 2 : 3) : 9;   // ceil(ceil(half/2) * (half+1) / NT): half <= 48 with V in LDS, <= 80 packed
  // ceil(half * ceil(nv/32) / VG) for the sizes each variant is launched with (n <= 96 when V is in LDS)
  constexpr int MAXU = PK ? 15 : (NT == 1024) ? (V_LDS ? 5 : 9) : 7;   // V in HBM: blocks up to n = 128 (160 packed: 80 x 5 units over 28 groups)
  constexpr int VG = (NT - kParamLanes) / 32;
#ifdef NNSDP_STAMPS
  long long st_acc[4] = {0, 0, 0, 0};
#define STAMP(i, tprev) { long long tn_ = clock64(); st_acc[i] += tn_ - tprev; tprev = tn_; }
#else
#define STAMP(i, tprev)
#endif
  int sweeps = 0;
  int pofs = 0;   // systolic sweeps: eigen-index j of the result sits at position j + pofs (the padded index travels)
  // ---- refinement stage (ping-pong variant, warm blocks): GEMM-only update of the PERSISTENT eigenbasis.
  // Once the ADMM iterate moves slowly, B = V'AV is diagonal up to a perturbation E far below the spectral gaps, and ONE simultaneous
  // first-order rotation of all pairs, K_ij = B_ij / (d_j - d_i), does what a Jacobi sweep does (quadratic: off(B) -> ~ |E| |K|) at
  // the price of two more products on the matrix cores instead of np rounds of plane rotations.  The step is Ogita & Aishima's
  // refinement (Japan J. Indust. Appl. Math. 35, 2018): E~_ij = (B_ij + l_j R_ij) / (l_j - l_i), E~_ii = R_ii / 2 with
  // R = I - V'V, V <- V (I + E~), which restores orthogonality to second order at the same time.  Pairs it cannot resolve
  // (|B_ij| > kcap |d_j - d_i|: near-degenerate pairs) get E~_ij = R_ij / 2 and their coupling counts as unresolved; it is far
  // below the tolerance late in a solve (both eigenvalues sit in the near-zero cluster) and large early, when the block goes on to
  // the exact sweeps below.  Acceptance without a check of the result: predicted off = 1.5 |E|_F |K|_F + unresolved <=
  // refine_acc x tol |A|_F (measured on oracle iterates: tests/experiments/refine_proj2.py).
  bool refined = false;
  int side_force = 0;     // refinement step accepted for ONE side of the spectrum: the reconstruction must use it (+1 positive, -1 negative)
  if constexpr (PP) {
    int wait = 0, level = 0, credit = 0, gcred = 0;
    double rdef = 0.0;      // estimate (upper bound) of the basis' defect |I - V'V|_F since it was last measured
    if (a.rstate) {
      const int rs = a.rstate[4 * k]; wait = rs & 255; level = (rs >> 8) & 255; credit = (rs >> 16) & 255; gcred = (rs >> 24) & 15;
      rdef = *reinterpret_cast<const double*>(a.rstate + 4 * k + 2);
    }
    // EVERY wave must have read the block's back-off state before thread 0 rewrites it (the skip branch below does so at once): a
    // wave that loaded the word after that store saw wait - 1, entered the stage and its barriers while the others skipped it - a
    // barrier mismatch that showed only when three processes shared the card (the two-rank tests: one solve in five diverged)
    __syncthreads();
    const int rmode = a.refine;
    if (warm && rmode != 0 && wait == 0) {
      constexpr int NW = NT / 64;
      const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lc = lane >> 4;
      const int nt = npg >> 4, ntl = nt * (nt + 1) / 2, ks = (np + 3) >> 2;
      double* const dvec = desc;            // B_ii
      double* const rdg = desc + npg;       // R_ii
      double* const cs1 = desc + 2 * npg;   // column sums of E~^2
      double* const cs2 = desc + 3 * npg;   // column sums of E~^2 d
      int tti[2], ttj[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int t = wv + m * NW;
        int ti = -1, tj = -1;
        if (t < ntl) { ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; tj = t - ti * (ti + 1) / 2; }
        tti[m] = ti; ttj[m] = tj;
      }
      d4_t g[2];
      // Gram matrix V'V on the wave's lower tiles (registers)
      auto gram = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          d4_t c = {0.0, 0.0, 0.0, 0.0};
          if (tti[m] >= 0) {
            const double* ap = V + lc + (size_t)(16 * tti[m] + lr) * ldv;
            const double* bp = V + lc + (size_t)(16 * ttj[m] + lr) * ldv;
            for (int kk = 0; kk < ks; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk], bp[4 * kk], c, 0, 0, 0);
          }
          g[m] = c;
        }
      };
      // R_ii and the squared norm of R from the tiles
      auto gram_diag = [&]() {
        double r2 = 0.0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i < n && j < n) {
                if (i == j) { const double v = 1.0 - g[m][r]; rdg[i] = v; r2 += v * v; }
                else if (i > j) r2 += 2.0 * g[m][r] * g[m][r];
              } else if (i == j && i < npg) rdg[i] = 0.0;
            }
        return block_sum(r2, red);
      };
      // V <- V + V X with X in the A buffer (all tiles; accumulators in registers, V updated in place)
      auto v_update = [&]() {
        constexpr int MAXT = 3;
        d4_t acc[MAXT];
#pragma unroll
        for (int m = 0; m < MAXT; ++m) {
          const int t = wv + m * NW;
          d4_t c = {0.0, 0.0, 0.0, 0.0};
          if (t < nt * nt) {
            const int ti = t / nt, tj = t - ti * nt;
            const double* ap = V + 16 * ti + lr + (size_t)lc * ldv;       // V(i, k)
            const double* bp = A + lc * lda + 16 * tj + lr;               // X(k, j)
            for (int kk = 0; kk < ks; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(size_t)4 * kk * ldv], bp[4 * kk * lda], c, 0, 0, 0);
          }
          acc[m] = c;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MAXT; ++m) {
          const int t = wv + m * NW;
          if (t < nt * nt) {
            const int ti = t / nt, tj = t - ti * nt;
#pragma unroll
            for (int r = 0; r < 4; ++r) V[16 * ti + lc + 4 * r + (size_t)(16 * tj + lr) * ldv] += acc[m][r];
          }
        }
        __syncthreads();
      };
      const double kcap = a.refine_kcap;
      const double T = tolv * sqrt(fro2), accT = a.refine_acc * T;
      double r2 = 0.0, off2 = 0.0, k2 = 0.0, unpp = 0.0, unnn = 0.0, unx = 0.0, kd2 = 0.0, cpos = 0.0, cneg = 0.0;
      // analysis of the wave's pairs (i > j): nothing is written, so a rejected block reaches the sweeps untouched.
      // An unresolved coupling between two eigenvalues of the SAME sign costs nothing when the projection is rebuilt from the OTHER
      // side of the spectrum (W = sum over the positive side, or sym(nu) minus the sum over the negative side): the untouched side
      // only has to keep its inertia (b^2 < d_i d_j).  So the unresolved mass is kept per side and the side to rebuild from is the
      // smaller one when that passes, the other one when only that passes (measured on W40-D20 iterates at residual 4e-5: rejected
      // blocks 25 % -> 7 %; the near-degenerate pairs sit almost always among the negative eigenvalues).
      auto analyse = [&]() {
        double o2 = 0.0, q2 = 0.0, upp = 0.0, unn = 0.0, ux = 0.0, qd2 = 0.0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i > j && i < n) {
                const double b = A[i * lda + j], rr = -g[m][r];
                const double li = dvec[i] * (1.0 + rdg[i]), lj = dvec[j] * (1.0 + rdg[j]);
                const double gap = lj - li;
                o2 += 2.0 * b * b;
                if (fabs(b) <= kcap * fabs(gap) && gap != 0.0) {
                  const double e = (b + lj * rr) * rcp_nr2(gap), f = rr - e;
                  q2 += e * e + f * f;
                  qd2 += e * e * lj * lj + f * f * li * li;
                  A[j * lda + i] = e;            // kept for the step in the mirror position (nothing reads the upper triangle of B)
                } else {
                  A[j * lda + i] = __longlong_as_double(0x7ff8000000000000LL);      // "not resolved"
                  q2 += 0.5 * rr * rr;
                  const double dd = dvec[i] * dvec[j];
                  if (b * b < dd) { if (dvec[i] > 0.0) upp += 2.0 * b * b; else unn += 2.0 * b * b; }   // same sign, inertia kept
                  else ux += 2.0 * b * b;
                }
              }
            }
        // the six sums in ONE reduction (two barriers instead of sixteen): wave sums into rsc[q * 16 + wave] (96 doubles behind this
        // block's V: the V region is a bordered matrix of (npg + 2) x 100 doubles for the sweeps, V itself takes npg x 99 of them),
        // 16-lane butterflies over the waves, totals into red[0..5] (free after the first barrier: every wave is through gram_diag's reduction by then); the sign counts by
        // ballot into rsc[96..99]
        double* const rsc = V + (size_t)npg * ldv;
        static_assert(NW == 16, "the refinement stage's reductions assume 16 waves");
        o2 = wave_sum(o2); q2 = wave_sum(q2); upp = wave_sum(upp); unn = wave_sum(unn); ux = wave_sum(ux); qd2 = wave_sum(qd2);
        if (lane == 0) { rsc[wv] = o2; rsc[16 + wv] = q2; rsc[32 + wv] = upp; rsc[48 + wv] = unn; rsc[64 + wv] = ux; rsc[80 + wv] = qd2; }
        if (tid < 128) {
          const double dv = tid < n ? dvec[tid] : 0.0;
          const unsigned long long bp = __ballot(dv > 0.0), bn = __ballot(dv < 0.0);
          if (lane == 0) { rsc[96 + 2 * wv] = (double)__popcll(bp); rsc[97 + 2 * wv] = (double)__popcll(bn); }
        }
        __syncthreads();
        if (tid < 96) {
          const double v = row_sum16(rsc[tid]);
          if ((tid & 15) == 0) red[tid >> 4] = v;
        }
        __syncthreads();
        off2 = uniform(red[0]); k2 = uniform(red[1]); unpp = uniform(red[2]); unnn = uniform(red[3]); unx = uniform(red[4]); kd2 = uniform(red[5]);
        cpos = uniform(rsc[96] + rsc[98]); cneg = uniform(rsc[97] + rsc[99]);
        __syncthreads();      // (the reduction scratch is free again: the paths that follow reuse it without another barrier)
      };
      // the step: X = E~ + E~^2 / 2 over B (both triangles and the diagonal), V <- V (I + X); with_sums: column sums of E~^2 and
      // E~^2 d for the second-order eigenvalues (only the unchecked acceptance needs them)
      auto step = [&](const bool with_sums) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i > j) {
                double e = 0.0, f = 0.0;
                if (i < n) {
                  const double rr = -g[m][r];
                  e = A[j * lda + i];            // from the analysis: E~_ij of a resolved pair, NaN otherwise
                  if (e == e) f = rr - e;
                  else { e = 0.5 * rr; f = e; }
                }
                A[i * lda + j] = e; A[j * lda + i] = f;
              } else if (i == j) A[i * lda + i] = 0.5 * rdg[i];
            }
        __syncthreads();
        RST(4)
        if (with_sums) {
          const int col = tid >> 3, part = tid & 7;     // 8 lanes per column
          double s1 = 0.0, s2 = 0.0;
          if (col < n)
            for (int kx = part; kx < n; kx += 8)
              if (kx != col) { const double e = A[kx * lda + col]; s1 += e * e; s2 += e * e * dvec[kx]; }
          s1 = row_sum8(s1); s2 = row_sum8(s2);
          if (part == 0 && col < npg) { cs1[col] = s1; cs2[col] = s2; }
        }
        RST(5)
        // second-order term of the rotation: X = E~ + E~^2 / 2 ~ exp(K) - I, so that V (I + X) is orthogonal to THIRD order - with
        // the first-order step alone the projection carries an error |K^2 D| that no later iteration takes back.  E~^2 is symmetric
        // (E~ is antisymmetric up to R): lower tiles on the matrix cores, added to both triangles
        d4_t sq[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          d4_t c = {0.0, 0.0, 0.0, 0.0};
          if (tti[m] >= 0) {
            const double* ap = A + (16 * tti[m] + lr) * lda + lc;      // E~(i, k)
            const double* bp = A + lc * lda + 16 * ttj[m] + lr;        // E~(k, j)
            for (int kk = 0; kk < ks; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk], bp[4 * kk * lda], c, 0, 0, 0);
          }
          sq[m] = c;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i < n && j < n) {
                if (i > j) { A[i * lda + j] += 0.5 * sq[m][r]; A[j * lda + i] += 0.5 * sq[m][r]; }
                else if (i == j) A[i * lda + i] += 0.5 * sq[m][r];
              }
            }
        __syncthreads();
        RST(6)
        v_update();
        RST(7)
      };
      // B = V'AV again from the matrix in HBM (the A buffer held X)
      auto rebuild_B = [&]() {
        for (int j = wv; j < npg; j += NW)
          for (int i = lane; i < npg; i += 64) {
            double v = 0.0;
            if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
            A[i * lda + j] = v;
          }
        __syncthreads();
        congruence(0);
      };
      // predicted error of the projection after one step: second-order residual coupling |[E, K]| / 2 <= |E| |K| (measured: 0.3 |E| |K|
      // at the median, 1.3 at the worst), the couplings left alone, and the third-order defect of exp(K) ~ I + K + K^2 / 2, which hits
      // the projection as |K^3 D| / 3 <= |K|^2 |K D| / 3 (D = eigenvalues: large rotations between eigenvalues near zero cost nothing).
      // Round 0 looks at the block as it arrives.  A block whose prediction fails by less than 30 x takes the step anyway and is
      // CHECKED: B is rebuilt with the new basis (round 1) and accepted when its measured off(B) is inside the accepted level - a
      // third of the price of a sweep.  Whatever is left goes on to the exact sweeps from a valid (B, V).
      int outcome = 2;      // 0: converged as it arrived, 1: one step (unchecked), 2: on to the sweeps, 3: step + check
      RST(0)
      bool check = false, far = true, loose = false;      // far: the prediction missed by more than 10 x (the iterate still moves fast)
      double pred0 = 0.0;
      auto decide = [&]() {
        if (off2 <= T * T && r2 <= tolv * tolv) { outcome = 0; refined = true; }
        else if (r2 <= 1e-4) {
          pred0 = 1.5 * sqrt(off2) * sqrt(k2) + k2 * sqrt(kd2) * (1.0 / 3.0);
          const double pred_pos = pred0 + sqrt(unpp + unx), pred_neg = pred0 + sqrt(unnn + unx);    // rebuilding from the positive / negative side
          const bool prefer_pos = cpos <= cneg;
          const bool kok = k2 <= a.refine_k2cap;      // |K|_F <= 0.3: the step is a rotation to |K|^3 / 6 < 5e-3 whatever the eigenvalues it touches
          if (kok && (prefer_pos ? pred_pos : pred_neg) <= accT) side_force = prefer_pos ? 1 : -1;
          else if (kok && (prefer_pos ? pred_neg : pred_pos) <= accT) side_force = prefer_pos ? -1 : 1;
          // an isolated near miss: a block that took the step at the regular level on its last 16 visits may take ONE step whose prediction
          // is up to refine_loose x higher (the prediction is an upper bound: the error itself is 0.06 x it at the median) instead of
          // holding the whole launch up for a sweep
          else if (kok && credit >= 16 && fmin(pred_pos, pred_neg) <= a.refine_loose * accT) { side_force = pred_pos <= pred_neg ? 1 : -1; loose = true; }
          // (the checked form needs a step that is a rotation at all: |K|_F <= 0.3 keeps V (I + X) orthogonal to |K|^3 / 6 < 5e-3,
          // which the Newton-Schulz repair below takes back if the check then fails)
          else check = rmode >= 2 && k2 <= a.refine_k2cap && fmin(pred_pos, pred_neg) <= 30.0 * accT;
          far = fmin(pred_pos, pred_neg) > 10.0 * accT;
        }
      };
      // The Gram product is not taken on every visit: a step with an antisymmetric K leaves (I + K + K^2 / 2)'(I + K + K^2 / 2) =
      // I + K^4 / 4, so the defect of V grows by |K|_F^4 / 4 per step at most - the estimate rdef carries that bound from the last
      // measurement, three visits in four run with R = 0 (no re-orthogonalisation term, a fifth of the stage's matrix work saved),
      // and any visit whose estimate comes near the accepted error level measures again.  Zero-initialised state measures first.
      const bool do_gram = !a.rstate || gcred == 0 || !(rdef <= 0.03 * a.refine_acc * tolv);
      bool measured = do_gram;
      if (do_gram) {
        gram();
        RST(1)
        if (tid < npg) dvec[tid] = tid < n ? A[tid * lda + tid] : 0.0;
        r2 = uniform(gram_diag());           // (two barriers inside: dvec / rdg are visible afterwards)
      } else {
        g[0] = d4_t{0.0, 0.0, 0.0, 0.0}; g[1] = d4_t{0.0, 0.0, 0.0, 0.0};
        if (tid < npg) { dvec[tid] = tid < n ? A[tid * lda + tid] : 0.0; rdg[tid] = 0.0; }
        r2 = rdef * rdef;
        __syncthreads();
      }
      RST(2)
      analyse();
      decide();
      RST(3)
      // What first order cannot resolve late in a solve is almost always ONE pair of eigenvalues on either side of zero (two thirds of
      // the rejections of W40-D20 after iteration 12 000): when the rest of the prediction passes, rotate that pair exactly - two columns
      // of V, the pair's rows of B (lower triangle: the strict upper tiles still hold the first product of the congruence); B and V
      // stay consistent - and look again (at most refine_pivots times; Gram product and analysis only).
      int pivots = 0;
      if (__builtin_expect(!refined && side_force == 0 && r2 <= 1e-4 && k2 <= a.refine_k2cap && unx > 0.0 && a.refine_pivots > 0, 0)) {
        for (;;) {
          // the largest coupling that counts against both sides
          double pm = 0.0;
          int pidx = 0;
#pragma unroll
          for (int m = 0; m < 2; ++m)
            if (tti[m] >= 0)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
                if (i > j && i < n) {
                  const double b = A[i * lda + j];
                  const double li = dvec[i] * (1.0 + rdg[i]), lj = dvec[j] * (1.0 + rdg[j]);
                  const double gap = lj - li;
                  if (!(fabs(b) <= kcap * fabs(gap) && gap != 0.0) && !(b * b < dvec[i] * dvec[j]) && b * b > pm) { pm = b * b; pidx = (i << 8) | j; }
                }
              }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(pm, o, 64);
            const int oi = __shfl_xor(pidx, o, 64);
            if (om > pm || (om == pm && oi > pidx)) { pm = om; pidx = oi; }
          }
          if (lane == 0) { cs1[wv] = pm; cs2[wv] = (double)pidx; }
          __syncthreads();
          pm = 0.0; pidx = 0;
          for (int w_ = 0; w_ < NW; ++w_) { const double om = cs1[w_]; const int oi = (int)cs2[w_]; if (om > pm || (om == pm && oi > pidx)) { pm = om; pidx = oi; } }
          pm = uniform(pm); pidx = __builtin_amdgcn_readfirstlane(pidx);
          if (!(pm > 0.0) || !(pred0 + sqrt(fmax(fmin(unpp, unnn) + unx - 2.0 * pm, 0.0)) <= accT)) break;     // (uniform)
          const int p_ = pidx >> 8, q_ = pidx & 255;       // p_ > q_
          const double bpp = A[p_ * lda + p_], bqq = A[q_ * lda + q_], bpq = A[p_ * lda + q_];
          const double tau = (bqq - bpp) / (2.0 * bpq);
          const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double cc = 1.0 / sqrt(1.0 + tt * tt), ss = tt * cc;
          __syncthreads();
          if (tid < n) {
            // B <- J'BJ, J = [c s; -s c] in the (p, q) plane: entries (k, p), (k, q) of the lower triangle, and the 2 x 2 block
            if (tid != p_ && tid != q_) {
              double* xp = tid > p_ ? A + tid * lda + p_ : A + p_ * lda + tid;
              double* xq = tid > q_ ? A + tid * lda + q_ : A + q_ * lda + tid;
              const double x = *xp, y = *xq; *xp = cc * x - ss * y; *xq = ss * x + cc * y;
            } else if (tid == p_) { A[p_ * lda + p_] = bpp - tt * bpq; A[q_ * lda + q_] = bqq + tt * bpq; A[p_ * lda + q_] = 0.0; }
          } else if (tid >= 128 && tid < 128 + n) {
            double* vp = V + (tid - 128) + (size_t)p_ * ldv; double* vq = V + (tid - 128) + (size_t)q_ * ldv;
            const double x = *vp, y = *vq; *vp = cc * x - ss * y; *vq = ss * x + cc * y;
          }
          __syncthreads();
          ++pivots;
          if (do_gram) {      // (a plane rotation leaves R = I - V'V rotated, its norm unchanged)
            gram();
            if (tid < npg) dvec[tid] = tid < n ? A[tid * lda + tid] : 0.0;
            r2 = uniform(gram_diag());
          } else {
            if (tid < npg) dvec[tid] = tid < n ? A[tid * lda + tid] : 0.0;
            __syncthreads();
          }
          analyse();
          decide();
          if (pivots >= a.refine_pivots || refined || side_force != 0 || !(r2 <= 1e-4) || !(k2 <= a.refine_k2cap) || !(unx > 0.0)) break;
        }
      }
      if (side_force == 0 && !refined && r2 <= 1e-4 && tid == 0 && a.stats) {      // diagnostic: which term of the prediction rejected the block
        const double ua = sqrt(unx), ub = sqrt(fmin(unpp, unnn));
        atomicAdd(&a.stats[pred0 >= ua && pred0 >= ub ? 9 : (ua >= ub ? 10 : 11)], 1);
      }
      if (side_force != 0) {
        step(true);
        {
          const int col = tid >> 3, part = tid & 7;
          double nr = 0.0;
          if (col < n)
            for (int rx = part; rx < n; rx += 8) { const double v = V[rx + (size_t)col * ldv]; nr += v * v; }
          nr = row_sum8(nr);
          if (part == 0 && col < npg) {
            double lamn = 0.0;
            if (col < n) { const double d = dvec[col]; lamn = (d * (1.0 + rdg[col] + cs1[col]) - cs2[col]) / nr; }
            A[col * lda + col] = lamn;
          }
        }
        __syncthreads();
        RST(8)
        outcome = 1;
        refined = true;
      } else if (check) {
        // checked form: the step, B rebuilt with the new basis, accepted when the MEASURED off(B) and defect of V are inside the
        // accepted level (the eigenvalues are then the diagonal of B itself)
        step(false);
        rebuild_B();
        gram();
        r2 = uniform(gram_diag());
        measured = true;
        double o2 = 0.0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i > j && i < n) { const double b = A[i * lda + j]; o2 += 2.0 * b * b; }
            }
        off2 = uniform(block_sum(o2, red));
        if (off2 <= accT * accT && r2 <= a.refine_acc * a.refine_acc * tolv * tolv) { outcome = 3; refined = true; }
      }
      if (!refined) {
        outcome = 2;
        if (!(r2 <= 0.01)) {
          // (cannot happen with the guards above; Newton-Schulz only converges from |R| < 1) restart the block cold: V = I, B = sym(nu)
          for (int j = wv; j < npg; j += NW)
            for (int i = lane; i < npg; i += 64) {
              double v = 0.0;
              if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
              A[i * lda + j] = v;
              V[i + (size_t)j * ldv] = (i == j) ? 1.0 : 0.0;
            }
          __syncthreads();
          r2 = 0.0; measured = true;
        } else if (r2 > a.refine_acc * a.refine_acc * tolv * tolv) {
          // on to the sweeps, which keep V only as orthogonal as they find it: a defect above the error level accepted for refinement
          // steps is first put right by Newton-Schulz steps V <- V (I + R / 2) (R -> 3/8 R^2 each), then B is rebuilt.  (A smaller
          // one goes to the sweeps as it is: they diagonalise B exactly, the projection carries that defect once, and the next
          // refinement step's R term removes it.)
          for (int pass = 0; pass < 3 && r2 > 1e-24; ++pass) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
              if (tti[m] >= 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
                  if (i > j) { const double e = (i < n) ? -0.5 * g[m][r] : 0.0; A[i * lda + j] = e; A[j * lda + i] = e; }
                  else if (i == j) A[i * lda + i] = 0.5 * rdg[i];
                }
            __syncthreads();
            v_update();
            gram();
            r2 = uniform(gram_diag());
          }
          rebuild_B();
        }
      }
      if (tid == 0) {
        if (a.stats) { atomicAdd(&a.stats[outcome == 3 || loose ? 8 : 4 + outcome], 1); if (pivots) atomicAdd(&a.stats[outcome == 1 ? 12 : 13], pivots); }
        if (a.rstate) {
          // back-off: a block whose prediction misses by more than 10 x TWICE IN A ROW skips the attempt for 2, 4, 8, 16 iterations (the
          // Gram product and the analysis are wasted work while the iterate still moves fast); near misses - the isolated failures
          // late in a solve - try again at once; a success resets the level
          int word;
          if (outcome == 2 && far) { const int lv = min(level + 1, 5); word = (lv << 8) | (lv >= 2 ? (1 << (lv - 1)) : 0); }
          else if (outcome == 2) word = (credit << 16) | (level << 8);
          else word = loose ? 0 : (min(credit + 1, 255) << 16);
          // the defect estimate: measured where a Gram product was taken (a step with the R term removes it to second order), otherwise
          // carried; every step adds its |K|_F^4 / 4
          double rnew = measured ? sqrt(r2) : rdef;
          if (outcome == 1) rnew = (do_gram ? r2 + 2.0 * sqrt(r2 * k2) : rdef * (1.0 + 2.2 * sqrt(k2))) + 0.25 * k2 * k2;     // ((I + X)'R(I + X) of an uncorrected R)
          a.rstate[4 * k] = word | ((do_gram ? a.gram_credit : gcred - 1) << 24);
          *reinterpret_cast<double*>(a.rstate + 4 * k + 2) = rnew;
        }
      }
    } else if (warm && rmode != 0 && wait > 0 && tid == 0) {
      a.rstate[4 * k] = (gcred << 24) | (credit << 16) | (level << 8) | (wait - 1);
      if (a.stats) atomicAdd(&a.stats[7], 1);
    }
  }
  // ---- refinement stage of the packed variant (blocks 129 .. 160).  The same step as above in the form these sizes allow: the basis
  // lives in HBM / L2 and the packed triangle has no room for a second triangle, so the rotation is the antisymmetric K alone
  // (K_ij = B_ij / (d_j - d_i) in the packed triangle, K_ji = -K_ij implied), orthogonality is restored by a separate pass
  // V <- V (I + R / 2) on the visits that measure the Gram matrix and find a defect worth removing, and every product streams its
  // operands once through the 32 x 161 LDS panel: X = I + K + K^2 / 2 (both factors gathered from the packed LDS) goes to the n^2
  // scratch, V X to the second scratch and back.  Same acceptance rule, same state word, the packed sweeps as the exact fall-back.
  if constexpr (PK) {
    int wait = 0, level = 0, credit = 0, gcred = 0;
    double rdef = 0.0;
    if (a.rstate) {
      const int rs = a.rstate[4 * k]; wait = rs & 255; level = (rs >> 8) & 255; credit = (rs >> 16) & 255; gcred = (rs >> 24) & 15;
      rdef = *reinterpret_cast<const double*>(a.rstate + 4 * k + 2);
    }
    __syncthreads();      // (every wave has the state word before thread 0 rewrites it: see the ping-pong stage)
    const int rmode = a.refine;
    if (warm && rmode != 0 && a.Ug && wait == 0) {
      constexpr int NW = NT / 64, kPanelLd = 161, kTilesPerWave = 4;
      const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lc = lane >> 4;
      const int nt = (n + 15) >> 4, ntl = nt * (nt + 1) / 2, ks = (n + 3) >> 2;
      double* const Pn = red + 16 + (npg >> 1) + 2;
      double* const Tk = a.Tg + a.coff[k];
      double* const Uk = a.Ug + a.coff[k];
      double* const dvec = desc;                 // B_ii
      double* const cs1 = desc + npg;            // column sums of K^2
      double* const cs2 = desc + 2 * npg;        // column sums of K^2 d
      double* const nrm = desc + 3 * npg;        // squared column norms of the new basis
      int tti[kTilesPerWave], ttj[kTilesPerWave];
#pragma unroll
      for (int m = 0; m < kTilesPerWave; ++m) {
        const int t = wv + m * NW;
        int ti = -1, tj = -1;
        if (t < ntl) { ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; tj = t - ti * (ti + 1) / 2; }
        tti[m] = ti; ttj[m] = tj;
      }
      // V <- V M with M (n x n, row-major, identity included) in the first scratch: 16-column chunks of V against 16-row chunks of M,
      // all n^2 output tiles in two halves of the tile columns (four accumulators per wave), the result through the second scratch;
      // the copy back also takes the squared column norms
      auto apply_M = [&]() {
        const int hsplit = (nt + 1) >> 1;
        for (int hcol = 0; hcol < 2; ++hcol) {
          const int tj0 = hcol ? hsplit : 0, ntile = nt * (hcol ? nt - hsplit : hsplit);
          d4_t acc[kTilesPerWave];
#pragma unroll
          for (int m = 0; m < kTilesPerWave; ++m) acc[m] = d4_t{0.0, 0.0, 0.0, 0.0};
          double* const Vc = Pn;
          double* const Mc = Pn + 16 * kPanelLd;
          for (int l0 = 0; l0 < n; l0 += 16) {
            {
              const int ll = tid >> 6, l = l0 + ll;
              for (int i = tid & 63; i < 16 * nt; i += 64) {
                Vc[ll * kPanelLd + i] = (l < n && i < n) ? V[i + (size_t)l * ldv] : 0.0;
                Mc[ll * kPanelLd + i] = (l < n && i < n) ? Tk[(size_t)l * n + i] : 0.0;
              }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < kTilesPerWave; ++m) {
              const int t = wv + m * NW;
              if (t < ntile) {
                const int tjl = t / nt, ti = t - tjl * nt, tj = tj0 + tjl;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                  const int kx = 4 * k4 + lc;
                  acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vc[kx * kPanelLd + 16 * ti + lr], Mc[kx * kPanelLd + 16 * tj + lr], acc[m], 0, 0, 0);
                }
              }
            }
            __syncthreads();
          }
#pragma unroll
          for (int m = 0; m < kTilesPerWave; ++m) {
            const int t = wv + m * NW;
            if (t < ntile) {
              const int tjl = t / nt, ti = t - tjl * nt, tj = tj0 + tjl;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lc + 4 * r, col = 16 * tj + lr;
                if (row < n && col < n) Uk[row + (size_t)col * n] = acc[m][r];
              }
            }
          }
        }
        __syncthreads();      // (the block's writes of the new basis are visible to the block)
        for (int j = wv; j < n; j += NW) {
          double nr = 0.0;
          for (int i = lane; i < n; i += 64) { const double v = Uk[i + (size_t)j * n]; V[i + (size_t)j * ldv] = v; nr += v * v; }
          nr = wave_sum(nr);
          if (lane == 0) nrm[j] = nr;
        }
        __syncthreads();
      };
      const double kcap = a.refine_kcap;
      const double T = tolv * sqrt(fro2), accT = a.refine_acc * T;
      int outcome = 2;      // 0: converged as it arrived, 1: one step, 2: on to the sweeps
      double r2 = rdef * rdef, k2 = 0.0;
      bool measured = false, far = true;
      const bool do_gram = !a.rstate || gcred == 0 || !(rdef <= 0.03 * a.refine_acc * tolv);
      RST(0)
      if (do_gram) {
        // lower tiles of G = V'V over 16-row chunks of V
        d4_t g[kTilesPerWave];
#pragma unroll
        for (int m = 0; m < kTilesPerWave; ++m) g[m] = d4_t{0.0, 0.0, 0.0, 0.0};
        for (int l0 = 0; l0 < n; l0 += 16) {
          {
            const int lv = tid & 15, l2 = l0 + lv;
            for (int j = tid >> 4; j < 16 * nt; j += NT >> 4) Pn[lv * kPanelLd + j] = (l2 < n && j < n) ? V[l2 + (size_t)j * ldv] : 0.0;
          }
          __syncthreads();
#pragma unroll
          for (int m = 0; m < kTilesPerWave; ++m)
            if (tti[m] >= 0) {
#pragma unroll
              for (int k4 = 0; k4 < 4; ++k4) {
                const int kx = 4 * k4 + lc;
                g[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pn[kx * kPanelLd + 16 * tti[m] + lr], Pn[kx * kPanelLd + 16 * ttj[m] + lr], g[m], 0, 0, 0);
              }
            }
          __syncthreads();
        }
        double rr2 = 0.0;
#pragma unroll
        for (int m = 0; m < kTilesPerWave; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * tti[m] + lc + 4 * r, j = 16 * ttj[m] + lr;
              if (i < n && j < n) {
                if (i == j) { const double v = 1.0 - g[m][r]; rr2 += v * v; }
                else if (i > j) rr2 += 2.0 * g[m][r] * g[m][r];
              }
            }
        r2 = uniform(block_sum(rr2, red));
        measured = true;
        const double rep = 0.01 * a.refine_acc * tolv;
        if (r2 > rep * rep && r2 <= 0.01) {
          // a defect worth removing: V <- V (I + R / 2) = V (3 I - G) / 2 (Newton-Schulz), then B = V'AV again from the matrix in HBM
#pragma unroll
          for (int m = 0; m < kTilesPerWave; ++m)
            if (tti[m] >= 0)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = 16 * tti[m] + lc + 4 * r, col = 16 * ttj[m] + lr;
                if (row < n && col < n) {
                  const double v = (row == col ? 1.5 : 0.0) - 0.5 * g[m][r];
                  Tk[(size_t)row * n + col] = v;
                  if (tti[m] != ttj[m]) Tk[(size_t)col * n + row] = v;
                }
              }
          __syncthreads();
          apply_M();
          for (int j = tid >> 6; j < npg; j += NT >> 6)
            for (int i = (tid & 63) + j; i < npg; i += 64) {
              double v = 0.0;
              if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
              A[ixl(i, j)] = v;
            }
          __syncthreads();
          pk_congruence();
          r2 = r2 * r2;         // (the defect is squared by the step, up to a factor 3/8)
        }
      }
      RST(1)
      double off2 = 0.0, unpp = 0.0, unnn = 0.0, unx = 0.0, kd2 = 0.0, cpos = 0.0, cneg = 0.0, pred0 = 0.0;
      // analysis of all pairs (nothing is written: a rejected block reaches the sweeps untouched) and the decision, as in the
      // ping-pong form
      auto pk_analyse = [&]() {
        if (tid < npg) dvec[tid] = tid < n ? A[ixl(tid, tid)] : 0.0;
        __syncthreads();
        double o2 = 0.0, q2 = 0.0, upp = 0.0, unn = 0.0, ux = 0.0, qd2 = 0.0;
        for (int j = tid >> 6; j < n; j += NW)
          for (int i = (tid & 63) + j + 1; i < n; i += 64) {
            const double b = A[ixl(i, j)], di = dvec[i], dj = dvec[j], gap = dj - di;
            o2 += 2.0 * b * b;
            if (fabs(b) <= kcap * fabs(gap) && gap != 0.0) {
              const double e = b * rcp_nr2(gap);
              q2 += 2.0 * e * e;
              qd2 += e * e * (dj * dj + di * di);
            } else {
              const double dd = di * dj;
              if (b * b < dd) { if (di > 0.0) upp += 2.0 * b * b; else unn += 2.0 * b * b; }   // same sign, inertia kept
              else ux += 2.0 * b * b;
            }
          }
        off2 = uniform(block_sum(o2, red)); k2 = uniform(block_sum(q2, red)); unpp = uniform(block_sum(upp, red));
        unnn = uniform(block_sum(unn, red)); unx = uniform(block_sum(ux, red)); kd2 = uniform(block_sum(qd2, red));
        cpos = uniform(block_sum((tid < n && dvec[tid] > 0.0) ? 1.0 : 0.0, red));
        cneg = uniform(block_sum((tid < n && dvec[tid] < 0.0) ? 1.0 : 0.0, red));
        __syncthreads();
        if (off2 <= T * T && r2 <= tolv * tolv) { outcome = 0; refined = true; }
        else if (r2 <= 1e-4 && k2 <= a.refine_k2cap) {
          pred0 = 1.5 * sqrt(off2) * sqrt(k2) + k2 * sqrt(kd2) * (1.0 / 3.0) + sqrt(r2) * sqrt(fro2);   // (an uncorrected defect shows in the projection to first order)
          const double pred_pos = pred0 + sqrt(unpp + unx), pred_neg = pred0 + sqrt(unnn + unx);
          const bool prefer_pos = cpos <= cneg;
          if ((prefer_pos ? pred_pos : pred_neg) <= accT) side_force = prefer_pos ? 1 : -1;
          else if ((prefer_pos ? pred_neg : pred_pos) <= accT) side_force = prefer_pos ? -1 : 1;
          far = fmin(pred_pos, pred_neg) > 10.0 * accT;
        }
      };
      pk_analyse();
      RST(2)
      // the one pair across zero that first order cannot resolve (see the ping-pong form): rotated exactly - its two rows of the packed
      // B, its two columns of V in HBM - and the block analysed again, at most refine_pivots times
      int pivots = 0;
      if (__builtin_expect(!refined && side_force == 0 && r2 <= 1e-4 && k2 <= a.refine_k2cap && unx > 0.0 && a.refine_pivots > 0, 0)) {
        for (;;) {
          double pm = 0.0;
          int pidx = 0;
          for (int j = tid >> 6; j < n; j += NW)
            for (int i = (tid & 63) + j + 1; i < n; i += 64) {
              const double b = A[ixl(i, j)], di = dvec[i], dj = dvec[j], gap = dj - di;
              if (!(fabs(b) <= kcap * fabs(gap) && gap != 0.0) && !(b * b < di * dj) && b * b > pm) { pm = b * b; pidx = (i << 8) | j; }
            }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(pm, o, 64);
            const int oi = __shfl_xor(pidx, o, 64);
            if (om > pm || (om == pm && oi > pidx)) { pm = om; pidx = oi; }
          }
          if (lane == 0) { cs1[wv] = pm; cs2[wv] = (double)pidx; }
          __syncthreads();
          pm = 0.0; pidx = 0;
          for (int w_ = 0; w_ < NW; ++w_) { const double om = cs1[w_]; const int oi = (int)cs2[w_]; if (om > pm || (om == pm && oi > pidx)) { pm = om; pidx = oi; } }
          pm = uniform(pm); pidx = __builtin_amdgcn_readfirstlane(pidx);
          if (!(pm > 0.0) || !(pred0 + sqrt(fmax(fmin(unpp, unnn) + unx - 2.0 * pm, 0.0)) <= accT)) break;     // (uniform)
          const int p_ = pidx >> 8, q_ = pidx & 255;       // p_ > q_
          const double bpp = A[ixl(p_, p_)], bqq = A[ixl(q_, q_)], bpq = A[ixl(p_, q_)];
          const double tau = (bqq - bpp) / (2.0 * bpq);
          const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double cc = 1.0 / sqrt(1.0 + tt * tt), ss = tt * cc;
          __syncthreads();
          if (tid < n) {
            if (tid != p_ && tid != q_) {
              double* xp = A + ixs(tid, p_);
              double* xq = A + ixs(tid, q_);
              const double x = *xp, y = *xq; *xp = cc * x - ss * y; *xq = ss * x + cc * y;
            } else if (tid == p_) { A[ixl(p_, p_)] = bpp - tt * bpq; A[ixl(q_, q_)] = bqq + tt * bpq; A[ixl(p_, q_)] = 0.0; }
          } else if (tid >= 256 && tid < 256 + n) {
            double* vp = V + (tid - 256) + (size_t)p_ * ldv; double* vq = V + (tid - 256) + (size_t)q_ * ldv;
            const double x = *vp, y = *vq; *vp = cc * x - ss * y; *vq = ss * x + cc * y;
          }
          __syncthreads();
          ++pivots;
          pk_analyse();
          if (pivots >= a.refine_pivots || refined || side_force != 0 || !(r2 <= 1e-4) || !(k2 <= a.refine_k2cap) || !(unx > 0.0)) break;
        }
      }
      if (side_force != 0) {
        // K into the packed triangle (0 for the pairs first order cannot resolve)
        for (int j = tid >> 6; j < n; j += NW)
          for (int i = (tid & 63) + j + 1; i < n; i += 64) {
            const double b = A[ixl(i, j)], gap = dvec[j] - dvec[i];
            A[ixl(i, j)] = (fabs(b) <= kcap * fabs(gap) && gap != 0.0) ? b * rcp_nr2(gap) : 0.0;
          }
        __syncthreads();
        auto Ks = [&](int i, int l) { return (i < n && l < n && i != l) ? (i > l ? A[ixl(i, l)] : -A[ixl(l, i)]) : 0.0; };
        {
          // column sums for the second-order eigenvalues: 4 lanes per column
          const int col = tid >> 2, part = tid & 3;
          double s1 = 0.0, s2 = 0.0;
          if (col < n)
            for (int kx = part; kx < n; kx += 4)
              if (kx != col) { const double e = A[ixs(kx, col)]; s1 += e * e; s2 += e * e * dvec[kx]; }
          s1 += dpp_row<0xB1>(s1); s1 += dpp_row<0x4E>(s1);
          s2 += dpp_row<0xB1>(s2); s2 += dpp_row<0x4E>(s2);
          if (part == 0 && col < n) { cs1[col] = s1; cs2[col] = s2; }
        }
        RST(3)
        // K^2 (symmetric), lower tiles: both factors gathered from the packed triangle with their signs
        d4_t sq[kTilesPerWave];
#pragma unroll
        for (int m = 0; m < kTilesPerWave; ++m) {
          d4_t c = {0.0, 0.0, 0.0, 0.0};
          if (tti[m] >= 0)
            for (int kk = 0; kk < ks; ++kk) {
              const int kx = 4 * kk + lc;
              c = __builtin_amdgcn_mfma_f64_16x16x4f64(Ks(16 * tti[m] + lr, kx), Ks(kx, 16 * ttj[m] + lr), c, 0, 0, 0);
            }
          sq[m] = c;
        }
        // X = I + K + K^2 / 2 into the scratch (row-major, both triangles)
#pragma unroll
        for (int m = 0; m < kTilesPerWave; ++m)
          if (tti[m] >= 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * tti[m] + lc + 4 * r, col = 16 * ttj[m] + lr;
              if (row < n && col < n) {
                const double kv = Ks(row, col), h = 0.5 * sq[m][r];
                Tk[(size_t)row * n + col] = (row == col ? 1.0 : 0.0) + kv + h;
                if (tti[m] != ttj[m]) Tk[(size_t)col * n + row] = h - kv;
              }
            }
        __syncthreads();
        RST(4)
        apply_M();
        RST(5)
        if (tid < n) A[ixl(tid, tid)] = (dvec[tid] * (1.0 + cs1[tid]) - cs2[tid]) / nrm[tid];
        __syncthreads();
        outcome = 1;
        refined = true;
      }
      if (tid == 0) {
        if (a.stats) { atomicAdd(&a.stats[4 + outcome], 1); if (pivots) atomicAdd(&a.stats[outcome == 1 ? 12 : 13], pivots); }
        if (a.rstate) {
          int word;
          if (outcome == 2 && far) { const int lv = min(level + 1, 5); word = (lv << 8) | (lv >= 2 ? (1 << (lv - 1)) : 0); }
          else if (outcome == 2) word = (credit << 16) | (level << 8);
          else word = min(credit + 1, 255) << 16;
          double rnew = measured ? sqrt(r2) : rdef;
          if (outcome == 1) rnew = rnew * (1.0 + 2.2 * sqrt(k2)) + 0.25 * k2 * k2;     // (no R term in this form: the defect is carried, every step adds |K|_F^4 / 4)
          a.rstate[4 * k] = word | ((do_gram ? a.gram_credit : gcred - 1) << 24);
          *reinterpret_cast<double*>(a.rstate + 4 * k + 2) = rnew;
        }
      }
    } else if (warm && rmode != 0 && a.Ug && wait > 0 && tid == 0) {
      a.rstate[4 * k] = (gcred << 24) | (credit << 16) | (level << 8) | (wait - 1);
      if (a.stats) atomicAdd(&a.stats[7], 1);
    }
  }
  // (the sweeps' static work assignment is computed after the refinement stage: nothing of it is live across the stage)
  int blk[MAXB];
  {
    const int R = (half + 1) >> 1, Wd = half + 1;
#pragma unroll
    for (int u = 0; u < MAXB; ++u) {
      int b = tid + u * NT;
      int v = -1;
      if (b < R * Wd) {
        int row = b / Wd, m = b - row * Wd;
        int iahi = half - 1 - row;
        if (m <= row) v = (row << 8) | m;
        else if (iahi != row) v = (iahi << 8) | (m - row - 1);
      }
      blk[u] = v;
    }
  }
  const int nch = (nv + 31) >> 5;
  const int vr = tid & 31, vg = tid >> 5;
  int vunit[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    int un = vg + u * VG;
    int v = -1;
    if (plane < 0 && un < half * nch) {
      int ia = un / nch, ch = un - ia * nch;
      int row = ch * 32 + vr;
      if (row < nv) v = (ia << 8) | row;
    }
    vunit[u] = v;
  }
  if (refined) {
    // eigenvalues on the diagonal of A, eigenvectors in V: nothing left to do before the reconstruction
  } else
  if constexpr (PP) {
    // ---- ping-pong Jacobi: odd-even ordering on matrix POSITIONS (a rotation is followed by a swap of the two
    // positions, so pairs are always neighbours and no index tables exist).  Round type 0 pairs (2K, 2K+1), type 1
    // pairs (2K-1, 2K) with virtual positions -1 and np at the ends; np rounds make every pair meet once and reverse
    // the order.  The lower triangle lives in LDS twice, shifted by one cell so that the virtual positions are a
    // border and every 2x2 block of either round type is read and written unconditionally.  A round reads A_in and
    // writes A_out (the eigenvector copy is not needed during the sweeps, its storage is the second buffer), so the
    // LAST wave can compute round t+1's rotations from A_in - analytically for the two diagonal entries, from one 2x2
    // coupling block for the off-diagonal one - WHILE waves 0..14 apply round t.  One workgroup barrier per round.
    // Eigenvectors live in the registers of waves 0..14 (lane = column pair, neighbour columns over DPP).
    // Rotations are applied in scaled form (square-root-free "fast" rotations): stored values are A_ij / (d_i d_j)
    // and V_ij / d_j; with t = s/c the rotation + swap of positions (p, q) is
    //     x_p' = x_q + (t d_p/d_q) x_p,   x_q' = x_p - (t d_q/d_p) x_q,   d_p' = c d_q,   d_q' = c d_p
    // (2 FMAs per element pair instead of 4 operations; |theta| <= pi/4 keeps c >= 0.7, and one sweep shrinks a scale
    // by at most 0.7^np, far inside the fp64 range).  The scales live in the parameter wave; the true values are
    // restored at the end of every sweep, where convergence is measured.
    constexpr int UW = NT / 64 - 1;            // updater waves
    constexpr int VRW = RPW;                   // eigenvector rows per updater wave (5: n <= 74, 6: n <= 90, 7: n <= 96), dealt round robin
    constexpr int MAXI = 2;                    // items per updater thread: (m+1)(m+2)/2 <= 1225 <= 2 x 960
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool updater = wv < UW;
    const int m = half;
    double* const B0 = V;                      // shifted buffers: B0 holds the matrix at even rounds
    double* const B1 = A;
    // Sweep layout of both buffers (stride kPpLdp, even): even rows first, then the odd rows, so that the two rows of a
    // 2x2 block of EITHER round type are each one 16-byte aligned pair, read with one conflict-free ds_read_b128 per
    // row (consecutive lanes = consecutive column pairs).  B0 is read by type-0 rounds (blocks start at odd shifted
    // columns): its columns are stored one cell to the right; B1 is read by type-1 rounds (even columns): no shift.
    const int H = m + 1;                       // rows per parity class (shifted indices 0 .. np + 1)
    auto pidx = [&](int r, int c, int sh) { return ((r >> 1) + (r & 1) * H) * kPpLdp + c + sh; };
    double* tt = desc;                         // [2][m + 2] x {t d_p/d_q, t d_q/d_p}; fixed entries for idle lanes / virtual pairs
    const int ttld = 2 * (m + 2);
    double* dsc = desc + 2 * ttld;             // [np] scale of every position at the end of a sweep
    if (tid == 0) {
      *reinterpret_cast<double2*>(tt + 2 * m) = make_double2(0.0, 0.0);
      *reinterpret_cast<double2*>(tt + 2 * m + 2) = make_double2(0.0, 0.0);
      *reinterpret_cast<double2*>(tt + ttld + 2 * m + 2) = make_double2(0.0, 0.0);
    }
    // eigenvector rows -> registers (before the second matrix buffer overwrites the LDS copy)
    double vr[VRW][2];
#pragma unroll
    for (int j = 0; j < VRW; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        int r = j * UW + wv, col = 2 * lane + cc;   // rows dealt round robin over the updater waves
        vr[j][cc] = (updater && r < np && col < np) ? V[r + (size_t)col * ldv] : 0.0;
      }
    // static items of both round types: (ka << 8) | kb and the offset of the block's first element
    int ik[2][MAXI], eb[2][MAXI];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int mt = m + sg, Wd = mt + 1, R = (mt + 1) >> 1;
#pragma unroll
      for (int u = 0; u < MAXI; ++u) {
        int b = tid + u * (UW * 64);
        int v = -1;
        if (updater && b < R * Wd) {
          int row = b / Wd, c = b - row * Wd;
          int hi = mt - 1 - row;
          if (c <= row) v = (row << 8) | c;
          else if (hi != row) v = (hi << 8) | (c - row - 1);
        }
        ik[sg][u] = v;
        eb[sg][u] = pidx(2 * (v >> 8) + 1 - sg, 2 * (v & 255) + 1 - sg, 1 - sg);   // in the buffer the round type READS
      }
    }
    __syncthreads();
    // shifted copy (lower triangle) with a border of zeros; type-1 rounds read B1, whose border is never written
    for (int i = tid >> 6; i < np; i += NT >> 6)
      for (int j = tid & 63; j <= i; j += 64) B0[pidx(i + 1, j + 1, 1)] = A[i * lda + j];
    __syncthreads();
    for (int i = tid; i < np + 2; i += NT) { B1[pidx(i, 0, 0)] = 0.0; B1[pidx(np + 1, i, 0)] = 0.0; }
    // parameter wave state for pair `lane` of the current round: true (pp, qq, pq), rotation (pc, ps), scales of the
    // pair's first / second position before the round
    // (the parameter wave holds no eigenvector rows: its state lives in those registers, the kernel runs at the VGPR limit)
    double &st_pp = vr[0][0], &st_qq = vr[0][1], &st_pq = vr[1][0], &pc = vr[1][1], &ps = vr[2][0], &st_dp = vr[2][1], &st_dq = vr[3][0];
    // off(A)^2 before the first sweep; afterwards it comes out of the pass that restores the true values
    double off2 = 0.0;
    for (int i = (tid >> 6) + 2; i <= np; i += NT >> 6)
      for (int j = (tid & 63) + 1; j < i; j += 64) { double v = B0[pidx(i, j, 1)]; off2 += v * v; }
    off2 = 2.0 * block_sum(off2, red);
    for (;;) {
      if (off2 <= thresh2 || sweeps >= a.max_sweeps) break;
      if (!updater) {   // rotations of round 0 (type 0) straight from the matrix, all scales 1
        st_dp = 1.0; st_dq = 1.0;
        if (lane < m) {
          st_pp = B0[pidx(2 * lane + 1, 2 * lane + 1, 1)]; st_qq = B0[pidx(2 * lane + 2, 2 * lane + 2, 1)]; st_pq = B0[pidx(2 * lane + 2, 2 * lane + 1, 1)];
          double t;
          jacobi_cst(st_pp, st_qq, st_pq, rot_thr, pc, ps, t);
          nrot += (ps != 0.0);
          *reinterpret_cast<double2*>(tt + 2 * lane) = make_double2(t, t);
        } else { st_pp = st_qq = st_pq = 0.0; pc = 1.0; ps = 0.0; }
      }
      __syncthreads();
      auto round = [&](auto sgc, int t) {
        constexpr int sg = decltype(sgc)::value;   // round type as a constant: the static item tables stay in registers
        const double* Ain = sg ? B1 : B0;
        double* Aout = sg ? B0 : B1;
#ifdef NNSDP_STAMPS
        long long tprev = clock64();
#endif
        const double* ttr = tt + sg * ttld;
#ifdef PP_NO_UPDATE
        if (false) {
#else
        if (updater) {
#endif
          // A_out <- M_r' A_in M_c on the lower block triangle, scaled rotation + swap.  Source order: LDS loads of
          // the first block, eigenvector columns (registers; covers the LDS latency), block math and stores
          const int rstep = sg ? H * kPpLdp : (1 - H) * kPpLdp;   // from the block's first row to its second (other parity class)
          auto blk_math = [&](int it, int ebo, double2 pr, double2 pq, double b00, double b01r, double b10, double b11) {
            const bool dg = (it >> 8) == (it & 255);
            // the written buffer has the other column shift: same cells, one position to the left (type 0) / right (type 1)
            double* dst = Aout + ebo + (sg ? 1 : -1);
            const double b01 = dg ? b10 : b01r;
            const double t00 = b01 + pq.x * b00, t01 = b00 - pq.y * b01;
            const double t10 = b11 + pq.x * b10, t11 = b10 - pq.y * b11;
            dst[0] = t10 + pr.x * t00;
            dst[rstep] = t00 - pr.y * t10;
            dst[rstep + 1] = t01 - pr.y * t11;
            if (!dg) dst[1] = t11 + pr.x * t01;
          };
          const int it0 = ik[sg][0];
          // (deliberately not initialised: they are only read under the same it0 >= 0 below, and eight v_mov_b64 of zeros per
          // wave and round are 10 % of the round's VALU issue slots)
          double2 pr0, pq0;
          double a00, a01, a10, a11;
          if (it0 >= 0) {
            const double* src = Ain + eb[sg][0];
            pr0 = *reinterpret_cast<const double2*>(ttr + 2 * (it0 >> 8));
            pq0 = *reinterpret_cast<const double2*>(ttr + 2 * (it0 & 255));
            const double2 r0v = *reinterpret_cast<const double2*>(src), r1v = *reinterpret_cast<const double2*>(src + rstep);
            a00 = r0v.x; a01 = r0v.y; a10 = r1v.x; a11 = r1v.y;
          }
          // eigenvector columns (registers)
          if (sg == 0) {
            const double2 p = *reinterpret_cast<const double2*>(ttr + 2 * min(lane, m));
#pragma unroll
            for (int j = 0; j < VRW; ++j) {
              double x0 = vr[j][0], x1 = vr[j][1];
              vr[j][0] = x1 + p.x * x0;
              vr[j][1] = x0 - p.y * x1;
            }
          } else if (lane < m) {
            // position 2J is the second of pair J = (2J-1, 2J), position 2J+1 the first of pair J+1; idle lanes are
            // masked off, so the last real column reads 0 from its right-hand neighbour (DPP bound_ctrl)
            const double2 p0 = *reinterpret_cast<const double2*>(ttr + 2 * lane);
            const double2 p1 = *reinterpret_cast<const double2*>(ttr + 2 * lane + 2);
#pragma unroll
            for (int j = 0; j < VRW; ++j) {
              double x0 = vr[j][0], x1 = vr[j][1];
              double xp = lane_prev(x1), xq = lane_next(x0);
              vr[j][0] = xp - p0.y * x0;
              vr[j][1] = xq + p1.x * x1;
            }
          }
          if (it0 >= 0) blk_math(it0, eb[sg][0], pr0, pq0, a00, a01, a10, a11);
#pragma unroll
          for (int u = 1; u < MAXI; ++u) {
            const int it = ik[sg][u];
            if (it >= 0) {
              const double* src = Ain + eb[sg][u];
              const double2 pr = *reinterpret_cast<const double2*>(ttr + 2 * (it >> 8));
              const double2 pq = *reinterpret_cast<const double2*>(ttr + 2 * (it & 255));
              const double2 r0v = *reinterpret_cast<const double2*>(src), r1v = *reinterpret_cast<const double2*>(src + rstep);
              blk_math(it, eb[sg][u], pr, pq, r0v.x, r0v.y, r1v.x, r1v.y);
            }
          }
#ifdef PP_NO_PARAM
        } else if (false) {
#else
        } else if (!updater) {
#endif
          // rotations of round t+1 from A_in and this round's rotations.  New pair = (second position of pair Ka,
          // first position of pair Kb = Ka + 1) of this round; its off-diagonal entry is element [0][1] of the updated
          // coupling block (rows of Kb, columns of Ka).
          __builtin_amdgcn_s_setprio(3);   // the rotation chain is the round's critical path: issue ahead of the updaters on this SIMD
          const int L = lane;
          const bool edge_cur = (sg == 1) && (L == 0 || L == m);     // virtual pairs keep their real position and scale
          const double d1 = edge_cur ? st_dp : pc * st_dq;          // scales after this round: first / second position
          const double d2 = edge_cur ? st_dq : pc * st_dp;
          if (t + 1 < np) {
            const int mtn = m + 1 - sg;                      // pairs of round t+1
            const bool act = L < mtn;
            const bool edge = (sg == 0) && (L == 0 || L == m);   // virtual pairs of a type-1 round
            double b00 = 0.0, b01 = 0.0, b10 = 0.0, b11 = 0.0;
            if (act && !edge) {
              const int r0 = sg == 0 ? 2 * L + 1 : 2 * L + 2, c0 = sg == 0 ? 2 * L - 1 : 2 * L, sh = 1 - sg;
              b00 = Ain[pidx(r0, c0, sh)]; b01 = Ain[pidx(r0, c0 + 1, sh)]; b10 = Ain[pidx(r0 + 1, c0, sh)]; b11 = Ain[pidx(r0 + 1, c0 + 1, sh)];
            }
            const double cc_ = pc * pc, ss_ = ps * ps, sc2 = 2.0 * pc * ps * st_pq;
            const double app1 = cc_ * st_pp - sc2 + ss_ * st_qq;    // lands on the pair's second position
            const double aqq1 = ss_ * st_pp + sc2 + cc_ * st_qq;    // lands on the pair's first position
            double npp, nqq, ndp, ndq, ccol, scol, dca, dcb, crow, srow, dra, drb;
            if (sg == 0) {
              npp = lane_prev(app1); nqq = aqq1; ndp = lane_prev(d2); ndq = d1;
              ccol = lane_prev(pc); scol = lane_prev(ps); dca = lane_prev(st_dp); dcb = lane_prev(st_dq);
              crow = pc; srow = ps; dra = st_dp; drb = st_dq;
            } else {
              npp = app1; nqq = lane_next(aqq1); ndp = d2; ndq = lane_next(d1);
              ccol = pc; scol = ps; dca = st_dp; dcb = st_dq;
              crow = lane_next(pc); srow = lane_next(ps); dra = lane_next(st_dp); drb = lane_next(st_dq);
            }
            const double e1 = ccol * dca, e2 = scol * dcb;
            const double u01 = e1 * b00 - e2 * b01, u11 = e1 * b10 - e2 * b11;
            double npq = (srow * dra) * u01 + (crow * drb) * u11;
            if (sg == 0 && L == 0) { npp = 0.0; ndp = 1.0; }
            if (sg == 0 && L == m) { nqq = 0.0; ndq = 1.0; }
            if (!act) { npp = 0.0; nqq = 0.0; ndp = 1.0; ndq = 1.0; }
            if (!act || edge) npq = 0.0;
            const double rq = rcp_nr2(ndq), rp = rcp_nr2(ndp);
            double c, sn, tg;
            jacobi_cst(npp, nqq, npq, rot_thr, c, sn, tg);
            nrot += (sn != 0.0);
            double2 pub = make_double2(tg * ndp * rq, tg * ndq * rp);
            if (sg == 0 && L == 0) { c = 0.0; sn = -1.0; pub = make_double2(0.0, -1.0); }
            if (sg == 0 && L == m) { c = 0.0; sn = 1.0; pub = make_double2(1.0, 0.0); }
            if (act) *reinterpret_cast<double2*>(tt + (sg ^ 1) * ttld + 2 * L) = pub;
            st_pp = npp; st_qq = nqq; st_pq = npq; pc = c; ps = sn; st_dp = ndp; st_dq = ndq;
          } else if (sg == 1) {
            // last round of the sweep: publish the scale of every real position (pair L = positions 2L-1, 2L)
            if (L >= 1 && L <= m) dsc[2 * L - 1] = d1;
            if (L <= m - 1) dsc[2 * L] = d2;
          }
          __builtin_amdgcn_s_setprio(0);
        }
        STAMP(sg * 2, tprev)
        __syncthreads();
        STAMP(sg * 2 + 1, tprev)
      };
      for (int t = 0; t < np; t += 2) {
        round(std::integral_constant<int, 0>{}, t);
        round(std::integral_constant<int, 1>{}, t + 1);
      }
      // back to true values: A_ij = d_i d_j a_ij, V_ij = d_j v_ij; the same pass measures off(A)^2 for the next decision
      off2 = 0.0;
      for (int i = (tid >> 6) + 1; i <= np; i += NT >> 6) {
        const double di = dsc[i - 1];
        for (int j = (tid & 63) + 1; j <= i; j += 64) {
          const double v = B0[pidx(i, j, 1)] * (di * dsc[j - 1]);
          B0[pidx(i, j, 1)] = v;
          if (j < i) off2 += v * v;
        }
      }
      if (lane < m) {
        const double da = dsc[2 * lane], db = dsc[2 * lane + 1];
#pragma unroll
        for (int j = 0; j < VRW; ++j) { vr[j][0] *= da; vr[j][1] *= db; }
      }
      off2 = 2.0 * block_sum(off2, red);     // (two barriers: the rescaled matrix is visible to everybody afterwards)
      ++sweeps;
    }
    // ---- eigenvalues back to the unshifted diagonal of A, eigenvectors (position order) to V
    pofs = ((n & 1) && (sweeps & 1)) ? 1 : 0;
    double dsave = 0.0;
    if (tid < np) dsave = B0[pidx(tid + 1, tid + 1, 1)];
    __syncthreads();
    if (tid < np) A[tid * lda + tid] = dsave;
#pragma unroll
    for (int j = 0; j < VRW; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        int r = j * UW + wv, col = 2 * lane + cc;
        if (updater && r < np && col < np) V[r + (size_t)col * ldv] = vr[j][cc];
      }
    for (int idx = tid; idx < (npg - np) * np; idx += NT) { int r = np + idx / np, col = idx - (r - np) * np; V[r + (size_t)col * ldv] = 0.0; }
    __syncthreads();
  } else
  if constexpr (SYS) {
    // ---- register-resident systolic Jacobi.  The matrix (both triangles) and the eigenvectors live in VGPRs for all
    // sweeps: lane J of every wave owns the column pair (2J, 2J+1); wave w owns the row pairs ("slots") I = w SPW + i
    // of the matrix as 2x2 blocks ar[i][row][col], and rows w RPW + j of the eigenvector matrix.  Odd-even ordering:
    // round type A rotates the position pairs (2J, 2J+1) - column and row rotations are register-local - and type B
    // the pairs (2J+1, 2J+2): column partners come from the neighbouring lane over DPP (no LDS), row partners from the
    // neighbouring slot (registers; across a wave boundary one row per wave through LDS).  Every rotation is followed
    // by a swap of the two positions, so after np rounds every pair has met exactly once and the order is reversed.
    // Per round: ONE workgroup barrier; LDS carries only the 2x2 rotation inputs and the boundary rows.
    constexpr int NWV = NT / 64;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = half;
    const int I0 = wv * SPW;
    double ar[SPW][2][2];
    double vr[RPW][2];
#pragma unroll
    for (int i = 0; i < SPW; ++i)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          int pr = 2 * (I0 + i) + rr, pc = 2 * lane + cc;
          ar[i][rr][cc] = (pr < np && pc < np) ? A[sym_at(pr, pc, lda)] : 0.0;
        }
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        int r = wv * RPW + j, col = 2 * lane + cc;
        double v = 0.0;
        if (r < np && col < np) {
          if (V_LDS) v = V[r + (size_t)col * ldv];
          else if (r < n && col < n) v = warm ? V[r + (size_t)col * ldv] : (r == col ? 1.0 : 0.0);
          else v = (r == col) ? 1.0 : 0.0;
        }
        vr[j][cc] = v;
      }
    __syncthreads();                       // A / V storage is dead from here until the write-back: exchange scratch
    double* prm = A;                       // [2][65][4] = {B[2I][2I], B[2I+1][2I+1], B[2I][2I+1], B[2I+1][2I+2]} of slot I
    double* halo = A + kSysPrm;            // [NWV][2][64][2]: first slot's row 0, last slot's row 1 of every wave
    auto publish = [&](int buf) {
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        const int I = I0 + i;
        double* dst = prm + (buf * 65 + I) * 4;
        if (lane == I) { dst[0] = ar[i][0][0]; dst[1] = ar[i][1][1]; dst[2] = ar[i][0][1]; }
        if (lane == I + 1) dst[3] = ar[i][1][0];
      }
    };
    // eigenvector columns: type A (x0, x1) <- (s x0 + c x1, c x0 - s x1); type B with the neighbours' columns
    auto vrotA = [&](double c, double s) {
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        double x0 = vr[j][0], x1 = vr[j][1];
        vr[j][0] = s * x0 + c * x1;
        vr[j][1] = c * x0 - s * x1;
      }
    };
    auto colB = [&](double& x0, double& x1, double c, double s, double cm, double sm) {
      double xq = lane_next(x0), xp = lane_prev(x1);
      double n1 = s * x1 + c * xq;       // position 2J+1: pair J = (own column 1, next lane's column 0)
      double n0 = cm * xp - sm * x0;     // position 2J  : pair J-1 = (previous lane's column 1, own column 0)
      x0 = n0; x1 = n1;
    };
    auto vrotB = [&](double c, double s, double cm, double sm) {
#pragma unroll
      for (int j = 0; j < RPW; ++j) colB(vr[j][0], vr[j][1], c, s, cm, sm);
    };
    for (;;) {
      double off2 = 0.0;
#pragma unroll
      for (int i = 0; i < SPW; ++i) {
        const bool dg = (lane == I0 + i);
        off2 += ar[i][0][1] * ar[i][0][1] + ar[i][1][0] * ar[i][1][0];
        if (!dg) off2 += ar[i][0][0] * ar[i][0][0] + ar[i][1][1] * ar[i][1][1];
      }
      off2 = block_sum(off2, red);
      if (off2 <= thresh2 || sweeps >= a.max_sweeps) break;
      publish(0);
      __syncthreads();
      // pending eigenvector update of the previous round (applied after the barrier so that it overlaps the next
      // round's rotation set-up); starts as the identity of type B
      double cB = 0.0, sB = 1.0, cmB = 0.0, smB = -1.0;
      for (int t = 0; t < np; t += 2) {
        double cA, sA;
        {  // ===== round t, type A: pairs (2J, 2J+1)
          const double2 dg = *reinterpret_cast<const double2*>(prm + lane * 4);
          const double oa = prm[lane * 4 + 2];
          jacobi_cs(dg.x, dg.y, oa, rot_thr, cA, sA);
          if (lane >= m) { cA = 1.0; sA = 0.0; }
          nrot += (sA != 0.0);
          vrotB(cB, sB, cmB, smB);
#pragma unroll
          for (int i = 0; i < SPW; ++i) {
            const int I = I0 + i;
            const double cI = lane_bcast(cA, I), sI = lane_bcast(sA, I);
            double t00 = sA * ar[i][0][0] + cA * ar[i][0][1], t01 = cA * ar[i][0][0] - sA * ar[i][0][1];
            double t10 = sA * ar[i][1][0] + cA * ar[i][1][1], t11 = cA * ar[i][1][0] - sA * ar[i][1][1];
            ar[i][0][0] = sI * t00 + cI * t10; ar[i][1][0] = cI * t00 - sI * t10;
            ar[i][0][1] = sI * t01 + cI * t11; ar[i][1][1] = cI * t01 - sI * t11;
            if (lane == I) { ar[i][0][1] = 0.0; ar[i][1][0] = 0.0; }
          }
          publish(1);
          double* hw = halo + ((wv * 2) * 64 + lane) * 2;
          *reinterpret_cast<double2*>(hw) = make_double2(ar[0][0][0], ar[0][0][1]);
          *reinterpret_cast<double2*>(hw + 128) = make_double2(ar[SPW - 1][1][0], ar[SPW - 1][1][1]);
          __syncthreads();
        }
        {  // ===== round t + 1, type B: pairs (2J+1, 2J+2); edge lanes / slots get the no-move parameters
          const double app = prm[(65 + lane) * 4 + 1], aqq = prm[(65 + lane + 1) * 4], ob = prm[(65 + lane) * 4 + 3];
          jacobi_cs(app, aqq, ob, rot_thr, cB, sB);
          if (lane >= m - 1) { cB = 0.0; sB = 1.0; } else nrot += (sB != 0.0);
          cmB = lane_prev(cB); smB = lane_prev(sB);
          if (lane == 0) { cmB = 0.0; smB = -1.0; }
          double hp0 = 0.0, hp1 = 0.0, hn0 = 0.0, hn1 = 0.0;
          if (wv > 0) { double2 h = *reinterpret_cast<const double2*>(halo + (((wv - 1) * 2 + 1) * 64 + lane) * 2); hp0 = h.x; hp1 = h.y; }
          if (wv < NWV - 1) { double2 h = *reinterpret_cast<const double2*>(halo + (((wv + 1) * 2) * 64 + lane) * 2); hn0 = h.x; hn1 = h.y; }
          vrotA(cA, sA);
          // columns
#pragma unroll
          for (int i = 0; i < SPW; ++i) {
            colB(ar[i][0][0], ar[i][0][1], cB, sB, cmB, smB);
            colB(ar[i][1][0], ar[i][1][1], cB, sB, cmB, smB);
          }
          colB(hp0, hp1, cB, sB, cmB, smB);
          colB(hn0, hn1, cB, sB, cmB, smB);
          // rows: pair I = (row 1 of slot I, row 0 of slot I + 1)
          {
            double cP = 0.0, sP = -1.0;
            if (wv > 0) { cP = lane_bcast(cB, I0 - 1); sP = lane_bcast(sB, I0 - 1); }
            ar[0][0][0] = cP * hp0 - sP * ar[0][0][0];
            ar[0][0][1] = cP * hp1 - sP * ar[0][0][1];
          }
#pragma unroll
          for (int i = 0; i < SPW; ++i) {
            const int I = I0 + i;
            const double cI = lane_bcast(cB, I), sI = lane_bcast(sB, I);
            if (i + 1 < SPW) {
#pragma unroll
              for (int cc = 0; cc < 2; ++cc) {
                double x = ar[i][1][cc], y = ar[i + 1][0][cc];
                ar[i][1][cc] = sI * x + cI * y;
                ar[i + 1][0][cc] = cI * x - sI * y;
              }
            } else {
              ar[i][1][0] = sI * ar[i][1][0] + cI * hn0;
              ar[i][1][1] = sI * ar[i][1][1] + cI * hn1;
            }
          }
#pragma unroll
          for (int i = 0; i < SPW; ++i) {
            const int I = I0 + i;
            if (lane == I + 1) ar[i][1][0] = 0.0;
            if (lane == I - 1) ar[i][0][1] = 0.0;
          }
          publish(0);
          __syncthreads();
        }
      }
      vrotB(cB, sB, cmB, smB);
      ++sweeps;
    }
    // ---- write back: eigenvalues to the diagonal of A, eigenvectors to V (position order); rows np .. npg-1 of the
    // LDS copy are re-zeroed because the exchange scratch may have covered them
    pofs = ((n & 1) && (sweeps & 1)) ? 1 : 0;   // an odd number of reversals leaves the padded index at position 0
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int I = I0 + i;
      if (lane == I && I < m) { A[(2 * I) * lda + 2 * I] = ar[i][0][0]; A[(2 * I + 1) * lda + 2 * I + 1] = ar[i][1][1]; }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        int r = wv * RPW + j, col = 2 * lane + cc;
        if (V_LDS) { if (r < np && col < np) V[r + (size_t)col * ldv] = vr[j][cc]; }
        else { int cg = col - pofs; if (r < n && col < np && cg >= 0 && cg < n) V[r + (size_t)cg * ldv] = vr[j][cc]; }
      }
    if (V_LDS)
      for (int idx = tid; idx < (npg - np) * np; idx += NT) { int r = np + idx / np, col = idx - (r - np) * np; V[r + (size_t)col * ldv] = 0.0; }
    __syncthreads();
  } else
  if (BLOCK) {
    constexpr int NW = NT / 64;
    const int nb = npg >> 3, hb = nb >> 1, Mb = nb - 1, ntile = npg >> 4;
    const int lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, lc = lane >> 4;
    for (;;) {
      double off2 = 0.0;
      for (int j = tid >> 6; j < np; j += NT >> 6)
        for (int i = (tid & 63) + j + 1; i < np; i += 64) { double v = A[i * lda + j]; off2 += v * v; }
      off2 = 2.0 * block_sum(off2, red);
      if (off2 <= thresh2 || sweeps >= a.max_sweeps) break;
      for (int br = 0; br < Mb; ++br) {
#ifdef NNSDP_STAMPS
        long long tprev = clock64();
#endif
        // ---- phase S: one wave per block pair diagonalises A[I,I] in place (one cyclic sweep), J accumulates
        if (wv < hb) {
          const int P = pair_top(wv, br, Mb, hb), Q = pair_bot(wv, br, Mb, hb);
          double* Jb = Jall + (size_t)wv * 272;
          double* ics = Jb + 256;
#pragma unroll
          for (int t = 0; t < 4; ++t) { int e = lane + 64 * t; Jb[e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0; }
          auto gi = [&](int l) { return l < 8 ? 8 * P + l : 8 * Q + l - 8; };   // local 0..15 -> matrix index
          wave_lds_sync();
          for (int ir = 0; ir < 15; ++ir) {
            if (lane < 8) {
              int p = gi(pair_top(lane, ir, 15, 8)), q = gi(pair_bot(lane, ir, 15, 8));
              double c, sn;
              jacobi_cs(A[p * lda + p], A[q * lda + q], A[p * lda + q], rot_thr, c, sn);
              ics[2 * lane] = c; ics[2 * lane + 1] = sn;
            }
            wave_lds_sync();
            {
              int sa = lane >> 3, sb = lane & 7;
              int l1p = pair_top(sa, ir, 15, 8), l1q = pair_bot(sa, ir, 15, 8);
              int l2p = pair_top(sb, ir, 15, 8), l2q = pair_bot(sb, ir, 15, 8);
              int p1 = gi(l1p), q1 = gi(l1q), p2 = gi(l2p), q2 = gi(l2q);
              double c1 = ics[2 * sa], s1 = ics[2 * sa + 1], c2 = ics[2 * sb], s2 = ics[2 * sb + 1];
              double b00 = A[p1 * lda + p2], b01 = A[p1 * lda + q2], b10 = A[q1 * lda + p2], b11 = A[q1 * lda + q2];
              double t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
              double t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
              double n00 = t00 * c2 - t01 * s2, n01 = t00 * s2 + t01 * c2;
              double n10 = t10 * c2 - t11 * s2, n11 = t10 * s2 + t11 * c2;
              if (sa == sb) { n01 = 0.0; n10 = 0.0; }
              A[p1 * lda + p2] = n00; A[p1 * lda + q2] = n01; A[q1 * lda + p2] = n10; A[q1 * lda + q2] = n11;
              // J <- J R : (row, slot) items, 2 per lane
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                int it = lane + 64 * t;
                int row = it & 15, sl = it >> 4;
                int jp = pair_top(sl, ir, 15, 8), jq = pair_bot(sl, ir, 15, 8);
                double c = ics[2 * sl], sn = ics[2 * sl + 1];
                double xp = Jb[row * 16 + jp], xq = Jb[row * 16 + jq];
                Jb[row * 16 + jp] = c * xp - sn * xq;
                Jb[row * 16 + jq] = sn * xp + c * xq;
              }
            }
            wave_lds_sync();
          }
        }
        STAMP(0, tprev)
        __syncthreads();
        STAMP(1, tprev)
        // ---- phase U1: A[:, I] <- A[:, I] J (rows outside I) and V[:, I] <- V[:, I] J, one 16-row tile per task
        for (int task = wv; task < 2 * hb * ntile; task += NW) {
          const bool isV = task >= hb * ntile;
          int tt = isV ? task - hb * ntile : task;
          int pi = tt / ntile, tile = tt - pi * ntile;
          const int P = pair_top(pi, br, Mb, hb), Q = pair_bot(pi, br, Mb, hb);
          const double* Jb = Jall + (size_t)pi * 272;
          const int r0 = 16 * tile;
          d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            int kl = 4 * kk + lc;
            int kg = kl < 8 ? 8 * P + kl : 8 * Q + kl - 8;
            double av = isV ? V[(r0 + lr) + (size_t)kg * ldv] : A[(r0 + lr) * lda + kg];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Jb[kl * 16 + lr], c, 0, 0, 0);
          }
          int cg = lr < 8 ? 8 * P + lr : 8 * Q + lr - 8;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = r0 + lc + 4 * r;
            if (isV) V[row + (size_t)cg * ldv] = c[r];
            else if ((row >> 3) != P && (row >> 3) != Q) A[row * lda + cg] = c[r];
          }
        }
        __syncthreads();
        STAMP(2, tprev)
        // ---- phase U2: A[I, :] <- J' A[I, :] (columns outside I), one 16-column tile per task
        for (int task = wv; task < hb * ntile; task += NW) {
          int pi = task / ntile, tile = task - pi * ntile;
          const int P = pair_top(pi, br, Mb, hb), Q = pair_bot(pi, br, Mb, hb);
          const double* Jb = Jall + (size_t)pi * 272;
          const int c0 = 16 * tile;
          d4_t c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            int kl = 4 * kk + lc;
            int kg = kl < 8 ? 8 * P + kl : 8 * Q + kl - 8;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(Jb[kl * 16 + lr], A[kg * lda + c0 + lr], c, 0, 0, 0);
          }
          int col = c0 + lr;
          if ((col >> 3) != P && (col >> 3) != Q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int il = lc + 4 * r;
              int ig = il < 8 ? 8 * P + il : 8 * Q + il - 8;
              A[ig * lda + col] = c[r];
            }
          }
        }
        __syncthreads();
        STAMP(3, tprev)
      }
      ++sweeps;
    }
  } else {
  double* const Tlog = PK ? a.Tg + a.coff[k] : nullptr;     // packed variant: the sweep's rotation log (the warm-start scratch, free by now)
  for (;;) {
    // direct measurement of off(A)^2 (no cancellation): ~1/40 of a sweep
    double off2 = 0.0;
    for (int j = tid >> 6; j < np; j += NT >> 6)
      for (int i = (tid & 63) + j + 1; i < np; i += 64) { double v = A[ixl(i, j)]; off2 += v * v; }
    off2 = 2.0 * block_sum(off2, red);
    if (off2 <= thresh2 || sweeps >= a.max_sweeps) break;
    // parameters of round 0
    if (plane >= 0 && plane < half) {
      int p = pair_top(plane, 0, M, half), q = pair_bot(plane, 0, M, half);
      double c, s;
      jacobi_cs(A[ixl(p, p)], A[ixl(q, q)], A[ixs(p, q)], rot_thr, c, s);
      nrot += (s != 0.0);
      double* dd = desc + 4 * plane;
      dd[0] = c; dd[1] = s;
      reinterpret_cast<int*>(dd + 2)[0] = p; reinterpret_cast<int*>(dd + 2)[1] = q;
    }
    __syncthreads();
    int buf = 0;
    for (int r = 0; r < M; ++r) {
#ifdef NNSDP_STAMPS
      long long tprev = clock64();
#endif
      const double* dsc = desc + buf * 4 * half;
      // phase 1: A <- J' A J on the lower block triangle (every 2x2 block is read and written by its one owner, so the blocks of a
      // thread can be done in chunks: the packed variant holds 4 per thread and would spill with all of them in flight)
      constexpr int CH = PK ? 2 : MAXB;
#pragma unroll
      for (int u0 = 0; u0 < MAXB; u0 += CH) {
        int e00[CH], e01[CH], e10[CH], e11[CH];
        double b00[CH], b01[CH], b10[CH], b11[CH];
        double2 r1[CH], r2[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (blk[u0 + u] >= 0) {
            int ia = blk[u0 + u] >> 8, ib = blk[u0 + u] & 255;
            const double* d1 = dsc + 4 * ia;
            const double* d2 = dsc + 4 * ib;
            r1[u] = *reinterpret_cast<const double2*>(d1);
            r2[u] = *reinterpret_cast<const double2*>(d2);
            int2 pq1 = *reinterpret_cast<const int2*>(d1 + 2), pq2 = *reinterpret_cast<const int2*>(d2 + 2);
            e00[u] = ixs(pq1.x, pq2.x); e01[u] = ixs(pq1.x, pq2.y);
            e10[u] = ixs(pq1.y, pq2.x); e11[u] = ixs(pq1.y, pq2.y);
            b00[u] = A[e00[u]]; b01[u] = A[e01[u]]; b10[u] = A[e10[u]]; b11[u] = A[e11[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (blk[u0 + u] >= 0) {
            double c1 = r1[u].x, s1 = r1[u].y, c2 = r2[u].x, s2 = r2[u].y;
            double t00 = c1 * b00[u] - s1 * b10[u], t01 = c1 * b01[u] - s1 * b11[u];
            double t10 = s1 * b00[u] + c1 * b10[u], t11 = s1 * b01[u] + c1 * b11[u];
            double n00 = t00 * c2 - t01 * s2, n01 = t00 * s2 + t01 * c2;
            double n10 = t10 * c2 - t11 * s2, n11 = t10 * s2 + t11 * c2;
            if ((blk[u0 + u] >> 8) == (blk[u0 + u] & 255)) { A[e00[u]] = n00; A[e11[u]] = n11; A[e01[u]] = 0.0; }
            else { A[e00[u]] = n00; A[e01[u]] = n01; A[e10[u]] = n10; A[e11[u]] = n11; }
          }
        }
      }
      STAMP(0, tprev)
      __syncthreads();
      STAMP(1, tprev)
      // phase 2: V <- V J (all waves but the last)  ||  next round's parameters (last wave)
      if (plane >= 0) {
        if (plane < half && r + 1 < M) {
          int p = pair_top(plane, r + 1, M, half), q = pair_bot(plane, r + 1, M, half);
          double c, s;
          jacobi_cs(A[ixl(p, p)], A[ixl(q, q)], A[ixs(p, q)], rot_thr, c, s);
          nrot += (s != 0.0);
          double* dd = desc + (buf ^ 1) * 4 * half + 4 * plane;
          dd[0] = c; dd[1] = s;
          reinterpret_cast<int*>(dd + 2)[0] = p; reinterpret_cast<int*>(dd + 2)[1] = q;
        }
      } else if constexpr (PK) {
        // packed variant: the eigenvectors live in HBM / L2, and rotating them there every round (the whole V read and written 151
        // times per sweep through one CU's ~25 KB in flight) was 5/6 of the sweep.  The rounds only LOG their (c, s) - the first
        // M - 1 of them into the warm-start scratch (exactly (np - 2) half 2 <= n^2 doubles; the last round's are still in LDS when
        // the sweep ends) - and V takes all rotations of the sweep afterwards, 16 rows at a time through LDS (below).
        if (r < M - 1 && tid < 2 * half) Tlog[(size_t)r * 2 * half + tid] = dsc[4 * (tid >> 1) + (tid & 1)];
      } else {
        {
#pragma unroll
          for (int u = 0; u < MAXU; ++u) {
            if (vunit[u] >= 0) {
              const double* d1 = dsc + 4 * (vunit[u] >> 8);
              int row = vunit[u] & 255;
              double2 cs2 = *reinterpret_cast<const double2*>(d1);
              double c = cs2.x, sn = cs2.y;
              int2 pq = *reinterpret_cast<const int2*>(d1 + 2);
              if (V_LDS || (pq.x < nv && pq.y < nv)) {       // padded index: rotation is the identity
                double* vp_ = V + (size_t)pq.x * ldv + row;
                double* vq_ = V + (size_t)pq.y * ldv + row;
                double xp = *vp_, xq = *vq_;
                *vp_ = c * xp - sn * xq;
                *vq_ = sn * xp + c * xq;
              }
            }
          }
        }
      }
      STAMP(2, tprev)
      __syncthreads();
      STAMP(3, tprev)
      buf ^= 1;
    }
    RST(15)
    if constexpr (PK) {
      // V <- V J_0 J_1 ... J_{M-1}, 32 rows at a time through LDS: thread = (row, stream), stream s takes the pair slots s, s + 32, ...
      // of every round (the pairs of a round are disjoint; one barrier per round orders the rounds); the (c, s) of kLogBatch rounds at a
      // time are staged in LDS (next batch prefetched into registers while this one runs).  Panel stride 161: the 32 rows of a column
      // sit in 32 different banks.  Panels are loaded and stored as 256-byte pieces of V's columns.
      constexpr int kLogBatch = 4, kPanelLd = 161, kRows = 32;
      double* const Pn = red + 16 + (npg >> 1) + 2;                   // kRows x kPanelLd panel
      double* const ring = Pn + kRows * kPanelLd;                     // 2 x kLogBatch x 2 half
      const double* const lastdsc = desc + ((M - 1) & 1) * 4 * half;  // parameters of the sweep's last round
      const int prow = tid & (kRows - 1), stream = tid >> 5;
      const int per = 2 * half, nbatch = (M + kLogBatch - 1) / kLogBatch;
      // this thread's entry of batch b of the log (kLogBatch x 2 half <= 640 doubles: threads [0, 640)); the pair's two indices ride
      // along in a third ring plane, computed once per slot and round here instead of once per ROW by every thread of the panel
      const int e_r = tid / per, e_w = tid - e_r * per;
      auto fetch = [&](int b) {
        const int r = b * kLogBatch + e_r;
        double v = 0.0;
        if (tid < kLogBatch * per && r < M) v = r < M - 1 ? Tlog[(size_t)r * per + e_w] : lastdsc[4 * (e_w >> 1) + (e_w & 1)];
        return v;
      };
      int* const ringpq = reinterpret_cast<int*>(ring + 2 * kLogBatch * per);      // 2 x kLogBatch x half packed (p << 16 | q), -1: identity
      auto stash_pq = [&](int b) {
        const int r = b * kLogBatch + tid / half, slot = tid % half;
        if (tid < kLogBatch * half) {
          int v = -1;
          if (r < M) { const int pp = pair_top(slot, r, M, half), qq = pair_bot(slot, r, M, half); if (pp < nv && qq < nv) v = (pp << 16) | qq; }
          ringpq[(b & 1) * kLogBatch * half + tid] = v;
        }
      };
      for (int g0 = 0; g0 < n; g0 += kRows) {
        for (int j = tid >> 5; j < n; j += NT >> 5) { const int i = g0 + prow; Pn[prow * kPanelLd + j] = i < n ? V[i + (size_t)j * ldv] : 0.0; }
        double st = fetch(0);
        if (tid < kLogBatch * per) ring[tid] = st;
        stash_pq(0);
        __syncthreads();
        double* const row = Pn + prow * kPanelLd;
        for (int b = 0; b < nbatch; ++b) {
          if (b + 1 < nbatch) st = fetch(b + 1);
          const double* cs = ring + (b & 1) * kLogBatch * per;
          const int* pqs = ringpq + (b & 1) * kLogBatch * half;
          const int r1 = min(kLogBatch, M - b * kLogBatch);
          for (int rr = 0; rr < r1; ++rr) {
            // (requesting the operands of the thread's three pairs together before writing any - one LDS round trip per round - costs 13
            // spilled VGPRs and is slower, 1.58 M against 1.44 M cycles per sweep; one pair through FOUR rows per thread - a third fewer LDS
            // bytes per row rotation - is slower too, 1.08 against 1.02 ms per warm launch, and so are two rows per thread, 1.12: only the layout with
            // the 32 rows of the panel across the lanes is free of bank conflicts)
            for (int slot = stream; slot < half; slot += NT >> 5) {
              const int pq = pqs[rr * half + slot];
              if (pq >= 0) {                  // (padded index: the rotation is the identity)
                const double c = cs[rr * per + 2 * slot], sn = cs[rr * per + 2 * slot + 1];
                const int p_ = pq >> 16, q_ = pq & 0xffff;
                const double xp = row[p_], xq = row[q_];
                row[p_] = c * xp - sn * xq;
                row[q_] = sn * xp + c * xq;
              }
            }
            if (rr == r1 - 1 && b + 1 < nbatch) {      // (the other ring half: nobody reads it in batch b)
              if (tid < kLogBatch * per) ring[((b + 1) & 1) * kLogBatch * per + tid] = st;
              stash_pq(b + 1);
            }
            __syncthreads();
          }
        }
        for (int j = tid >> 5; j < n; j += NT >> 5) { const int i = g0 + prow; if (i < n) V[i + (size_t)j * ldv] = Pn[prow * kPanelLd + j]; }
        __syncthreads();
      }
    }
    RST(16)
    ++sweeps;
  }
  }
#ifdef NNSDP_STAMPS
  sec_t[3] = clock64();
  if (k == 0 && (tid & 63) == 0 && a.eig) {   // debug build only: per-wave phase cycles into the tail of eig[]
    long long* dbg = reinterpret_cast<long long*>(a.eig + 4096);
    for (int i = 0; i < 4; ++i) dbg[(tid >> 6) * 4 + i] = st_acc[i];
    if (tid == 0) { dbg[64] = sweeps; dbg[65] = sec_t[1] - sec_t[0]; dbg[66] = sec_t[2] - sec_t[1]; dbg[67] = sec_t[3] - sec_t[2]; dbg[68] = sec_t[3]; }
  }
#endif
  if (tid == 0 && a.stats && sweeps > 0) { atomicAdd(&a.stats[0], sweeps); atomicMax(&a.stats[1], sweeps); }
  if (a.stats && sweeps > 0 && plane >= 0 && plane < half) { atomicAdd(&a.stats[2], nrot); atomicAdd(&a.stats[3], sweeps * M); }

  // ---- eigenvalues on the diagonal; the smaller side of the spectrum gives the rank-k update
  {
    // selection list in index order, built by two (three: packed variant) waves with ballots (positions 0..127 / 0..191)
    const int nl = (SYS || PP) ? np : n;   // positions that can hold an eigenvalue (the padded one is exactly 0: never selected)
    constexpr int kSelThreads = PK ? 192 : 128;
    int* cnt = reinterpret_cast<int*>(red);
    const int ln = tid & 63, w2 = tid >> 6;
    double l = 0.0;
    if (tid < kSelThreads) {
      if (tid < nl) l = A[ixl(tid, tid)];
      const unsigned long long bp = __ballot(l > 0.0), bn = __ballot(l < 0.0);
      if (ln == 0) { cnt[2 * w2] = __popcll(bp); cnt[2 * w2 + 1] = __popcll(bn); }
    }
    if (!PK && tid == 0) { cnt[4] = 0; cnt[5] = 0; }
    __syncthreads();
    const int npos = cnt[0] + cnt[2] + cnt[4], nneg = cnt[1] + cnt[3] + cnt[5];
    const bool up = side_force != 0 ? side_force > 0 : npos <= nneg;
    if (tid < kSelThreads) {
      const bool me = up ? (l > 0.0) : (l < 0.0);
      const unsigned long long mk = __ballot(me);
      int before = 0;
      for (int q = 0; q < w2; ++q) before += up ? cnt[2 * q] : cnt[2 * q + 1];
      const int rank = __popcll(mk & ((1ull << ln) - 1ull)) + before;
      if (me) sel[rank] = tid;
    }
    if (tid == 0) { sel[npg] = up ? npos : nneg; sel[npg + 1] = up ? 1 : 0; }
  }
  __syncthreads();
  RST(10)
  const int nsel = sel[npg];
  const bool use_pos = sel[npg + 1] != 0;
  if (a.eig && tid < n) a.eig[a.eoff[k] + tid] = A[ixl(tid + pofs, tid + pofs)];
  const double kap = a.kappa ? *a.kappa : 1.0;
  double* wk = a.w + a.coff[k];
  double* nuw = a.nu + a.coff[k];
  if (V_LDS) {
    // W = sum_t lam_t v_t v_t' over the selected eigenpairs as MFMA tiles (K = nsel padded to 4); lower
    // tiles are computed and mirrored; !use_pos: W = sym(nu) - (negative part)
    constexpr int NW = NT / 64;
    const int nt = (n + 15) >> 4, ntl = nt * (nt + 1) / 2, ks = (nsel + 3) >> 2;
    const int lane = tid & 63, wv = tid >> 6;
    const int lr = lane & 15, lc = lane >> 4;
    for (int t = wv; t < ntl; t += NW) {
      int ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      int tj = t - ti * (ti + 1) / 2;
      d4_t c = {0.0, 0.0, 0.0, 0.0};
      double sn[4] = {0.0, 0.0, 0.0, 0.0};      // sym(nu) of the tile, requested before the products (L2 latency behind the matrix work)
      if (!use_pos) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + lc + 4 * r, col = 16 * tj + lr;
          if (row < n && col < n) sn[r] = 0.5 * (nuk[(size_t)col * n + row] + nuk[(size_t)row * n + col]);
        }
      }
      for (int kk = 0; kk < ks; ++kk) {
        int tt = 4 * kk + lc;
        double av = 0.0, bv = 0.0;
        if (tt < nsel) {
          int l = sel[tt];
          av = A[l * lda + l] * V[16 * ti + lr + (size_t)l * ldv];
          bv = V[16 * tj + lr + (size_t)l * ldv];
        }
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 16 * ti + lc + 4 * r, col = 16 * tj + lr;
        if (row < n && col < n) {
          double v = c[r];
          if (!use_pos) v = sn[r] - v;
          wk[(size_t)col * n + row] = v;
          if (ti != tj) wk[(size_t)row * n + col] = v;
        }
      }
    }
  } else if constexpr (PK) {
    // packed variant: W = sum_t lam_t v_t v_t' on the matrix cores, the selected eigenvectors staged through LDS 16 at a time
    // (k-major panel in the space the sweeps' eigenvector panel used; every thread re-reading its 2 nsel operands from L2 - the form
    // below - was 16 % of a warm launch of the 151-wide blocks).  Lower tiles, four per wave at most (55 tiles at n = 160).
    constexpr int NW = NT / 64, kPanelLd = 161, kTilesPerWave = 4;
    double* const Pn = red + 16 + (npg >> 1) + 2;       // 16 x kPanelLd
    double* const lamc = Pn + 16 * kPanelLd;            // the chunk's 16 eigenvalues (0 past the selection)
    const int nt = (n + 15) >> 4, ntl = nt * (nt + 1) / 2;
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lc = lane >> 4;
    int tti[kTilesPerWave], ttj[kTilesPerWave];
    d4_t acc[kTilesPerWave];
#pragma unroll
    for (int m = 0; m < kTilesPerWave; ++m) {
      const int t = wv + m * NW;
      int ti = -1, tj = -1;
      if (t < ntl) { ti = 0; while ((ti + 1) * (ti + 2) / 2 <= t) ++ti; tj = t - ti * (ti + 1) / 2; }
      tti[m] = ti; ttj[m] = tj;
      acc[m] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();      // (the selection list is complete; the panel space is free)
    for (int c0 = 0; c0 < nsel; c0 += 16) {
      {
        const int kk = tid >> 6, l = c0 + kk < nsel ? sel[c0 + kk] : -1;
        for (int i = tid & 63; i < 16 * nt; i += 64) Pn[kk * kPanelLd + i] = (l >= 0 && i < n) ? V[i + (size_t)l * ldv] : 0.0;
        if ((tid & 63) == 0) lamc[kk] = l >= 0 ? A[ixl(l, l)] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kTilesPerWave; ++m)
        if (tti[m] >= 0) {
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int kx = 4 * k4 + lc;
            const double av = lamc[kx] * Pn[kx * kPanelLd + 16 * tti[m] + lr], bv = Pn[kx * kPanelLd + 16 * ttj[m] + lr];
            acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m], 0, 0, 0);
          }
        }
      __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < kTilesPerWave; ++m)
      if (tti[m] >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tti[m] + lc + 4 * r, col = 16 * ttj[m] + lr;
          if (row < n && col < n) {
            double v = acc[m][r];
            if (!use_pos) v = 0.5 * (nuk[(size_t)col * n + row] + nuk[(size_t)row * n + col]) - v;
            wk[(size_t)col * n + row] = v;
            if (tti[m] != ttj[m]) wk[(size_t)row * n + col] = v;
          }
        }
      }
  } else {
    for (int j = tid >> 6; j < n; j += NT >> 6)
      for (int i = tid & 63; i < n; i += 64) {
        double s = 0.0;
        for (int t = 0; t < nsel; ++t) {
          int l = sel[t], lv = l - pofs;   // V in HBM is stored without the padded column
          s += A[ixl(l, l)] * V[i + (size_t)lv * ldv] * V[j + (size_t)lv * ldv];
        }
        double nij = nuk[(size_t)j * n + i], nji = nuk[(size_t)i * n + j];
        if (!use_pos) s = 0.5 * (nij + nji) - s;
        wk[(size_t)j * n + i] = s;
      }
  }
  RST(11)
  if (kap != 1.0) {
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += NT) { double wv = wk[idx]; nuw[idx] = wv + kap * (nuw[idx] - wv); }
  }
  if (V_LDS) {
    double* vg = a.Vg + a.coff[k];
    for (int j = tid >> 6; j < n; j += NT >> 6)
      for (int i = tid & 63; i < n; i += 64) vg[(size_t)j * n + i] = V[i + (j + pofs) * ldv];
  }
#ifdef NNSDP_STAMPS
  if (k == 0 && tid == 0 && a.eig) { long long* dbg = reinterpret_cast<long long*>(a.eig + 4096); dbg[69] = clock64() - dbg[68]; }
  if (PK && k == 0 && tid == 0 && !a.eig) {
    if (rst[5]) printf("[stamps packed stage n=%d] Gram %lld analysis %lld K + sums %lld K^2 + X %lld V X + copy %lld\n", n, rst[1] - rst[0], rst[2] - rst[1], rst[3] - rst[2], rst[4] - rst[3], rst[5] - rst[4]);
    printf("[stamps packed n=%d] (last sweep: V update %lld) load %lld congruence %lld sweeps %lld (%d) select %lld reconstruct + stores %lld | total %lld\n", n, rst[16] - rst[15], sec_t[1] - sec_t[0], sec_t[2] - sec_t[1],
           sec_t[3] - sec_t[2], sweeps, rst[10] - sec_t[3], clock64() - rst[10], clock64() - sec_t[0]);
  }
  if (PP && k == 0 && (tid == 0 || tid == NT - 64) && !a.eig) {
    const long long te = clock64();
    printf("[stamps n=%d wave %d] (load: to LDS %lld barrier %lld symmetrise %lld sum %lld) load %lld congruence %lld to-stage %lld | gram %lld diag %lld analyse %lld | E~ %lld sums %lld E~^2 %lld V-update %lld lambda %lld | to-select %lld select %lld W %lld V-store %lld | total %lld\n",
           n, tid >> 6, rst[12] - sec_t[0], rst[13] - rst[12], rst[14] - rst[13], sec_t[1] - rst[14], sec_t[1] - sec_t[0], sec_t[2] - sec_t[1], rst[0] - sec_t[2], rst[1] - rst[0], rst[2] - rst[1], rst[3] - rst[2], rst[4] - rst[3], rst[5] - rst[4],
           rst[6] - rst[5], rst[7] - rst[6], rst[8] - rst[7], sec_t[3] - (rst[8] ? rst[8] : rst[3]), rst[10] - sec_t[3], rst[11] - rst[10], te - rst[11], te - sec_t[0]);
  }
#endif
}

template <bool V_LDS, int NT, int ALG = 0, int SPW = 1, int RPW = 1>
__global__ __launch_bounds__(NT) void k_proj_jacobi(ProjArgs a) {
  int k = blockIdx.x, role = 0;
  if (ALG == 3 && a.split) {
    // helpers first (they never wait), leaders from the next multiple of 8: workgroup ids go round robin over the 8 XCDs, so a block's
    // two workgroups share an XCD - and its L2, through which the helper's tiles travel
    const int nb8 = (a.nblk + 7) & ~7;
    if (k < nb8) { if (k >= a.nblk) return; role = 2; }
    else { k -= nb8; role = 1; }
  }
  proj_body<V_LDS, NT, ALG, SPW, RPW>(a, k, role);
}
// one launch for the blocks of SEVERAL independent SDPs: map[blockIdx.x] = (SDP, block of that SDP)
template <bool V_LDS, int NT, int ALG = 0, int SPW = 1, int RPW = 1>
__global__ __launch_bounds__(NT) void k_proj_jacobi_b(const ProjArgs* __restrict__ args, const int2* __restrict__ map) {
  const int2 m = map[blockIdx.x];
  const int sdp = __builtin_amdgcn_readfirstlane(m.x), blk = __builtin_amdgcn_readfirstlane(m.y);   // wave-uniform: keep the arguments in SGPRs
  const ProjArgs a = args[sdp];
  proj_body<V_LDS, NT, ALG, SPW, RPW>(a, blk);
}

// launch: projection algorithm by the largest block of the launch.
//   kProjSystolic (default for 49 <= nmax <= 128): 512 threads, matrix + eigenvectors in registers
//   kProjRoundRobin: LDS-resident round robin, NT = 1024 above kSmallBlock, 256 below (default for small blocks)
//   kProjBlock: block Jacobi on MFMA (diagnostic; slower)
enum { kProjRoundRobin = 0, kProjBlock = 1, kProjSystolic = 2, kProjPingPong = 3, kProjPacked = 4 };
static constexpr int kMaxLdsBlock = 160;      // largest block the LDS-resident kernel takes (packed variant); above: library path
inline bool proj_packed_ok(int nmax) { return nmax > 128 && nmax <= kMaxLdsBlock; }
static constexpr int kSmallBlock = 40;
static constexpr int kSysMin = 49;
inline bool proj_sys_ok(int nmax) { return nmax >= kSysMin && nmax <= 128; }
inline bool proj_pp_ok(int nmax) { return nmax > kSmallBlock && nmax <= 96; }   // V in LDS, 1024 threads
#define NNSDP_PROJ_VARIANTS(X) \
  X((k_proj_jacobi<true, 1024, 1>)) X((k_proj_jacobi<true, 256, 1>)) X((k_proj_jacobi<true, 1024>)) X((k_proj_jacobi<false, 1024>)) \
  X((k_proj_jacobi<true, 256>)) X((k_proj_jacobi<false, 256>)) X((k_proj_jacobi<true, 512, 2, 6, 12>)) X((k_proj_jacobi<false, 512, 2, 8, 16>)) X((k_proj_jacobi<false, 512, 2, 7, 14>)) X((k_proj_jacobi<true, 1024, 3, 1, 5>)) X((k_proj_jacobi<true, 1024, 3, 1, 6>)) X((k_proj_jacobi<true, 1024, 3, 1, 7>)) \
  X((k_proj_jacobi<false, 1024, 4>)) X((k_proj_jacobi_b<false, 1024, 4>)) \
  X((k_proj_jacobi_b<true, 1024>)) X((k_proj_jacobi_b<false, 1024>)) X((k_proj_jacobi_b<true, 256>)) X((k_proj_jacobi_b<false, 256>)) \
  X((k_proj_jacobi_b<true, 512, 2, 6, 12>)) X((k_proj_jacobi_b<false, 512, 2, 8, 16>)) X((k_proj_jacobi_b<false, 512, 2, 7, 14>)) X((k_proj_jacobi_b<true, 1024, 3, 1, 5>)) X((k_proj_jacobi_b<true, 1024, 3, 1, 6>)) X((k_proj_jacobi_b<true, 1024, 3, 1, 7>))
inline void launch_proj(const ProjArgs& a, int nblocks, int nmax, bool v_lds, size_t lds, hipStream_t st, int alg = kProjRoundRobin) {
  if (alg == kProjPacked) { hipLaunchKernelGGL((k_proj_jacobi<false, 1024, 4>), dim3(nblocks), dim3(1024), lds, st, a); return; }
  if (alg == kProjPingPong && proj_pp_ok(nmax) && v_lds) {
    const bool split = a.split != 0 && a.warm != 0 && a.nblk == nblocks;
    ProjArgs b = a;
    b.split = split ? 1 : 0;
    const int grid = split ? ((nblocks + 7) & ~7) + nblocks : nblocks;
    if (nmax <= 74) hipLaunchKernelGGL((k_proj_jacobi<true, 1024, 3, 1, 5>), dim3(grid), dim3(1024), lds, st, b);
    else if (nmax <= 90) hipLaunchKernelGGL((k_proj_jacobi<true, 1024, 3, 1, 6>), dim3(grid), dim3(1024), lds, st, b);
    else hipLaunchKernelGGL((k_proj_jacobi<true, 1024, 3, 1, 7>), dim3(grid), dim3(1024), lds, st, b);
    return;
  }
  if (alg == kProjSystolic && proj_sys_ok(nmax)) {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi<true, 512, 2, 6, 12>), dim3(nblocks), dim3(512), lds, st, a);
    else if (nmax <= 110) hipLaunchKernelGGL((k_proj_jacobi<false, 512, 2, 7, 14>), dim3(nblocks), dim3(512), lds, st, a);   // width-50 path cliques (101..103)
    else hipLaunchKernelGGL((k_proj_jacobi<false, 512, 2, 8, 16>), dim3(nblocks), dim3(512), lds, st, a);
    return;
  }
  if (alg == kProjBlock && v_lds) {
    if (nmax > kSmallBlock) hipLaunchKernelGGL((k_proj_jacobi<true, 1024, 1>), dim3(nblocks), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((k_proj_jacobi<true, 256, 1>), dim3(nblocks), dim3(256), lds, st, a);
    return;
  }
  if (nmax > kSmallBlock) {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi<true, 1024>), dim3(nblocks), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((k_proj_jacobi<false, 1024>), dim3(nblocks), dim3(1024), lds, st, a);
  } else {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi<true, 256>), dim3(nblocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_proj_jacobi<false, 256>), dim3(nblocks), dim3(256), lds, st, a);
  }
}
// batched form: `nblocks` blocks in total over the SDPs of a batch handle (device arrays args / map)
inline void launch_proj_batched(const ProjArgs* dargs, const int2* dmap, int nblocks, int nmax, bool v_lds, size_t lds, hipStream_t st,
                                int alg) {
  if (alg == kProjPacked) { hipLaunchKernelGGL((k_proj_jacobi_b<false, 1024, 4>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap); return; }
  if (alg == kProjPingPong && proj_pp_ok(nmax) && v_lds) {
    if (nmax <= 74) hipLaunchKernelGGL((k_proj_jacobi_b<true, 1024, 3, 1, 5>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap);
    else if (nmax <= 90) hipLaunchKernelGGL((k_proj_jacobi_b<true, 1024, 3, 1, 6>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap);
    else hipLaunchKernelGGL((k_proj_jacobi_b<true, 1024, 3, 1, 7>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap);
    return;
  }
  if (alg == kProjSystolic && proj_sys_ok(nmax)) {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi_b<true, 512, 2, 6, 12>), dim3(nblocks), dim3(512), lds, st, dargs, dmap);
    else if (nmax <= 110) hipLaunchKernelGGL((k_proj_jacobi_b<false, 512, 2, 7, 14>), dim3(nblocks), dim3(512), lds, st, dargs, dmap);
    else hipLaunchKernelGGL((k_proj_jacobi_b<false, 512, 2, 8, 16>), dim3(nblocks), dim3(512), lds, st, dargs, dmap);
    return;
  }
  if (nmax > kSmallBlock) {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi_b<true, 1024>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap);
    else hipLaunchKernelGGL((k_proj_jacobi_b<false, 1024>), dim3(nblocks), dim3(1024), lds, st, dargs, dmap);
  } else {
    if (v_lds) hipLaunchKernelGGL((k_proj_jacobi_b<true, 256>), dim3(nblocks), dim3(256), lds, st, dargs, dmap);
    else hipLaunchKernelGGL((k_proj_jacobi_b<false, 256>), dim3(nblocks), dim3(256), lds, st, dargs, dmap);
  }
}
inline hipError_t proj_allow_big_lds() {
  hipError_t e = hipSuccess;
#define NNSDP_SET_LDS(K) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  NNSDP_PROJ_VARIANTS(NNSDP_SET_LDS)
#undef NNSDP_SET_LDS
  return e;
}

inline size_t proj_lds_bytes(int nmax, bool v_lds, int alg = kProjRoundRobin) {
  int np = (nmax + 15) & ~15;
  if (alg == kProjPacked)    // packed lower triangle, desc[2][half][4], red, sel, the 32-row eigenvector panel (stride 161) and the rotation-log ring (2 x 4 rounds of (c, s) and of the packed pair indices)
    return ((((size_t)np * (np + 1)) / 2 + 1) + 4 * (size_t)np + 16 + (np >> 1) + 2 + 2 + 32 * 161 + 10 * (size_t)np) * sizeof(double);
  if (alg == kProjSystolic) {
    size_t am = ((size_t)np * (np + 1) + 1) & ~(size_t)1;
    size_t d = am + (v_lds ? (size_t)np * (np + 1) : 0);
    size_t scratch = (size_t)sys_scratch_doubles(8);
    return (kSysHead + (d > scratch ? d : scratch)) * sizeof(double);
  }
  if (alg == kProjPingPong) {   // two bordered matrices (np + 2 rows, stride kPpLda), rotation table, red, sel
    size_t am = (size_t)(np + 2) * kPpLdp;
    return (2 * am + 4 * (size_t)np + 16 + (np >> 1) + 2) * sizeof(double);
  }
  size_t d = (size_t)np * (np + 1) + 1 + 16 + (np >> 1) + 2;           // A, red, sel[np+2] (ints)
  if (v_lds) d += (size_t)np * (np + 1);
  if (alg == kProjBlock) d += (size_t)(np >> 4) * 272;                 // J + rotation parameters per block pair
  else d += 4 * (size_t)np;                                            // desc[2][half][4]
  return d * sizeof(double);
}
// block mode needs V in LDS and its scratch next to it
inline bool proj_block_ok(int nmax) { return proj_lds_bytes(nmax, true, kProjBlock) <= 160 * 1024; }

// everything one plain ADMM iteration of one SDP needs besides the projection (batch handles: one launch per stage
// for all SDPs, blockIdx.y = SDP)
struct IterArgs {
  int ng, NE, ldm, nlong;
  long long nmat;
  const int* sptr; const long long* soff; const unsigned char* isdiag;
  const int* csc_ptr; const int* csc_row; const double* csc_val;
  const int* csr_ptr; const int* csr_col; const double* csr_val;
  const int* longrows; const unsigned int* gidx;
  const int* medrows; const int* medsrc; int nmed, nmsrc, nnz;   // rows of A with 3 .. kLongRow nonzeros / pattern entries with more than 2 sources
  const int* colcls; int ncs;                                      // multipliers by column length: the first ncs of colcls have <= kShortCol nonzeros
  const double* z0; const double* Dinv; const double* c; const double* Minv;
  double* nu; double* w; double* g; double* p; double* qv; double* ww; double* x;
  const double* sigma; double* kappa;
  double alpha;
  double* symv_part;     // [nb][nb][64] partial products of the tiled symmetric M^-1 q (batch handles), nb = ceil(ng / 64)
};

// ---------------------------------------------------------------------------------------------
// multiplier block: w_s = max(nu_s, 0), reflection p = 2 w_s - nu_s - c   (also applies kappa)
// fused into the A' product: qv[g] = sum_e A[e,g] gvec[e] - p[g]; one wave per multiplier.
// ---------------------------------------------------------------------------------------------
// Columns of A (one per multiplier) by length (round 4): two thirds of them hold at most 64 nonzeros - a wave per column left most lanes
// idle and, in the batch handle, 29 k waves for three dependent round trips each (16.9 us for 13 SDPs).  Columns of up to kShortCol
// nonzeros take 16 lanes (blocks [0, nsb): 16 columns per workgroup, the whole column in flight at once), the others a wave with four
// chunks of 64 in flight; colcls lists the short ones first.
static constexpr int kShortCol = 64;
__device__ __forceinline__ void spmv_At_body(const int bid, const int nsb, int ng, const int* __restrict__ ptr, const int* __restrict__ row,
                                                       const double* __restrict__ val, const double* __restrict__ gvec,
                                                       double* __restrict__ nus, const double* __restrict__ c,
                                                       const double* __restrict__ kappa, double* __restrict__ p,
                                                       double* __restrict__ qv, const int* __restrict__ colcls, const int ncs) {
  int g, lane;
  double s = 0.0;
  const bool shortc = bid < nsb;
  if (shortc) {
    const int gi = bid * (kThreads / 16) + (threadIdx.x >> 4);
    lane = threadIdx.x & 15;
    if (gi >= ncs) return;            // (whole 16-lane groups leave together: the DPP row sums below stay inside a group)
    g = colcls[gi];
  } else {
    const int wi = (bid - nsb) * (kThreads / 64) + (threadIdx.x >> 6);
    lane = threadIdx.x & 63;
    if (wi >= ng - ncs) return;
    g = colcls[ncs + wi];
  }
  // (everything lane 0 needs at the end is requested before the column loop: a dependent round trip less)
  const double v0 = nus[g], cg = c[g];
  const double kap = kappa ? *kappa : 1.0;
  const int q0 = ptr[g], q1 = ptr[g + 1];
  if (shortc) {
    double vv[4]; int rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int qq = max(min(q0 + lane + 16 * u, q1 - 1), 0); vv[u] = val[qq]; rr[u] = row[qq]; }      // (an empty column reads a neighbour's entry and discards it)
    double gg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) gg[u] = gvec[rr[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (q0 + lane + 16 * u < q1) ? vv[u] * gg[u] : 0.0;
    s = row_sum16(s);
  } else {
    // (a tenth of the columns hold 250 .. 530 nonzeros - the sector multipliers' W' diag W blocks: four chunks of 64 are requested
    // together, index / value loads first and the gathers behind them, instead of one dependent chain per chunk)
    for (int q = q0 + lane; q < q1; q += 256) {
      double vv[4]; int rr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int qq = min(q + 64 * u, q1 - 1); vv[u] = val[qq]; rr[u] = row[qq]; }
      double gg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) gg[u] = gvec[rr[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (q + 64 * u < q1) ? vv[u] * gg[u] : 0.0;
    }
    s = wave_sum(s);
  }
  if (lane == 0) {
    double v = v0, wv = v > 0.0 ? v : 0.0;
    if (kap != 1.0) { v = wv + kap * (v - wv); nus[g] = v; }
    double pp = 2.0 * wv - v - cg;
    p[g] = pp;
    qv[g] = s - pp;
  }
}
__global__ __launch_bounds__(kThreads) void k_spmv_At(int nsb, int ng, const int* __restrict__ ptr, const int* __restrict__ row,
                                                       const double* __restrict__ val, const double* __restrict__ gvec,
                                                       double* __restrict__ nus, const double* __restrict__ c,
                                                       const double* __restrict__ kappa, double* __restrict__ p,
                                                       double* __restrict__ qv, const int* __restrict__ colcls, int ncs) {
  spmv_At_body(blockIdx.x, nsb, ng, ptr, row, val, gvec, nus, c, kappa, p, qv, colcls, ncs);
}
__global__ __launch_bounds__(kThreads) void k_spmv_At_b(const IterArgs* __restrict__ A, const int nsb_max) {
  const IterArgs a = A[blockIdx.y];
  spmv_At_body(blockIdx.x, nsb_max, a.ng, a.csc_ptr, a.csc_row, a.csc_val, a.g, a.nu, a.c, a.kappa, a.p, a.qv, a.colcls, a.ncs);
}

// g[e] = Dinv[e] (z0[e] / sigma + wgt * sum_src (2 w - nu)[src]); kGatherLanes lanes per pattern entry: the entries of the
// block every clique shares ((x_K, a): 19 sources at W40-D20, 39 at W40-D40) are a dependent chain of 19-39 indirect loads for
// ONE thread otherwise, and the launch lasts as long as that chain (11.1 us for 27.6 k entries)
static constexpr int kGatherLanes = 8;
__device__ __forceinline__ double gather_sum(double s) {   // sum over the kGatherLanes lanes of an entry, valid in all of them
#pragma unroll
  for (int o = 1; o < kGatherLanes; o <<= 1) s += __shfl_xor(s, o, kGatherLanes);
  return s;
}
// Round 4: most pattern entries have ONE source (an element of one clique) or two (an overlap); only the block every clique shares
// and the overlap strips have more.  Eight lanes per entry for all of them meant 8 x NE threads whose launches, in the batch handle,
// ran at 1.6 TB/s for want of resident waves (13 SDPs: 11 k workgroups, five rounds of three dependent round trips).  Entries with
// up to two sources now take ONE thread (blocks [0, nshort)), the others 16 lanes each from a list built at set-up (blocks behind).
__device__ __forceinline__ void gather_g_body(const int bid, const int nshort, int NE, const int* __restrict__ sptr, const long long* __restrict__ soff,
                                                        const unsigned char* __restrict__ isdiag,
                                                        const double* __restrict__ nuk, const double* __restrict__ wk,
                                                        const double* __restrict__ z0, const double* __restrict__ Dinv,
                                                        const double* __restrict__ sigma, double* __restrict__ g,
                                                        const int* __restrict__ medsrc, const int nmsrc) {
  if (bid < nshort) {
    const int e = bid * kThreads + threadIdx.x;
    if (e >= NE) return;
    const int p0 = sptr[e], ns = sptr[e + 1] - p0;
    const double zz = z0[e], dd = Dinv[e], sg = *sigma;
    const bool dg = isdiag[e] != 0;
    if (ns > 2) return;
    // (both source slots are read whatever ns is - clamped, masked afterwards: no load waits behind a branch)
    const long long o0 = soff[p0], o1 = soff[p0 + (ns > 1 ? 1 : 0)];
    const double a0 = 2.0 * wk[o0] - nuk[o0], a1 = 2.0 * wk[o1] - nuk[o1];
    double s = (ns > 0 ? a0 : 0.0) + (ns > 1 ? a1 : 0.0);
    if (!dg) s *= kSqrt2;
    g[e] = dd * (zz / sg + s);
    return;
  }
  const int t = (bid - nshort) * kThreads + threadIdx.x;
  const int r = t >> 4, sub = t & 15;
  if (r >= nmsrc) return;   // (the lanes of one entry leave together)
  const int e = medsrc[r];
  double s = 0.0;
  for (int q = sptr[e] + sub; q < sptr[e + 1]; q += 16) { long long o = soff[q]; s += 2.0 * wk[o] - nuk[o]; }
  s = row_sum16(s);
  if (sub != 0) return;
  if (!isdiag[e]) s *= kSqrt2;
  g[e] = Dinv[e] * (z0[e] / (*sigma) + s);
}
__global__ __launch_bounds__(kThreads) void k_gather_g(int nshort, int NE, const int* __restrict__ sptr, const long long* __restrict__ soff,
                                                        const unsigned char* __restrict__ isdiag,
                                                        const double* __restrict__ nuk, const double* __restrict__ wk,
                                                        const double* __restrict__ z0, const double* __restrict__ Dinv,
                                                        const double* __restrict__ sigma, double* __restrict__ g,
                                                        const int* __restrict__ medsrc, int nmsrc) { gather_g_body(blockIdx.x, nshort, NE, sptr, soff, isdiag, nuk, wk, z0, Dinv, sigma, g, medsrc, nmsrc); }
// batch handles: blocks [0, nshort_max) one thread per entry, blocks behind 16 lanes per listed entry (every member leaves the blocks
// past its own counts at once)
__global__ __launch_bounds__(kThreads) void k_gather_g_b(const IterArgs* __restrict__ A, int nshort_max) {
  const IterArgs a = A[blockIdx.y];
  const int mine = (a.NE + kThreads - 1) / kThreads;
  int bid = blockIdx.x;
  if (bid < nshort_max) { if (bid >= mine) return; }
  else { bid = bid - nshort_max + mine; if ((long long)(bid - mine) * kThreads >= (long long)a.nmsrc * 16) return; }
  gather_g_body(bid, mine, a.NE, a.sptr, a.soff, a.isdiag, a.nu + a.ng, a.w + a.ng, a.z0, a.Dinv, a.sigma, a.g, a.medsrc, a.nmsrc);
}

// clique-sharded mode: h[e] = wgt * sum over the rank's OWN sources of (2 w - nu), or of (nu - w) when dual != 0;
// summed over ranks with one all-reduce, then k_finish_g
__global__ __launch_bounds__(kThreads) void k_gather_h(int NE, const int* __restrict__ sptr, const long long* __restrict__ soff,
                                                        const unsigned char* __restrict__ isdiag, const double* __restrict__ nuk,
                                                        const double* __restrict__ wk, int dual, double* __restrict__ h,
                                                        const unsigned long long* __restrict__ ipc_ctr) {
  // (device-side transport: the partial sum goes into the slot of the NEXT exchange of this rank's mapped buffer; the exchange
  // number lives on the device, so a replayed hipGraph picks the right slot whatever ran between two replays)
  if (ipc_ctr) h += (size_t)((*ipc_ctr + 1) & 1) * NE;
  const int t = blockIdx.x * kThreads + threadIdx.x;
  const int e = t / kGatherLanes, sub = t % kGatherLanes;
  if (e >= NE) return;
  double s = 0.0;
  for (int q = sptr[e] + sub; q < sptr[e + 1]; q += kGatherLanes) {
    long long o = soff[q];
    s += dual ? nuk[o] - wk[o] : 2.0 * wk[o] - nuk[o];
  }
  s = gather_sum(s);
  if (sub != 0) return;
  if (!isdiag[e]) s *= kSqrt2;
  h[e] = s;
}

// ---------------------------------------------------------------------------------------------
// Clique-sharded mode, device-side transport (round 4): the ranks are processes of one node whose exchange buffers are mapped into
// each other's address space with hipIpc (one card, or peers over xGMI).  Every rank writes its partial consensus sum into slot
// (exchange number & 1) of ITS buffer, publishes the exchange number behind a system-scope release, and then sums the slots of all
// ranks in rank order - one-shot all-gather + local reduce, the same bits on every rank, no library kernel
// in the iteration.  Two slots are enough: a rank can only reach exchange c + 2 after every peer has published c + 1, i.e. after it
// has finished reading slot c & 1.  The wait is a bounded spin (a peer that never publishes sets the error word and the caller
// throws at its next check instead of hanging the card).
//   layout of a rank's buffer: [2][NE] doubles | flags[2] (unsigned long long) at double offset 2 NE
// ---------------------------------------------------------------------------------------------
struct IpcArgs {
  double* peer[8];              // every rank's buffer in THIS process's address space (own buffer included), rank order
  int nranks, rank, NE;
  unsigned long long* ctr;      // this rank's exchange counter (device)
  int* err;                     // set to 1 when a wait ran out
  long long spin_limit;
};
__global__ void k_ipc_publish(IpcArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long c = *a.ctr + 1;
  *a.ctr = c;
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(a.peer[a.rank] + 2 * (size_t)a.NE);
  __threadfence_system();                                   // the slot's stores (previous kernel) are complete and visible
  __hip_atomic_store(flags + (c & 1), c, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(kThreads) void k_ipc_reduce(IpcArgs a, double* __restrict__ out) {
  __shared__ int bad;
  const unsigned long long c = *a.ctr;                      // (published by this rank earlier on the stream)
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  if ((int)threadIdx.x < a.nranks && (int)threadIdx.x != a.rank) {      // one lane per peer: the waits (a fabric round trip each) overlap
    const int r = threadIdx.x;
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(a.peer[r] + 2 * (size_t)a.NE) + (c & 1);
    long long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < c) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > a.spin_limit) { atomicExch(a.err, 1); bad = 1; break; }
    }
    __threadfence_system();
  }
  __syncthreads();
  const int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= a.NE) return;
  const size_t o = (size_t)(c & 1) * a.NE + e;
  double s = 0.0;
  for (int r = 0; r < a.nranks; ++r)      // (system-scope loads: a peer's lines may sit stale in this XCD's L2 from two exchanges ago)
    s += __hip_atomic_load(a.peer[r] + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  out[e] = s;
}

__global__ __launch_bounds__(kThreads) void k_finish_g(int NE, const double* __restrict__ h, const double* __restrict__ z0,
                                                        const double* __restrict__ Dinv, const double* __restrict__ sigma,
                                                        double* __restrict__ g) {
  int e = blockIdx.x * kThreads + threadIdx.x;
  if (e < NE) g[e] = Dinv[e] * (z0[e] / (*sigma) + h[e]);
}

// ww = Minv qv  (Minv symmetric, column-major): one wave per output row, coalesced column reads
__device__ __forceinline__ void gemv_sym_body(const int bid, int n, int ldm, const double* __restrict__ Minv, const double* __restrict__ x,
                                                        double* __restrict__ y) {
  int i = (bid * kThreads + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (i >= n) return;
  const double* col = Minv + (size_t)i * ldm;  // ldm even: 16-byte aligned columns
  double s0 = 0.0, s1 = 0.0;
  int j = lane * 2;
  for (; j + 1 < n; j += 128) {
    double2 m = *reinterpret_cast<const double2*>(col + j);
    double2 xv = *reinterpret_cast<const double2*>(x + j);
    s0 += m.x * xv.x;
    s1 += m.y * xv.y;
  }
  if (j < n) s0 += col[j] * x[j];
  double s = wave_sum(s0 + s1);
  if (lane == 0) y[i] = s;
}
__global__ __launch_bounds__(kThreads) void k_gemv_sym(int n, int ldm, const double* __restrict__ Minv, const double* __restrict__ x,
                                                        double* __restrict__ y) { gemv_sym_body(blockIdx.x, n, ldm, Minv, x, y); }
__global__ __launch_bounds__(kThreads) void k_gemv_sym_b(const IterArgs* __restrict__ A) {
  const IterArgs a = A[blockIdx.y];
  if ((long long)blockIdx.x * kThreads >= (long long)a.ng * 64) return;
  gemv_sym_body(blockIdx.x, a.ng, a.ldm, a.Minv, a.qv, a.ww);
}

// ---------------------------------------------------------------------------------------------
// Batch handles: ww = Minv qv reading only the LOWER triangle of the symmetric Minv (half the HBM traffic of
// k_gemv_sym; with 13 SDPs the product is 26 % of a lockstep iteration).  64 x 64 tiles (I >= J), one wave per tile:
// lane r owns row r of the tile and adds T[r][c] x_J[c] over the columns (direct part, y_I); the transposed part
// y_J[c] = sum_r T[r][c] x_I[r] needs a sum ACROSS lanes per column - done 8 columns at a time with a halving
// exchange (4 + 2 + 1 values, then three full steps on one value: 10 exchanges per 8 columns instead of 48).
// Every tile writes its two 64-vectors into its own slot of part[block][other block][64]; a second launch adds the
// nb slots of every block in a fixed order, so the result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_symv_tiles_b(const IterArgs* __restrict__ A) {
  const IterArgs a = A[blockIdx.y];
  const int n = a.ng, nb = (n + 63) >> 6;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (t >= nb * (nb + 1) / 2) return;
  int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int r0 = I << 6, c0 = J << 6;
  const int row = r0 + lane;
  const bool rv = row < n;
  const double* __restrict__ x = a.qv;
  const double* __restrict__ Mp = a.Minv + row;
  const double xi = rv ? x[row] : 0.0;
  double acc = 0.0;        // direct part: row `row` of y_I
  double outT = 0.0;       // transposed part: column tcol(lane) of y_J
  const int ldm = a.ldm;
#pragma unroll 1
  for (int g8 = 0; g8 < 8; ++g8) {
    double p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + 8 * g8 + i;
      const double v = (rv && c < n) ? Mp[(size_t)c * ldm] : 0.0;
      const double xc = c < n ? x[c] : 0.0;      // wave-uniform
      acc += v * xc;
      p[i] = v * xi;
    }
    if (I != J) {
      // 8 column sums over 64 lanes.  Halving steps over lane bits 5, 4, 3 ...
      const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
      double q[4], r[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double send = h5 ? p[i] : p[4 + i], keep = h5 ? p[4 + i] : p[i]; q[i] = keep + __shfl_xor(send, 32, 64); }
#pragma unroll
      for (int i = 0; i < 2; ++i) { const double send = h4 ? q[i] : q[2 + i], keep = h4 ? q[2 + i] : q[i]; r[i] = keep + __shfl_xor(send, 16, 64); }
      double sv;
      { const double send = h3 ? r[0] : r[1], keep = h3 ? r[1] : r[0]; sv = keep + __shfl_xor(send, 8, 64); }
      // ... then plain butterfly steps over bits 2, 1, 0
      sv += __shfl_xor(sv, 4, 64);
      sv += __shfl_xor(sv, 2, 64);
      sv += __shfl_xor(sv, 1, 64);
      // lane holds the sum of column 8 g8 + 4 h5 + 2 h4 + h3 of the tile; lanes with (lane & 7) == g8 keep it
      if ((lane & 7) == g8) outT = sv;
    }
  }
  double* part = a.symv_part;
  part[((size_t)I * nb + J) * 64 + lane] = acc;
  if (I != J) {
    const int tc = 8 * (lane & 7) + ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    part[((size_t)J * nb + I) * 64 + tc] = outT;
  }
}
__global__ __launch_bounds__(64) void k_symv_reduce_b(const IterArgs* __restrict__ A) {
  const IterArgs a = A[blockIdx.y];
  const int n = a.ng, nb = (n + 63) >> 6;
  const int b = blockIdx.x;
  if (b >= nb) return;
  const double* part = a.symv_part + (size_t)b * nb * 64 + threadIdx.x;
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= nb; k += 8) {          // loads of 8 slots in flight, added in slot order
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = part[(size_t)(k + i) * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  for (; k < nb; ++k) s += part[(size_t)k * 64];
  const int row = b * 64 + threadIdx.x;
  if (row < n) a.ww[row] = s;
}

// x[e] = g[e] - Dinv[e] * sum_g A[e,g] ww[g]; 4 lanes per pattern entry (W40-D20: 27.6 k rows, 60 % of them with no
// multiplier at all and 37 % with one; only 2 880 rows carry more than 16 nonzeros)
static constexpr int kRowLanes = 8;
static constexpr int kLongRow = 256;   // rows of A with more nonzeros go to the block-per-row kernel
// Round 4: 60 % of the rows of A have no multiplier at all and 37 % one (W40-D20); rows with up to two nonzeros take ONE thread
// (blocks [0, nshort)), rows with 3 .. kLongRow nonzeros 16 lanes each from a list built at set-up (the first form gave every row 8
// lanes: 8 x NE threads per SDP, and the batched launch ran at 0.6 TB/s - resident-wave-bound, five rounds of dependent round trips)
__device__ __forceinline__ void spmv_A_x_body(const int bid, const int nshort, int NE, const int* __restrict__ ptr, const int* __restrict__ col,
                                                        const double* __restrict__ val, const double* __restrict__ ww,
                                                        const double* __restrict__ g, const double* __restrict__ Dinv,
                                                        double* __restrict__ x, const int* __restrict__ medrows, const int nmed, const int nnz) {
  if (bid < nshort) {
    const int e = bid * kThreads + threadIdx.x;
    if (e >= NE) return;
    const int p0 = ptr[e], n = ptr[e + 1] - p0;
    const double ge = g[e], de = Dinv[e];
    if (n > 2) return;
    // (both slots read whatever n is - clamped into the table, masked afterwards)
    const int q0 = min(p0, nnz - 1), q1 = min(p0 + 1, nnz - 1);
    const double t0 = val[q0] * ww[col[q0]], t1 = val[q1] * ww[col[q1]];
    const double s = (n > 0 ? t0 : 0.0) + (n > 1 ? t1 : 0.0);
    x[e] = ge - de * s;
    return;
  }
  const int t = (bid - nshort) * kThreads + threadIdx.x;
  const int r = t >> 4, sub = t & 15;
  if (r >= nmed) return;
  const int e = medrows[r];
  const double ge = g[e], de = Dinv[e];
  double s = 0.0;
  const int q1 = ptr[e + 1];
  for (int q = ptr[e] + sub; q < q1; q += 64) {      // (four chunks of 16 in flight)
    double vv[4]; int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int qq = min(q + 16 * u, q1 - 1); vv[u] = val[qq]; cc[u] = col[qq]; }
    double xx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xx[u] = ww[cc[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (q + 16 * u < q1) ? vv[u] * xx[u] : 0.0;
  }
  s = row_sum16(s);
  if (sub == 0) x[e] = ge - de * s;
}

// the few long rows (the affine-affine entry touches every multiplier): one workgroup per row
__device__ __forceinline__ void spmv_A_x_long_body(const int bid, int nlong, const int* __restrict__ rows, const int* __restrict__ ptr,
                                                             const int* __restrict__ col, const double* __restrict__ val,
                                                             const double* __restrict__ ww, const double* __restrict__ g,
                                                             const double* __restrict__ Dinv, double* __restrict__ x) {
  __shared__ double red[8];
  int e = rows[bid];
  const double ge = g[e], de = Dinv[e];
  double s = 0.0;
  const int q1 = ptr[e + 1];
  for (int q = ptr[e] + threadIdx.x; q < q1; q += 8 * kThreads) {      // (eight chunks in flight: the affine-affine row touches every multiplier)
    double vv[8]; int cc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int qq = min(q + kThreads * u, q1 - 1); vv[u] = val[qq]; cc[u] = col[qq]; }
    double xx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xx[u] = ww[cc[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (q + kThreads * u < q1) ? vv[u] * xx[u] : 0.0;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) x[e] = ge - de * s;
}
__global__ __launch_bounds__(kThreads) void k_spmv_A_x_long(int nlong, const int* __restrict__ rows, const int* __restrict__ ptr,
                                                             const int* __restrict__ col, const double* __restrict__ val,
                                                             const double* __restrict__ ww, const double* __restrict__ g,
                                                             const double* __restrict__ Dinv, double* __restrict__ x) { spmv_A_x_long_body(blockIdx.x, nlong, rows, ptr, col, val, ww, g, Dinv, x); }
__global__ __launch_bounds__(kThreads) void k_spmv_A_x_long_b(const IterArgs* __restrict__ A) {
  const IterArgs a = A[blockIdx.y];
  if ((long long)blockIdx.x * kThreads >= (long long)a.nlong * kThreads) return;
  spmv_A_x_long_body(blockIdx.x, a.nlong, a.longrows, a.csr_ptr, a.csr_col, a.csr_val, a.ww, a.g, a.Dinv, a.x);
}

// all forms in one launch: blocks [0, nshort) one thread per row (rows with up to two nonzeros), [nshort, nreg) 16 lanes per listed
// medium row, [nreg, nreg + nlong) one workgroup per long row
__global__ __launch_bounds__(kThreads) void k_spmv_A_x_all(int NE, int nshort, int nreg, int nlong, const int* __restrict__ rows,
                                                            const int* __restrict__ ptr, const int* __restrict__ col,
                                                            const double* __restrict__ val, const double* __restrict__ ww,
                                                            const double* __restrict__ g, const double* __restrict__ Dinv,
                                                            double* __restrict__ x, const int* __restrict__ medrows, int nmed, int nnz) {
  if ((int)blockIdx.x < nreg) spmv_A_x_body(blockIdx.x, nshort, NE, ptr, col, val, ww, g, Dinv, x, medrows, nmed, nnz);
  else spmv_A_x_long_body(blockIdx.x - nreg, nlong, rows, ptr, col, val, ww, g, Dinv, x);
}
// batch handles: [0, nshort_max) short rows, [nshort_max, nshort_max + nmedb_max) medium rows, behind them the long rows
__global__ __launch_bounds__(kThreads) void k_spmv_A_x_all_b(const IterArgs* __restrict__ A, int nshort_max, int nmedb_max, int nnz_dummy) {
  const IterArgs a = A[blockIdx.y];
  const int mine = (a.NE + kThreads - 1) / kThreads;
  const int b = blockIdx.x;
  (void)nnz_dummy;
  if (b < nshort_max) {
    if (b >= mine) return;
    spmv_A_x_body(b, mine, a.NE, a.csr_ptr, a.csr_col, a.csr_val, a.ww, a.g, a.Dinv, a.x, a.medrows, a.nmed, a.nnz);
  } else if (b < nshort_max + nmedb_max) {
    const int bm = b - nshort_max;
    if ((long long)bm * kThreads >= (long long)a.nmed * 16) return;
    spmv_A_x_body(mine + bm, mine, a.NE, a.csr_ptr, a.csr_col, a.csr_val, a.ww, a.g, a.Dinv, a.x, a.medrows, a.nmed, a.nnz);
  } else {
    const int bl = b - nshort_max - nmedb_max;
    if (bl >= a.nlong) return;
    spmv_A_x_long_body(bl, a.nlong, a.longrows, a.csr_ptr, a.csr_col, a.csr_val, a.ww, a.g, a.Dinv, a.x);
  }
}

// nu <- nu + alpha (K x + q - w).  acc (may be null) accumulates at check iterations:
//   |Kx+q-w|^2, |Kx+q|^2, |w|^2 as per-workgroup partial sums acc[{0,1,2} * astride + block]; k_acc_reduce adds the partials
//   in block order, so the residuals (and with them every stopping / penalty decision) are reproducible run to run
__device__ __forceinline__ void update_nu_body(const int bid, int ng, long long nmat, const double* __restrict__ p,
                                                         const double* __restrict__ ww, const double* __restrict__ c,
                                                         const double* __restrict__ x, const unsigned int* __restrict__ gidx,
                                                         double* __restrict__ nu, const double* __restrict__ w,
                                                         double alpha, double* __restrict__ kappa, double* __restrict__ acc,
                                                         long long lo, long long hi, int acc_s, int astride) {
  // [lo, hi): the clique elements this rank owns (everything when not sharded); acc_s: count the
  // multiplier block in the residual sums (rank 0 only when sharded, the sums are all-reduced)
  __shared__ double red[8];
  long long i = (long long)bid * kThreads + threadIdx.x;
  double r2 = 0.0, k2 = 0.0, w2 = 0.0;
  if (i < ng) {
    double v = nu[i], wv = v > 0.0 ? v : 0.0;
    double kxq = p[i] + ww[i] + c[i];
    double res = kxq - wv;
    nu[i] = v + alpha * res;
    if (acc_s) { r2 = res * res; k2 = kxq * kxq; w2 = wv * wv; }
  } else if (i < ng + nmat && i - ng >= lo && i - ng < hi) {
    long long m = i - ng;
    unsigned int gi = gidx[m];
    double xv = x[gi & 0x7fffffffu];
    if (!(gi >> 31)) xv *= kInvSqrt2;
    double wv = w[i];
    double res = xv - wv;
    nu[i] += alpha * res;
    r2 = res * res; k2 = xv * xv; w2 = wv * wv;
  }
  if (acc) {
    r2 = block_sum(r2, red);
    k2 = block_sum(k2, red + 4);
    w2 = block_sum(w2, red);
    if (threadIdx.x == 0) { acc[bid] = r2; acc[astride + bid] = k2; acc[2 * astride + bid] = w2; }
  }
  // the one-shot penalty rescale is consumed: kappa is READ only by the projection and the A' kernels, which precede this
  // launch on the iteration's single stream, and never by this kernel - so resetting it here cannot race with a reader
  if (i == 0 && kappa) *kappa = 1.0;
}
__global__ __launch_bounds__(kThreads) void k_update_nu(int ng, long long nmat, const double* __restrict__ p,
                                                         const double* __restrict__ ww, const double* __restrict__ c,
                                                         const double* __restrict__ x, const unsigned int* __restrict__ gidx,
                                                         double* __restrict__ nu, const double* __restrict__ w,
                                                         double alpha, double* __restrict__ kappa, double* __restrict__ acc,
                                                         long long lo, long long hi, int acc_s, int astride) { update_nu_body(blockIdx.x, ng, nmat, p, ww, c, x, gidx, nu, w, alpha, kappa, acc, lo, hi, acc_s, astride); }
__global__ __launch_bounds__(kThreads) void k_update_nu_b(const IterArgs* __restrict__ A) {
  const IterArgs a = A[blockIdx.y];
  if ((long long)blockIdx.x * kThreads >= (long long)a.ng + a.nmat) return;
  update_nu_body(blockIdx.x, a.ng, a.nmat, a.p, a.ww, a.c, a.x, a.gidx, a.nu, a.w, a.alpha, a.kappa, nullptr, 0LL, a.nmat, 1, 0);
}

// check iteration, dual side: t[e] = (K'y)[e] = sum_g A[e,g] ys[g] + wgt * sum_src y_k[src],
// y = sigma (nu - w).  acc[3] += |t - z0|^2, acc[4] += |t|^2; 16 lanes per entry.
__global__ __launch_bounds__(kThreads) void k_check_dual(int NE, int ng, int nreg, int nlong, const int* __restrict__ longrows,
                                                          const int* __restrict__ ptr, const int* __restrict__ col,
                                                          const double* __restrict__ val, const int* __restrict__ sptr,
                                                          const long long* __restrict__ soff, const unsigned char* __restrict__ isdiag,
                                                          const double* __restrict__ nu, const double* __restrict__ w,
                                                          const double* __restrict__ z0, const double* __restrict__ sigma,
                                                          double* __restrict__ acc, const double* __restrict__ hsum, int astride) {
  // hsum != null (clique-sharded mode): the clique part of K'y/sigma, already summed over ranks.
  // Blocks [0, nreg): 16 lanes per entry, rows with more than kLongRow nonzeros skipped; blocks [nreg, nreg + nlong): one
  // workgroup per long row (the (a,a) entry meets every multiplier: 3 203 nonzeros at W40-D20, 200 steps for 16 lanes)
  __shared__ double red[8];
  double sg = *sigma;
  double d2 = 0.0, t2 = 0.0;
  if ((int)blockIdx.x < nreg) {
    int e = (blockIdx.x * kThreads + threadIdx.x) >> 4;
    int sub = threadIdx.x & 15;
    double s = 0.0, h = 0.0;
    const bool mine = e < NE && ptr[e + 1] - ptr[e] <= kLongRow;
    if (mine) {
      for (int q = ptr[e] + sub; q < ptr[e + 1]; q += 16) { double v = nu[col[q]]; s += val[q] * (v < 0.0 ? v : 0.0); }
      if (hsum) { if (sub == 0) h = hsum[e]; }
      else {
        for (int q = sptr[e] + sub; q < sptr[e + 1]; q += 16) { long long o = soff[q]; h += nu[ng + o] - w[ng + o]; }
        if (!isdiag[e]) h *= kSqrt2;
      }
    }
    s += h;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 16);
    if (mine && sub == 0) { double t = sg * s; double d = t - z0[e]; d2 = d * d; t2 = t * t; }
  } else {
    const int b = blockIdx.x - nreg;
    const int e = longrows[b];      // (the grid is exactly nreg + nlong workgroups)
    double s = 0.0, h = 0.0;
    for (int q = ptr[e] + threadIdx.x; q < ptr[e + 1]; q += kThreads) { double v = nu[col[q]]; s += val[q] * (v < 0.0 ? v : 0.0); }
    if (hsum) { if (threadIdx.x == 0) h = hsum[e]; }
    else {
      for (int q = sptr[e] + threadIdx.x; q < sptr[e + 1]; q += kThreads) { long long o = soff[q]; h += nu[ng + o] - w[ng + o]; }
      if (!isdiag[e]) h *= kSqrt2;
    }
    s = block_sum(s + h, red);
    if (threadIdx.x == 0) { double t = sg * s; double d = t - z0[e]; d2 = d * d; t2 = t * t; }
    __syncthreads();
  }
  d2 = block_sum(d2, red);
  t2 = block_sum(t2, red + 4);
  if (threadIdx.x == 0) { acc[3 * astride + blockIdx.x] = d2; acc[4 * astride + blockIdx.x] = t2; }
}

// acc[5] += -c' ys  (scaled primal objective), acc[6] += z0' x (scaled dual objective)
__global__ __launch_bounds__(kThreads) void k_check_obj(int ng, int NE, const double* __restrict__ nu, const double* __restrict__ c,
                                                         const double* __restrict__ z0, const double* __restrict__ x,
                                                         const double* __restrict__ sigma, double* __restrict__ acc, int astride) {
  __shared__ double red[8];
  int i = blockIdx.x * kThreads + threadIdx.x;
  double a = 0.0, b = 0.0;
  if (i < ng) { double v = nu[i]; a = -c[i] * (*sigma) * (v < 0.0 ? v : 0.0); }
  if (i < NE) b = z0[i] * x[i];
  a = block_sum(a, red);
  b = block_sum(b, red + 4);
  if (threadIdx.x == 0) { acc[5 * astride + blockIdx.x] = a; acc[6 * astride + blockIdx.x] = b; }
}

// second stage of the residual sums: acc[q] = sum of the nb[q] workgroup partials of quantity q, added in a fixed order
// (thread t takes partials t, t + 256, ...; then the block tree) - no atomics anywhere on the way to a stopping decision.
__global__ __launch_bounds__(kThreads) void k_acc_reduce(const double* __restrict__ accp, int astride, int nb_upd, int nb_dual, int nb_obj,
                                                          double* __restrict__ acc) {
  __shared__ double red[8];
#pragma unroll 1
  for (int q = 0; q < 7; ++q) {
    const int nb = q < 3 ? nb_upd : (q < 5 ? nb_dual : nb_obj);
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += kThreads) s += accp[(size_t)q * astride + b];
    s = block_sum(s, red);
    if (threadIdx.x == 0) acc[q] = s;
  }
}


// gs[g] = max(-ys[g], 0), ys = sigma * min(nu_s, 0)
__global__ void k_extract_gamma(int ng, const double* __restrict__ nu, const double* __restrict__ sigma, double* __restrict__ gs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ng) { double v = nu[i]; gs[i] = v < 0.0 ? -(*sigma) * v : 0.0; }
}

// K1 (assembly, reference coordinates): z[e] = z0[e] + sum_g A[e,g] gamma[g]; then scatter to dense
__global__ __launch_bounds__(kThreads) void k_apply_A(int NE, const int* __restrict__ ptr, const int* __restrict__ col,
                                                       const double* __restrict__ val, const double* __restrict__ gam,
                                                       const double* __restrict__ z0, double* __restrict__ z) {
  int e = (blockIdx.x * kThreads + threadIdx.x) >> 4;
  int sub = threadIdx.x & 15;
  double s = 0.0;
  if (e < NE)
    for (int q = ptr[e] + sub; q < ptr[e + 1]; q += 16) s += val[q] * gam[col[q]];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 16);
  if (e < NE && sub == 0) z[e] = (z0 ? z0[e] : 0.0) + s;
}

__global__ void k_scatter_dense(int NE, int n, const int* __restrict__ erow, const int* __restrict__ ecol,
                                const double* __restrict__ z, double* __restrict__ Z) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NE) return;
  int i = erow[e], j = ecol[e];
  double v = z[e];
  if (i != j) v *= kInvSqrt2;
  Z[(size_t)j * n + i] = v;
  Z[(size_t)i * n + j] = v;
}

// svec of a dense symmetric matrix on the pattern (for the adjoint test entry point)
__global__ void k_gather_dense(int NE, int n, const int* __restrict__ erow, const int* __restrict__ ecol,
                               const double* __restrict__ X, double* __restrict__ xv) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= NE) return;
  int i = erow[e], j = ecol[e];
  double v = 0.5 * (X[(size_t)j * n + i] + X[(size_t)i * n + j]);
  xv[e] = (i != j) ? v * kSqrt2 : v;
}

// out[g] = sum_e A[e,g] xv[e]  (K2, adjoint of the generator part); one wave per multiplier
__global__ __launch_bounds__(kThreads) void k_apply_At(int ng, const int* __restrict__ ptr, const int* __restrict__ row,
                                                        const double* __restrict__ val, const double* __restrict__ xv,
                                                        double* __restrict__ out) {
  int g = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (g >= ng) return;
  double s = 0.0;
  for (int q = ptr[g] + lane; q < ptr[g + 1]; q += 64) s += val[q] * xv[row[q]];
  s = wave_sum(s);
  if (lane == 0) out[g] = s;
}

}  // namespace nnsdp
