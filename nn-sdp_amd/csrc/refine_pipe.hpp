// K3, tile-parallel form of the refinement stage (round 4): the GEMM-only update of the persistent eigenbasis of every PSD block
// (kernels.hip, proj_body: B = V'AV, E~ from B, X = E~ + E~^2 / 2, V <- V (I + X), second-order eigenvalues, W = sum lambda v v')
// spread over the WHOLE chip instead of one CU per block.  Every 16 x 16 output tile of every product is ONE WAVE that pulls its two
// operand strips straight from L2 into registers in the MFMA operand layout (a strip is 16 x n doubles; the working set of an SDP is a
// few MB, L2 / MALL resident), runs one chain of v_mfma_f64_16x16x4_f64 and writes its tile.  What crosses tiles - the diagonal of B,
// the decision sums, E~ for E~^2, V' for the reconstruction - crosses at a KERNEL BOUNDARY: five short launches on the solver's stream
// (replayed inside its hipGraph), no in-kernel spin, nothing to dead-lock when other processes share the card, deterministic
// (fixed-order reductions), any block size up to 160 in one code path:
//
//   P1 k_pipe_T   T = A V            all tiles      + d = diag(V'T), R_ii = 1 - |v_i|^2, |A|_F^2, Vt = V', snapshot of the block's state
//   P2 k_pipe_B   B = V'T (G = V'V)  lower tiles    + pair analysis -> E~ and E~' (both triangles), 7 partial sums per tile
//   P3 k_pipe_X   X = E~ + E~^2 / 2  lower tiles    + the block's accept / reject decision from the sums (recorded for P4, P5)
//   P4 k_pipe_V   V' = V + V X       all tiles      + column norms, second-order eigenvalues
//   P5 k_pipe_W   W = sum mu v'v''   lower tiles    + w, nu rescale, V' -> Vg, state word, pmode[block] = handled
//
// A block whose step is not accepted (early in a solve, penalty changes, pairs first order cannot resolve) is left untouched with
// pmode = 0 and the one-CU kernel launched behind the pipeline runs it exactly as before (stage attempt with exact pair rotations,
// Newton-Schulz repair, sweeps); handled blocks return from that kernel at once.  The decision rule, its thresholds and the state word
// are those of proj_body's stage.
//
// Workgroup = (block, tile column tj) with one wave per tile row ti; the map is built on the host so that the workgroups of one
// block land on ONE XCD (workgroup id mod 8: MI355X_MICROARCH.md, dispatch) and share that XCD's L2.
// Access pattern (the first form of these kernels let a lane walk its own run of a strided operand - 64 cache lines per wave
// instruction, ~1 500 line requests per wave - and every launch took 9-15 us bound by the CU's address unit): every matrix a product
// reads is kept in the orientation in which the TILE's index is the fast one, so that an operand load is 16 lanes x 8 contiguous
// bytes per lane group (four 128-byte segments per wave instruction): the producing kernel writes the transposed copy where the
// consumer needs one (Vt by P1 from an LDS strip, Et by P2 through a 16 x 16 LDS transpose).  nu is taken as it is stored (it is
// symmetric to rounding by construction of the update; sym(nu) is formed where it matters, in W).
#pragma once

namespace nnsdp {

struct PipeArgs {
  const int* cn;            // block sizes (blocks of this launch)
  const long long* coff;    // element offsets of the blocks in the packed clique storage
  const int2* wgmap;        // workgroup -> (block, tile column); block < 0: idle
  double* nu;               // packed matrices (rescaled in place when kappa != 1, as the one-CU kernel does)
  double* w;                // packed projections (out)
  double* Vg;               // packed eigenbases (in / out)
  double* T;                // n^2 per block: T = A V (row-major), then X (row-major)
  double* E;                // n^2 per block: E~ (row-major)
  double* U;                // n^2 per block: the new basis (column-major)
  double* Vt;               // n^2 per block: V' (row-major copy of the basis, written by P1)
  double* Et;               // n^2 per block: E~' (row-major)
  double* drec;             // [block] PipeRec: the decision, recorded by P3
  double* dvec;             // [block][vs] diag(V'AV)
  double* rdg;              // [block][vs] 1 - |v_i|^2 (0 on visits without the Gram product)
  double* lam;              // [block][vs] second-order eigenvalues of the stepped basis
  double* fro;              // [block] |A|_F^2
  double* psum;             // [block][pt][8] partial sums of the lower tiles: off2, k2, unpp, unnn, unx, kd2, r2
  int* vrec;                // [block][4] snapshot of rstate taken by P1: word, do_gram, rdef (double)
  int* pmode;               // [block] 0: not handled (the one-CU kernel runs it), 1: stepped, 2: converged as it arrived
  int* rstate;              // refinement state per block (ProjArgs::rstate)
  int* stats;               // ProjArgs::stats
  const double* kappa;      // device scalar, may be null
  const double* tol_dev;    // device scalar, may be null
  double tol;
  double refine_acc, refine_kcap, refine_loose;
  int gram_credit;
  int vs;                   // stride of the per-block vectors (multiple of 16, >= largest block)
  int pt;                   // stride of psum in tiles (>= lower tiles of the largest block)
  double* eig;              // optional eigenvalue output (test entry), with eoff
  const long long* eoff;
  long long* dbg;           // (-DNNSDP_STAMPS builds only) [5 kernels][workgroup][8] wall-clock stamps of wave 0
};
#ifdef NNSDP_STAMPS
#define PST(kern, slot) { if (a.dbg && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (kern == 0 || kern == 3 ? 0 : blockIdx.x >= 0 ? (int)(a.wgmap[blockIdx.x].y) : 0)) a.dbg[((size_t)(kern) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); }
#define PWAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PDEP(x) asm volatile("v_mov_b64 %0, %0" : "+v"(x))
#else
#define PST(kern, slot)
#define PWAIT()
#define PDEP(x)
#endif

struct PipeDecision {
  int mode;                 // 0 not handled, 1 step, 2 converged as it arrived
  bool up;                  // rebuild W from the positive side
  bool loose, do_gram;
  double r2, k2, rdef;
  int credit, gcred;
};

// The block's accept / reject decision from the partial sums of P2: executed by one full wave, every lane returns the same bits
// (fixed-order sums), and every wave of P3 .. P5 that calls it gets the same answer.
__device__ __forceinline__ PipeDecision pipe_decide(const PipeArgs& a, const int b, const int n) {
  const int lane = threadIdx.x & 63;
  const int nt = (n + 15) >> 4, ntl = nt * (nt + 1) / 2;
  PipeDecision D;
  const int rs = a.vrec[4 * b];
  D.do_gram = a.vrec[4 * b + 1] != 0;
  D.rdef = *reinterpret_cast<const double*>(a.vrec + 4 * b + 2);
  D.credit = (rs >> 16) & 255; D.gcred = (rs >> 24) & 15;
  D.mode = 0; D.up = true; D.loose = false; D.r2 = 0.0; D.k2 = 0.0;
  if ((rs & 255) != 0) return D;                       // back-off: not attempted
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = lane; t < ntl; t += 64) {
    const double* p = a.psum + ((size_t)b * a.pt + t) * 8;
#pragma unroll
    for (int q = 0; q < 7; ++q) s[q] += p[q];
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) s[q] = wave_sum(s[q]);
  int cpos = 0, cneg = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const double dv = (i0 + lane < n) ? a.dvec[(size_t)b * a.vs + i0 + lane] : 0.0;
    cpos += __popcll(__ballot(dv > 0.0)); cneg += __popcll(__ballot(dv < 0.0));
  }
  const double tolv = a.tol_dev ? *a.tol_dev : a.tol;
  const double fro2 = a.fro[b];
  const double Tl = tolv * sqrt(fro2), accT = a.refine_acc * Tl;
  const double off2 = s[0], k2 = s[1], unpp = s[2], unnn = s[3], unx = s[4], kd2 = s[5];
  const double r2 = D.do_gram ? s[6] : D.rdef * D.rdef;
  D.r2 = r2; D.k2 = k2;
  D.up = cpos <= cneg;
  if (off2 <= Tl * Tl && r2 <= tolv * tolv) { D.mode = 2; return D; }
  if (r2 <= 1e-4) {
    const double pred0 = 1.5 * sqrt(off2) * sqrt(k2) + k2 * sqrt(kd2) * (1.0 / 3.0);
    const double pred_pos = pred0 + sqrt(unpp + unx), pred_neg = pred0 + sqrt(unnn + unx);
    const bool prefer_pos = cpos <= cneg;
    const bool kok = k2 <= 0.09;
    int side = 0;
    if (kok && (prefer_pos ? pred_pos : pred_neg) <= accT) side = prefer_pos ? 1 : -1;
    else if (kok && (prefer_pos ? pred_neg : pred_pos) <= accT) side = prefer_pos ? -1 : 1;
    else if (kok && D.credit >= 16 && fmin(pred_pos, pred_neg) <= a.refine_loose * accT) { side = pred_pos <= pred_neg ? 1 : -1; D.loose = true; }
    if (side != 0) { D.mode = 1; D.up = side > 0; }
  }
  return D;
}

// the products' chain of MFMAs over the contraction index k = 4 kk + lc, operands already in registers.  A wave is alone (or nearly)
// on its SIMD here, and ONE dependent chain of v_mfma_f64_16x16x4_f64 advances at ~150 cycles per instruction (stamps: 22 steps in
// 1.3 - 1.5 us) against the pipe's 64: even and odd steps go to two accumulators, summed at the end (fixed order: deterministic).
template <int KSQ>
__device__ __forceinline__ d4_t pipe_chain(const double (&av)[KSQ], const double (&bv)[KSQ], const int ksq) {
  d4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < KSQ; kk += 2) {
    if (kk < ksq) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], c0, 0, 0, 0);
    if (kk + 1 < KSQ && kk + 1 < ksq) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk + 1], bv[kk + 1], c1, 0, 0, 0);
  }
  return c0 + c1;
}
// operand strip of a matrix stored with the tile's own index fastest (M[k * n + base + lr]): 16 lanes x 8 contiguous bytes per
// lane group, four 128-byte segments per wave instruction - the only global access pattern the products use.
// EVERY LOAD IS UNCONDITIONAL, from a clamped address: the first form guarded each one (`in range ? M[..] : 0.0`), the compiler
// turned that into a wait for the value right behind every load, and a wave spent 3 - 5.7 us issuing its 44 loads one L2 round trip
// at a time (profiles/r04_pipe_stamps_*.log).  Out-of-range k is masked afterwards on ONE operand of the product (pipe_mask_k: the
// other then multiplies a zero); out-of-range rows / columns of a tile compute garbage that no store or sum ever takes.
// Addressing: a per-lane 32-bit byte offset computed ONCE ((lc n + idx) 8) plus a wave-uniform 64-bit base that advances by 32 n bytes
// per step on the scalar unit (global_load ..., v_off, s[base:base+1]).  The first clamped form rebuilt a 64-bit address per load
// on the vector unit (v_min, v_mad_i64_i32, v_lshlrev_b64, v_lshl_add_u64: ~90 cycles per load for a wave alone on its SIMD - the
// 'issue' phase of the stamps, 2.2 - 4 us per wave whatever the block count).  Only the LAST step (kk = ksq - 1) can reach past row
// n - 1: it takes a second per-lane offset with the row clamped; steps past ksq re-read the last one (never used).
template <int KSQ>
__device__ __forceinline__ void pipe_load(double (&v)[KSQ], const double* __restrict__ M, const int n, const int ksq, const int idx, const int lc) {
  const unsigned ic = (unsigned)min(idx, n - 1);
  const unsigned voff = ((unsigned)(lc * n) + ic) * 8u;
  const unsigned voff_last = ((unsigned)((min(4 * (ksq - 1) + lc, n - 1) - 4 * (ksq - 1)) * n) + ic) * 8u;
  const char* base = reinterpret_cast<const char*>(M);
  const unsigned step = 32u * (unsigned)n;
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) {
    const int kc = min(kk, ksq - 1);                                  // (wave-uniform)
    const char* sb = base + (size_t)((unsigned)kc * step);
    v[kk] = *reinterpret_cast<const double*>(sb + (kk >= ksq - 1 ? voff_last : voff));
  }
}
template <int KSQ>
__device__ __forceinline__ void pipe_mask_k(double (&v)[KSQ], const int n, const int lc) {
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) v[kk] = (4 * kk + lc < n) ? v[kk] : 0.0;
}

// decision record written by P3's first wave for P4 / P5
struct PipeRec { int mode, up, loose, do_gram, credit, gcred; double r2, k2, rdef; };

// ---- P1: T = A V (all tiles), d = diag(V'T), R_ii, |A|_F^2, state snapshot, Vt = V' (row-major copy of the basis) ------------------
template <int KSQ>
__global__ __launch_bounds__(16 * KSQ) void k_pipe_T(PipeArgs a) {
  constexpr int NTW = KSQ / 4, K4 = 4 * KSQ;
  __shared__ double Vs[K4 * 17];          // the workgroup's 16 columns of V, k-major (stride 17: conflict-free stores and operand reads)
  __shared__ double red[NTW][64];
  __shared__ double red2[64];
  __shared__ double redf[NTW];
  const int2 m = a.wgmap[blockIdx.x];
  const int b = m.x, tj = m.y;
  if (b < 0) return;
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2;
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  const int rs = a.rstate[4 * b];
  const double rdef = *reinterpret_cast<const double*>(a.rstate + 4 * b + 2);
  const double tolv = a.tol_dev ? *a.tol_dev : a.tol;
  const int gcred = (rs >> 24) & 15;
  const bool do_gram = gcred == 0 || !(rdef <= 0.03 * a.refine_acc * tolv);
  if (tj == 0 && threadIdx.x == 0) {      // (nobody writes rstate while the pipeline's first four launches run)
    a.vrec[4 * b] = rs; a.vrec[4 * b + 1] = do_gram ? 1 : 0;
    *reinterpret_cast<double*>(a.vrec + 4 * b + 2) = rdef;
  }
  if ((rs & 255) != 0) return;            // back-off: the one-CU kernel counts it down (uniform over the workgroup)
  const bool active = ti < nt;
  const double* nuk = a.nu + a.coff[b];
  const double* vk = a.Vg + a.coff[b];
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  double av[KSQ], bv[KSQ];
  pipe_load<KSQ>(av, nuk, n, ksq, i, lc);          // A[i][k] = nu[k n + i]
  // the strip V[:, J]: columns are contiguous in memory, one wave per column (loads first, unconditional; the stores mask)
  {
    constexpr int CW = (16 + NTW - 1) / NTW, RW = (4 * KSQ + 63) / 64;
    double sv[CW][RW];
#pragma unroll
    for (int q = 0; q < CW; ++q)
#pragma unroll
      for (int u = 0; u < RW; ++u) sv[q][u] = vk[(size_t)min(16 * tj + ti + q * NTW, n - 1) * n + min(lane + 64 * u, n - 1)];
#pragma unroll
    for (int q = 0; q < CW; ++q)
#pragma unroll
      for (int u = 0; u < RW; ++u) {
        const int jl = ti + q * NTW, col = 16 * tj + jl, k = lane + 64 * u;
        if (jl < 16 && k < 4 * ksq) Vs[k * 17 + jl] = (k < n && col < n) ? sv[q][u] : 0.0;
      }
  }
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) av[kk] = (active && 4 * kk + lc < n && i < n) ? av[kk] : 0.0;      // (|A|_F^2 sums this strip)
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) bv[kk] = kk < ksq ? Vs[(4 * kk + lc) * 17 + lr] : 0.0;
  {
    // Vt[k][16 tj .. 16 tj + 15] = the strip's rows: what P2 reads as its row-operand
    double* vt = a.Vt + a.coff[b];
    for (int k = 4 * ti + lc; k < n; k += 4 * NTW)
      if (j < n) vt[(size_t)k * n + j] = Vs[k * 17 + lr];
  }
  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);
  double pd = 0.0;
  if (active) {
    double* Tk = a.T + a.coff[b];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ti + lc + 4 * r;
      if (row < n && j < n) { Tk[(size_t)row * n + j] = c[r]; pd += Vs[row * 17 + lr] * c[r]; }
    }
  }
  red[ti][lane] = pd;
  if (ti == 0) {
    double s = 0.0;
#pragma unroll
    for (int kk = 0; kk < KSQ; ++kk) s += bv[kk] * bv[kk];
    red2[lane] = s;
  }
  if (tj == 0) {
    double f = 0.0;
#pragma unroll
    for (int kk = 0; kk < KSQ; ++kk) f += av[kk] * av[kk];
    f = wave_sum(f);
    if (lane == 0) redf[ti] = f;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int t = threadIdx.x, col = 16 * tj + t;
    double d = 0.0, nrm = 0.0;
    for (int w_ = 0; w_ < nt; ++w_)
#pragma unroll
      for (int q = 0; q < 4; ++q) d += red[w_][q * 16 + t];
#pragma unroll
    for (int q = 0; q < 4; ++q) nrm += red2[q * 16 + t];
    if (col < a.vs) {
      a.dvec[(size_t)b * a.vs + col] = col < n ? d : 0.0;
      a.rdg[(size_t)b * a.vs + col] = (col < n && do_gram) ? 1.0 - nrm : 0.0;
    }
  }
  if (tj == 0 && threadIdx.x == 0) {
    double f = 0.0;
    for (int w_ = 0; w_ < nt; ++w_) f += redf[w_];
    a.fro[b] = f;
  }
}

// ---- P2: B = V'T (and G = V'V on Gram visits), pair analysis, E~ and its transpose ------------------------------------------------
template <int KSQ>
__global__ __launch_bounds__(16 * KSQ) void k_pipe_B(PipeArgs a) {
  constexpr int NTW = KSQ / 4;
  __shared__ double tre[NTW][16][17];
  __shared__ double trf[NTW][16][17];
  const int2 m = a.wgmap[blockIdx.x];
  const int b = m.x, tj = m.y;
  if (b < 0) return;
  const int rs = a.vrec[4 * b];
  if ((rs & 255) != 0) return;
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2;
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  if (ti >= nt || ti < tj) return;        // (no workgroup barrier in this kernel)
  const bool do_gram = a.vrec[4 * b + 1] != 0;
  const double* vt = a.Vt + a.coff[b];
  const double* Tk = a.T + a.coff[b];
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  double av[KSQ], bv[KSQ];
  pipe_load<KSQ>(av, vt, n, ksq, i, lc);       // V'[i][k] = Vt[k n + i]
  pipe_load<KSQ>(bv, Tk, n, ksq, j, lc);       // T[k][j]
  // (the Gram product's second operand rides in the same batch of requests when the registers allow it)
  constexpr bool kGramEarly = KSQ <= 24;
  double gv[kGramEarly ? KSQ : 1];
  if constexpr (kGramEarly) { if (do_gram) pipe_load<KSQ>(gv, vt, n, ksq, j, lc); }
  double di[4], ri[4];
  const double* dv = a.dvec + (size_t)b * a.vs;
  const double* rv = a.rdg + (size_t)b * a.vs;
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int row = 16 * ti + lc + 4 * r; di[r] = dv[row]; ri[r] = rv[row]; }
  const double dj = dv[j], rj = rv[j];
  pipe_mask_k<KSQ>(av, n, lc);
  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);
  d4_t gc = {0.0, 0.0, 0.0, 0.0};
  if (do_gram) {
    if constexpr (kGramEarly) gc = pipe_chain<KSQ>(av, gv, ksq);
    else {
      pipe_load<KSQ>(bv, vt, n, ksq, j, lc);     // V[k][j]
      gc = pipe_chain<KSQ>(av, bv, ksq);
    }
  }
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
  // lane layout): E <- e, Et <- f; at address j n + row (through a 16 x 16 transpose in LDS): E <- f, Et <- e.
  double* Ek = a.E + a.coff[b];
  double* Etk = a.Et + a.coff[b];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (row < n && j < n && row >= j) { Ek[(size_t)row * n + j] = eo[r]; Etk[(size_t)row * n + j] = fo[r]; }
    tre[ti][lc + 4 * r][lr] = eo[r]; trf[ti][lc + 4 * r][lr] = fo[r];
  }
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ii = 16 * ti + lr, jj = 16 * tj + lc + 4 * r;     // pair (ii, jj), ii > jj
    if (ii < n && jj < n && ii > jj) { Ek[(size_t)jj * n + ii] = trf[ti][lr][lc + 4 * r]; Etk[(size_t)jj * n + ii] = tre[ti][lr][lc + 4 * r]; }
  }
}

// ---- P3: the decision (every wave, the first one records it), X = E~ + E~^2 / 2 (lower tiles, both triangles written) ---------------
template <int KSQ>
__global__ __launch_bounds__(16 * KSQ) void k_pipe_X(PipeArgs a) {
  constexpr int NTW = KSQ / 4;
  __shared__ double tr[NTW][16][17];
  const int2 m = a.wgmap[blockIdx.x];
  const int b = m.x, tj = m.y;
  if (b < 0) return;
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2;
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  const bool writer = tj == 0 && ti == 0;
  PipeRec* rec = reinterpret_cast<PipeRec*>(a.drec) + b;
  if ((a.vrec[4 * b] & 255) != 0) { if (writer && lane == 0) rec->mode = 0; return; }
  if (ti >= nt || ti < tj) return;
  const double* Ek = a.E + a.coff[b];
  const double* Etk = a.Et + a.coff[b];
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  // (operand loads are requested before the decision is known: plain reads of scratch that always exists; the decision's own
  // dependent loads then run in their shadow)
  double av[KSQ], bv[KSQ];
  pipe_load<KSQ>(av, Etk, n, ksq, i, lc);      // E~[i][k] = Et[k n + i]
  pipe_load<KSQ>(bv, Ek, n, ksq, j, lc);       // E~[k][j]
  double ec[4], et[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(16 * ti + lc + 4 * r, n - 1), rowt = min(16 * tj + lc + 4 * r, n - 1);
    ec[r] = Ek[(size_t)row * n + min(j, n - 1)];
    et[r] = Ek[(size_t)rowt * n + min(i, n - 1)];
  }
  const PipeDecision D = pipe_decide(a, b, n);
  if (writer && lane == 0) {
    rec->mode = D.mode; rec->up = D.up ? 1 : 0; rec->loose = D.loose ? 1 : 0; rec->do_gram = D.do_gram ? 1 : 0;
    rec->credit = D.credit; rec->gcred = D.gcred; rec->r2 = D.r2; rec->k2 = D.k2; rec->rdef = D.rdef;
  }
  if (D.mode != 1) return;
  pipe_mask_k<KSQ>(av, n, lc);
  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);
  double* Xk = a.T + a.coff[b];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (row < n && j < n) Xk[(size_t)row * n + j] = ec[r] + 0.5 * c[r];
    tr[ti][lc + 4 * r][lr] = c[r];
  }
  if (ti != tj) {
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowt = 16 * tj + lc + 4 * r;
      if (rowt < n && i < n) Xk[(size_t)rowt * n + i] = et[r] + 0.5 * tr[ti][lr][lc + 4 * r];      // E~^2 is symmetric
    }
  }
}

// ---- P4: V' = V + V X (all tiles, computed transposed so that the stores are coalesced), column norms, eigenvalues -----------------
template <int KSQ>
__global__ __launch_bounds__(16 * KSQ) void k_pipe_V(PipeArgs a) {
  constexpr int NTW = KSQ / 4;
  __shared__ double red[NTW][16];
  __shared__ double red2[NTW][64];
  __shared__ double red3[NTW][64];
  const int2 m = a.wgmap[blockIdx.x];
  const int b = m.x, tj = m.y;
  if (b < 0) return;
  const PipeRec rec = reinterpret_cast<const PipeRec*>(a.drec)[b];
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2;
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  const bool active = ti < nt;
  const double* vk = a.Vg + a.coff[b];
  const double* Xk = a.T + a.coff[b];
  const double* Ek = a.E + a.coff[b];
  double* Uk = a.U + a.coff[b];
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  double av[KSQ], bv[KSQ];
  pipe_load<KSQ>(av, Xk, n, ksq, j, lc);     // X'[j][k] = X[k n + j]
  pipe_load<KSQ>(bv, vk, n, ksq, i, lc);     // V'[k][i] = V[i][k] = Vg[k n + i]
  double v0[4], ee[4], dk[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = min(16 * tj + lc + 4 * r, n - 1), row = min(16 * ti + lc + 4 * r, n - 1);
    v0[r] = vk[(size_t)col * n + min(i, n - 1)];
    ee[r] = Ek[(size_t)row * n + min(j, n - 1)];     // E~[k = row][j]
    dk[r] = a.dvec[(size_t)b * a.vs + row];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r;
    if (!(active && row < n && j < n && row != j)) ee[r] = 0.0;
  }
  if (rec.mode == 0) return;               // (uniform over the workgroup)
  if (rec.mode == 2) {
    // converged as it arrived: the basis and diag(B) are the result; P5 reads U and lam whatever the mode
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 16 * tj + lc + 4 * r;
      if (active && col < n && i < n) Uk[(size_t)col * n + i] = v0[r];
    }
    if (threadIdx.x < 16 && 16 * tj + threadIdx.x < a.vs) a.lam[(size_t)b * a.vs + 16 * tj + threadIdx.x] = a.dvec[(size_t)b * a.vs + 16 * tj + threadIdx.x];
    return;
  }
  pipe_mask_k<KSQ>(av, n, lc);
  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);
  double nr[4];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = 16 * tj + lc + 4 * r;
    const double vn = v0[r] + c[r];
    if (active && col < n && i < n) Uk[(size_t)col * n + i] = vn;
    nr[r] = row_sum16((active && col < n && i < n) ? vn * vn : 0.0);
    s1 += ee[r] * ee[r]; s2 += ee[r] * ee[r] * dk[r];
  }
  if (lr == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[ti][lc + 4 * r] = nr[r];
  }
  red2[ti][lane] = s1; red3[ti][lane] = s2;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int t = threadIdx.x, col = 16 * tj + t;
    double nrm = 0.0, c1 = 0.0, c2 = 0.0;
    for (int w_ = 0; w_ < nt; ++w_) {
      nrm += red[w_][t];
#pragma unroll
      for (int q = 0; q < 4; ++q) { c1 += red2[w_][q * 16 + t]; c2 += red3[w_][q * 16 + t]; }
    }
    if (col < a.vs) {
      double l = 0.0;
      if (col < n) {
        const double d = a.dvec[(size_t)b * a.vs + col], rr = a.rdg[(size_t)b * a.vs + col];
        l = (d * (1.0 + rr + c1) - c2) / nrm;
      }
      a.lam[(size_t)b * a.vs + col] = l;
    }
  }
}

// ---- P5: W = sum over the chosen side (lower tiles, mirrored), nu rescale, V' -> Vg, state ---------------------------------------
template <int KSQ>
__global__ __launch_bounds__(16 * KSQ) void k_pipe_W(PipeArgs a) {
  constexpr int NTW = KSQ / 4;
  __shared__ double tr[NTW][16][17];
  const int2 m = a.wgmap[blockIdx.x];
  const int b = m.x, tj = m.y;
  if (b < 0) return;
  const PipeRec rec = reinterpret_cast<const PipeRec*>(a.drec)[b];
  const int n = a.cn[b], nt = (n + 15) >> 4, ksq = (n + 3) >> 2;
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  const bool writer = tj == 0 && ti == 0;
  if (ti >= nt || ti < tj) return;        // (no workgroup barrier in this kernel)
  const double* Uk = a.U + a.coff[b];
  const double* lamv = a.lam + (size_t)b * a.vs;
  double* nuk = a.nu + a.coff[b];
  double* wk = a.w + a.coff[b];
  const int i = 16 * ti + lr, j = 16 * tj + lr;
  // (every load below is issued whatever the mode: U, lam and nu always exist)
  double av[KSQ], bv[KSQ], mu[KSQ];
  pipe_load<KSQ>(av, Uk, n, ksq, i, lc);       // V'[i][k] = U[k n + i]
  pipe_load<KSQ>(bv, Uk, n, ksq, j, lc);       // V'[j][k]
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) mu[kk] = lamv[min(4 * kk + lc, n - 1)];
  // nu at the two places this wave writes: nd at [(16 ti + lc + 4r) n + 16 tj + lr] - the MIRROR of the entry the lane computes -
  // and nt_ at [(16 tj + lc + 4r) n + 16 ti + lr], the entry itself in the transposed lane layout
  const double kap = a.kappa ? *a.kappa : 1.0;
  double nd[4], nt_[4], u1[4], u2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * ti + lc + 4 * r, rowt = 16 * tj + lc + 4 * r;
    const size_t ixd = (size_t)min(row, n - 1) * n + min(j, n - 1), ixt = (size_t)min(rowt, n - 1) * n + min(i, n - 1);
    nd[r] = nuk[ixd];
    nt_[r] = nuk[ixt];
    u1[r] = Uk[ixt];       // the basis' tile (rows ti, columns tj) and its mirror tile
    u2[r] = Uk[ixd];
  }
  if (rec.mode == 0) { if (writer && lane == 0) a.pmode[b] = 0; return; }
  const bool up = rec.up != 0;
#pragma unroll
  for (int kk = 0; kk < KSQ; ++kk) av[kk] = (4 * kk + lc < n) ? av[kk] * (up ? fmax(mu[kk], 0.0) : fmin(mu[kk], 0.0)) : 0.0;
  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);
  // value of entry (i' = 16 ti + lc + 4r, j' = 16 tj + lr), i' >= j'
  double vd[4];
  if (!up) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[ti][lr][lc + 4 * r] = nt_[r];      // nu(i' = 16 ti + lr, j' = 16 tj + lc + 4r) -> transposed
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) vd[r] = 0.5 * (nd[r] + tr[ti][lc + 4 * r][lr]) - c[r];
    wave_lds_sync();
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) vd[r] = c[r];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tr[ti][lc + 4 * r][lr] = vd[r];
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
      const double v = tr[ti][lr][lc + 4 * r];
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
  if (a.eig && ti == tj && lane < 16 && 16 * ti + lane < n) a.eig[a.eoff[b] + 16 * ti + lane] = lamv[16 * ti + lane];
  if (writer && lane == 0) {
    const int outcome = rec.mode == 1 ? 1 : 0;
    if (a.stats) atomicAdd(&a.stats[rec.loose ? 8 : 4 + outcome], 1);
    const int word = rec.loose ? 0 : (min(rec.credit + 1, 255) << 16);
    double rnew = rec.do_gram ? sqrt(rec.r2) : rec.rdef;
    if (outcome == 1) rnew = (rec.do_gram ? rec.r2 + 2.0 * sqrt(rec.r2 * rec.k2) : rec.rdef * (1.0 + 2.2 * sqrt(rec.k2))) + 0.25 * rec.k2 * rec.k2;
    a.rstate[4 * b] = word | ((rec.do_gram ? a.gram_credit : rec.gcred - 1) << 24);
    *reinterpret_cast<double*>(a.rstate + 4 * b + 2) = rnew;
    a.pmode[b] = rec.mode;
  }
}

// largest block the pipeline takes (KSQ = 40: ten tile rows, 640 threads per workgroup)
static constexpr int kPipeMaxBlock = 160;

// host side: scratch and the workgroup map of one set of blocks; the five launches
struct RefinePipe {
  int nblocks = 0, nmax = 0, nwg = 0, vs = 16, pt = 1;
  int2* wgmap = nullptr;
  double *T = nullptr, *E = nullptr, *U = nullptr, *Vt = nullptr, *Et = nullptr, *drec = nullptr, *dvec = nullptr, *rdg = nullptr, *lam = nullptr, *fro = nullptr, *psum = nullptr;
  int *vrec = nullptr, *pmode = nullptr;
  bool ready = false;
  ~RefinePipe() { release(); }
  void release() {
    void* ps[] = {wgmap, T, E, U, Vt, Et, drec, dvec, rdg, lam, fro, psum, vrec, pmode};
    for (void* p : ps) if (p) (void)hipFree(p);
    wgmap = nullptr; T = E = U = Vt = Et = drec = dvec = rdg = lam = fro = psum = nullptr; vrec = pmode = nullptr; ready = false;
  }
  // cn[nb]: sizes of the blocks of the launch in launch order; nmat: elements of their packed storage
  hipError_t build(const int* cn, int nb, long long nmat) {
    release();
    nblocks = nb; nmax = 0;
    for (int k = 0; k < nb; ++k) nmax = nmax > cn[k] ? nmax : cn[k];
    if (nb <= 0 || nmax > kPipeMaxBlock) return hipSuccess;
    vs = (nmax + 15) & ~15;
    const int ntm = vs >> 4;
    pt = ntm * (ntm + 1) / 2;
    // workgroups of one block on one XCD: bucket = block mod 8, workgroup id = 8 x position + bucket
    std::vector<std::vector<int2>> bucket(8);
    for (int k = 0; k < nb; ++k)
      for (int tj = 0; tj < (cn[k] + 15) / 16; ++tj) bucket[k & 7].push_back(int2{k, tj});
    size_t depth = 0;
    for (auto& bk : bucket) depth = depth > bk.size() ? depth : bk.size();
    std::vector<int2> map(8 * depth, int2{-1, 0});
    for (int x = 0; x < 8; ++x)
      for (size_t p = 0; p < bucket[x].size(); ++p) map[8 * p + x] = bucket[x][p];
    nwg = (int)map.size();
    hipError_t e = hipSuccess;
    auto al = [&](void** p, size_t bytes) { if (e == hipSuccess) { e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) e = hipMemset(*p, 0, bytes ? bytes : 8); } };
    al((void**)&wgmap, map.size() * sizeof(int2));
    al((void**)&T, (size_t)nmat * 8); al((void**)&E, (size_t)nmat * 8); al((void**)&U, (size_t)nmat * 8);
    al((void**)&Vt, (size_t)nmat * 8); al((void**)&Et, (size_t)nmat * 8); al((void**)&drec, (size_t)nb * 64);
    al((void**)&dvec, (size_t)nb * vs * 8); al((void**)&rdg, (size_t)nb * vs * 8); al((void**)&lam, (size_t)nb * vs * 8);
    al((void**)&fro, (size_t)nb * 8); al((void**)&psum, (size_t)nb * pt * 8 * 8);
    al((void**)&vrec, (size_t)nb * 4 * sizeof(int)); al((void**)&pmode, (size_t)nb * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(wgmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    ready = e == hipSuccess;
    return e;
  }
  // the pipeline's arguments from the one-CU kernel's (same block list, same state)
  PipeArgs args(const ProjArgs& p) const {
    PipeArgs a{};
    a.cn = p.cn; a.coff = p.coff; a.wgmap = wgmap; a.nu = p.nu; a.w = p.w; a.Vg = p.Vg;
    a.T = T; a.E = E; a.U = U; a.Vt = Vt; a.Et = Et; a.drec = drec; a.dvec = dvec; a.rdg = rdg; a.lam = lam; a.fro = fro; a.psum = psum; a.vrec = vrec; a.pmode = pmode;
    a.rstate = p.rstate; a.stats = p.stats; a.kappa = p.kappa; a.tol_dev = p.tol_dev; a.tol = p.tol;
    a.refine_acc = p.refine_acc; a.refine_kcap = p.refine_kcap; a.refine_loose = p.refine_loose; a.gram_credit = p.gram_credit;
    a.vs = vs; a.pt = pt; a.eig = p.eig; a.eoff = p.eoff;
    return a;
  }
  void launch(const PipeArgs& a, hipStream_t st) const {
    if (nmax <= 96) {
      hipLaunchKernelGGL((k_pipe_T<24>), dim3(nwg), dim3(384), 0, st, a);
      hipLaunchKernelGGL((k_pipe_B<24>), dim3(nwg), dim3(384), 0, st, a);
      hipLaunchKernelGGL((k_pipe_X<24>), dim3(nwg), dim3(384), 0, st, a);
      hipLaunchKernelGGL((k_pipe_V<24>), dim3(nwg), dim3(384), 0, st, a);
      hipLaunchKernelGGL((k_pipe_W<24>), dim3(nwg), dim3(384), 0, st, a);
    } else {
      hipLaunchKernelGGL((k_pipe_T<40>), dim3(nwg), dim3(640), 0, st, a);
      hipLaunchKernelGGL((k_pipe_B<40>), dim3(nwg), dim3(640), 0, st, a);
      hipLaunchKernelGGL((k_pipe_X<40>), dim3(nwg), dim3(640), 0, st, a);
      hipLaunchKernelGGL((k_pipe_V<40>), dim3(nwg), dim3(640), 0, st, a);
      hipLaunchKernelGGL((k_pipe_W<40>), dim3(nwg), dim3(640), 0, st, a);
    }
  }
};

}  // namespace nnsdp
