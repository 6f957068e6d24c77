// Structured application of the Woodbury core M^-1, M = I + A' D^-1 A (ng x ng), for large multiplier counts.
//
// The dense inverse costs 8 ng^2 bytes per ADMM iteration and O(ng^3) set-up: 40 MB at W40-D20 beta=0 (served from L2/MALL),
// but 1 GB at W20-D100 beta=7 and 2.5 GB at W40-D40 beta=7 - the beta sweep the reference runs (experiments/scale.jl:28).
// M is not dense: a multiplier of network layer k only touches the pattern entries of (x_k, x_{k+1}, affine index), so with the
// multipliers ordered by layer M is BLOCK-BANDED (half-bandwidth 1 layer; 2 when repeated-nonlinearity pairs straddle a layer
// boundary) plus a low-rank term from the very few pattern entries every generator touches (the affine-affine entry):
//        M = T + U diag(d) U'          T block-banded, U = ng x r (r = 1 here)
// T^-1 is applied by a two-level domain decomposition with everything precomputed as small dense inverses: layers are cut into
// chunks I_j separated by `bw` separator layers S; chunks do not couple with each other, so with P_j = T[I_j,I_j]^-1,
// H_j = P_j T[I_j,S], Sc = T[S,S] - sum_j T[S,I_j] H_j:
//        x_S = Sc^-1 (q_S - sum_j H_j' q_Ij),      x_Ij = P_j q_Ij - H_j x_S
// and the low-rank term by Woodbury with v = T^-1 U:  M^-1 q = T^-1 q - v (diag(1/d) + U'v)^-1 (v'q).
// Three dependent launches per application, operands sum_j n_j^2 + 2 sum_j n_j w_j + n_S^2 doubles (W20-D100 beta=7: ~120 MB
// against 968 MB), set-up O(sum n_j^3).  The structure is COMPUTED from the generator table, not assumed: if chunks turn out to
// couple, or too many rows are global, plan() reports failure and the caller keeps the dense inverse.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "setup.hpp"

namespace nnsdp {

struct MinvPlan {
  bool ok = false;
  int ng = 0, nlayers = 0, bw = 1, r = 0;
  std::vector<int> lstart;               // generator range of every layer (nlayers + 1)
  std::vector<int> grow;                 // global pattern rows (low-rank part)
  // partition: pieces in generator order; chunk j = [clo[j], chi[j]), separators between them
  int nchunk = 0, nS = 0, ldS = 0;
  std::vector<int> clo, chi;             // chunk generator ranges
  std::vector<int> slo, shi;             // separator generator ranges (nchunk - 1 of them; separator j sits after chunk j)
  std::vector<int> soff;                 // offset of separator j inside the packed separator vector (nchunk entries, last = nS)
  std::vector<int> w0, w1;               // packed separator columns [w0[j], w1[j]) adjacent to chunk j
  std::vector<long long> poff, hoff;     // offsets of P_j (n_j^2) and H_j (n_j * w_j) in their packed buffers
  long long ptot = 0, htot = 0;
  std::vector<int> sep_of;               // per generator: packed separator index or -1
  std::vector<int> chunk_of;             // per generator: chunk or -1
};

// structure analysis + partition (host, integers only)
inline MinvPlan plan_minv(const ScaledOperator& S, int target_chunk = 1024) {
  MinvPlan P;
  P.ng = S.ng;
  if (S.layer.size() != (size_t)S.ng || S.ng == 0) return P;
  P.nlayers = S.layer.back() + 1;
  P.lstart.assign(P.nlayers + 1, 0);
  for (int g = 0; g < S.ng; ++g) P.lstart[S.layer[g] + 1]++;
  for (int l = 0; l < P.nlayers; ++l) P.lstart[l + 1] += P.lstart[l];
  // rows: layer span of their nonzeros; rows spanning more than 2 layers apart are "global" (low-rank part)
  int bw = 0;
  for (int e = 0; e < S.NE; ++e) {
    const int lo = S.csr_ptr[e], hi = S.csr_ptr[e + 1];
    if (hi - lo < 2) continue;
    int mn = 1 << 30, mx = -1;
    for (int k = lo; k < hi; ++k) { const int l = S.layer[S.csr_col[k]]; mn = std::min(mn, l); mx = std::max(mx, l); }
    if (mx - mn > 6) P.grow.push_back(e);       // (groups: two per network layer, see ProblemCopy::generator_layers)
    else bw = std::max(bw, mx - mn);
  }
  P.r = (int)P.grow.size();
  if (P.r > 8) return P;                       // not the structure this scheme is for
  P.bw = std::max(bw, 1);
  // chunks of whole layers with about target_chunk generators, separated by bw layers
  std::vector<int> piece_lo, piece_hi, piece_is_sep;   // in layers
  int l = 0;
  while (l < P.nlayers) {
    int l1 = l, cnt = 0;
    while (l1 < P.nlayers && (cnt == 0 || cnt + (P.lstart[l1 + 1] - P.lstart[l1]) <= target_chunk)) { cnt += P.lstart[l1 + 1] - P.lstart[l1]; ++l1; }
    // a chunk ends on a whole-layer group (even index), so that the separator behind it is {straddling, layer, straddling}
    if (l1 < P.nlayers && (l1 & 1) == 0 && l1 - 1 > l) --l1;
    piece_lo.push_back(l); piece_hi.push_back(l1); piece_is_sep.push_back(0);
    l = l1;
    if (l < P.nlayers) {
      const int s1 = std::min(l + P.bw, P.nlayers);
      if (s1 == P.nlayers) { piece_hi.back() = P.nlayers; break; }   // no room for a chunk behind the separator: absorb
      piece_lo.push_back(l); piece_hi.push_back(s1); piece_is_sep.push_back(1);
      l = s1;
    }
  }
  for (size_t i = 0; i < piece_lo.size(); ++i) {
    const int g0 = P.lstart[piece_lo[i]], g1 = P.lstart[piece_hi[i]];
    if (piece_is_sep[i]) { P.slo.push_back(g0); P.shi.push_back(g1); }
    else { P.clo.push_back(g0); P.chi.push_back(g1); }
  }
  P.nchunk = (int)P.clo.size();
  if (P.nchunk < 2 || (int)P.slo.size() != P.nchunk - 1) return P;   // too few layers: the dense inverse is the right tool
  P.soff.assign(P.nchunk, 0);
  for (int j = 0; j + 1 < P.nchunk; ++j) P.soff[j + 1] = P.soff[j] + (P.shi[j] - P.slo[j]);
  P.nS = P.soff[P.nchunk - 1];
  P.sep_of.assign(S.ng, -1);
  P.chunk_of.assign(S.ng, -1);
  for (int j = 0; j < P.nchunk; ++j)
    for (int g = P.clo[j]; g < P.chi[j]; ++g) P.chunk_of[g] = j;
  for (int j = 0; j + 1 < P.nchunk; ++j)
    for (int g = P.slo[j]; g < P.shi[j]; ++g) P.sep_of[g] = P.soff[j] + (g - P.slo[j]);
  P.w0.resize(P.nchunk); P.w1.resize(P.nchunk); P.poff.resize(P.nchunk); P.hoff.resize(P.nchunk);
  for (int j = 0; j < P.nchunk; ++j) {
    P.w0[j] = j > 0 ? P.soff[j - 1] : 0;
    P.w1[j] = j + 1 < P.nchunk ? P.soff[j + 1] : P.nS;
    if (j == 0) P.w0[j] = 0;
    const long long n = P.chi[j] - P.clo[j], w = P.w1[j] - P.w0[j];
    const long long ldn = (n + 1) & ~1LL, ldw = (w + 1) & ~1LL;     // even leading dimensions: 16-byte aligned columns / rows
    P.poff[j] = P.ptot; P.ptot += ldn * n;
    P.hoff[j] = P.htot; P.htot += std::max(ldn * w, ldw * n);       // one offset serves H_j (ldn x w) and its row-major copy (n x ldw)
  }
  P.ldS = (P.nS + 1) & ~1;
  // chunks must not couple with each other, nor with a separator that is not adjacent
  std::vector<char> glob(S.NE, 0);
  for (int e : P.grow) glob[e] = 1;
  for (int e = 0; e < S.NE; ++e) {
    if (glob[e]) continue;
    const int lo = S.csr_ptr[e], hi = S.csr_ptr[e + 1];
    int cj = -1, smin = 1 << 30, smax = -1;
    for (int k = lo; k < hi; ++k) {
      const int g = S.csr_col[k];
      if (P.chunk_of[g] >= 0) { if (cj >= 0 && cj != P.chunk_of[g]) return P; cj = P.chunk_of[g]; }
      else { smin = std::min(smin, P.sep_of[g]); smax = std::max(smax, P.sep_of[g]); }
    }
    if (cj >= 0 && smax >= 0 && (smin < P.w0[cj] || smax >= P.w1[cj])) return P;
  }
  P.ok = true;
  return P;
}

// dense pieces of T (column-major): T[I_j,I_j], T[I_j, S(w0..w1)], T[S,S]; T = I + sum over NON-global rows of Dinv a a'
struct MinvBlocks {
  std::vector<double> Tjj, Tjs, Tss, U, dU;   // U: ng x r (column-major), dU: r weights Dinv[e_r]
};
inline MinvBlocks assemble_minv_blocks(const ScaledOperator& S, const MinvPlan& P) {
  MinvBlocks B;
  B.Tjj.assign((size_t)P.ptot, 0.0);
  B.Tjs.assign((size_t)P.htot, 0.0);
  B.Tss.assign((size_t)P.ldS * std::max(P.nS, 1), 0.0);
  B.U.assign((size_t)P.ng * std::max(P.r, 1), 0.0);
  B.dU.assign(std::max(P.r, 1), 0.0);
  std::vector<char> glob(S.NE, 0);
  for (int i = 0; i < P.r; ++i) {
    const int e = P.grow[i];
    glob[e] = 1;
    B.dU[i] = S.Dinv[e];
    for (int k = S.csr_ptr[e]; k < S.csr_ptr[e + 1]; ++k) B.U[(size_t)i * P.ng + S.csr_col[k]] = S.csr_val[k];
  }
  auto add = [&](int gx, int gy, double v) {     // T[gx, gy] += v (both triangles are filled by symmetric calls)
    const int cx = P.chunk_of[gx], cy = P.chunk_of[gy];
    if (cx >= 0 && cy >= 0) {
      const long long ldn = (P.chi[cx] - P.clo[cx] + 1) & ~1LL;
      B.Tjj[P.poff[cx] + (long long)(gy - P.clo[cx]) * ldn + (gx - P.clo[cx])] += v;
    } else if (cx >= 0) {          // row in chunk, column in separator
      const long long ldn = (P.chi[cx] - P.clo[cx] + 1) & ~1LL;
      B.Tjs[P.hoff[cx] + (long long)(P.sep_of[gy] - P.w0[cx]) * ldn + (gx - P.clo[cx])] += v;
    } else if (cy < 0) {
      B.Tss[(size_t)P.sep_of[gy] * P.ldS + P.sep_of[gx]] += v;
    }                              // (separator row, chunk column) is the transpose of the case above: not stored
  };
  for (int e = 0; e < S.NE; ++e) {
    if (glob[e]) continue;
    const int lo = S.csr_ptr[e], hi = S.csr_ptr[e + 1];
    const double d = S.Dinv[e];
    for (int x = lo; x < hi; ++x) {
      const double vx = d * S.csr_val[x];
      for (int y = lo; y < hi; ++y) add(S.csr_col[x], S.csr_col[y], vx * S.csr_val[y]);
    }
  }
  for (int g = 0; g < P.ng; ++g) add(g, g, 1.0);
  return B;
}

// ---------------------------------------------------------------------------------------------- device side
struct MinvDev {
  int ng, nchunk, nS, ldS, r, nslots;
  const int *clo, *chi, *w0, *w1, *hslot0;   // per chunk
  const long long *poff, *hoff;              // per chunk
  const int* chunk_of;                       // per generator (-1: separator)
  const int* sep_of;                         // per generator (-1: chunk)
  const int* sep_gen;                        // packed separator index -> generator
  const int* slot_chunk;                     // per H column slot: its chunk (slot = hslot0[j] + c)
  const int *slotA, *slotB;                  // per separator column: its slots in the left / right chunk
  const double *Pinv, *H, *HT, *Scinv, *v, *kap;   // v: ng x r, kap: r x r
  double *t, *rpart, *rvec, *xS, *coef;      // work: t[ng], rpart[nslots], rvec[ldS], xS[nS], coef[8]
  double* dpart;                             // [8][kMinvParts] partial sums of v_a'q (stage 1), added in slot order (stage 2a)
};
static constexpr int kMinvParts = 32;

// dot product of two 16-byte aligned vectors of length n by one wave (the pattern of k_gemv_sym: 16-byte loads, two chains)
__device__ __forceinline__ double wave_dot2(const double* __restrict__ a, const double* __restrict__ b, int n, int lane) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = lane * 2;
  for (; j + 129 < n; j += 256) {
    const double2 a0 = *reinterpret_cast<const double2*>(a + j), b0 = *reinterpret_cast<const double2*>(b + j);
    const double2 a1 = *reinterpret_cast<const double2*>(a + j + 128), b1 = *reinterpret_cast<const double2*>(b + j + 128);
    s0 += a0.x * b0.x; s1 += a0.y * b0.y; s2 += a1.x * b1.x; s3 += a1.y * b1.y;
  }
  for (; j + 1 < n; j += 128) {
    const double2 a0 = *reinterpret_cast<const double2*>(a + j), b0 = *reinterpret_cast<const double2*>(b + j);
    s0 += a0.x * b0.x; s1 += a0.y * b0.y;
  }
  if (j < n) s0 += a[j] * b[j];
  return wave_sum((s0 + s1) + (s2 + s3));
}

// stage 1: t = P_j q_Ij (one wave per chunk row), rpart[slot] = H_j[:, c]' q_Ij (one wave per slot).  q_Ij starts at an even
// generator index only by luck, so the chunk's slice of q is read through an aligned copy-free path when clo[j] is even and
// through scalar loads otherwise.
__global__ __launch_bounds__(kThreads) void k_minv_stage1(MinvDev m, const double* __restrict__ q) {
  const int wid = (int)(((long long)blockIdx.x * kThreads + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  int j, n, lo;
  const double* col;
  double* dst;
  if (wid < m.ng) {
    j = m.chunk_of[wid];
    if (j < 0) return;
    lo = m.clo[j]; n = m.chi[j] - lo;
    col = m.Pinv + m.poff[j] + (size_t)(wid - lo) * ((n + 1) & ~1);      // symmetric: column = row
    dst = m.t + wid;
  } else {
    const int slot = wid - m.ng;
    if (slot >= m.nslots) {
      // kMinvParts extra waves: partial sums of the low-rank dots v_a'q over a slice of the multipliers
      const int part = slot - m.nslots;
      if (part >= kMinvParts) return;
      const int per = (m.ng + kMinvParts - 1) / kMinvParts, i0 = part * per, i1 = min(i0 + per, m.ng);
      for (int a = 0; a < m.r; ++a) {
        double s = 0.0;
        for (int i = i0 + lane; i < i1; i += 64) s += m.v[(size_t)a * m.ng + i] * q[i];
        s = wave_sum(s);
        if (lane == 0) m.dpart[a * kMinvParts + part] = s;
      }
      return;
    }
    j = m.slot_chunk[slot];
    lo = m.clo[j]; n = m.chi[j] - lo;
    col = m.H + m.hoff[j] + (size_t)(slot - m.hslot0[j]) * ((n + 1) & ~1);
    dst = m.rpart + slot;
  }
  const double* qq = q + lo;
  double s;
  if ((lo & 1) == 0) s = wave_dot2(col, qq, n, lane);
  else {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = lane;
    for (; i + 192 < n; i += 256) { s0 += col[i] * qq[i]; s1 += col[i + 64] * qq[i + 64]; s2 += col[i + 128] * qq[i + 128]; s3 += col[i + 192] * qq[i + 192]; }
    for (; i < n; i += 64) s0 += col[i] * qq[i];
    s = wave_sum((s0 + s1) + (s2 + s3));
  }
  if (lane == 0) *dst = s;
}

// stage 2a: r = q_S - the two adjacent chunks' contributions (one thread per separator entry); the LAST block adds the partial
// low-rank dots of stage 1 in slot order and applies kap: coef = kap (v'q) (deterministic)
__global__ __launch_bounds__(kThreads) void k_minv_resid(MinvDev m, const double* __restrict__ q) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x == 0) {
      double d[8];
      for (int a = 0; a < m.r; ++a) { double s = 0.0; for (int p = 0; p < kMinvParts; ++p) s += m.dpart[a * kMinvParts + p]; d[a] = s; }
      for (int a = 0; a < m.r; ++a) { double s = 0.0; for (int b = 0; b < m.r; ++b) s += m.kap[a * m.r + b] * d[b]; m.coef[a] = s; }
    }
    return;
  }
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c < m.nS) m.rvec[c] = q[m.sep_gen[c]] - m.rpart[m.slotA[c]] - m.rpart[m.slotB[c]];
}

// stage 2b: xS = Sc^-1 r (symmetric, even leading dimension: one wave per row)
__global__ __launch_bounds__(kThreads) void k_minv_schur(MinvDev m) {
  const int srow = (int)(((long long)blockIdx.x * kThreads + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (srow >= m.nS) return;
  const double s = wave_dot2(m.Scinv + (size_t)srow * m.ldS, m.rvec, m.nS, lane);
  if (lane == 0) m.xS[srow] = s;
}

// stage 3: x_Ij = t - H_j x_S (row-major copy of H_j, rows padded to an even length: one wave per row), x_S as is; minus the
// low-rank term
__global__ __launch_bounds__(kThreads) void k_minv_stage3(MinvDev m, double* __restrict__ out) {
  const int g = (int)(((long long)blockIdx.x * kThreads + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (g >= m.ng) return;
  const int j = m.chunk_of[g];
  double x;
  if (j >= 0) {
    const int lo = m.clo[j], wj = m.w1[j] - m.w0[j], ldw = (wj + 1) & ~1;
    const double* row = m.HT + m.hoff[j] + (size_t)(g - lo) * ldw;
    const double* xs = m.xS + m.w0[j];
    double s;
    if ((m.w0[j] & 1) == 0) s = wave_dot2(row, xs, wj, lane);
    else { double s0 = 0.0; for (int c = lane; c < wj; c += 64) s0 += row[c] * xs[c]; s = wave_sum(s0); }
    x = m.t[g] - s;
  } else {
    x = m.xS[m.sep_of[g]];
  }
  if (lane == 0) {
    for (int a = 0; a < m.r; ++a) x -= m.v[(size_t)a * m.ng + g] * m.coef[a];
    out[g] = x;
  }
}

// row-major copy (rows padded to ldw) of a column-major n x w block with leading dimension ldn
__global__ void k_minv_transpose(int n, int w, int ldn, int ldw, const double* __restrict__ H, double* __restrict__ HT) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * w) return;
  const int i = (int)(idx % n), c = (int)(idx / n);
  HT[(size_t)i * ldw + c] = H[(size_t)c * ldn + i];
}

}  // namespace nnsdp
