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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnsdp {

static constexpr int kThreads = 256;
static constexpr double kSqrt2 = 1.41421356237309504880;
static constexpr double kInvSqrt2 = 0.70710678118654752440;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum, result valid in every thread; scratch >= 4 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wv] = v;
  __syncthreads();
  double s = 0.0;
  int nw = (blockDim.x + 63) >> 6;
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
  const double* kappa;    // device scalar: nu <- w + kappa (nu - w) (penalty change), may be null
  int* stats;             // [0] += sweeps used (atomic), [1] = max sweeps seen
  int warm;               // 1: use Vg as the starting basis
  int max_sweeps;
  double tol;             // relative off-diagonal tolerance
};

template <bool V_LDS>
__global__ __launch_bounds__(kThreads) void k_proj_jacobi(ProjArgs a) {
  extern __shared__ double lds[];
  const int k = blockIdx.x;
  const int n = a.cn[k];
  const int np = (n + 1) & ~1;   // even
  const int half = np >> 1;
  const int lda = np + 1;        // odd stride (in doubles): column walks hit distinct banks
  const int tid = threadIdx.x;
  double* A = lds;
  double* cs = A + (size_t)np * lda;  // 2 * half rotation parameters
  double* red = cs + np;              // 8 doubles of reduction scratch
  double* V;
  int ldv;
  if (V_LDS) { V = red + 8; ldv = np + 1; }
  else { V = a.Vg + a.coff[k]; ldv = n; }
  const double* nuk = a.nu + a.coff[k];

  // ---- load: A = sym(nu_k), padded row/col zero
  double fro2 = 0.0;
  for (int idx = tid; idx < np * np; idx += kThreads) {
    int i = idx % np, j = idx / np;
    double v = 0.0;
    if (i < n && j < n) v = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]);
    A[i * lda + j] = v;
    fro2 += v * v;
  }
  fro2 = block_sum(fro2, red);
  // ---- starting basis
  const bool warm = a.warm != 0;
  if (V_LDS) {
    for (int idx = tid; idx < np * np; idx += kThreads) {
      int i = idx % np, j = idx / np;
      double v = (i == j) ? 1.0 : 0.0;
      if (warm && i < n && j < n) v = a.Vg[a.coff[k] + (size_t)j * n + i];
      V[i + j * ldv] = v;
    }
  } else if (!warm) {
    for (int idx = tid; idx < n * n; idx += kThreads) {
      int i = idx % n, j = idx / n;
      V[i + (size_t)j * ldv] = (i == j) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  const int nv = V_LDS ? np : n;  // rows/cols of V that exist
  if (warm) {
    // A <- V' A V in two passes through registers (A is overwritten by T = A V, then by V' T).
    // Static trip counts keep acc[] in VGPRs (a runtime-indexed array would go to scratch).
    constexpr int kMaxOwn = (130 * 130 + kThreads - 1) / kThreads;
    double acc[kMaxOwn];
    const int tot = np * np;
#pragma unroll
    for (int m = 0; m < kMaxOwn; ++m) {
      int idx = tid + m * kThreads;
      double s = 0.0;
      if (idx < tot) {
        int i = idx % np, j = idx / np;
        if (j < nv)
          for (int l = 0; l < nv; ++l) s += A[i * lda + l] * V[l + (size_t)j * ldv];
        else
          s = A[i * lda + j];
      }
      acc[m] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kMaxOwn; ++m) {
      int idx = tid + m * kThreads;
      if (idx < tot) A[(idx % np) * lda + idx / np] = acc[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kMaxOwn; ++m) {
      int idx = tid + m * kThreads;
      double s = 0.0;
      if (idx < tot) {
        int i = idx % np, j = idx / np;
        if (i < nv)
          for (int l = 0; l < nv; ++l) s += V[l + (size_t)i * ldv] * A[l * lda + j];
        else
          s = A[i * lda + j];
      }
      acc[m] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kMaxOwn; ++m) {
      int idx = tid + m * kThreads;
      if (idx < tot) A[(idx % np) * lda + idx / np] = acc[m];
    }
    __syncthreads();
    // enforce exact symmetry (the two passes round differently)
    for (int idx = tid; idx < tot; idx += kThreads) {
      int i = idx % np, j = idx / np;
      if (i > j) A[i * lda + j] = 0.5 * (A[i * lda + j] + A[j * lda + i]);
    }
    __syncthreads();
    for (int idx = tid; idx < tot; idx += kThreads) {
      int i = idx % np, j = idx / np;
      if (i < j) A[i * lda + j] = A[j * lda + i];
    }
    __syncthreads();
  }

  // ---- Jacobi sweeps
  const double thresh2 = a.tol * fro2;  // stop after a sweep that STARTED below sqrt(tol): it ends near tol (quadratic)
  int sweeps = 0;
  for (; sweeps < a.max_sweeps; ++sweeps) {
    double off2 = 0.0;
    for (int r = 0; r < np - 1; ++r) {
      // (a) rotation parameters of this round's pairs
      if (tid < half) {
        int p, q;
        if (tid == 0) { p = np - 1; q = r; }
        else { p = (r + tid) % (np - 1); q = (r + np - 1 - tid) % (np - 1); }
        double app = A[p * lda + p], aqq = A[q * lda + q], apq = A[p * lda + q];
        double c = 1.0, s = 0.0;
        off2 += 2.0 * apq * apq;
        if (apq != 0.0) {
          double tau = (aqq - app) / (2.0 * apq);
          double t = 1.0 / (fabs(tau) + sqrt(1.0 + tau * tau));
          if (tau < 0.0) t = -t;
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
        }
        cs[2 * tid] = c;
        cs[2 * tid + 1] = s;
      }
      __syncthreads();
      // (b) A <- J' A J by 2x2 blocks; block (ia, ib) = rows {p,q}(ia) x cols {p,q}(ib)
      for (int blk = tid; blk < half * half; blk += kThreads) {
        int ia = blk / half, ib = blk % half;
        int p1, q1, p2, q2;
        if (ia == 0) { p1 = np - 1; q1 = r; } else { p1 = (r + ia) % (np - 1); q1 = (r + np - 1 - ia) % (np - 1); }
        if (ib == 0) { p2 = np - 1; q2 = r; } else { p2 = (r + ib) % (np - 1); q2 = (r + np - 1 - ib) % (np - 1); }
        double c1 = cs[2 * ia], s1 = cs[2 * ia + 1], c2 = cs[2 * ib], s2 = cs[2 * ib + 1];
        double b00 = A[p1 * lda + p2], b01 = A[p1 * lda + q2], b10 = A[q1 * lda + p2], b11 = A[q1 * lda + q2];
        double t00 = c1 * b00 - s1 * b10, t01 = c1 * b01 - s1 * b11;
        double t10 = s1 * b00 + c1 * b10, t11 = s1 * b01 + c1 * b11;
        b00 = t00 * c2 - t01 * s2;
        b01 = t00 * s2 + t01 * c2;
        b10 = t10 * c2 - t11 * s2;
        b11 = t10 * s2 + t11 * c2;
        if (ia == ib) { b01 = 0.0; b10 = 0.0; }
        A[p1 * lda + p2] = b00; A[p1 * lda + q2] = b01; A[q1 * lda + p2] = b10; A[q1 * lda + q2] = b11;
      }
      // (c) V <- V J by (row, pair) items
      for (int it = tid; it < nv * half; it += kThreads) {
        int row = it % nv, ia = it / nv;
        int p, q;
        if (ia == 0) { p = np - 1; q = r; } else { p = (r + ia) % (np - 1); q = (r + np - 1 - ia) % (np - 1); }
        if (p >= nv || q >= nv) continue;  // padded index: rotation is the identity
        double c = cs[2 * ia], s = cs[2 * ia + 1];
        double vp = V[row + (size_t)p * ldv], vq = V[row + (size_t)q * ldv];
        V[row + (size_t)p * ldv] = c * vp - s * vq;
        V[row + (size_t)q * ldv] = s * vp + c * vq;
      }
      __syncthreads();
    }
    off2 = block_sum(off2, red);
    if (off2 <= thresh2) { ++sweeps; break; }
  }
  if (tid == 0 && a.stats) { atomicAdd(&a.stats[0], sweeps); atomicMax(&a.stats[1], sweeps); }

  // ---- eigenvalues on the diagonal; pick the smaller side of the spectrum for the rank-k update
  int npos = 0, nneg = 0;
  for (int i = 0; i < n; ++i) { double l = A[i * lda + i]; npos += (l > 0.0); nneg += (l < 0.0); }
  const bool use_pos = npos <= nneg;
  if (a.eig && tid < n) a.eig[a.eoff[k] + tid] = A[tid * lda + tid];
  const double kap = a.kappa ? *a.kappa : 1.0;
  double* wk = a.w + a.coff[k];
  double* nuw = a.nu + a.coff[k];
  for (int idx = tid; idx < n * n; idx += kThreads) {
    int i = idx % n, j = idx / n;
    double s = 0.0;
    if (use_pos) {
      for (int l = 0; l < n; ++l) { double lam = A[l * lda + l]; if (lam > 0.0) s += lam * V[i + (size_t)l * ldv] * V[j + (size_t)l * ldv]; }
    } else {
      for (int l = 0; l < n; ++l) { double lam = A[l * lda + l]; if (lam < 0.0) s += lam * V[i + (size_t)l * ldv] * V[j + (size_t)l * ldv]; }
      s = 0.5 * (nuk[(size_t)j * n + i] + nuk[(size_t)i * n + j]) - s;
    }
    wk[idx] = s;
  }
  if (kap != 1.0) {
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kThreads) { double wv = wk[idx]; nuw[idx] = wv + kap * (nuw[idx] - wv); }
  }
  if (V_LDS) {
    double* vg = a.Vg + a.coff[k];
    for (int idx = tid; idx < n * n; idx += kThreads) vg[idx] = V[(idx % n) + (idx / n) * ldv];
  }
}

inline size_t proj_lds_bytes(int nmax, bool v_lds) {
  int np = (nmax + 1) & ~1;
  size_t d = (size_t)np * (np + 1) + np + 8;
  if (v_lds) d += (size_t)np * (np + 1);
  return d * sizeof(double);
}

// ---------------------------------------------------------------------------------------------
// multiplier block: w_s = max(nu_s, 0), reflection p = 2 w_s - nu_s - c   (also applies kappa)
// fused into the A' product: qv[g] = sum_e A[e,g] gvec[e] - p[g]; one wave per multiplier.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_spmv_At(int ng, const int* __restrict__ ptr, const int* __restrict__ row,
                                                       const double* __restrict__ val, const double* __restrict__ gvec,
                                                       double* __restrict__ nus, const double* __restrict__ c,
                                                       const double* __restrict__ kappa, double* __restrict__ p,
                                                       double* __restrict__ qv) {
  int g = (blockIdx.x * kThreads + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (g >= ng) return;
  double s = 0.0;
  for (int q = ptr[g] + lane; q < ptr[g + 1]; q += 64) s += val[q] * gvec[row[q]];
  s = wave_sum(s);
  if (lane == 0) {
    double v = nus[g], wv = v > 0.0 ? v : 0.0;
    double kap = kappa ? *kappa : 1.0;
    if (kap != 1.0) { v = wv + kap * (v - wv); nus[g] = v; }
    double pp = 2.0 * wv - v - c[g];
    p[g] = pp;
    qv[g] = s - pp;
  }
}

// g[e] = Dinv[e] (z0[e] / sigma + wgt * sum_src (2 w - nu)[src]); one thread per pattern entry
__global__ __launch_bounds__(kThreads) void k_gather_g(int NE, const int* __restrict__ sptr, const long long* __restrict__ soff,
                                                        const unsigned char* __restrict__ isdiag,
                                                        const double* __restrict__ nuk, const double* __restrict__ wk,
                                                        const double* __restrict__ z0, const double* __restrict__ Dinv,
                                                        const double* __restrict__ sigma, double* __restrict__ g) {
  int e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= NE) return;
  double s = 0.0;
  for (int q = sptr[e]; q < sptr[e + 1]; ++q) { long long o = soff[q]; s += 2.0 * wk[o] - nuk[o]; }
  if (!isdiag[e]) s *= kSqrt2;
  g[e] = Dinv[e] * (z0[e] / (*sigma) + s);
}

// ww = Minv qv  (Minv symmetric, column-major): one wave per output row, coalesced column reads
__global__ __launch_bounds__(kThreads) void k_gemv_sym(int n, int ldm, const double* __restrict__ Minv, const double* __restrict__ x,
                                                        double* __restrict__ y) {
  int i = (blockIdx.x * kThreads + threadIdx.x) >> 6;
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

// x[e] = g[e] - Dinv[e] * sum_g A[e,g] ww[g]; 16 lanes per pattern entry
__global__ __launch_bounds__(kThreads) void k_spmv_A_x(int NE, const int* __restrict__ ptr, const int* __restrict__ col,
                                                        const double* __restrict__ val, const double* __restrict__ ww,
                                                        const double* __restrict__ g, const double* __restrict__ Dinv,
                                                        double* __restrict__ x) {
  int e = (blockIdx.x * kThreads + threadIdx.x) >> 4;
  int sub = threadIdx.x & 15;
  double s = 0.0;
  if (e < NE)
    for (int q = ptr[e] + sub; q < ptr[e + 1]; q += 16) s += val[q] * ww[col[q]];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 16);
  if (e < NE && sub == 0) x[e] = g[e] - Dinv[e] * s;
}

// nu <- nu + alpha (K x + q - w).  acc (may be null) accumulates at check iterations:
//   acc[0] += |Kx+q-w|^2, acc[1] += |Kx+q|^2, acc[2] += |w|^2
__global__ __launch_bounds__(kThreads) void k_update_nu(int ng, long long nmat, const double* __restrict__ p,
                                                         const double* __restrict__ ww, const double* __restrict__ c,
                                                         const double* __restrict__ x, const unsigned int* __restrict__ gidx,
                                                         double* __restrict__ nu, const double* __restrict__ w,
                                                         double alpha, double* __restrict__ kappa, double* __restrict__ acc) {
  __shared__ double red[8];
  long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  double r2 = 0.0, k2 = 0.0, w2 = 0.0;
  if (i < ng) {
    double v = nu[i], wv = v > 0.0 ? v : 0.0;
    double kxq = p[i] + ww[i] + c[i];
    double res = kxq - wv;
    nu[i] = v + alpha * res;
    r2 = res * res; k2 = kxq * kxq; w2 = wv * wv;
  } else if (i < ng + nmat) {
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
    if (threadIdx.x == 0) { atomicAdd(&acc[0], r2); atomicAdd(&acc[1], k2); atomicAdd(&acc[2], w2); }
  }
  if (i == 0 && kappa) *kappa = 1.0;
}

// check iteration, dual side: t[e] = (K'y)[e] = sum_g A[e,g] ys[g] + wgt * sum_src y_k[src],
// y = sigma (nu - w).  acc[3] += |t - z0|^2, acc[4] += |t|^2; 16 lanes per entry.
__global__ __launch_bounds__(kThreads) void k_check_dual(int NE, int ng, const int* __restrict__ ptr, const int* __restrict__ col,
                                                          const double* __restrict__ val, const int* __restrict__ sptr,
                                                          const long long* __restrict__ soff, const unsigned char* __restrict__ isdiag,
                                                          const double* __restrict__ nu, const double* __restrict__ w,
                                                          const double* __restrict__ z0, const double* __restrict__ sigma,
                                                          double* __restrict__ acc) {
  __shared__ double red[8];
  int e = (blockIdx.x * kThreads + threadIdx.x) >> 4;
  int sub = threadIdx.x & 15;
  double sg = *sigma;
  double s = 0.0, h = 0.0;
  if (e < NE) {
    for (int q = ptr[e] + sub; q < ptr[e + 1]; q += 16) { double v = nu[col[q]]; s += val[q] * (v < 0.0 ? v : 0.0); }
    for (int q = sptr[e] + sub; q < sptr[e + 1]; q += 16) { long long o = soff[q]; h += nu[ng + o] - w[ng + o]; }
    if (!isdiag[e]) h *= kSqrt2;
  }
  s += h;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 16);
  double d2 = 0.0, t2 = 0.0;
  if (e < NE && sub == 0) { double t = sg * s; double d = t - z0[e]; d2 = d * d; t2 = t * t; }
  d2 = block_sum(d2, red);
  t2 = block_sum(t2, red + 4);
  if (threadIdx.x == 0) { atomicAdd(&acc[3], d2); atomicAdd(&acc[4], t2); }
}

// acc[5] += -c' ys  (scaled primal objective), acc[6] += z0' x (scaled dual objective)
__global__ __launch_bounds__(kThreads) void k_check_obj(int ng, int NE, const double* __restrict__ nu, const double* __restrict__ c,
                                                         const double* __restrict__ z0, const double* __restrict__ x,
                                                         const double* __restrict__ sigma, double* __restrict__ acc) {
  __shared__ double red[8];
  int i = blockIdx.x * kThreads + threadIdx.x;
  double a = 0.0, b = 0.0;
  if (i < ng) { double v = nu[i]; a = -c[i] * (*sigma) * (v < 0.0 ? v : 0.0); }
  if (i < NE) b = z0[i] * x[i];
  a = block_sum(a, red);
  b = block_sum(b, red + 4);
  if (threadIdx.x == 0) { atomicAdd(&acc[5], a); atomicAdd(&acc[6], b); }
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
