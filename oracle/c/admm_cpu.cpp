// CPU port of the chordal ADMM iteration (oracle/admm.py AdmmState.step, statement by statement) in C++ / OpenMP with
// LAPACK dsyevd per clique - the "same-box CPU restatement" SURVEY.md section 8d(2) asks for beside every GPU number.
// TEST / MEASUREMENT INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg and tests/ call it; the product never does.
//
// The reference's own CPU path (JuMP -> MOSEK, src/Methods/Methods.jl:61,64,83) cannot run here (no Julia, no MOSEK);
// this is the first-order method the HIP library runs, on the host cores: one OpenMP thread per clique, LAPACK from the
// OpenBLAS that ships inside the image's scipy wheel (dlopen'ed: the image has no system LAPACK).
#include <dlfcn.h>
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
typedef void (*dsyevd_t)(const char*, const char*, const int*, double*, const int*, double*, double*, const int*, int*, const int*, int*);
typedef void (*dpotrs_t)(const char*, const int*, const int*, const double*, const int*, double*, const int*, int*);
typedef void (*setthr_t)(int);
dsyevd_t p_dsyevd = nullptr;
dpotrs_t p_dpotrs = nullptr;
setthr_t p_setthr = nullptr;

void* sym(void* h, const char* a, const char* b) {
  void* p = dlsym(h, a);
  return p ? p : dlsym(h, b);
}
}  // namespace

struct AdmmCpuProblem {
  int NE, ng, ncl;
  const int *csr_ptr, *csr_col;      // A (NE x ng), rows = pattern entries
  const double* csr_val;
  const int *csc_ptr, *csc_row;      // the same matrix by columns
  const double* csc_val;
  const double *z0, *c, *Dinv;
  const double* Mchol;               // M^-1, M = I + A' D^-1 A: symmetric ng x ng (explicit inverse, as on the GPU)
  const int* nk;                     // block dimensions
  const long long* off;              // ncl + 1 offsets of the blocks inside nu (first block at ng)
  const int* gidx;                   // per block element (column-major n x n): pattern entry
  const int *sptr;                   // NE + 1: lower-triangle sources of every pattern entry ...
  const long long* soff;             // ... as offsets into nu
  const unsigned char* isdiag;       // NE
};

extern "C" int admm_cpu_init(const char* lapack_path) {
  void* h = dlopen(lapack_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { std::fprintf(stderr, "admm_cpu: cannot dlopen %s: %s\n", lapack_path, dlerror()); return 1; }
  p_dsyevd = (dsyevd_t)sym(h, "scipy_dsyevd_", "dsyevd_");
  p_dpotrs = (dpotrs_t)sym(h, "scipy_dpotrs_", "dpotrs_");
  p_setthr = (setthr_t)sym(h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
  return (p_dsyevd && p_dpotrs) ? 0 : 2;
}

// `iters` iterations on the state nu (length off[ncl]); stats[0..4] = pres, dres, primal objective, dual objective (scaled
// problem), seconds.  Returns 0, or the LAPACK info of a failed eigendecomposition.
extern "C" int admm_cpu_run(const AdmmCpuProblem* P, double sigma, double alpha, int iters, int threads, double* nu, double* stats) {
  const int NE = P->NE, ng = P->ng, ncl = P->ncl;
  const long long N = P->off[ncl];
  const double kSqrt2 = std::sqrt(2.0), kInv = 1.0 / kSqrt2;
  if (threads > 0) omp_set_num_threads(threads);
  if (p_setthr) p_setthr(1);        // one LAPACK thread per clique; the cliques run side by side
  int nmax = 0;
  for (int k = 0; k < ncl; ++k) nmax = std::max(nmax, P->nk[k]);
  std::vector<double> w(N), g(NE), p(ng), ww(ng), qv(ng), x(NE), res(N), kxq(N);
  int bad = 0;
  auto t0 = std::chrono::steady_clock::now();
  double rp = 0, rd = 0, pobj = 0, dobj = 0;
  const int nthr = std::max(1, omp_get_max_threads());
  const int lwork = 1 + 6 * nmax + 2 * nmax * nmax, liwork = 3 + 5 * nmax;
  std::vector<double> wsA((size_t)nthr * nmax * nmax), wsV((size_t)nthr * nmax), wsW((size_t)nthr * lwork);
  std::vector<int> wsI((size_t)nthr * liwork);
  for (int it = 0; it < iters; ++it) {
    // w = proj_C(nu): multiplier block, then one eigendecomposition per clique
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ng; ++i) w[i] = nu[i] > 0.0 ? nu[i] : 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < ncl; ++k) {
      const int n = P->nk[k], t = omp_get_thread_num();
      double* A = &wsA[(size_t)t * nmax * nmax];
      double* ev = &wsV[(size_t)t * nmax];
      const double* V = nu + P->off[k];
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) A[(size_t)j * n + i] = 0.5 * (V[(size_t)j * n + i] + V[(size_t)i * n + j]);
      int info = 0, lw = lwork, liw = liwork;
      p_dsyevd("V", "L", &n, A, &n, ev, &wsW[(size_t)t * lwork], &lw, &wsI[(size_t)t * liwork], &liw, &info);
      if (info != 0) {
#pragma omp atomic write
        bad = info;
      }
      double* Wk = &w[P->off[k]];
      std::fill(Wk, Wk + (size_t)n * n, 0.0);
      for (int e = 0; e < n; ++e) {
        if (!(ev[e] > 0.0)) continue;
        const double* q = A + (size_t)e * n;
        for (int j = 0; j < n; ++j) {
          const double s = ev[e] * q[j];
          for (int i = 0; i < n; ++i) Wk[(size_t)j * n + i] += s * q[i];
        }
      }
    }
    if (bad) return bad;
    // g = Dinv (z0 / sigma + h),  h[e] = wgt * sum over sources of (2 w - nu)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < NE; ++e) {
      double s = 0.0;
      for (int q = P->sptr[e]; q < P->sptr[e + 1]; ++q) { const long long o = P->soff[q]; s += 2.0 * w[o] - nu[o]; }
      if (!P->isdiag[e]) s *= kSqrt2;
      g[e] = P->Dinv[e] * (P->z0[e] / sigma + s);
    }
    // p = 2 w_s - nu_s - c;  qv = A' g - p;  ww = M^-1 qv
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ng; ++j) {
      double s = 0.0;
      for (int q = P->csc_ptr[j]; q < P->csc_ptr[j + 1]; ++q) s += P->csc_val[q] * g[P->csc_row[q]];
      p[j] = 2.0 * w[j] - nu[j] - P->c[j];
      ww[j] = s - p[j];
    }
    {
      // ww = M^-1 qv with the explicit symmetric inverse (as the HIP library does): one row per thread chunk
      std::swap(ww, qv);
      const double* Mi = P->Mchol;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < ng; ++i) {
        const double* col = Mi + (size_t)i * ng;
        double s = 0.0;
        for (int j = 0; j < ng; ++j) s += col[j] * qv[j];
        ww[i] = s;
      }
    }
    // x = g - Dinv (A ww)
#pragma omp parallel for schedule(static)
    for (int e = 0; e < NE; ++e) {
      double s = 0.0;
      for (int q = P->csr_ptr[e]; q < P->csr_ptr[e + 1]; ++q) s += P->csr_val[q] * ww[P->csr_col[q]];
      x[e] = g[e] - P->Dinv[e] * s;
    }
    // K x + q, residual, nu update
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ng; ++j) { kxq[j] = p[j] + ww[j] + P->c[j]; res[j] = kxq[j] - w[j]; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < ncl; ++k) {
      const int n = P->nk[k];
      const int* gi = P->gidx + (P->off[k] - ng);
      for (long long m = 0; m < (long long)n * n; ++m) {
        const long long o = P->off[k] + m;
        const bool diag = (m / n) == (m % n);
        kxq[o] = x[gi[m]] * (diag ? 1.0 : kInv);
        res[o] = kxq[o] - w[o];
      }
    }
    if (it == iters - 1) {
      // residuals of this iteration, as oracle/admm.py admm_solve computes them (y = sigma (nu_prev - w))
      double r2 = 0, k2 = 0, w2 = 0;
      for (long long i = 0; i < N; ++i) { r2 += res[i] * res[i]; k2 += kxq[i] * kxq[i]; w2 += w[i] * w[i]; }
      rp = std::sqrt(r2) / std::max({std::sqrt(k2), std::sqrt(w2), 1e-300});
      double d2 = 0, t2 = 0, z2 = 0;
      for (int e = 0; e < NE; ++e) {
        double s = 0.0;
        for (int q = P->csr_ptr[e]; q < P->csr_ptr[e + 1]; ++q) { const int j = P->csr_col[q]; s += P->csr_val[q] * sigma * (nu[j] - w[j]); }
        double h = 0.0;
        for (int q = P->sptr[e]; q < P->sptr[e + 1]; ++q) { const long long o = P->soff[q]; h += sigma * (nu[o] - w[o]); }
        if (!P->isdiag[e]) h *= kSqrt2;
        const double t = s + h, dd = t - P->z0[e];
        d2 += dd * dd; t2 += t * t; z2 += P->z0[e] * P->z0[e];
        dobj += P->z0[e] * x[e];
      }
      rd = std::sqrt(d2) / std::max({std::sqrt(t2), std::sqrt(z2), 1e-300});
      for (int j = 0; j < ng; ++j) pobj -= P->c[j] * sigma * (nu[j] - w[j]);
    }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < N; ++i) nu[i] += alpha * res[i];
  }
  stats[0] = rp; stats[1] = rd; stats[2] = pobj; stats[3] = dobj;
  stats[4] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}
