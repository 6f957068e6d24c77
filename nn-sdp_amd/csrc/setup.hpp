// Host-side problem setup: clique index sets, clique sparsity pattern, generator table of the
// affine LMI Z(gamma) = Z0 + sum_i gamma_i G_i in scaled-svec coordinates, solver normalisation.
// Integer / index work only; all floating-point hot loops live in kernels.hip.
//
// Reference behaviour restated here (numeric gamma instead of JuMP variables):
//   makeCliques            src/Methods/chordal_cliques.jl:13-59
//   setupZs! index sets    src/Methods/chordal_sdp.jl:19-57
//   makeZin                src/Qc/input.jl:19-42
//   makeSide / makeZout    src/Qc/output.jl:34-106
//   makeA/makeb/makeB/makeZac   src/Qc/activ.jl:7-42
//   makeQ bounded / sector src/Qc/activ_bounded.jl:13-24, src/Qc/activ_sector.jl:23-60
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nnsdp.h"

namespace nnsdp {

struct SpVec {
  std::vector<int> idx;
  std::vector<double> val;
};

// deep copy of the caller's problem (the ABI says: copy in, retain nothing of the caller's)
struct ProblemCopy {
  int K = 0, beta = 0, query_kind = 0, out_kind = 0, activ = 0;
  std::vector<int> xdims;
  std::vector<std::vector<double>> W;  // W[k] row-major n_{k+1} x n_k
  std::vector<std::vector<double>> b;  // b[k]
  std::vector<double> x1min, x1max, acymin, acymax, smin, smax, normal, yc, invP, S;
  int Zdim = 0, acdim = 0, nin = 0, nout = 0, n1 = 0, npairs = 0, n2 = 0, ng = 0, mout = 0;
  std::vector<int> offs;  // z-offset of block k (size K+1, offs[K] = a = Zdim-1)
  std::vector<int> neuron_layer;   // hidden layer (0-based) of every neuron t

  // ordering group of every multiplier: 2k for the multipliers of hidden layer k (their generators touch the blocks of x_k,
  // x_{k+1} and the affine index only), 2k - 1 for the few repeated-nonlinearity pairs that straddle the boundary between
  // layers k-1 and k (they touch x_{k-1}, x_k, x_{k+1}): kept apart so that a separator of the structured M^-1 (minv.hpp) is
  // one layer plus its two small straddling groups instead of two whole layers
  std::vector<int> generator_layers() const {
    std::vector<int> L(ng, 0);
    const int o1 = nin + nout, ol = o1 + n1, ov = ol + acdim, oe = ov + npairs, on = oe + acdim;
    for (int t = 0; t < acdim; ++t) {
      L[o1 + t] = L[ol + t] = 2 * neuron_layer[t];
      if (activ == NNSDP_ACTIV_RELU) L[oe + t] = L[on + t] = 2 * neuron_layer[t];
    }
    int r = 0;
    for (int i = 0; i < acdim - 1; ++i)
      for (int j = i + 1; j < acdim && j - i <= beta; ++j, ++r)
        L[ov + r] = neuron_layer[i] == neuron_layer[j] ? 2 * neuron_layer[j] : 2 * neuron_layer[j] - 1;
    return L;
  }

  void load(const nnsdp_problem* p) {
    if (!p || !p->xdims || !p->M) throw std::invalid_argument("null problem / xdims / M");
    K = p->K;
    if (K < 2) throw std::invalid_argument("K must be >= 2 (length(xdims) >= 3)");
    xdims.assign(p->xdims, p->xdims + K + 1);
    for (int v : xdims)
      if (v < 1) throw std::invalid_argument("xdims entries must be >= 1");
    beta = p->beta;
    if (beta < 0) throw std::invalid_argument("beta must be >= 0");
    query_kind = p->query_kind;
    out_kind = p->out_kind;
    activ = p->activ;
    if (activ != NNSDP_ACTIV_RELU && activ != NNSDP_ACTIV_TANH) throw std::invalid_argument("unsupported activation (activ_sector.jl:88)");
    offs.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) offs[k + 1] = offs[k] + xdims[k];
    Zdim = offs[K] + 1;
    acdim = 0;
    for (int k = 1; k < K; ++k) acdim += xdims[k];
    if (beta > acdim) throw std::invalid_argument("beta exceeds acdim");
    mout = xdims[K];
    const double* Mp = p->M;
    W.resize(K);
    b.resize(K);
    for (int k = 0; k < K; ++k) {
      int r = xdims[k + 1], c = xdims[k];
      W[k].assign((size_t)r * c, 0.0);
      b[k].assign(r, 0.0);
      for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) W[k][(size_t)i * c + j] = Mp[(size_t)j * r + i];
      for (int i = 0; i < r; ++i) b[k][i] = Mp[(size_t)c * r + i];
      Mp += (size_t)r * (c + 1);
    }
    auto need = [](const double* q, const char* name) {
      if (!q) throw std::invalid_argument(std::string("null pointer: ") + name);
    };
    need(p->x1min, "x1min"); need(p->x1max, "x1max");
    need(p->acymin, "acymin"); need(p->acymax, "acymax"); need(p->smin, "smin"); need(p->smax, "smax");
    x1min.assign(p->x1min, p->x1min + xdims[0]);
    x1max.assign(p->x1max, p->x1max + xdims[0]);
    acymin.assign(p->acymin, p->acymin + acdim);
    acymax.assign(p->acymax, p->acymax + acdim);
    smin.assign(p->smin, p->smin + acdim);
    smax.assign(p->smax, p->smax + acdim);
    for (int i = 0; i < xdims[0]; ++i)
      if (!(x1min[i] <= x1max[i])) throw std::invalid_argument("x1min <= x1max violated");
    for (int i = 0; i < acdim; ++i)
      if (!(acymin[i] <= acymax[i])) throw std::invalid_argument("acymin <= acymax violated (activ_bounded.jl:8)");
    // `@assert smin <= smax` (activ_sector.jl:13) compares two Julia vectors, i.e. lexicographically: the first entry that
    // differs decides.  (The tanh branch of makeSectorMinMax returns smin > smax on negative intervals, :76-77; the sector
    // generator is symmetric in the two slopes, so that is harmless, and it passes the reference's assertion.)
    for (int i = 0; i < acdim; ++i)     // every entry: the lexicographic loop below stops at the first differing one
      if (smin[i] != smin[i] || smax[i] != smax[i]) throw std::invalid_argument("smin / smax contain NaN");
    for (int i = 0; i < acdim; ++i) {
      if (smin[i] == smax[i]) continue;
      if (smin[i] > smax[i]) throw std::invalid_argument("smin <= smax violated (activ_sector.jl:13)");
      break;
    }
    nin = xdims[0];
    if (query_kind == NNSDP_QUERY_REACH) {
      nout = 1;
      if (out_kind == NNSDP_OUT_HPLANE) { need(p->normal, "normal"); normal.assign(p->normal, p->normal + mout); }
      else if (out_kind == NNSDP_OUT_CIRCLE) { need(p->yc, "yc"); yc.assign(p->yc, p->yc + mout); }
      else if (out_kind == NNSDP_OUT_ELLIPSOID) {
        need(p->yc, "yc"); need(p->invP, "invP");
        yc.assign(p->yc, p->yc + mout);
        invP.assign(p->invP, p->invP + (size_t)mout * mout);
      } else throw std::invalid_argument("reach query needs out_kind hplane/circle/ellipsoid");
    } else if (query_kind == NNSDP_QUERY_SAFETY) {
      nout = 0;
      if (out_kind != NNSDP_OUT_SAFETY_S) throw std::invalid_argument("safety query needs out_kind SAFETY_S");
      need(p->S, "S");
      int sd = xdims[0] + mout + 1;
      S.assign(p->S, p->S + (size_t)sd * sd);
    } else throw std::invalid_argument("unrecognized query_kind");
    n1 = acdim;
    neuron_layer.clear();
    for (int k = 1; k < K; ++k)
      for (int i = 0; i < xdims[k]; ++i) neuron_layer.push_back(k - 1);
    // lambda_dim = sum((acdim-beta):acdim) (activ_sector.jl:18) = acdim + #pairs
    npairs = beta * acdim - beta * (beta + 1) / 2;
    n2 = acdim + npairs + (activ == NNSDP_ACTIV_RELU ? 2 * acdim : 0);   // vardim (activ_sector.jl:19)
    ng = nin + nout + n1 + n2;
  }
};

// makeCliques + setupZs! index sets; 0-based z-indices
inline std::vector<std::vector<int>> clique_index_sets(int K, const int* xdims, int beta, int mode) {
  std::vector<int> S(K + 2, 0);  // S[k] = sum(xdims[0..k-1])
  for (int k = 1; k <= K + 1; ++k) S[k] = S[k - 1] + xdims[k - 1];
  int Zdim = S[K] + 1;
  std::vector<std::vector<int>> out;
  if (mode == NNSDP_DECOMP_DENSE) {
    std::vector<int> all(Zdim);
    std::iota(all.begin(), all.end(), 0);
    out.push_back(all);
    return out;
  }
  if (mode == NNSDP_DECOMP_PATH) {
    // extension (not in the reference): without an x_1 -- x_K coupling (S12 = 0: every reach query and hyperplane
    // safety sets) the sparsity graph of Z is the path x_1 - x_2 - ... - x_K plus the affine index, which is
    // chordal as it stands: cliques {x_k, x_{k+1} (+ beta spill), a} of size ~2W+1 instead of 3W+1.  The
    // generator builder rejects the mode when an entry falls outside these blocks.
    for (int k = 1; k <= K - 1; ++k) {
      std::vector<int> Ck;
      int hi = std::min(S[k + 1] + beta, S[K]);
      for (int i = S[k - 1]; i < hi; ++i) Ck.push_back(i);
      Ck.push_back(S[K]);
      out.push_back(Ck);
    }
    return out;
  }
  int p = 1;
  for (int i = 1; i <= K; ++i)
    if (S[i + 1] + beta >= S[K - 1]) { p = i; break; }
  for (int k = 1; k < p; ++k) {
    std::vector<int> Ck;
    for (int i = S[k - 1]; i < S[k + 1] + beta; ++i) Ck.push_back(i);
    int ck1 = (int)Ck.size();
    if (Ck.back() > S[K - 1]) throw std::runtime_error("clique overlap assertion (chordal_cliques.jl:35)");
    for (int i = S[K - 1]; i < S[K] + 1; ++i)
      if (i != Ck[ck1 - 1]) Ck.push_back(i);
    int n = (int)Ck.size();
    if (mode == NNSDP_DECOMP_SINGLE || k == 1) {
      out.push_back(Ck);
    } else {
      int nk = xdims[k - 1], nk1 = xdims[k];
      std::vector<int> D1, D2;
      for (int i = 0; i < nk + nk1 + beta; ++i) D1.push_back(Ck[i]);
      D1.push_back(Ck[n - 1]);
      for (int i = nk + nk1; i < n; ++i) D2.push_back(Ck[i]);
      out.push_back(D1);
      out.push_back(D2);
    }
  }
  std::vector<int> Cp;
  for (int i = S[p - 1]; i < S[K] + 1; ++i) Cp.push_back(i);
  out.push_back(Cp);
  return out;
}

struct Pattern {
  int n = 0;                 // dimension of the (reduced) z vector
  int NE = 0;
  std::vector<int> erow, ecol;     // i >= j
  std::vector<int> index;          // n*n -> entry or -1 (symmetric)
  std::vector<double> count;       // cliques containing the entry
  std::vector<std::vector<int>> cliques;
  int pos(int i, int j) const { return index[(size_t)i * n + j]; }
};

inline Pattern build_pattern(int n, const std::vector<std::vector<int>>& cliques, bool dense_all = false) {
  Pattern P;
  P.n = n;
  P.cliques = cliques;
  std::vector<int> cnt((size_t)n * n, 0);
  if (dense_all) {
    std::fill(cnt.begin(), cnt.end(), 1);
  } else {
    for (auto& c : cliques)
      for (int i : c)
        for (int j : c) cnt[(size_t)i * n + j] += 1;
  }
  P.index.assign((size_t)n * n, -1);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i)
      if (cnt[(size_t)i * n + j] > 0) {
        int e = P.NE++;
        P.erow.push_back(i);
        P.ecol.push_back(j);
        P.count.push_back((double)cnt[(size_t)i * n + j]);
        P.index[(size_t)i * n + j] = e;
        P.index[(size_t)j * n + i] = e;
      }
  return P;
}

// z = T zhat: z_i = h_i zhat_{newpos i} + m_i zhat_a  (solver-internal normalisation)
struct Congruence {
  std::vector<double> m, h;
  std::vector<int> newpos;
  int nfull = 0, nred = 0;
  bool factored = false;   // true: box generators are written in their exact factored form -2 h^2 (zhat_i^2 - zhat_a^2)
  SpVec tvec(const SpVec& v) const {
    SpVec o;
    double acc = 0.0;
    int a = nfull - 1;
    for (size_t t = 0; t < v.idx.size(); ++t) {
      int i = v.idx[t];
      double x = v.val[t];
      if (i == a) acc += x;
      else {
        acc += x * m[i];
        if (newpos[i] >= 0) { o.idx.push_back(newpos[i]); o.val.push_back(x * h[i]); }
      }
    }
    o.idx.push_back(nred - 1);
    o.val.push_back(acc);
    return o;
  }
};

// `guard` (nnsdp_options.interval_guard): a neuron interval narrower than guard x |midpoint| is widened to that.  The
// solver then works with the weaker, still valid, QC, and a gamma feasible for the widened LMI is feasible for the
// original one: Z_orig(gamma) = Z_wid(gamma) - 2 gac1_t (h_wid^2 - h^2) e_a e_a'.  Needed because the reference's
// float32 CROWN boxes of collapsed deep nets (bench/rand W10-D80, W20-D70, ...) are narrower than their own rounding
// error: they exclude the float64 trajectories by up to 2e-5 relative, the QC set is then empty and rho = 0 "optimal".
inline Congruence make_congruence(const ProblemCopy& P, bool normalize, double guard = 0.0) {
  Congruence C;
  C.nfull = P.Zdim;
  C.m.assign(P.Zdim - 1, 0.0);
  C.h.assign(P.Zdim - 1, 1.0);
  C.newpos.assign(P.Zdim, 0);
  if (!normalize) {
    std::iota(C.newpos.begin(), C.newpos.end(), 0);
    C.nred = P.Zdim;
    return C;
  }
  bool eliminate = (P.query_kind == NNSDP_QUERY_REACH);  // only cost-free multipliers may go to infinity
  double hmax = 0.0;
  for (int i = 0; i < P.Zdim - 1; ++i) {
    double lo = i < P.nin ? P.x1min[i] : P.acymin[i - P.nin];
    double hi = i < P.nin ? P.x1max[i] : P.acymax[i - P.nin];
    C.m[i] = 0.5 * (lo + hi);
    C.h[i] = 0.5 * (hi - lo);
    if (i >= P.nin) C.h[i] = std::max(C.h[i], guard * std::fabs(C.m[i]));
    hmax = std::max(hmax, C.h[i]);
  }
  C.factored = true;
  if (!eliminate)
    for (auto& v : C.h) v = std::max(v, 1e-6 * std::max(1.0, hmax));
  int r = 0;
  for (int i = 0; i < P.Zdim - 1; ++i) C.newpos[i] = (C.h[i] > 0.0) ? r++ : -1;
  C.newpos[P.Zdim - 1] = r++;
  C.nred = r;
  return C;
}

struct Triplet {
  int pos, gen;
  double val;
};

struct Operator {
  Pattern pat;
  int ng = 0;                       // number of generator columns (full gamma length)
  std::vector<Triplet> trip;        // svec-scaled entries, duplicates summed, sorted by (pos, gen)
  std::vector<double> z0;           // NE
  std::vector<double> c;            // ng
};

class OperatorBuilder {
 public:
  OperatorBuilder(const ProblemCopy& P, const Congruence& C, Pattern pat) : P_(P), C_(C) { op_.pat = std::move(pat); }

  Operator build() {
    const ProblemCopy& P = P_;
    op_.ng = P.ng;
    int a = P.Zdim - 1;
    SpVec ea = C_.tvec(unit(a));
    // box generator -2 (z_i - l)(z_i - u) (input.jl:24-26, activ_bounded.jl:19-21); in solver coordinates it is
    // exactly -2 h^2 (zhat_i^2 - zhat_a^2): written in that factored form, free of cancellation
    auto add_box = [&](int gen, int i, double l, double u) {
      if (!C_.factored) {
        SpVec ei = C_.tvec(unit(i));
        add_sym(gen, ei, ei, -1.0);
        add_sym(gen, ei, ea, l + u);
        add_sym(gen, ea, ea, -l * u);
      } else if (C_.newpos[i] >= 0) {
        const double hh = C_.h[i] * C_.h[i];
        SpVec ri{{C_.newpos[i]}, {1.0}}, ra{{C_.nred - 1}, {1.0}};
        add_sym(gen, ri, ri, -hh);
        add_sym(gen, ra, ra, hh);
      }
    };
    for (int i = 0; i < P.nin; ++i) add_box(i, i, P.x1min[i], P.x1max[i]);
    // gout (output.jl:75,84,93)
    if (P.nout) add_sym(P.nin, ea, ea, P.out_kind == NNSDP_OUT_HPLANE ? -1.0 : -0.5);
    // activations
    int o1 = P.nin + P.nout, ol = o1 + P.n1, ov = ol + P.acdim, oe = ov + P.npairs, on = oe + P.acdim;
    std::vector<SpVec> ut(P.acdim), yt(P.acdim);
    int t = 0;
    for (int k = 0; k < P.K - 1; ++k) {
      int nk = P.xdims[k], nk1 = P.xdims[k + 1];
      for (int r = 0; r < nk1; ++r, ++t) {
        SpVec u;
        for (int i = 0; i < nk; ++i) { u.idx.push_back(P.offs[k] + i); u.val.push_back(P.W[k][(size_t)r * nk + i]); }
        u.idx.push_back(a);
        u.val.push_back(P.b[k][r]);
        ut[t] = C_.tvec(u);
        yt[t] = C_.tvec(unit(P.offs[k + 1] + r));
      }
    }
    for (t = 0; t < P.acdim; ++t) {
      double sn = P.smin[t], sx = P.smax[t];
      add_box(o1 + t, P.nin + t, P.acymin[t], P.acymax[t]);   // activ_bounded.jl:19-21
      add_sym(ol + t, ut[t], ut[t], -sn * sx);             // activ_sector.jl:42 (Q11)
      add_sym(ol + t, ut[t], yt[t], sn + sx);              // activ_sector.jl:43 (Q12)
      if (P.activ == NNSDP_ACTIV_RELU) {                   // eta, nu exist for ReLU only (activ_sector.jl:49-57)
        add_sym(oe + t, ut[t], ea, -sn);                   // activ_sector.jl:55
        add_sym(oe + t, yt[t], ea, 1.0);                   // activ_sector.jl:56
        add_sym(on + t, ut[t], ea, -sx);
        add_sym(on + t, yt[t], ea, 1.0);
      }
    }
    // repeated-nonlinearity pairs, i-major order (activ_sector.jl:29-35)
    int r = 0;
    for (int i = 0; i < P.acdim - 1; ++i)
      for (int j = i + 1; j < P.acdim && j - i <= P.beta; ++j, ++r) {
        SpVec du = diff(ut[i], ut[j]), dy = diff(yt[i], yt[j]);
        add_sym(ov + r, du, dy, 1.0);
        add_sym(ov + r, dy, dy, -1.0);
      }
    if (r != P.npairs) throw std::runtime_error("pair count mismatch (activ_sector.jl:32)");
    finalize_triplets();
    build_z0();
    build_cost();
    return std::move(op_);
  }

 private:
  static SpVec unit(int i) { return SpVec{{i}, {1.0}}; }
  static SpVec diff(const SpVec& a, const SpVec& b) {
    SpVec o = a;
    for (size_t t = 0; t < b.idx.size(); ++t) { o.idx.push_back(b.idx[t]); o.val.push_back(-b.val[t]); }
    return o;
  }
  // alpha * (p q' + q p') into generator `gen`
  void add_sym(int gen, const SpVec& p, const SpVec& q, double alpha) {
    if (alpha == 0.0) return;
    const Pattern& pat = op_.pat;
    for (size_t s = 0; s < p.idx.size(); ++s) {
      double pv = alpha * p.val[s];
      if (pv == 0.0) continue;
      int a = p.idx[s];
      for (size_t t = 0; t < q.idx.size(); ++t) {
        double v = pv * q.val[t];
        if (v == 0.0) continue;
        int b = q.idx[t];
        int e = pat.pos(a, b);
        if (e < 0) throw std::runtime_error("generator entry outside the clique pattern");
        op_.trip.push_back(Triplet{e, gen, a == b ? 2.0 * v : v * M_SQRT2});
      }
    }
  }
  void finalize_triplets() {
    auto& T = op_.trip;
    std::sort(T.begin(), T.end(), [](const Triplet& x, const Triplet& y) {
      return x.pos != y.pos ? x.pos < y.pos : x.gen < y.gen;
    });
    size_t w = 0;
    for (size_t i = 0; i < T.size();) {
      size_t j = i;
      double s = 0.0;
      while (j < T.size() && T[j].pos == T[i].pos && T[j].gen == T[i].gen) s += T[j++].val;
      if (s != 0.0) T[w++] = Triplet{T[i].pos, T[i].gen, s};
      i = j;
    }
    T.resize(w);
  }
  // Z0 = (R Eout T)' S(gout = 0) (R Eout T), rows of R Eout: [x_1; W_K x_K + b_K; 1] (output.jl:34-49)
  void build_z0() {
    const ProblemCopy& P = P_;
    int d1 = P.xdims[0], m = P.mout, sd = d1 + m + 1, a = P.Zdim - 1, K = P.K;
    std::vector<double> S0((size_t)sd * sd, 0.0);  // column-major
    auto Sset = [&](int i, int j, double v) { S0[(size_t)j * sd + i] = v; };
    if (P.out_kind == NNSDP_OUT_SAFETY_S) {
      S0 = P.S;
    } else if (P.out_kind == NNSDP_OUT_HPLANE) {
      for (int i = 0; i < m; ++i) { Sset(d1 + i, sd - 1, P.normal[i]); Sset(sd - 1, d1 + i, P.normal[i]); }
    } else if (P.out_kind == NNSDP_OUT_CIRCLE) {
      double ycyc = 0;
      for (int i = 0; i < m; ++i) {
        Sset(d1 + i, d1 + i, 1.0);
        Sset(d1 + i, sd - 1, -P.yc[i]);
        Sset(sd - 1, d1 + i, -P.yc[i]);
        ycyc += P.yc[i] * P.yc[i];
      }
      Sset(sd - 1, sd - 1, ycyc);
    } else {  // ellipsoid: S22 = invP' invP, S23 = -invP' yc, S33 = yc'yc (output.jl:91-93)
      double ycyc = 0;
      for (int i = 0; i < m; ++i) ycyc += P.yc[i] * P.yc[i];
      for (int i = 0; i < m; ++i) {
        for (int j = 0; j < m; ++j) {
          double s = 0;
          for (int k = 0; k < m; ++k) s += P.invP[(size_t)i * m + k] * P.invP[(size_t)j * m + k];  // (invP'invP)_ij
          Sset(d1 + i, d1 + j, s);
        }
        double s = 0;
        for (int k = 0; k < m; ++k) s += P.invP[(size_t)i * m + k] * P.yc[k];  // (invP' yc)_i = sum_k invP[k,i] yc_k
        Sset(d1 + i, sd - 1, -s);
        Sset(sd - 1, d1 + i, -s);
      }
      Sset(sd - 1, sd - 1, ycyc);
    }
    // rows of R Eout as sparse vectors in full coordinates, then through the congruence
    std::vector<SpVec> rows(sd);
    for (int i = 0; i < d1; ++i) rows[i] = C_.tvec(unit(i));
    int nK = P.xdims[K - 1];
    for (int r = 0; r < m; ++r) {
      SpVec u;
      for (int i = 0; i < nK; ++i) { u.idx.push_back(P.offs[K - 1] + i); u.val.push_back(P.W[K - 1][(size_t)r * nK + i]); }
      u.idx.push_back(a);
      u.val.push_back(P.b[K - 1][r]);
      rows[d1 + r] = C_.tvec(u);
    }
    rows[sd - 1] = C_.tvec(unit(a));
    const Pattern& pat = op_.pat;
    op_.z0.assign(pat.NE, 0.0);
    for (int r1 = 0; r1 < sd; ++r1)
      for (int r2 = 0; r2 < sd; ++r2) {
        double s = S0[(size_t)r2 * sd + r1];
        if (s == 0.0) continue;
        const SpVec &p = rows[r1], &q = rows[r2];
        for (size_t x = 0; x < p.idx.size(); ++x)
          for (size_t y = 0; y < q.idx.size(); ++y) {
            int i = p.idx[x], j = q.idx[y];
            if (i < j) continue;  // lower triangle only: Z0[i][j] = sum_{r1,r2} R[r1][i] S[r1][r2] R[r2][j]
            double v = s * p.val[x] * q.val[y];
            if (v == 0.0) continue;
            int e = pat.pos(i, j);
            if (e < 0) throw std::runtime_error("Z0 entry outside the clique pattern");
            op_.z0[e] += (i == j) ? v : v * M_SQRT2;
          }
      }
  }
  void build_cost() {
    // reach: obj_func = x -> x[1] on gout (NnSdp.jl:46); safety: sum(gin) + sum(gac) (deep_sdp.jl:25)
    op_.c.assign(P_.ng, P_.nout ? 0.0 : 1.0);
    if (P_.nout) op_.c[P_.nin] = 1.0;
  }
  const ProblemCopy& P_;
  const Congruence& C_;
  Operator op_;
};

// Solver-scaled operator in CSR (rows = pattern entries) and CSC (columns = kept generators).
struct ScaledOperator {
  int NE = 0, ng = 0, ng_full = 0;
  std::vector<int> keep;            // kept generator -> full index
  std::vector<int> layer;           // (when ordered by layer) network layer of every kept generator, nondecreasing
  std::vector<double> ecol;         // g_full = ecol * g_scaled / zscale
  double zscale = 1.0, cscale = 1.0;
  std::vector<double> z0, c, Dinv;
  std::vector<int> csr_ptr, csr_col;
  std::vector<double> csr_val;
  std::vector<int> csc_ptr, csc_row;
  std::vector<double> csc_val;
};

// `layer_of` (optional, per full generator index): kept generators are ordered by it (stable), so that the multipliers of one
// network layer are contiguous - the order the structured M^-1 (minv.hpp) works in.  Any order gives the same iteration.
inline ScaledOperator scale_operator(const Operator& op, bool normalize_columns, const std::vector<int>* layer_of = nullptr) {
  ScaledOperator S;
  S.NE = op.pat.NE;
  S.ng_full = op.ng;
  std::vector<double> cn(op.ng, 0.0);
  for (auto& t : op.trip) cn[t.gen] += t.val * t.val;
  std::vector<int> newgen(op.ng, -1);
  for (int g = 0; g < op.ng; ++g) {
    cn[g] = std::sqrt(cn[g]);
    if (cn[g] > 1e-150) S.keep.push_back(g);   // only identically-zero generators are dropped; tiny-width neurons keep theirs
  }
  if (layer_of) std::stable_sort(S.keep.begin(), S.keep.end(), [&](int a, int b) { return (*layer_of)[a] < (*layer_of)[b]; });
  for (size_t i = 0; i < S.keep.size(); ++i) {
    const int g = S.keep[i];
    newgen[g] = (int)i;
    S.ecol.push_back(normalize_columns ? 1.0 / cn[g] : 1.0);
    if (layer_of) S.layer.push_back((*layer_of)[g]);
  }
  S.ng = (int)S.keep.size();
  double zn = 0, cnrm = 0;
  for (double v : op.z0) zn += v * v;
  S.c.resize(S.ng);
  for (int g = 0; g < S.ng; ++g) { S.c[g] = op.c[S.keep[g]] * S.ecol[g]; cnrm += S.c[g] * S.c[g]; }
  zn = std::sqrt(zn);
  cnrm = std::sqrt(cnrm);
  if (normalize_columns) {
    S.zscale = zn > 0 ? 1.0 / zn : 1.0;
    S.cscale = cnrm > 0 ? 1.0 / cnrm : 1.0;
  }
  S.z0 = op.z0;
  for (auto& v : S.z0) v *= S.zscale;
  for (auto& v : S.c) v *= S.cscale;
  S.Dinv.resize(S.NE);
  for (int e = 0; e < S.NE; ++e) S.Dinv[e] = 1.0 / op.pat.count[e];
  // CSR (triplets are sorted by pos then gen)
  S.csr_ptr.assign(S.NE + 1, 0);
  for (auto& t : op.trip)
    if (newgen[t.gen] >= 0) S.csr_ptr[t.pos + 1]++;
  for (int e = 0; e < S.NE; ++e) S.csr_ptr[e + 1] += S.csr_ptr[e];
  size_t nnz = S.csr_ptr[S.NE];
  S.csr_col.resize(nnz);
  S.csr_val.resize(nnz);
  S.csc_ptr.assign(S.ng + 1, 0);
  {
    std::vector<int> fill(S.csr_ptr.begin(), S.csr_ptr.end() - 1);
    for (auto& t : op.trip) {
      int g = newgen[t.gen];
      if (g < 0) continue;
      int k = fill[t.pos]++;
      S.csr_col[k] = g;
      S.csr_val[k] = t.val * S.ecol[g];
      S.csc_ptr[g + 1]++;
    }
  }
  if (layer_of) {
    // the kept order is a permutation of the full order: re-sort every row by kept index (build_M relies on sorted columns)
    std::vector<std::pair<int, double>> tmp;
    for (int e = 0; e < S.NE; ++e) {
      const int lo = S.csr_ptr[e], hi = S.csr_ptr[e + 1];
      tmp.clear();
      for (int k = lo; k < hi; ++k) tmp.emplace_back(S.csr_col[k], S.csr_val[k]);
      std::sort(tmp.begin(), tmp.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      for (int k = lo; k < hi; ++k) { S.csr_col[k] = tmp[k - lo].first; S.csr_val[k] = tmp[k - lo].second; }
    }
  }
  for (int g = 0; g < S.ng; ++g) S.csc_ptr[g + 1] += S.csc_ptr[g];
  S.csc_row.resize(nnz);
  S.csc_val.resize(nnz);
  {
    std::vector<int> fill(S.csc_ptr.begin(), S.csc_ptr.end() - 1);
    for (int e = 0; e < S.NE; ++e)
      for (int k = S.csr_ptr[e]; k < S.csr_ptr[e + 1]; ++k) {
        int g = S.csr_col[k];
        int q = fill[g]++;
        S.csc_row[q] = e;
        S.csc_val[q] = S.csr_val[k];
      }
  }
  return S;
}

// dense M = I + A' D^-1 A (column-major ng x ng, both triangles)
// M = I + A' D^-1 A, dense, LOWER triangle (column-major, leading dimension ng; the upper triangle is zero: the device factorises with
// fill_lower and mirrors the inverse itself) into caller memory that need not be initialised.  Host threads own disjoint column ranges:
// every entry is accumulated by ONE thread in the row order of A, so the bits do not depend on the thread count; each thread also
// first-touches its own columns (W40-D40, 150 MB: 31 ms -> see DESIGN.md section 5).
inline void build_M_lower(const ScaledOperator& S, double* M) {
  const size_t n = S.ng;
  int T = (int)std::max<size_t>(1, std::min<size_t>(4, n / 512));
  if (const char* e = std::getenv("NNSDP_HOST_THREADS")) T = std::max(1, std::min(std::atoi(e), 16));      // (diagnostic / test)
  auto work = [&](int t) {
    // columns [c0, c1): the cost of a column range is ~ the lower triangle below it, so the ranges are balanced by area
    auto cut = [&](int q) { return (size_t)((double)n * (1.0 - std::sqrt(1.0 - (double)q / T))); };
    const size_t c0 = t == 0 ? 0 : cut(t), c1 = t == T - 1 ? n : cut(t + 1);
    if (c1 <= c0) return;
    std::fill(M + c0 * n, M + c1 * n, 0.0);
    for (int e = 0; e < S.NE; ++e) {
      const int lo = S.csr_ptr[e], hi = S.csr_ptr[e + 1];
      const double d = S.Dinv[e];
      for (int x = lo; x < hi; ++x) {
        const size_t cx = S.csr_col[x];
        if (cx < c0) continue;
        if (cx >= c1) break;                                 // (columns of a row are sorted)
        const double vx = d * S.csr_val[x];
        double* col = M + cx * n;
        for (int y = x; y < hi; ++y) col[S.csr_col[y]] += vx * S.csr_val[y];  // lower: row >= col
      }
    }
    for (size_t j = c0; j < c1; ++j) M[j * n + j] += 1.0;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

}  // namespace nnsdp
