// Interval pre-processing on the host: CROWN-sliced bounds of every
// hidden layer, the direct inputs of the LMI assembler (acymin/acymax of QcActivBounded, smin/smax of
// QcActivSector).  Replaces Intervals.intervalsAutoLirpaSliced (src/Intervals/intervals_auto_lirpa.jl:12-64),
// which reaches auto_LiRPA 0.2 through PyCall + ONNX once per layer (exts/auto_lirpa_bridge.py:97-112).
// Rules restated from the vendored library (exts/auto_LiRPA/operators/activation.py:306-323,386-388,
// bound_general.py:1078-1079): backward LiRPA for the final node and for every intermediate pre-activation;
// ReLU relaxation lb_r = min(l,0), ub_r = max(max(u,0), lb_r + 1e-8), upper slope d = ub_r/(ub_r - lb_r) with
// intercept -lb_r d, lower slope 1 if d > 0.5 else 0.  float32 like the reference's torch path
// (exts/NNet/converters/nnet2onnx.py:47,51), the Julia post-fix lb = min(lb,ub), ub = max(lb,ub)
// (intervals_auto_lirpa.jl:38-39), then one float64 interval step per layer for the pre-activations (:55-62).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace nnsdp {

struct DenseF {             // row-major float matrix
  int r = 0, c = 0;
  std::vector<float> a;
  DenseF() {}
  DenseF(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0f) {}
  float& at(int i, int j) { return a[(size_t)i * c + j]; }
  float at(int i, int j) const { return a[(size_t)i * c + j]; }
};

struct IntervalsOut {
  std::vector<std::vector<double>> xlo, xhi;      // K+1 post-activation intervals (x_1 .. x_K, output)
  std::vector<std::vector<double>> plo, phi;      // K-1 pre-activation intervals (float64 interval step)
};

// ---- tanh relaxation (exts/auto_LiRPA/operators/activation.py BoundTanh: dtanh :863-868, precompute_relaxation :872-917,
// bound_relax_impl, non-optimised branch :925-956,992-1016; the reference's bridge maps torch.nn.Tanh onto it,
// exts/auto_lirpa_bridge.py:31-37,86-87).  float32 like the library's default dtype.
inline float dtanh_f(float x) {
  if (!(std::fabs(x) < 25.0f)) return 0.0f;        // (mask: cosh(25)^2 overflows float32)
  const float c = std::cosh(x);
  return 1.0f / (c * c);
}
// the table entries the relaxation looks up, computed on demand: tangent point d <= 0 whose tangent stays below tanh at
// U = 0.01 (max(0, int(u / 0.01)) + 1), and its mirror image for the lower end (bisection, 100 halvings as the library's)
inline float tanh_d_lower(float upper) {
  const long idx = std::max(0L, (long)(upper / 0.01f)) + 1;
  const float U = 0.01f * (float)idx, fU = std::tanh(U);
  auto ok = [&](float d) { return dtanh_f(d) * (U - d) + std::tanh(d) <= fU; };
  float l = -1.0f, r = 0.0f;
  for (int it = 0; it < 64 && !ok(l); ++it) l *= 2.0f;
  for (int it = 0; it < 100; ++it) { const float m = (l + r) / 2.0f; if (ok(m)) l = m; else r = m; }
  return l;
}
inline float tanh_d_upper(float lower) {
  const long idx = std::max(0L, (long)(lower / -0.01f)) + 1;
  const float Lw = -0.01f * (float)idx, fL = std::tanh(Lw);
  auto ok = [&](float d) { return dtanh_f(d) * (Lw - d) + std::tanh(d) >= fL; };
  float l = 0.0f, r = 1.0f;
  for (int it = 0; it < 64 && !ok(r); ++it) r *= 2.0f;
  for (int it = 0; it < 100; ++it) { const float m = (l + r) / 2.0f; if (ok(m)) r = m; else l = m; }
  return r;
}
// lw x + lb <= tanh(x) <= uw x + ub on [l, u]
inline void tanh_relax(float l, float u, float& lw, float& lb, float& uw, float& ub) {
  const float lower = std::max(l, -500.0f), upper = std::min(u, 500.0f);
  const float yl = std::tanh(lower), yu = std::tanh(upper);
  const float kd = (upper - lower) < 1e-6f ? dtanh_f(upper) : (yu - yl) / std::max(upper - lower, 1e-6f);
  const bool pos = l >= 0.0f, neg = u <= 0.0f;
  const float m = (lower + upper) / 2.0f, ym = std::tanh(m), km = dtanh_f(m);
  lw = lb = uw = ub = 0.0f;
  auto line = [](float k, float x0, float y0, float& w, float& b) { w += k; b += -x0 * k + y0; };
  // (a neuron with l >= 0 and u <= 0, i.e. l = u = 0, carries BOTH masks in the library - mask_both = 1 - pos - neg = -1 there;
  // the three contributions are added exactly as the library adds them)
  const float fpos = pos ? 1.0f : 0.0f, fneg = neg ? 1.0f : 0.0f, fboth = 1.0f - fpos - fneg;
  if (neg) { line(kd, lower, yl, uw, ub); line(km, m, ym, lw, lb); }
  if (pos) { line(kd, lower, yl, lw, lb); line(km, m, ym, uw, ub); }
  if (fboth != 0.0f) {
    const float dl = tanh_d_lower(upper), du = tanh_d_upper(lower);
    float w = 0.0f, b = 0.0f;
    if (kd < dtanh_f(lower)) line(kd, lower, yl, w, b); else line(dtanh_f(dl), dl, std::tanh(dl), w, b);
    lw += fboth * w; lb += fboth * b;
    w = b = 0.0f;
    if (kd < dtanh_f(upper)) line(kd, lower, yl, w, b); else line(dtanh_f(du), du, std::tanh(du), w, b);
    uw += fboth * w; ub += fboth * b;
  }
}

// backward bound of  Ws[L-1] act(... act(Ws[0] x + bs[0]) ...) + bs[L-1]  over the box [lo, hi] (act = relu, or tanh when tanh_act);
// pre[j] = pre-activation bounds of layer j (j < L-1)
inline void crown_backward(const std::vector<const DenseF*>& Ws, const std::vector<const std::vector<float>*>& bs,
                           const std::vector<std::vector<float>>& prel, const std::vector<std::vector<float>>& preu,
                           const std::vector<float>& lo, const std::vector<float>& hi, std::vector<float>& out_lo,
                           std::vector<float>& out_hi, bool tanh_act = false) {
  const int L = (int)Ws.size();
  DenseF lA = *Ws[L - 1], uA = *Ws[L - 1];
  const int nout = lA.r;
  std::vector<float> lb(*bs[L - 1]), ub(*bs[L - 1]);
  for (int j = L - 2; j >= 0; --j) {
    const int d = lA.c;                       // width of layer j's output
    std::vector<float> du(d), dl(d), bu(d), bl(d, 0.0f);
    for (int t = 0; t < d; ++t) {
      if (tanh_act) { tanh_relax(prel[j][t], preu[j][t], dl[t], bl[t], du[t], bu[t]); continue; }
      float lr = std::min(prel[j][t], 0.0f);
      float ur = std::max(std::max(preu[j][t], 0.0f), lr + 1e-8f);
      du[t] = ur / (ur - lr);
      dl[t] = du[t] > 0.5f ? 1.0f : 0.0f;
      bu[t] = -lr * du[t];
    }
    for (int i = 0; i < nout; ++i) {
      float sl = 0.0f, su = 0.0f;
      for (int t = 0; t < d; ++t) {
        float a = lA.at(i, t), b = uA.at(i, t);
        sl += std::min(a, 0.0f) * bu[t] + std::max(a, 0.0f) * bl[t];
        su += std::max(b, 0.0f) * bu[t] + std::min(b, 0.0f) * bl[t];
        lA.at(i, t) = std::max(a, 0.0f) * dl[t] + std::min(a, 0.0f) * du[t];
        uA.at(i, t) = std::max(b, 0.0f) * du[t] + std::min(b, 0.0f) * dl[t];
      }
      lb[i] += sl; ub[i] += su;
      float tl = 0.0f, tu = 0.0f;
      for (int t = 0; t < d; ++t) { tl += lA.at(i, t) * (*bs[j])[t]; tu += uA.at(i, t) * (*bs[j])[t]; }
      lb[i] += tl; ub[i] += tu;
    }
    const DenseF& W = *Ws[j];
    DenseF nl(nout, W.c), nu(nout, W.c);
    for (int i = 0; i < nout; ++i)
      for (int t = 0; t < d; ++t) {
        const float a = lA.at(i, t), b = uA.at(i, t);
        if (a == 0.0f && b == 0.0f) continue;
        const float* w = &W.a[(size_t)t * W.c];
        float* pl = &nl.a[(size_t)i * W.c];
        float* pu = &nu.a[(size_t)i * W.c];
        for (int q = 0; q < W.c; ++q) { pl[q] += a * w[q]; pu[q] += b * w[q]; }
      }
    lA = std::move(nl); uA = std::move(nu);
  }
  const int n0 = lA.c;
  out_lo.assign(nout, 0.0f); out_hi.assign(nout, 0.0f);
  for (int i = 0; i < nout; ++i) {
    float sl = 0.0f, su = 0.0f, rl = 0.0f, ru = 0.0f;
    for (int q = 0; q < n0; ++q) {
      const float c = (hi[q] + lo[q]) / 2.0f, r = (hi[q] - lo[q]) / 2.0f;
      sl += lA.at(i, q) * c; rl += std::fabs(lA.at(i, q)) * r;
      su += uA.at(i, q) * c; ru += std::fabs(uA.at(i, q)) * r;
    }
    out_lo[i] = sl - rl + lb[i];
    out_hi[i] = su + ru + ub[i];
  }
}

// M: K matrices [W_k b_k], column-major xdims[k+1] x (xdims[k]+1), back to back (include/nnsdp.h, nnsdp_problem::M)
inline IntervalsOut make_intervals(int K, const int32_t* xdims, const double* M, const double* x1min, const double* x1max,
                                   bool tanh_act = false) {
  if (K < 2 || !xdims || !M || !x1min || !x1max) throw std::invalid_argument("make_intervals: bad arguments");
  std::vector<DenseF> W(K);
  std::vector<std::vector<float>> b(K);
  std::vector<std::vector<double>> Wd(K), bd(K);
  size_t off = 0;
  for (int k = 0; k < K; ++k) {
    const int r = xdims[k + 1], c = xdims[k];
    if (r <= 0 || c <= 0) throw std::invalid_argument("make_intervals: layer widths must be positive");
    W[k] = DenseF(r, c); b[k].resize(r); Wd[k].resize((size_t)r * c); bd[k].resize(r);
    for (int j = 0; j < c; ++j)
      for (int i = 0; i < r; ++i) { double v = M[off + (size_t)j * r + i]; W[k].at(i, j) = (float)v; Wd[k][(size_t)i * c + j] = v; }
    for (int i = 0; i < r; ++i) { double v = M[off + (size_t)c * r + i]; b[k][i] = (float)v; bd[k][i] = v; }
    off += (size_t)r * (c + 1);
  }
  const int n0 = xdims[0];
  std::vector<float> lo(n0), hi(n0);
  for (int i = 0; i < n0; ++i) {
    if (!(x1min[i] <= x1max[i])) throw std::invalid_argument("make_intervals: x1min must be <= x1max");
    lo[i] = (float)x1min[i]; hi[i] = (float)x1max[i];
  }
  IntervalsOut out;
  out.xlo.emplace_back(x1min, x1min + n0); out.xhi.emplace_back(x1max, x1max + n0);
  std::vector<std::vector<float>> prel, preu;
  auto fix = [&](const std::vector<float>& l, const std::vector<float>& u) {
    std::vector<double> L(l.size()), U(l.size());
    for (size_t i = 0; i < l.size(); ++i) { L[i] = std::min((double)l[i], (double)u[i]); U[i] = std::max(L[i], (double)u[i]); }
    out.xlo.push_back(std::move(L)); out.xhi.push_back(std::move(U));
  };
  for (int k = 1; k <= K; ++k) {
    std::vector<const DenseF*> Ws;
    std::vector<const std::vector<float>*> bs;
    for (int j = 0; j < std::min(k, K); ++j) { Ws.push_back(&W[j]); bs.push_back(&b[j]); }
    std::vector<float> l, u;
    if (k < K) {
      // slice k (intervals_auto_lirpa.jl:12-28): pre-activation of layer k, then its post-activation through an identity head
      crown_backward(Ws, bs, prel, preu, lo, hi, l, u, tanh_act);
      prel.push_back(l); preu.push_back(u);
      const int n = xdims[k];
      DenseF I(n, n);
      for (int i = 0; i < n; ++i) I.at(i, i) = 1.0f;
      std::vector<float> zero(n, 0.0f);
      Ws.push_back(&I); bs.push_back(&zero);
      crown_backward(Ws, bs, prel, preu, lo, hi, l, u, tanh_act);
      fix(l, u);
    } else {
      crown_backward(Ws, bs, prel, preu, lo, hi, l, u, tanh_act);
      fix(l, u);
    }
  }
  for (int k = 0; k + 1 < K; ++k) {           // float64 interval step per layer (intervals_auto_lirpa.jl:55-62)
    const int r = xdims[k + 1], c = xdims[k];
    std::vector<double> l(r), u(r);
    for (int i = 0; i < r; ++i) {
      double sl = bd[k][i], su = bd[k][i];
      for (int j = 0; j < c; ++j) {
        const double w = Wd[k][(size_t)i * c + j];
        if (w >= 0) { sl += w * out.xlo[k][j]; su += w * out.xhi[k][j]; }
        else { sl += w * out.xhi[k][j]; su += w * out.xlo[k][j]; }
      }
      l[i] = sl; u[i] = su;
    }
    out.plo.push_back(std::move(l)); out.phi.push_back(std::move(u));
  }
  return out;
}

}  // namespace nnsdp
