// Sampled forward pass of the network on the GPU: Y[:, s] = f(X[:, s]) for N points, fp64 on v_mfma_f64_16x16x4_f64.
// Replaces the 1e5 calls of evalFeedFwdNet inside Utils.sampleTrajs (src/Utils/qc.jl:40-47; evalFeedFwdNet:
// src/MyNeuralNetwork/MyNeuralNetwork.jl:40-48), the step that shapes the ellipsoid of NnSdp.findEllipsoid
// (approxEllipsoid, src/Utils/qc.jl:50-67).  SURVEY.md section 8, row f2.
//
// One wave per 16 samples; the activations of those samples stay in LDS for the whole network (two buffers of
// wp x 16 doubles, row i = neuron, column = sample).  A layer  x+ = act(W x + b)  is the product [W b] [x; 1]: the bias is the
// column after the weights in M_k (the reference's own packing, MyNeuralNetwork.jl:21,26) and the activations carry a
// constant-one row, so each 16-neuron output tile is a chain of ceil((in + 1) / 4) MFMAs with the weight operand read straight
// from HBM / L2 (12.8 KB per layer at width 40, shared by all waves) and the activation operand from LDS (conflict-free:
// 16 consecutive doubles per k).  Operand maps as in kernels.hip: lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15]; result
// register r of lane l is C[(l >> 4) + 4 r][l & 15].
#pragma once
#include <hip/hip_runtime.h>

namespace nnsdp {

struct FwdArgs {
  int K;                    // affine layers
  const int* xdims;         // K + 1 widths
  const long long* moff;    // offset of M_k = [W_k b_k] (xdims[k+1] x (xdims[k] + 1), column-major) in M
  const double* M;
  const double* X;          // xdims[0] x N, column-major
  double* Y;                // xdims[K] x N, column-major
  long long N;
  int activ;                // NNSDP_ACTIV_RELU / NNSDP_ACTIV_TANH on the K - 1 hidden layers
  int wp;                   // rows of one LDS buffer: max_k of (xdims[k] + 1) rounded up to 4
};

static constexpr int kFwdChunk = 12;   // k-steps whose weight operands are in flight together (width 40: all 11)
__global__ __launch_bounds__(64) void k_forward_mfma(FwdArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, lr = lane & 15, lc = lane >> 4;
  const long long s0 = 16LL * blockIdx.x, s = s0 + lr;
  double* cur = lds;
  double* nxt = lds + (size_t)a.wp * 16;
  {
    const int d0 = a.xdims[0], rows = (d0 + 1 + 3) & ~3;
    for (int idx = lane; idx < rows * 16; idx += 64) {
      const int i = idx >> 4;
      const long long sc = s0 + (idx & 15);
      double v = 0.0;
      if (i < d0) { if (sc < a.N) v = a.X[(size_t)sc * d0 + i]; }
      else if (i == d0) v = 1.0;
      cur[idx] = v;
    }
  }
  __syncthreads();
  for (int k = 0; k < a.K; ++k) {
    const int in = a.xdims[k], out = a.xdims[k + 1];
    const double* Mk = a.M + a.moff[k];
    const int ks = (in + 1 + 3) >> 2, ots = (out + 15) >> 4;
    const bool last = k == a.K - 1;
    for (int ot = 0; ot < ots; ++ot) {
      d4_t c = {0.0, 0.0, 0.0, 0.0};
      const int o = 16 * ot + lr;
      const bool orow_ok = o < out;
      // the weight operands of kFwdChunk k-steps are requested together: one L2 round trip per chunk instead of one per MFMA
      // (0.535 -> 0.353 ms at W40-D20; interleaving the output tiles' chains on top costs registers and is slower: 0.38-0.46 ms)
      for (int k0 = 0; k0 < ks; k0 += kFwdChunk) {
        double av[kFwdChunk];
#pragma unroll
        for (int u = 0; u < kFwdChunk; ++u) {
          const int kin = 4 * (k0 + u) + lc;
          av[u] = (orow_ok && k0 + u < ks && kin <= in) ? Mk[(size_t)kin * out + o] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kFwdChunk; ++u)
          if (k0 + u < ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], cur[(4 * (k0 + u) + lc) * 16 + lr], c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = 16 * ot + lc + 4 * r;
        double v = c[r];
        if (last) {
          if (orow < out && s < a.N) a.Y[(size_t)s * out + orow] = v;
        } else if (orow < out) {
          v = a.activ == NNSDP_ACTIV_TANH ? tanh(v) : (v > 0.0 ? v : 0.0);
          nxt[orow * 16 + lr] = v;
        }
      }
    }
    if (!last) {
      const int pad_end = (out + 1 + 3) & ~3;   // constant-one row for the next layer's bias, zero rows up to its K padding
      for (int idx = lane; idx < (pad_end - out) * 16; idx += 64) {
        const int i = out + (idx >> 4);
        nxt[i * 16 + (idx & 15)] = i == out ? 1.0 : 0.0;
      }
      __syncthreads();
      double* t = cur; cur = nxt; nxt = t;
    }
  }
}

}  // namespace nnsdp
