// (diagnostic, never shipped) what bounds the operand loads of the tile-parallel refinement pipeline: a wave issues L global loads of
// the pipeline's pattern (four 128-byte segments per wave instruction, scalar base + 32-bit lane offset) from an L2-resident buffer
// and stamps the wall clock (100 MHz) at entry, after the last load is issued and when all have landed.
//   hipcc -O3 --offload-arch=gfx950 tools/load_issue_probe.hip -o gpurun_out/load_issue_probe && gpurun_out/load_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int L, int WIDTH>   // WIDTH: 1 = dwordx2 per lane (8 B), 2 = dwordx4 per lane (16 B)
__global__ void probe(const double* __restrict__ M, int n, long long* out, double* sink, int dep_chain) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lc = lane >> 4;
  long long t0 = wall_clock64();
  const char* base = reinterpret_cast<const char*>(M) + (size_t)blockIdx.x * 65536 + (size_t)wv * 8192;
  double acc = 0.0;
  if (WIDTH == 1) {
    double v[L];
    const unsigned voff = (unsigned)(lc * n + lr) * 8u;
#pragma unroll
    for (int k = 0; k < L; ++k) v[k] = *reinterpret_cast<const double*>(base + (size_t)((unsigned)k * 32u * (unsigned)n) + voff);
    long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = wall_clock64();
#pragma unroll
    for (int k = 0; k < L; ++k) acc += v[k];
    if (lane == 0) { long long* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + wv) * 4; o[0] = t0; o[1] = t1; o[2] = t2; }
  } else {
    double2 v[L / 2];
    const unsigned voff = (unsigned)(lc * n + (lr & 7) * 2) * 8u + (unsigned)(lr >> 3) * 16u * (unsigned)n;
#pragma unroll
    for (int k = 0; k < L / 2; ++k) v[k] = *reinterpret_cast<const double2*>(base + (size_t)((unsigned)k * 64u * (unsigned)n) + voff);
    long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = wall_clock64();
#pragma unroll
    for (int k = 0; k < L / 2; ++k) acc += v[k].x + v[k].y;
    if (lane == 0) { long long* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + wv) * 4; o[0] = t0; o[1] = t1; o[2] = t2; }
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

template <int L, int WIDTH>
void run(const double* dM, long long* dout, double* dsink, int wgs, int waves, int n) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<L, WIDTH>), dim3(wgs), dim3(64 * waves), 0, nullptr, dM, n, dout, dsink, 0);
  hipDeviceSynchronize();
  std::vector<long long> h((size_t)wgs * waves * 4);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double iss = 0, land = 0; long long lo = h[0], hi = 0;
  for (int w = 0; w < wgs * waves; ++w) { iss += h[4 * w + 1] - h[4 * w]; land += h[4 * w + 2] - h[4 * w + 1]; lo = std::min(lo, h[4 * w]); hi = std::max(hi, h[4 * w + 2]); }
  printf("L=%2d loads/wave x %2d B/lane, %3d workgroups x %d waves: issue %.2f us (%.0f ns per load), landed +%.2f us, span %.2f us\n", L, 8 * WIDTH, wgs, waves,
         0.01 * iss / (wgs * waves), 10.0 * iss / (wgs * waves) / (WIDTH == 1 ? L : L / 2), 0.01 * land / (wgs * waves), 0.01 * (hi - lo));
}

int main() {
  const int n = 85;
  double* dM; long long* dout; double* dsink;
  hipMalloc(&dM, (size_t)64 << 20); hipMemset(dM, 0, (size_t)64 << 20);
  hipMalloc(&dout, 1 << 20); hipMalloc(&dsink, 64);
  for (int waves : {1, 2, 4, 6}) {
    run<48, 1>(dM, dout, dsink, 114, waves, n);
    run<48, 2>(dM, dout, dsink, 114, waves, n);
  }
  run<16, 1>(dM, dout, dsink, 114, 6, n);
  run<24, 1>(dM, dout, dsink, 114, 6, n);
  run<48, 1>(dM, dout, dsink, 684, 1, n);
  run<48, 1>(dM, dout, dsink, 12, 6, n);
  return 0;
}
