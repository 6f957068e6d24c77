// Diagnostic build (never shipped): per-phase cycle shares of the Jacobi projection kernel.
// hipcc -O3 --offload-arch=gfx950 -DNNSDP_STAMPS tools/prof_jacobi.hip -o gpurun_out/prof_jacobi
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../nn-sdp_amd/csrc/kernels.hip"
using namespace nnsdp;
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 85, batch = argc > 2 ? atoi(argv[2]) : 19;
  std::mt19937_64 rng(1); std::normal_distribution<double> nd;
  std::vector<double> h((size_t)batch * n * n);
  for (int b = 0; b < batch; ++b) for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = nd(rng); h[(size_t)b*n*n + j*n + i] = v; h[(size_t)b*n*n + i*n + j] = v; }
  std::vector<int> cn(batch, n); std::vector<long long> coff(batch + 1), eoff(batch + 1);
  for (int b = 0; b <= batch; ++b) { coff[b] = (long long)b * n * n; eoff[b] = (long long)b * n; }
  int *dcn; long long *dco, *deo; double *dnu, *dw, *dV, *dE;
  hipMalloc(&dcn, batch * 4); hipMalloc(&dco, (batch + 1) * 8); hipMalloc(&deo, (batch + 1) * 8);
  hipMalloc(&dnu, h.size() * 8); hipMalloc(&dw, h.size() * 8); hipMalloc(&dV, h.size() * 8); hipMalloc(&dE, (8192 + batch * n) * 8);
  hipMemcpy(dcn, cn.data(), batch * 4, hipMemcpyHostToDevice); hipMemcpy(dco, coff.data(), (batch + 1) * 8, hipMemcpyHostToDevice);
  hipMemcpy(deo, eoff.data(), (batch + 1) * 8, hipMemcpyHostToDevice); hipMemcpy(dnu, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  int alg = argc > 5 ? atoi(argv[5]) : (proj_sys_ok(n) ? 2 : 0);   // 0 round robin (LDS), 1 block, 2 systolic (registers)
  if (alg == 1 && !proj_block_ok(n)) alg = 0;
  if (alg == 2 && !proj_sys_ok(n)) alg = 0;
  if (alg == 3 && !proj_pp_ok(n)) alg = 0;
  const bool blockmode = alg == 1;
  bool v_lds = proj_lds_bytes(n, true, alg) <= 160 * 1024; size_t lds = proj_lds_bytes(n, v_lds, alg);
  proj_allow_big_lds();
  ProjArgs a{}; a.cn = dcn; a.coff = dco; a.eoff = deo; a.nu = dnu; a.w = dw; a.Vg = dV; a.eig = dE; a.kappa = nullptr; a.tol_dev = nullptr; int* dstats; hipMalloc(&dstats, 64); a.stats = dstats; a.warm = 0; a.max_sweeps = 30; a.tol = 1e-13;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    if (rep >= 2) {  // warm regime: perturb the matrices slightly, start from the stored eigenvectors, solver tolerance
      double eps = argc > 3 ? atof(argv[3]) : 1e-3;
      for (auto& v : h) v *= (1.0 + eps * nd(rng));
      for (int b = 0; b < batch; ++b) for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) h[(size_t)b*n*n + i*n + j] = h[(size_t)b*n*n + j*n + i];
      hipMemcpy(dnu, h.data(), h.size() * 8, hipMemcpyHostToDevice);
      a.warm = 1; a.tol = argc > 4 ? atof(argv[4]) : 1e-6;
    }
    hipMemset(dstats, 0, 64);
    hipEventRecord(e0); launch_proj(a, batch, n, v_lds, lds, nullptr, alg); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long dbg[70] = {0}; hipMemcpy(dbg, dE + 4096, sizeof(dbg), hipMemcpyDeviceToHost);
    int st[4]; hipMemcpy(st, dstats, 16, hipMemcpyDeviceToHost);
    printf("n=%d batch=%d alg=%d v_lds=%d kernel %.3f ms, max sweeps %d (avg %.2f) -> %.1f us/sweep(max)\n", n, batch, alg, (int)v_lds, ms, st[1], (double)st[0] / batch, 1e3 * ms / (st[1] > 0 ? st[1] : 1));
#ifdef NNSDP_STAMPS
    printf("  sections (cycles, wave 0): load+basis %lld  warm-GEMM %lld  sweeps %lld  reconstruct+store %lld\n", dbg[65], dbg[66], dbg[67], dbg[69]);
    if (rep != 1 && rep != 3) continue;
    int nw = n > kSmallBlock ? 16 : 4;
    long long rounds = blockmode ? dbg[64] * ((((n + 15) & ~15) >> 3) - 1) : dbg[64] * (((n + 1) & ~1) - 1);
    for (int w = 0; w < nw; ++w) printf("  wave %2d cycles/round: phase1 %6.0f barrier %6.0f phase2 %6.0f barrier %6.0f\n", w,
      (double)dbg[w*4]/rounds, (double)dbg[w*4+1]/rounds, (double)dbg[w*4+2]/rounds, (double)dbg[w*4+3]/rounds);
#endif
  }
  return 0;
}
