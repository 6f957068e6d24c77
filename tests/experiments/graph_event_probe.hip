// (experiment, not a test) do event-record nodes inside a replayed hipGraph deliver timestamps on this runtime?
// hipcc --offload-arch=gfx950 -O2 -o graph_event_probe graph_event_probe.hip && ./graph_event_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(int* p, int n) { int v = 0; for (int i = 0; i < n; ++i) v += i * (threadIdx.x + 1); if (v == 12345) p[0] = v; }
#define SHOW(x) do { hipError_t e_ = (x); std::printf("%-60s -> %s\n", #x, hipGetErrorName(e_)); } while (0)
int main() {
  hipStream_t st; SHOW(hipStreamCreate(&st));
  int* d; SHOW(hipMalloc(&d, 4));
  for (int variant = 0; variant < 3; ++variant) {
    std::printf("--- variant %d (%s)\n", variant, variant == 0 ? "capture, memset" : variant == 1 ? "capture, kernel" : "capture, kernel, events created with hipEventDefault flags + warm record");
    hipEvent_t a, b;
    if (variant == 2) { SHOW(hipEventCreateWithFlags(&a, hipEventDefault)); SHOW(hipEventCreateWithFlags(&b, hipEventDefault)); SHOW(hipEventRecord(a, st)); SHOW(hipEventRecord(b, st)); SHOW(hipStreamSynchronize(st)); }
    else { SHOW(hipEventCreate(&a)); SHOW(hipEventCreate(&b)); }
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    SHOW(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    SHOW(hipEventRecord(a, st));
    if (variant == 0) SHOW(hipMemsetAsync(d, 0, 4, st));
    else { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 200000); SHOW(hipGetLastError()); }
    SHOW(hipEventRecord(b, st));
    SHOW(hipStreamEndCapture(st, &g));
    size_t nn = 0; SHOW(hipGraphGetNodes(g, nullptr, &nn)); std::printf("nodes: %zu\n", nn);
    SHOW(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      SHOW(hipGraphLaunch(ge, st));
      SHOW(hipStreamSynchronize(st));
      SHOW(hipEventQuery(a)); SHOW(hipEventQuery(b));
      float ms = -1; SHOW(hipEventElapsedTime(&ms, a, b)); std::printf("elapsed %f ms\n", ms);
    }
  }
  {
    std::printf("--- variant 3 (explicit hipGraphAddEventRecordNode around a kernel node)\n");
    hipEvent_t a, b; SHOW(hipEventCreate(&a)); SHOW(hipEventCreate(&b));
    hipGraph_t g; SHOW(hipGraphCreate(&g, 0));
    hipGraphNode_t n0, nk, n1;
    SHOW(hipGraphAddEventRecordNode(&n0, g, nullptr, 0, a));
    int n = 200000;
    void* args[] = {&d, &n};
    hipKernelNodeParams kp = {};
    kp.func = reinterpret_cast<void*>(spin); kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
    SHOW(hipGraphAddKernelNode(&nk, g, &n0, 1, &kp));
    SHOW(hipGraphAddEventRecordNode(&n1, g, &nk, 1, b));
    hipGraphExec_t ge; SHOW(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      n = 200000 * (rep + 1);
      SHOW(hipGraphLaunch(ge, st));
      SHOW(hipStreamSynchronize(st));
      float ms = -1; SHOW(hipEventElapsedTime(&ms, a, b)); std::printf("elapsed %f ms\n", ms);
    }
    // eager reference
    SHOW(hipEventRecord(a, st)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 200000); SHOW(hipEventRecord(b, st)); SHOW(hipStreamSynchronize(st));
    float ms = -1; SHOW(hipEventElapsedTime(&ms, a, b)); std::printf("eager elapsed %f ms\n", ms);
  }
  return 0;
}
