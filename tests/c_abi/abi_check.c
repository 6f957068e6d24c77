/* The drop-in boundary from plain C: include/nnsdp.h must compile as C99 and the host-only entry points must be callable
 * without a GPU (the Julia ccall stub of INTEGRATION.md binds the same symbols).  Built and run by tests/test_host_api.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nnsdp.h"

int main(void) {
  int32_t xdims[4] = {2, 10, 10, 2};
  int32_t ncl = 0, total = 0;
  if (nnsdp_version() <= 0) return 1;
  if (nnsdp_make_cliques(3, xdims, 0, NNSDP_DECOMP_SINGLE, &ncl, &total, NULL, NULL) != 0) { printf("cliques: %s\n", nnsdp_last_error()); return 2; }
  int32_t* ptr = (int32_t*)malloc((size_t)(ncl + 1) * sizeof(int32_t));
  int32_t* idx = (int32_t*)malloc((size_t)total * sizeof(int32_t));
  if (nnsdp_make_cliques(3, xdims, 0, NNSDP_DECOMP_SINGLE, &ncl, &total, ptr, idx) != 0) return 3;
  /* one hidden-layer pair: cliques over z = [x_1 (2); x_2 (10); x_3 (10); 1] */
  printf("cliques %d total %d first %d last %d\n", (int)ncl, (int)total, (int)idx[0], (int)idx[total - 1]);
  /* interval pre-processing of a fixed tiny network: identity-like weights, positive box */
  double M[10 * 3 + 10 * 11 + 2 * 11];
  memset(M, 0, sizeof(M));
  for (int i = 0; i < 10; ++i) { M[(i % 2) * 10 + i] = 1.0; M[2 * 10 + i] = 0.1 * i - 0.3; }          /* layer 1: x_{i mod 2} + b_i */
  for (int i = 0; i < 10; ++i) M[30 + i * 10 + i] = 1.0;                                               /* layer 2: identity */
  M[30 + 110 + 0] = 1.0; M[30 + 110 + 2 + 1] = 1.0;                                                      /* output: first two */
  double lo[2] = {0.5, 0.5}, hi[2] = {1.5, 1.5};
  double acymin[20], acymax[20], smin[20], smax[20], ymin[2], ymax[2];
  if (nnsdp_make_intervals(3, xdims, M, lo, hi, acymin, acymax, NULL, NULL, smin, smax, ymin, ymax) != 0) { printf("intervals: %s\n", nnsdp_last_error()); return 4; }
  /* neuron 0 of layer 1 is x_0 - 0.3 on [0.5, 1.5]: [0.2, 1.2], always active */
  printf("acy0 [%.6f, %.6f] smin0 %.0f y0 [%.6f, %.6f]\n", acymin[0], acymax[0], smin[0], ymin[0], ymax[0]);
  if (!(acymin[0] > 0.19 && acymin[0] < 0.21 && acymax[0] > 1.19 && acymax[0] < 1.21 && smin[0] == 1.0)) return 5;
  if (nnsdp_make_cliques(0, xdims, 0, NNSDP_DECOMP_SINGLE, &ncl, &total, NULL, NULL) >= 0) return 6;   /* invalid argument -> negative */
  if (strlen(nnsdp_last_error()) == 0) return 7;
  nnsdp_options o;
  nnsdp_default_options(&o);
  if (!(o.max_iters > 0 && o.alpha > 0.0 && strcmp(nnsdp_status_string(NNSDP_STATUS_OPTIMAL), "OPTIMAL") == 0)) return 8;
  free(ptr); free(idx);
  printf("ok\n");
  return 0;
}
