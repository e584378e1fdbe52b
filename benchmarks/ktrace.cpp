// ktrace.cpp -- phase timeline of one workgroup of a kernel (development tool).
// Built with -DLYC_TRACE so that the LYC_STAMP() points of the kernel headers record the shader clock.
//   benchmarks/ktrace M I O [fwd|bwd|dw2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../lycoris_amd/csrc/kron3.h"
#include "../lycoris_amd/csrc/kron_dw2s.h"

using namespace lyc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 1024;
  const int I = argc > 2 ? atoi(argv[2]) : 1280, O = argc > 3 ? atoi(argv[3]) : 1280;
  const char* mode = argc > 4 ? argv[4] : "fwd";
  const int G = 8, c = O / G, d = I / G;
  void *x, *g, *y, *dx; float *w1, *w2, *dw1, *dw2;
  CK(hipMalloc(&x, M * I * 2)); CK(hipMalloc(&dx, M * I * 2)); CK(hipMalloc(&g, M * O * 2)); CK(hipMalloc(&y, M * O * 2));
  CK(hipMalloc(&w1, 256)); CK(hipMalloc(&dw1, 256)); CK(hipMalloc(&w2, (size_t)c * d * 4)); CK(hipMalloc(&dw2, (size_t)c * d * 4));
  CK(hipMemset(x, 0x3c, M * I * 2)); CK(hipMemset(g, 0x3c, M * O * 2)); CK(hipMemset(w1, 0, 256)); CK(hipMemset(w2, 0, (size_t)c * d * 4));
  CK(hipMemset(dw1, 0, 256)); CK(hipMemset(dw2, 0, (size_t)c * d * 4));
  auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
  const bool timing = getenv("KT_TIME") != nullptr;  // time 200 back-to-back launches instead of tracing one
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < (timing ? 203 : 3); ++rep) {
    if (timing && rep == 3) CK(hipEventRecord(e0, 0));
    if (!strcmp(mode, "fwd") || !strcmp(mode, "bwd")) {
      KronArgs ka{};
      const bool bw = !strcmp(mode, "bwd");
      if (!bw) { ka.x = x; ka.y = y; ka.w1 = w1; ka.w2 = w2; ka.M = M; ka.Gin = G; ka.K = d; ka.Gout = G; ka.N = c; ka.s1o = G; ka.s1i = 1; ka.s2n = d; ka.s2k = 1; }
      else { ka.x = g; ka.y = dx; ka.w1 = w1; ka.w2 = w2; ka.dw1 = dw1; ka.xref = x; ka.M = M; ka.Gin = G; ka.K = c; ka.Gout = G; ka.N = d; ka.s1o = 1; ka.s1i = G; ka.s2n = 1; ka.s2k = d; }
      ka.alpha = 1.f;
      const int ni = getenv("LYC_K3_NI") ? atoi(getenv("LYC_K3_NI")) : 4;
      dim3 grid((unsigned)cdiv(M, K3_RT / G), (unsigned)cdiv(ka.N, 16 * ni));
      if (ni == 4) {
        if (bw) hipLaunchKernelGGL((kron3_kernel<__bf16, 4, true, 0>), grid, dim3(NTHREADS), kron3_lds_bytes(4, 2), 0, ka);
        else hipLaunchKernelGGL((kron3_kernel<__bf16, 4, false, 0>), grid, dim3(NTHREADS), kron3_lds_bytes(4, 2), 0, ka);
      } else {
        if (bw) hipLaunchKernelGGL((kron3_kernel<__bf16, 2, true, 0>), grid, dim3(NTHREADS), kron3_lds_bytes(2, 2), 0, ka);
        else hipLaunchKernelGGL((kron3_kernel<__bf16, 2, false, 0>), grid, dim3(NTHREADS), kron3_lds_bytes(2, 2), 0, ka);
      }
      if (rep == 0) printf("grid %u x %u\n", grid.x, grid.y);
    }
    if (!strcmp(mode, "dw2")) {
      KronDw2sArgs da{};
      da.Q = g; da.P = x; da.W = w1; da.out = dw2; da.M = M; da.G = G; da.I = c; da.J = d; da.ws = G; da.wt = 1; da.os = d; da.alpha = 1.f;
      const long rows_total = M * G;
      const long tiles = cdiv(c, 32) * cdiv(d, 32);
      long split = cdiv(512, tiles), smax = 600000 / ((long)c * d); if (smax < 1) smax = 1;
      if (split > smax) split = smax;
      if (split > rows_total / 128) split = rows_total / 128 > 0 ? rows_total / 128 : 1;
      da.rows_per_block = cdiv(cdiv(rows_total, split), 32) * 32;
      da.nsplit = (int)cdiv(rows_total, da.rows_per_block);
      da.tiles_i = (int)cdiv(c, 32); da.tiles_j = (int)cdiv(d, 32);
      dim3 grid((unsigned)(((da.tiles_i * da.tiles_j * da.nsplit + 7) / 8) * 8));
      hipLaunchKernelGGL((kron_dw2s_kernel<__bf16, 2, 2, 4, false>), grid, dim3(NTHREADS), 0, 0, da);
      if (rep == 0) printf("grid %u rows/block %ld\n", grid.x, da.rows_per_block);
    }
    if (timing) continue;
#ifdef LYC_TRACE
    CK(hipDeviceSynchronize());
    unsigned long long h[32];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(lyc_trace_buf), sizeof(h)));
    printf("rep %d:", rep);
    unsigned long long prev = h[0];
    for (int i = 0; i < 32; ++i) if (h[i]) { printf(" [%d]+%llu", i, h[i] - h[0]); prev = h[i]; }
    printf("\n");
    unsigned long long z[32] = {0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(lyc_trace_buf), z, sizeof(z)));
#endif
  }
  if (timing) {
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.2f us per launch (200 eager back-to-back launches, same buffers)\n", ms * 1e3f / 200);
  }
  return 0;
}
