// ktrace_conv.cpp -- phase timeline of workgroup 0 of kconv_kernel (development tool; built with -DLYC_TRACE), or -- built without it as
// benchmarks/kcbench, KT_TIME=1 -- the plain launch time of the library's patch kernel.
//   benchmarks/ktrace_conv B C H O [bwd]        (3x3, stride 1, pad 1, bf16, factor 8)
// Includes the library's translation unit so that the production host plans (plan_kconv) drive the launch.
#include "../lycoris_amd/csrc/capi.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

int main(int argc, char** argv) {
  const long B = argc > 1 ? atol(argv[1]) : 1;
  const int C = argc > 2 ? atoi(argv[2]) : 320, H = argc > 3 ? atoi(argv[3]) : 128, O = argc > 4 ? atoi(argv[4]) : 320;
  const bool bwd = argc > 5 && !strcmp(argv[5], "bwd");
  const int G = 8, c = O / G, d = C / G, taps = 9;
  if (getenv("KT_RAND")) srand(1);
  void *x, *y, *g, *dx, *pf, *pb, *ws; float *w1, *w2, *dw1;
  const size_t nx = (size_t)B * H * H * C * 2, ny = (size_t)B * H * H * O * 2;
  CK(hipMalloc(&x, nx)); CK(hipMalloc(&dx, nx)); CK(hipMalloc(&y, ny)); CK(hipMalloc(&g, ny));
  CK(hipMalloc(&w1, 256)); CK(hipMalloc(&dw1, 256)); CK(hipMalloc(&w2, (size_t)c * d * taps * 4)); CK(hipMalloc(&ws, 64 << 20));
  CK(hipMemset(x, 0x3c, nx)); CK(hipMemset(g, 0x3c, ny)); CK(hipMemset(w1, 0, 256)); CK(hipMemset(w2, 0, (size_t)c * d * taps * 4)); CK(hipMemset(dw1, 0, 256));
  const long bf = lyc_lokr_planes_bytes(c, d, taps, 0), bb = lyc_lokr_planes_bytes(c, d, taps, 1);
  CK(hipMalloc(&pf, bf)); CK(hipMalloc(&pb, bb));
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("pack: c=%d d=%d planes %ld + %ld bytes\n", c, d, bf, bb);
  if (lyc_lokr_pack_w2(w2, (long)d * taps, taps, 1, nullptr, 0, 0, nullptr, 0, 0, 0, 0, c, d, taps, pf, pb, LYC_BF16, nullptr)) { fprintf(stderr, "%s\n", lyc_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  printf("packed\n");
  printf("planes_ok fwd MI=%d bwd MI=%d\n", lyc_lokr_conv2d_planes_ok(B, H, H, G, G, c, d, 3, 3, 1, 1, 1, 1, 1, 1, LYC_BF16, 0),
         lyc_lokr_conv2d_planes_ok(B, H, H, G, G, c, d, 3, 3, 1, 1, 1, 1, 1, 1, LYC_BF16, 1));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const bool timing = getenv("KT_TIME") != nullptr;
  // KT_W4=1: the 4-wave workgroups of rounds 3 - 5; KT_SERIAL=1: 8 waves, serial k loop; KT_MI=2/4/8: pin the row tile
  const int DT = LYC_BF16 | (getenv("KT_W4") ? LYC_KCONV_W4 : 0) | (getenv("KT_SERIAL") ? LYC_KCONV_SERIAL : 0) | (getenv("KT_MI") ? LYC_KCONV_ROW_TILE(atoi(getenv("KT_MI"))) : 0);
  for (int rep = 0; rep < (timing ? 103 : 3); ++rep) {
    if (timing && rep == 3) CK(hipEventRecord(e0, 0));
    int rc;
    if (!bwd) rc = lyc_lokr_conv2d_fwd_planes(x, w1, pf, y, B, H, H, G, G, c, d, 3, 3, 1, 1, 1, 1, 1, 1, 1.0f, DT, nullptr);
    else rc = lyc_lokr_conv2d_bwd_planes(g, x, w1, nullptr, pb, dx, dw1, nullptr, ws, B, H, H, G, G, c, d, 3, 3, 1, 1, 1, 1, 1, 1, 1.0f, DT | LYC_DEFER_WGRAD, nullptr);
    if (rc) { fprintf(stderr, "%s\n", lyc_last_error()); return 1; }
#ifdef LYC_TRACE
    if (!timing) {
      CK(hipDeviceSynchronize());
      unsigned long long t[32];
      CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(lyc::lyc_trace_buf), sizeof(t)));
      printf("rep %d:", rep);
      for (int i = 0; i < 32; ++i) if (t[i]) printf(" [%d]+%llu", i, t[i] - t[0]);
      printf("\n");
    }
#endif
  }
  if (timing) {
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.2f us per launch (100 back-to-back launches, same buffers)\n", ms * 10.f);
  }
  return 0;
}
