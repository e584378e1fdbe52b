// ktrace.cpp -- phase timeline of one workgroup of a kernel (development tool).
// Built with -DLYC_TRACE so that the LYC_STAMP() points of the kernel headers record the shader clock.
//   benchmarks/ktrace M I O [fwd|bwd|dw2 | lfwd|lbwd (rank-16 bneck_kernel; KT_NS = column slices, KT_NW = 4|8)]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../lycoris_amd/csrc/kron3.h"
#include "../lycoris_amd/csrc/kron_dw2s.h"
#include "../lycoris_amd/csrc/kron_conv.h"
#include "../lycoris_amd/csrc/lowrank.h"

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
      static void* planes = nullptr;  // KT_PL=1: the pre-packed operand planes (round 3), packed once
      const bool pl = getenv("KT_PL") != nullptr;
      if (pl && planes == nullptr) {
        const long nf = kron_plane_bytes(c, 1, d), nb = kron_plane_bytes(d, 1, c);
        CK(hipMalloc(&planes, nf + nb));
        KronPackArgs pa{};
        pa.w2 = w2; pa.sq = d; pa.sv = 1; pa.st = 0; pa.c = c; pa.d = d; pa.taps = 1; pa.fwd = planes; pa.bwd = (char*)planes + nf;
        pa.units_fwd = nf / 2048;
        hipLaunchKernelGGL((kron_pack_kernel<__bf16>), dim3((unsigned)cdiv((nf + nb) / 2048, NWAVES)), dim3(NTHREADS), 0, 0, pa);
        CK(hipDeviceSynchronize());
      }
      if (pl) ka.w2p = bw ? (char*)planes + kron_plane_bytes(c, 1, d) : planes;
      const int ni = getenv("KT_NI") ? atoi(getenv("KT_NI")) : 2;   // production picks 2 for the 1280-wide layers
      const int gm = getenv("KT_GM") ? atoi(getenv("KT_GM")) : 3;   // 3 = x through the per-wave LDS stage (production)
      dim3 grid((unsigned)cdiv(M, K3_RT / G), (unsigned)cdiv(ka.N, 16 * ni));
      const long nseg = cdiv(ka.K, kron3_kc(ni));
      const int lds = kron3_lds_bytes(ni, nseg > 1 ? 2 : 1) + (gm == 3 ? kron3_xs_bytes() : 0);
      auto go = [&](auto kern) {
        if (rep == 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, 0, ka);
        CK(hipGetLastError());
      };
      if (ni == 1 && gm == 3) { if (bw) go(kron3_kernel<__bf16, 1, true, 3>); else go(kron3_kernel<__bf16, 1, false, 3>); }
      else if (ni == 1) { if (bw) go(kron3_kernel<__bf16, 1, true, 0>); else go(kron3_kernel<__bf16, 1, false, 0>); }
      else if (ni == 4 && gm == 3) { if (bw) go(kron3_kernel<__bf16, 4, true, 3>); else go(kron3_kernel<__bf16, 4, false, 3>); }
      else if (ni == 4) { if (bw) go(kron3_kernel<__bf16, 4, true, 0>); else go(kron3_kernel<__bf16, 4, false, 0>); }
      else if (gm == 3 && pl) { if (bw) go(kron3_kernel<__bf16, 2, true, 3, false, true>); else go(kron3_kernel<__bf16, 2, false, 3, false, true>); }
      else if (gm == 3) { if (bw) go(kron3_kernel<__bf16, 2, true, 3>); else go(kron3_kernel<__bf16, 2, false, 3>); }
      else { if (bw) go(kron3_kernel<__bf16, 2, true, 0>); else go(kron3_kernel<__bf16, 2, false, 0>); }
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
    if (!strcmp(mode, "lfwd") || !strcmp(mode, "lbwd")) {
      const int r = 16;
      static float *dn = nullptr, *up = nullptr, *mid = nullptr;
      if (!dn) { CK(hipMalloc(&dn, (size_t)r * I * 4)); CK(hipMalloc(&up, (size_t)r * O * 4)); CK(hipMalloc(&mid, (size_t)M * r * 4));
                 CK(hipMemset(dn, 0, (size_t)r * I * 4)); CK(hipMemset(up, 0, (size_t)r * O * 4)); }
      const bool bw = !strcmp(mode, "lbwd");
      BneckArgs b{};
      if (!bw) { b.A = x; b.lda = I; b.K1 = I; b.F1 = dn; b.f1n = I; b.f1k = 1; b.F2 = up; b.f2n = r; b.f2k = 1; b.N2 = O; b.out = y; b.ldo = O; }
      else { b.A = g; b.lda = O; b.K1 = O; b.F1 = up; b.f1n = 1; b.f1k = r; b.F2 = dn; b.f2n = 1; b.f2k = I; b.N2 = I; b.out = dx; b.ldo = I; }
      b.M = M; b.R = r; b.mid = mid; b.alpha1 = b.alpha2 = 1.f;
      const int ns = getenv("KT_NS") ? atoi(getenv("KT_NS")) : 1, nw = getenv("KT_NW") ? atoi(getenv("KT_NW")) : 8;
      b.nsplit = ns;
      dim3 grid((unsigned)cdiv(M, 16), (unsigned)ns);
      if (nw == 8) {
        if (!bw) hipLaunchKernelGGL((bneck_kernel<__bf16, 8, 1, 1, true, true>), grid, dim3(512), 0, 0, b);
        else hipLaunchKernelGGL((bneck_kernel<__bf16, 8, 1, 1, false, false>), grid, dim3(512), 0, 0, b);
      } else {
        if (!bw && getenv("KT_STG")) hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 1, 1, true, true, false, true>), grid, dim3(256), 0, 0, b);  // production forward
        else if (!bw) hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 1, 1, true, true>), grid, dim3(256), 0, 0, b);
        else hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 1, 1, false, false>), grid, dim3(256), 0, 0, b);
      }
      if (rep == 0) printf("grid %u x %u, %d waves\n", grid.x, grid.y, nw);
    }
    if (!strcmp(mode, "ltn")) {  // both factor gradients (KT_TIME only; KT_CV = 2|4|8, KT_SPLIT = row slabs)
      const int r = 16;
      static float *t = nullptr, *dtm = nullptr, *dd = nullptr, *du = nullptr;
      if (!t) { CK(hipMalloc(&t, (size_t)M * r * 4)); CK(hipMalloc(&dtm, (size_t)M * r * 4)); CK(hipMalloc(&dd, (size_t)r * I * 4)); CK(hipMalloc(&du, (size_t)r * O * 4));
                CK(hipMemset(t, 0, (size_t)M * r * 4)); CK(hipMemset(dtm, 0, (size_t)M * r * 4)); }
      const int cv = getenv("KT_CV") ? atoi(getenv("KT_CV")) : 2, split = getenv("KT_SPLIT") ? atoi(getenv("KT_SPLIT")) : 14;
      LowrankTnArgs a{};
      a.M = M; a.R = r;
      a.p[0].act = g; a.p[0].ld = O; a.p[0].C = O; a.p[0].mid = t; a.p[0].out = du; a.p[0].os = r; a.p[0].oj = 1; a.p[0].alpha = 1.f;
      a.p[1].act = x; a.p[1].ld = I; a.p[1].C = I; a.p[1].mid = dtm; a.p[1].out = dd; a.p[1].os = 1; a.p[1].oj = I; a.p[1].alpha = 1.f;
      a.p[0].tiles = (int)cdiv(O, 16 * cv); a.p[1].tiles = (int)cdiv(I, 16 * cv);
      a.rows_per_slab = cdiv(cdiv(M, split), 4) * 4; a.nsplit = (int)cdiv(M, a.rows_per_slab);
      dim3 grid((unsigned)cdiv((long)(a.p[0].tiles + a.p[1].tiles) * a.nsplit, 4));
      if (cv == 8) hipLaunchKernelGGL((lowrank_tn_kernel<__bf16, 1, 8>), grid, dim3(256), 0, 0, a);
      else if (cv == 4) hipLaunchKernelGGL((lowrank_tn_kernel<__bf16, 1, 4>), grid, dim3(256), 0, 0, a);
      else if (cv == 1) hipLaunchKernelGGL((lowrank_tn_kernel<__bf16, 1, 1>), grid, dim3(256), 0, 0, a);
      else hipLaunchKernelGGL((lowrank_tn_kernel<__bf16, 1, 2>), grid, dim3(256), 0, 0, a);
      if (rep == 0) printf("grid %u (cv %d, %d slabs of %ld rows)\n", grid.x, cv, a.nsplit, a.rows_per_slab);
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
