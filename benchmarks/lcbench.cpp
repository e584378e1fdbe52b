// lcbench.cpp -- bneck_kernel (rounds 1-5, lowrank.h) vs bneck4_kernel (round 6, lowrank4.h) on the SDXL / SD1.5 LoCon Linear shapes:
// the new launch is checked against the old one (fp32 `mid` to 1e-5 of its range, 16-bit `out` to one unit in the last place of its
// range -- the two kernels sum K in different orders) and timed inside a hipGraph over rotating buffer sets (footprint > the 256 MiB
// Infinity Cache, so A / out stream from / to HBM as they do in a training step).
//   benchmarks/lcbench [filter] [--eager] [--trace]      (make -C lycoris_amd/csrc lcbench)
// Development tool: parity proper is tests/ (oracle); this only guards kernel-vs-kernel agreement while plans are tuned.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../lycoris_amd/csrc/lowrank4.h"

using namespace lyc;
#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float v = ((h & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
    unsigned u = __float_as_uint(v);
    p[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((h & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
  }
}
__global__ void empty_kernel(int* p) {
  if (p && threadIdx.x == 9999) *p = 1;
}

struct Shape {
  const char* tag;
  long M;
  int I, O;
};
static const Shape SHAPES[] = {
    {"attn1280", 1024, 1280, 1280}, {"ffup1280", 1024, 1280, 10240}, {"ffdn1280", 1024, 5120, 1280},
    {"attn640", 4096, 640, 640},    {"ffup640", 4096, 640, 5120},    {"ffdn640", 4096, 2560, 640},
    {"xattn1280", 77, 2048, 1280},  {"xattn640", 77, 2048, 640},     {"sd15_320", 16384, 320, 320},
    {"sd15_640", 4096, 640, 640},   {"temb", 1, 1280, 1280},         {"ragged", 1000, 328, 200},
};

static bool g_eager = false;
template <typename F>
static float bench(hipStream_t st, int nlaunch, int reps, F&& fn) {
  for (int i = 0; i < 2; ++i) fn(i);
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  if (!g_eager) {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nlaunch; ++i) fn(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
  }
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    if (g_eager) for (int i = 0; i < nlaunch; ++i) fn(i);
    else CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
  return best * 1e3f / nlaunch;
}
static long cdivl(long a, long b) { return (a + b - 1) / b; }

// the production plan of rounds 1-5 (capi.hip: launch_bneck_rt / launch_bneck_v), rank <= 16
static void launch_old(const BneckArgs& b0, bool fwd, hipStream_t st) {
  BneckArgs b = b0;
  const int mi = b.M >= 8192 ? 2 : 1;
  const int nw = (mi == 1 && b.K1 >= 8192) ? 8 : 4;
  const long rows = cdivl(b.M, 16 * mi);
  long ns = 256 / rows;
  if (ns > cdivl(b.N2, 16 * nw)) ns = cdivl(b.N2, 16 * nw);
  if (ns > 8) ns = 8;
  if (ns < 1) ns = 1;
  b.nsplit = (int)ns;
  const dim3 grid((unsigned)rows, (unsigned)ns);
  const bool vec = fwd && (b.R % 4) == 0 && (b.f1n % 4) == 0;
  if (mi == 2) {
    if (vec) hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 2, 1, true, true, false, true>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 2, 1, false, false>), grid, dim3(256), 0, st, b);
  } else if (nw == 8) {
    if (vec) hipLaunchKernelGGL((bneck_kernel<__bf16, 8, 1, 1, true, true>), grid, dim3(512), 0, st, b);
    else hipLaunchKernelGGL((bneck_kernel<__bf16, 8, 1, 1, false, false>), grid, dim3(512), 0, st, b);
  } else {
    if (vec) hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 1, 1, true, true, false, true>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((bneck_kernel<__bf16, 4, 1, 1, false, false>), grid, dim3(256), 0, st, b);
  }
}

typedef Bneck4Plan Plan;
// LCB_NW / LCB_MI / LCB_NS / LCB_D override the library's plan (tuning)
static Plan plan_new(long M, int K1, int N2) {
  Plan p{};
  if (!bneck4_make_plan(M, K1, N2, true, 1, p)) { fprintf(stderr, "no plan\n"); exit(1); }
  if (getenv("LCB_NW")) p.nw = atoi(getenv("LCB_NW"));
  if (getenv("LCB_MI")) p.mi = atoi(getenv("LCB_MI"));
  if (getenv("LCB_NS")) p.ns = atoi(getenv("LCB_NS"));
  if (getenv("LCB_NW") || getenv("LCB_NS") || getenv("LCB_MI")) p.D2 = (int)cdivl(cdivl(cdivl(N2, 32), p.ns), p.nw);
  if (getenv("LCB_D")) p.D = atoi(getenv("LCB_D"));
  if ((int)cdivl(cdivl(K1, 32), p.nw) < p.D) p.D = (int)cdivl(cdivl(K1, 32), p.nw);
  while (p.D > 1 && bneck4_lds_bytes(p.nw, p.mi, p.D, p.D2) > 160 * 1024) --p.D;
  p.lds = bneck4_lds_bytes(p.nw, p.mi, p.D, p.D2);
  if (p.lds > 160 * 1024) { fprintf(stderr, "plan does not fit the LDS\n"); exit(1); }
  return p;
}
template <bool FT>
static void launch_new(const Bneck4Args& a, const Plan& p, hipStream_t st) {
  const dim3 grid((unsigned)cdivl(a.M, 16 * p.mi), (unsigned)p.ns);
  if (p.nw == 16) {
    hipLaunchKernelGGL((bneck4_kernel<__bf16, 16, 1, FT>), grid, dim3(1024), p.lds, st, a);
  } else if (p.nw == 8) {
    if (p.mi == 2) hipLaunchKernelGGL((bneck4_kernel<__bf16, 8, 2, FT>), grid, dim3(512), p.lds, st, a);
    else hipLaunchKernelGGL((bneck4_kernel<__bf16, 8, 1, FT>), grid, dim3(512), p.lds, st, a);
  } else {
    if (p.mi == 2) hipLaunchKernelGGL((bneck4_kernel<__bf16, 4, 2, FT>), grid, dim3(256), p.lds, st, a);
    else hipLaunchKernelGGL((bneck4_kernel<__bf16, 4, 1, FT>), grid, dim3(256), p.lds, st, a);
  }
}

template <bool FT>
static void launch_group(const Bneck4GroupArgs& ga, const Plan& p, long M, hipStream_t st) {
  const dim3 grid((unsigned)cdivl(M, 16 * p.mi), (unsigned)p.ns, (unsigned)ga.n);
  if (p.nw == 8) hipLaunchKernelGGL((bneck4_group_kernel<__bf16, 8, 1, FT>), grid, dim3(512), p.lds, st, ga);
  else hipLaunchKernelGGL((bneck4_group_kernel<__bf16, 4, 2, FT>), grid, dim3(256), p.lds, st, ga);
}
template <bool FT>
static void launch_sum(const Bneck4GroupArgs& ga, void* out_sum, const Plan& p, long M, hipStream_t st) {
  const dim3 grid((unsigned)cdivl(M, 16 * p.mi), (unsigned)p.ns);
  if (p.nw == 8) hipLaunchKernelGGL((bneck4_sum_kernel<__bf16, 8, 1, FT>), grid, dim3(512), p.lds, st, ga, out_sum);
  else hipLaunchKernelGGL((bneck4_sum_kernel<__bf16, 4, 2, FT>), grid, dim3(256), p.lds, st, ga, out_sum);
}
__global__ void sum_bf16(const unsigned short* a, const unsigned short* b, const unsigned short* c, unsigned short* o, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float v = __uint_as_float((unsigned)a[i] << 16) + __uint_as_float((unsigned)b[i] << 16) + (c ? __uint_as_float((unsigned)c[i] << 16) : 0.f);
    unsigned u = __float_as_uint(v);
    o[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}

static float bf2f(unsigned short v) {
  unsigned u = (unsigned)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const char* filter = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--eager")) g_eager = true;
    else filter = argv[i];
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
#define ATTR(NW_, MI_, FT_) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck4_kernel<__bf16, NW_, MI_, FT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  ATTR(4, 1, false); ATTR(4, 1, true); ATTR(4, 2, false); ATTR(4, 2, true); ATTR(8, 1, false); ATTR(8, 1, true); ATTR(8, 2, false); ATTR(8, 2, true); ATTR(16, 1, false); ATTR(16, 1, true);
#define ATTRG(K_, NW_, MI_) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K_<__bf16, NW_, MI_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  ATTRG(bneck4_group_kernel, 8, 1); ATTRG(bneck4_group_kernel, 4, 2); ATTRG(bneck4_sum_kernel, 8, 1); ATTRG(bneck4_sum_kernel, 4, 2);
  {
    float us = bench(st, 200, 5, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, nullptr); });
    printf("# empty 256-workgroup launch inside a graph: %.2f us\n", us);
  }
  const int R = getenv("LCB_R") ? atoi(getenv("LCB_R")) : 16;
  for (const Shape& s : SHAPES) {
    if (filter && !strstr(s.tag, filter)) continue;
    const long M = s.M;
    const int I = s.I, O = s.O;
    const size_t xb = (size_t)M * I * 2, yb = (size_t)M * O * 2;
    int nsets = (int)((600u << 20) / (xb + yb)) + 1;
    if (nsets > 64) nsets = 64;
    if (nsets < 2) nsets = 2;
    struct Set { void *x, *g, *y, *dx; float *t, *dt; };
    std::vector<Set> sets(nsets);
    for (int i = 0; i < nsets; ++i) {
      Set& z = sets[i];
      CK(hipMalloc(&z.x, xb + 256)); CK(hipMalloc(&z.g, yb + 256)); CK(hipMalloc(&z.y, yb + 256)); CK(hipMalloc(&z.dx, xb + 256));
      CK(hipMalloc(&z.t, (size_t)M * R * 4 + 256)); CK(hipMalloc(&z.dt, (size_t)M * R * 4 + 256));
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)z.x, xb / 2, 11u + i, 1.0f);
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)z.g, yb / 2, 77u + i, 0.05f);
    }
    float *down, *up;
    CK(hipMalloc(&down, (size_t)R * I * 4)); CK(hipMalloc(&up, (size_t)O * R * 4));
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, down, (size_t)R * I, 5u, 0.05f);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, up, (size_t)O * R, 6u, 0.2f);
    const size_t big = xb > yb ? xb : yb;
    void *oref, *onew; float *mref, *mnew;
    CK(hipMalloc(&oref, big + 256)); CK(hipMalloc(&onew, big + 256));
    CK(hipMalloc(&mref, (size_t)M * R * 4)); CK(hipMalloc(&mnew, (size_t)M * R * 4));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned short> h_ref(big / 2), h_new(big / 2);
    std::vector<float> hm_ref((size_t)M * R), hm_new((size_t)M * R);

    for (int mode = 0; mode < 2; ++mode) {  // 0 forward (t = x down^T, y = alpha t up^T), 1 backward (dt = alpha g up, dx = dt down)
      const bool bw = mode == 1;
      const int K1 = bw ? O : I, N2 = bw ? I : O;
      const size_t outb = bw ? xb : yb;
      const double bytes = (double)xb + yb;
      auto old_args = [&](const void* in, float* mid, void* out) {
        BneckArgs b{};
        if (!bw) { b.A = in; b.lda = I; b.K1 = I; b.F1 = down; b.f1n = I; b.f1k = 1; b.F2 = up; b.f2n = R; b.f2k = 1; b.N2 = O; b.out = out; b.ldo = O; b.alpha1 = 1.f; b.alpha2 = 0.5f; }
        else { b.A = in; b.lda = O; b.K1 = O; b.F1 = up; b.f1n = 1; b.f1k = R; b.F2 = down; b.f2n = 1; b.f2k = I; b.N2 = I; b.out = out; b.ldo = I; b.alpha1 = 0.5f; b.alpha2 = 1.f; }
        b.M = M; b.R = R; b.mid = mid;
        return b;
      };
      auto new_args = [&](const void* in, float* mid, void* out, const Plan& p) {
        Bneck4Args a{};
        a.A = in; a.mid = mid; a.out = out; a.M = (int)M; a.R = R; a.K1 = K1; a.KS = (int)cdivl(K1, 32); a.N2 = N2; a.lda = K1; a.ldo = N2;
        a.a_bytes = (unsigned)((size_t)M * K1 * 2); a.out_bytes = (unsigned)((size_t)M * N2 * 2);
        if (!bw) { a.F1 = down; a.F2 = up; a.alpha1 = 1.f; a.alpha2 = 0.5f; }
        else { a.F1 = up; a.F2 = down; a.alpha1 = 0.5f; a.alpha2 = 1.f; }
        a.f1_bytes = (unsigned)((size_t)R * K1 * 4); a.f2_bytes = (unsigned)((size_t)R * N2 * 4);
        a.D = p.D; a.D2 = p.D2;
        return a;
      };
      const Plan p = plan_new(M, K1, N2);
      // ---- agreement ----
      CK(hipMemsetAsync(oref, 0xff, outb, st)); CK(hipMemsetAsync(onew, 0xee, outb, st));
      launch_old(old_args(bw ? sets[0].g : sets[0].x, mref, oref), !bw, st);
#ifdef LYC_TRACE
      CK(hipStreamSynchronize(st));
      {
        unsigned long long h[32];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(lyc_trace_buf), sizeof(h)));
        printf("  trace old:");
        for (int i = 1; i < 32; ++i) if (h[i]) printf(" [%d]+%llu", i, h[i] - h[0]);
        printf("\n");
        memset(h, 0, sizeof(h));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(lyc_trace_buf), h, sizeof(h)));
      }
#endif
      if (bw) launch_new<true>(new_args(sets[0].g, mnew, onew, p), p, st);
      else launch_new<false>(new_args(sets[0].x, mnew, onew, p), p, st);
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
#ifdef LYC_TRACE
      {
        unsigned long long h[32];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(lyc_trace_buf), sizeof(h)));
        printf("  trace new:");
        for (int i = 1; i < 32; ++i) if (h[i]) printf(" [%d]+%llu", i, h[i] - h[0]);
        printf("\n");
      }
#endif
      CK(hipMemcpy(h_ref.data(), oref, outb, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_new.data(), onew, outb, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hm_ref.data(), mref, (size_t)M * R * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm_new.data(), mnew, (size_t)M * R * 4, hipMemcpyDeviceToHost));
      double mmax = 0, mdiff = 0, omax = 0, odiff = 0;
      for (size_t i = 0; i < (size_t)M * R; ++i) { mmax = fmax(mmax, fabs(hm_ref[i])); mdiff = fmax(mdiff, fabs((double)hm_ref[i] - hm_new[i])); }
      for (size_t i = 0; i < outb / 2; ++i) { const double r = bf2f(h_ref[i]), n = bf2f(h_new[i]); omax = fmax(omax, fabs(r)); odiff = fmax(odiff, fabs(r - n)); if (n != n) odiff = 1e30; }
      const bool ok = mdiff <= 2e-5 * mmax + 1e-12 && odiff <= omax / 128.0 + 1e-12;
      // ---- time ----
      const int nl = nsets * 2;
      const float t_old = bench(st, nl, 5, [&](int i) { const Set& z = sets[i % nsets]; launch_old(old_args(bw ? z.g : z.x, bw ? z.dt : z.t, bw ? z.dx : z.y), !bw, st); });
      const float t_new = bench(st, nl, 5, [&](int i) {
        const Set& z = sets[i % nsets];
        if (bw) launch_new<true>(new_args(z.g, z.dt, z.dx, p), p, st);
        else launch_new<false>(new_args(z.x, z.t, z.y, p), p, st);
      });
      printf("%-10s M=%-6ld %5d->%-5d %s | old %7.2f us %7.1f GB/s | new W%d MI%d ns%d D%d D2 %d lds %3dK wgs %4ld %7.2f us %7.1f GB/s x%.2f | mid %.1e/%.1e out %.1e/%.1e %s\n",
             s.tag, M, I, O, bw ? "bwd" : "fwd", t_old, bytes / t_old * 1e-3, p.nw, p.mi, p.ns, p.D, p.D2, p.lds >> 10, cdivl(M, 16 * p.mi) * p.ns, t_new, bytes / t_new * 1e-3,
             t_old / t_new, mdiff, mmax, odiff, omax, ok ? "ok" : "MISMATCH");
      fflush(stdout);
    }
    // ---- the input gradient of a tensor that n sibling projections read: n problems + a summation pass vs bneck4_sum_kernel ----
    const int nsib = getenv("LCB_SUM") ? atoi(getenv("LCB_SUM")) : 0;
    if (nsib >= 2 && nsib <= 4 && nsets >= nsib) {
      float* ups[4]; float* downs[4];
      for (int i = 0; i < nsib; ++i) {
        CK(hipMalloc(&ups[i], (size_t)O * R * 4)); CK(hipMalloc(&downs[i], (size_t)R * I * 4));
        hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, downs[i], (size_t)R * I, 50u + i, 0.05f);
        hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, ups[i], (size_t)O * R, 60u + i, 0.2f);
      }
      const int K1 = O, N2 = I;
      Plan pg{}, ps{};
      if (!bneck4_make_plan(M, K1, N2, true, nsib, pg) || !bneck4_make_plan(M, K1, N2, true, 1, ps, nsib)) { fprintf(stderr, "no plan\n"); exit(1); }
      auto garg = [&](int set0, const Plan& p, bool own_out) {
        Bneck4GroupArgs ga{};
        ga.n = nsib;
        for (int i = 0; i < nsib; ++i) {
          const Set& z = sets[(set0 + i) % nsets];
          Bneck4Args& a = ga.p[i];
          a.A = z.g; a.mid = z.dt; a.out = own_out ? z.dx : oref; a.M = (int)M; a.R = R; a.K1 = K1; a.KS = (int)cdivl(K1, 32); a.N2 = N2; a.lda = K1; a.ldo = N2;
          a.a_bytes = (unsigned)((size_t)M * K1 * 2); a.out_bytes = (unsigned)((size_t)M * N2 * 2);
          a.F1 = ups[i]; a.F2 = downs[i]; a.alpha1 = 0.5f; a.alpha2 = 1.f;
          a.f1_bytes = (unsigned)((size_t)R * K1 * 4); a.f2_bytes = (unsigned)((size_t)R * N2 * 4);
          a.D = p.D; a.D2 = p.D2;
        }
        return ga;
      };
      // agreement: group launch + sum of the rounded results vs the fused sum (one rounding): a few units in the last place of the range
      launch_group<true>(garg(0, pg, true), pg, M, st);
      hipLaunchKernelGGL(sum_bf16, dim3(1024), dim3(256), 0, st, (const unsigned short*)sets[0].dx, (const unsigned short*)sets[1 % nsets].dx,
                         nsib > 2 ? (const unsigned short*)sets[2 % nsets].dx : nullptr, (unsigned short*)oref, xb / 2);
      CK(hipMemcpyAsync(hm_ref.data(), sets[1 % nsets].dt, (size_t)M * R * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      CK(hipMemsetAsync(sets[1 % nsets].dt, 0xff, (size_t)M * R * 4, st));
      launch_sum<true>(garg(0, ps, true), onew, ps, M, st);
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
      CK(hipMemcpy(h_ref.data(), oref, xb, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_new.data(), onew, xb, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hm_new.data(), sets[1 % nsets].dt, (size_t)M * R * 4, hipMemcpyDeviceToHost));
      double omax = 0, odiff = 0, mdiff = 0;
      for (size_t i = 0; i < xb / 2; ++i) { const double r = bf2f(h_ref[i]), n = bf2f(h_new[i]); omax = fmax(omax, fabs(r)); odiff = fmax(odiff, fabs(r - n)); if (n != n) odiff = 1e30; }
      for (size_t i = 0; i < (size_t)M * R; ++i) mdiff = fmax(mdiff, fabs((double)hm_ref[i] - hm_new[i]));
      const int nl = nsets * 2;
      const float t_grp = bench(st, nl, 5, [&](int i) { launch_group<true>(garg(i * nsib, pg, true), pg, M, st); });
      const float t_add = bench(st, nl, 5, [&](int i) {
        hipLaunchKernelGGL(sum_bf16, dim3(1024), dim3(256), 0, st, (const unsigned short*)sets[i % nsets].dx, (const unsigned short*)sets[(i + 1) % nsets].dx,
                           nsib > 2 ? (const unsigned short*)sets[(i + 2) % nsets].dx : nullptr, (unsigned short*)sets[(i + 3) % nsets].x, xb / 2);
      });
      const float t_sum = bench(st, nl, 5, [&](int i) { launch_sum<true>(garg(i * nsib, ps, true), sets[(i * nsib + 1) % nsets].dx, ps, M, st); });
      printf("%-10s M=%-6ld %5d->%-5d dx of %d siblings | group (ns%d D%d D2 %d) %7.2f us + sum pass %6.2f us | fused ns%d D%d D2 %d lds %3dK %7.2f us x%.2f | out %.1e/%.1e mid %.1e %s\n",
             s.tag, M, I, O, nsib, pg.ns, pg.D, pg.D2, t_grp, t_add, ps.ns, ps.D, ps.D2, ps.lds >> 10, t_sum, (t_grp + t_add) / t_sum, odiff, omax, mdiff,
             odiff <= omax / 64.0 && mdiff <= 1e-6 ? "ok" : "MISMATCH");
      for (int i = 0; i < nsib; ++i) { CK(hipFree(ups[i])); CK(hipFree(downs[i])); }
    }
    for (Set& z : sets) { CK(hipFree(z.x)); CK(hipFree(z.g)); CK(hipFree(z.y)); CK(hipFree(z.dx)); CK(hipFree(z.t)); CK(hipFree(z.dt)); }
    CK(hipFree(down)); CK(hipFree(up)); CK(hipFree(oref)); CK(hipFree(onew)); CK(hipFree(mref)); CK(hipFree(mnew));
  }
  return 0;
}
