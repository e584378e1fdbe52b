// lg16bench.cpp -- HadaWeight.backward (reference lycoris/functional/loha.py:18-30) on a dense fp32 gradient G:
//   loha_factor_grad_mfma_kernel (loha_mfma.h: fp32 matrix core)  vs  loha_factor_grad16_kernel (loha_grad16.h: bf16 hi / lo split)
// Both are checked against an fp64 host evaluation (norm-wise relative error per factor gradient) on small and ragged layers, then
// timed on SDXL layer shapes (one layer per launch and a 24-layer grouped launch, rotating G buffers beyond the Infinity Cache).
//   benchmarks/lg16bench            (make -C lycoris_amd/csrc lg16bench)
// Development tool: parity proper is tests/ (oracle).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../lycoris_amd/csrc/loha_grad16.h"

using namespace lyc;
#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

static unsigned rng_state = 12345u;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) / 16777216.0f - 0.5f) * 2.0f;
}
static long cdivl(long a, long b) { return (a + b - 1) / b; }

struct Layer {
  long O, I;
  int R;
  std::vector<float> w1a, w1b, w2a, w2b, G;
  float *d_w1a, *d_w1b, *d_w2a, *d_w2b, *d_G;       // device inputs
  float *g_w1a, *g_w1b, *g_w2a, *g_w2b;             // device gradients
};

static void make_layer(Layer& L, long O, long I, int R, bool host_ref) {
  L.O = O; L.I = I; L.R = R;
  L.w1a.resize(O * R); L.w2a.resize(O * R); L.w1b.resize(R * I); L.w2b.resize(R * I); L.G.resize(O * I);
  for (auto& v : L.w1a) v = 0.1f * frand();
  for (auto& v : L.w2a) v = 0.1f * frand();
  for (auto& v : L.w1b) v = frand();
  for (auto& v : L.w2b) v = frand();
  for (auto& v : L.G) v = 1e-2f * frand();
  auto up = [](const std::vector<float>& h) {
    float* d;
    CK(hipMalloc(&d, h.size() * 4 + 64));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
  };
  L.d_w1a = up(L.w1a); L.d_w2a = up(L.w2a); L.d_w1b = up(L.w1b); L.d_w2b = up(L.w2b); L.d_G = up(L.G);
  CK(hipMalloc(&L.g_w1a, O * R * 4 + 64)); CK(hipMalloc(&L.g_w2a, O * R * 4 + 64));
  CK(hipMalloc(&L.g_w1b, R * I * 4 + 64)); CK(hipMalloc(&L.g_w2b, R * I * 4 + 64));
  (void)host_ref;
}
static void zero_grads(Layer& L) {
  CK(hipMemset(L.g_w1a, 0, L.O * L.R * 4)); CK(hipMemset(L.g_w2a, 0, L.O * L.R * 4));
  CK(hipMemset(L.g_w1b, 0, L.R * L.I * 4)); CK(hipMemset(L.g_w2b, 0, L.R * L.I * 4));
}
static LohaArgs args_of(const Layer& L, float scale) {
  LohaArgs a{};
  a.w1a = L.d_w1a; a.w1b = L.d_w1b; a.w2a = L.d_w2a; a.w2b = L.d_w2b; a.G = L.d_G;
  a.d_w1a = L.g_w1a; a.d_w1b = L.g_w1b; a.d_w2a = L.g_w2a; a.d_w2b = L.g_w2b;
  a.O = L.O; a.I = L.I; a.R = L.R; a.scale = scale;
  return a;
}

// fp64 host evaluation
static void host_ref(const Layer& L, float scale, std::vector<double>& r1a, std::vector<double>& r1b, std::vector<double>& r2a,
                     std::vector<double>& r2b) {
  const long O = L.O, I = L.I;
  const int R = L.R;
  r1a.assign(O * R, 0); r2a.assign(O * R, 0); r1b.assign(R * I, 0); r2b.assign(R * I, 0);
  for (long o = 0; o < O; ++o)
    for (long i = 0; i < I; ++i) {
      double p1 = 0, p2 = 0;
      for (int r = 0; r < R; ++r) {
        p1 += (double)L.w1a[o * R + r] * L.w1b[r * I + i];
        p2 += (double)L.w2a[o * R + r] * L.w2b[r * I + i];
      }
      const double gs = (double)L.G[o * I + i] * scale, t1 = gs * p2, t2 = gs * p1;
      for (int r = 0; r < R; ++r) {
        r1a[o * R + r] += t1 * L.w1b[r * I + i];
        r1b[r * I + i] += t1 * L.w1a[o * R + r];
        r2a[o * R + r] += t2 * L.w2b[r * I + i];
        r2b[r * I + i] += t2 * L.w2a[o * R + r];
      }
    }
}
static double rel_err(const float* dev, const std::vector<double>& ref) {
  std::vector<float> h(ref.size());
  CK(hipMemcpy(h.data(), dev, ref.size() * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t k = 0; k < ref.size(); ++k) {
    num += ((double)h[k] - ref[k]) * ((double)h[k] - ref[k]);
    den += ref[k] * ref[k];
  }
  return sqrt(num / (den > 0 ? den : 1));
}

static void plan(long O, long I, int target, int& no, int& nt) {
  const long tiles_o = cdivl(O, LOHA_T), tiles_j = cdivl(I, LOHA_T);
  const long per = (tiles_o * tiles_j) / target;
  no = 1; nt = 1;
  if (per >= 2) {
    no = 2;
    if (no > tiles_o) no = 1;
    nt = (int)(per / no);
    if (nt < 1) nt = 1;
    if (nt > 8) nt = 8;
    if (nt > tiles_j) nt = (int)tiles_j;
  }
}

static void launch_old(const Layer& L, float scale, int no, int nt, hipStream_t st) {
  LohaArgs a = args_of(L, scale);
  LohaGradGeom gm{nt};
  dim3 grid((unsigned)cdivl(cdivl(L.O, LOHA_T), no), (unsigned)cdivl(cdivl(L.I, LOHA_T), nt));
  if (no == 2) hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<2, true>), grid, dim3(NTHREADS), 0, st, a, gm);
  else hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<1, true>), grid, dim3(NTHREADS), 0, st, a, gm);
}
static void launch_new(const Layer& L, float scale, int no, int nt, hipStream_t st) {
  LohaArgs a = args_of(L, scale);
  LohaGradGeom gm{nt};
  dim3 grid((unsigned)cdivl(cdivl(L.O, LOHA_T), no), (unsigned)cdivl(cdivl(L.I, LOHA_T), nt));
  constexpr int lds = loha_grad16_lds_bytes();
  if (no == 2) hipLaunchKernelGGL((loha_factor_grad16_kernel<2>), grid, dim3(NTHREADS), lds, st, a, gm);
  else hipLaunchKernelGGL((loha_factor_grad16_kernel<1>), grid, dim3(NTHREADS), lds, st, a, gm);
}
template <int ABL>
static void launch_abl(const Layer& L, float scale, int no, int nt, hipStream_t st) {
  LohaArgs a = args_of(L, scale);
  LohaGradGeom gm{nt};
  dim3 grid((unsigned)cdivl(cdivl(L.O, LOHA_T), no), (unsigned)cdivl(cdivl(L.I, LOHA_T), nt));
  constexpr int lds = loha_grad16_lds_bytes();
  static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(loha_factor_grad16_kernel<2, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)once;
  hipLaunchKernelGGL((loha_factor_grad16_kernel<2, ABL>), grid, dim3(NTHREADS), lds, st, a, gm);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  constexpr int lds = loha_grad16_lds_bytes();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loha_factor_grad16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loha_factor_grad16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loha_factor_grad16_group_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(loha_factor_grad16_group_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));

  // ---- correctness: small / ragged layers, every (NO, nt) plan
  struct Case { long O, I; int R, no, nt; };
  const Case cases[] = {{64, 64, 32, 1, 1},   {128, 128, 32, 2, 2}, {200, 136, 32, 2, 1}, {72, 264, 16, 1, 3},  {320, 320, 32, 2, 5},
                        {100, 72, 4, 1, 1},   {640, 640, 32, 2, 8}, {40, 8, 8, 1, 1},    {130, 520, 28, 2, 4}, {320, 2880, 32, 2, 8}};
  int bad = 0;
  for (const Case& c : cases) {
    Layer L;
    make_layer(L, c.O, c.I, c.R, true);
    std::vector<double> r1a, r1b, r2a, r2b;
    host_ref(L, 0.5f, r1a, r1b, r2a, r2b);
    double e_old[4], e_new[4];
    zero_grads(L);
    launch_old(L, 0.5f, c.no, c.nt, st);
    CK(hipStreamSynchronize(st));
    e_old[0] = rel_err(L.g_w1a, r1a); e_old[1] = rel_err(L.g_w1b, r1b); e_old[2] = rel_err(L.g_w2a, r2a); e_old[3] = rel_err(L.g_w2b, r2b);
    zero_grads(L);
    launch_new(L, 0.5f, c.no, c.nt, st);
    CK(hipStreamSynchronize(st));
    e_new[0] = rel_err(L.g_w1a, r1a); e_new[1] = rel_err(L.g_w1b, r1b); e_new[2] = rel_err(L.g_w2a, r2a); e_new[3] = rel_err(L.g_w2b, r2b);
    const bool ok = e_new[0] < 3e-5 && e_new[1] < 3e-5 && e_new[2] < 3e-5 && e_new[3] < 3e-5;
    bad += ok ? 0 : 1;
    printf("check O=%ld I=%ld R=%d no=%d nt=%d  fp32-mfma %.1e %.1e %.1e %.1e | split16 %.1e %.1e %.1e %.1e  %s\n", c.O, c.I, c.R, c.no, c.nt,
           e_old[0], e_old[1], e_old[2], e_old[3], e_new[0], e_new[1], e_new[2], e_new[3], ok ? "ok" : "MISMATCH");
  }
  printf("correctness: %s\n", bad ? "FAILED" : "all ok");

  // ---- timing: SDXL layer shapes, 24 layers back to back (distinct G buffers: 24 x O x I x 4 bytes rotate through HBM)
  struct Shape { long O, I; const char* name; };
  const Shape shapes[] = {{1280, 1280, "attn 1280x1280"}, {10240, 1280, "ff.net.0 10240x1280"}, {1280, 5120, "ff.net.2 1280x5120"},
                          {640, 640, "attn 640x640"},     {1280, 2048, "ctx kv 1280x2048"},     {1280, 11520, "conv3x3 1280"},
                          {320, 2880, "conv3x3 320"}};
  for (const Shape& s : shapes) {
    const int NL = s.O * s.I > 8000000 ? 8 : 24;
    std::vector<Layer> Ls(NL);
    for (auto& L : Ls) make_layer(L, s.O, s.I, 32, false);
    for (int variant = 0; variant < 2; ++variant) {
      for (int target : {32, 128, 512}) {
        int no, nt;
        plan(s.O, s.I, target, no, nt);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, st));
          for (auto& L : Ls) (variant ? launch_new : launch_old)(L, 1.0f, no, nt, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const double us = best * 1e3 / NL, gb = (double)s.O * s.I * 4 / 1e9;
        printf("%-22s %s no=%d nt=%-2d wgs=%-5ld %8.1f us/layer  G read %.0f GB/s\n", s.name, variant ? "split16  " : "fp32-mfma", no, nt,
               cdivl(cdivl(s.O, 64), no) * cdivl(cdivl(s.I, 64), nt), us, gb / (us * 1e-6));
      }
    }
    {  // ablations of the split16 kernel, NO = 2, nt = 8 (results are garbage)
      typedef void (*Fn)(const Layer&, float, int, int, hipStream_t);
      const Fn fns[] = {launch_abl<1>, launch_abl<2>, launch_abl<4>, launch_abl<8>, launch_abl<16>, launch_abl<3>, launch_abl<31>};
      const char* names[] = {"no atomics", "no G loads", "no d_wb part", "no rebuild", "no restaging", "no atomics, no G", "everything off"};
      for (int v = 0; v < 7; ++v) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, st));
          for (auto& L : Ls) fns[v](L, 1.0f, 2, 8, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("%-22s split16 no=2 nt=8 ablation %-18s %8.1f us/layer\n", s.name, names[v], best * 1e3 / NL);
      }
    }
    for (auto& L : Ls) {
      for (float* p : {L.d_w1a, L.d_w1b, L.d_w2a, L.d_w2b, L.d_G, L.g_w1a, L.g_w1b, L.g_w2a, L.g_w2b}) CK(hipFree(p));
    }
  }
  return bad ? 1 : 0;
}
