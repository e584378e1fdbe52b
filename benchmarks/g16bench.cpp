// g16bench.cpp -- gemm16 (round 3: register-staged 64 x 128 tiles) vs gemm16d (round 6: LDS-DMA ring, transposed LDS reads, register
// epilogue) on LoHa's three contractions over the SDXL / SD1.5 shapes:  mode 0  y = x dW^T,  mode 1  dx = g dW,  mode 2  G = g^T x (fp32).
// Every variant is checked against an fp64-accumulating reference kernel on the device (norm-wise relative error) and timed inside a
// hipGraph over rotating buffer sets (footprint beyond the 256 MiB Infinity Cache).
//   benchmarks/g16bench [filter]       (make -C lycoris_amd/csrc g16bench)
// Development tool: parity proper is tests/ (oracle); this guards kernel-vs-reference equality while tiles are tuned.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../lycoris_amd/csrc/gemm16d.h"

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
__device__ inline float bf(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
// C[m][n] = sum_k A(m,k) B(n,k); a_ks: A is [K, M]; b_ks: B is [K, N]; one thread per output, fp64 accumulation
__global__ void ref_gemm(const unsigned short* A, const unsigned short* B, double* C, int M, int N, int K, int lda, int ldb, int a_ks, int b_ks) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  double s = 0;
  for (int k = 0; k < K; ++k) {
    const float a = a_ks ? bf(A[(long)k * lda + m]) : bf(A[(long)m * lda + k]);
    const float b = b_ks ? bf(B[(long)k * ldb + n]) : bf(B[(long)n * ldb + k]);
    s += (double)a * b;
  }
  C[idx] = s;
}
__global__ void empty_kernel(int* p) {
  if (p && threadIdx.x == 9999) *p = 1;
}

template <typename F>
static float bench(hipStream_t st, int nlaunch, int reps, F&& fn) {
  for (int i = 0; i < 2; ++i) fn(i);
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  if (getenv("G16_EAGER")) {  // plain stream launches (rocprofv3 cannot sample inside graph replays)
    for (int r = 0; r < 2; ++r) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < nlaunch; ++i) fn(i);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    return best * 1e3f / nlaunch;
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nlaunch; ++i) fn(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return best * 1e3f / nlaunch;
}

struct Shape {
  const char* tag;
  int M, I, O;  // rows, in features (x k x k), out features
};
static const Shape SHAPES[] = {
    {"attn1280", 1024, 1280, 1280}, {"ffup1280", 1024, 1280, 10240}, {"ffdn1280", 1024, 5120, 1280},  {"attn640", 4096, 640, 640},
    {"ffup640", 4096, 640, 5120},   {"ffdn640", 4096, 2560, 640},    {"xattn1280", 77, 2048, 1280},   {"xattn640", 77, 2048, 640},
    {"conv1280", 1024, 11520, 1280}, {"conv320", 16384, 2880, 320},  {"conv640", 4096, 5760, 640},     {"ragged", 200, 192, 328},
};

typedef void (*GFn)(Gemm16Group);
struct Variant {
  const char* name;
  int BM, BN, D;
  GFn fn[3];
};
#define V(BM, BN, D)                                                                                                      \
  {"g16d " #BM "x" #BN " D" #D, BM, BN, D,                                                                              \
   {gemm16d_kernel<__bf16, BM, BN, false, false, D>, gemm16d_kernel<__bf16, BM, BN, false, true, D>, gemm16d_kernel<__bf16, BM, BN, true, true, D>}}
static Variant VARIANTS[] = {V(128, 128, 2), V(128, 128, 3), V(128, 128, 4), V(128, 64, 3), V(128, 64, 4), V(64, 128, 3), V(64, 64, 3), V(64, 64, 4), V(64, 64, 6)};

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : nullptr;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  {
    float us = bench(st, 200, 5, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, nullptr); });
    printf("# empty 256-workgroup launch inside a graph: %.2f us\n", us);
  }
  for (Variant& v : VARIANTS)
    for (int e = 0; e < 3; ++e) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn[e]), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (const Shape& s : SHAPES) {
    if (filter && !strstr(s.tag, filter)) continue;
    const int M = s.M, I = s.I, O = s.O;
    const size_t xb = (size_t)M * I * 2, gb = (size_t)M * O * 2, wb = (size_t)O * I * 2;
    const size_t maxout = (size_t)O * I * 4 > xb ? (size_t)O * I * 4 : xb;
    int nsets = (int)((600u << 20) / (xb + gb + wb)) + 1;
    if (nsets > 24) nsets = 24;
    if (nsets < 2) nsets = 2;
    struct Set { void *x, *g, *w, *y, *dx, *G; };
    std::vector<Set> sets(nsets);
    for (int i = 0; i < nsets; ++i) {
      Set& z = sets[i];
      CK(hipMalloc(&z.x, xb)); CK(hipMalloc(&z.g, gb)); CK(hipMalloc(&z.w, wb)); CK(hipMalloc(&z.y, gb)); CK(hipMalloc(&z.dx, xb));
      CK(hipMalloc(&z.G, (size_t)O * I * 4));
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)z.x, xb / 2, 11u + i, 1.0f);
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)z.g, gb / 2, 77u + i, 0.05f);
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)z.w, wb / 2, 99u + i, 0.05f);
    }
    CK(hipStreamSynchronize(st));
    double* cref;
    void* cout;
    const size_t nout_max = (size_t)O * I > (size_t)M * O ? ((size_t)O * I > (size_t)M * I ? (size_t)O * I : (size_t)M * I)
                                                           : ((size_t)M * O > (size_t)M * I ? (size_t)M * O : (size_t)M * I);
    CK(hipMalloc(&cref, nout_max * 8));
    CK(hipMalloc(&cout, maxout > gb ? maxout : gb));
    std::vector<double> h_ref(nout_max);
    std::vector<unsigned short> h16(nout_max);
    std::vector<float> h32(nout_max);
    const bool do_ref = (double)M * I * O < 4e10 || getenv("G16_REF_ALL");

    for (int mode = 0; mode < 3; ++mode) {
      if (getenv("G16_MODE") && atoi(getenv("G16_MODE")) != mode) continue;
      auto prob = [&](const Set& z, void* out) {
        Gemm16Prob p{};
        if (mode == 0) { p.A = z.x; p.B = z.w; p.C = out; p.M = M; p.N = O; p.K = I; p.lda = I; p.ldb = I; p.ldc = O; }
        else if (mode == 1) { p.A = z.g; p.B = z.w; p.C = out; p.M = M; p.N = I; p.K = O; p.lda = O; p.ldb = I; p.ldc = I; }
        else { p.A = z.g; p.B = z.x; p.C = out; p.M = O; p.N = I; p.K = M; p.lda = O; p.ldb = I; p.ldc = I; }
        p.alpha = 1.0f;
        return p;
      };
      const Gemm16Prob p0 = prob(sets[0], cout);
      const bool a_ks = mode == 2, b_ks = mode >= 1, f32 = mode == 2;
      const size_t nout = (size_t)p0.M * p0.N;
      const double flops = 2.0 * p0.M * p0.N * p0.K;
      double refn = 0;
      if (do_ref) {
        hipLaunchKernelGGL(ref_gemm, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, (const unsigned short*)p0.A, (const unsigned short*)p0.B, cref,
                           p0.M, p0.N, p0.K, p0.lda, p0.ldb, a_ks ? 1 : 0, b_ks ? 1 : 0);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h_ref.data(), cref, nout * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nout; ++i) refn += h_ref[i] * h_ref[i];
      }
      auto relerr = [&]() {
        if (!do_ref) return -1.0;
        double e = 0;
        if (f32) {
          CK(hipMemcpy(h32.data(), cout, nout * 4, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < nout; ++i) e += (h32[i] - h_ref[i]) * (h32[i] - h_ref[i]);
        } else {
          CK(hipMemcpy(h16.data(), cout, nout * 2, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < nout; ++i) {
            const double v = (double)__builtin_bit_cast(float, (unsigned)h16[i] << 16);
            e += (v - h_ref[i]) * (v - h_ref[i]);
          }
        }
        return sqrt(e / (refn + 1e-300));
      };
      const int nl = flops > 2e10 ? 20 : 60;
      // ---- gemm16 (old) ----
      float t_old = -1;
      if (gemm16_ok(p0, a_ks, b_ks)) {
        auto launch_old = [&](const Gemm16Prob& p) {
          Gemm16Group ga{};
          ga.n = 1; ga.out_f32 = f32; ga.p[0] = p;
          ga.wg_end[0] = ((p.M + G16_TM - 1) / G16_TM) * ((p.N + 127) / 128);
          const dim3 grid((unsigned)ga.wg_end[0]);
          if (mode == 0) hipLaunchKernelGGL((gemm16_kernel<__bf16, G16_TM, false, false>), grid, dim3(NTHREADS), gemm16_lds_bytes<G16_TM>(), st, ga);
          else if (mode == 1) hipLaunchKernelGGL((gemm16_kernel<__bf16, G16_TM, false, true>), grid, dim3(NTHREADS), gemm16_lds_bytes<G16_TM>(), st, ga);
          else hipLaunchKernelGGL((gemm16_kernel<__bf16, G16_TM, true, true>), grid, dim3(NTHREADS), gemm16_lds_bytes<G16_TM>(), st, ga);
        };
        CK(hipMemsetAsync(cout, 0xff, nout * (f32 ? 4 : 2), st));
        launch_old(p0);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        const double e = relerr();
        t_old = bench(st, nl, 5, [&](int i) { const Set& z = sets[i % nsets]; launch_old(prob(z, mode == 0 ? z.y : mode == 1 ? z.dx : z.G)); });
        printf("%-10s mode %d  M=%-5d N=%-5d K=%-5d | gemm16 64x128        %8.2f us %7.1f TF/s  relerr %.2e\n", s.tag, mode, p0.M, p0.N, p0.K, t_old,
               flops / t_old * 1e-6, e);
      }
      if (!gemm16d_ok(p0, a_ks, b_ks, f32)) {
        printf("%-10s mode %d: gemm16d does not take this problem (alignment / K %% 64)\n", s.tag, mode);
        continue;
      }
      for (const Variant& v : VARIANTS) {
        if (getenv("G16_ONLY") && !strstr(v.name, getenv("G16_ONLY"))) continue;
        const int lds = gemm16d_lds_bytes(v.BM, v.BN, v.D);
        if (lds > 160 * 1024) continue;
        auto launch = [&](const Gemm16Prob& p) {
          Gemm16Group ga{};
          ga.n = 1; ga.out_f32 = f32; ga.p[0] = p;
          ga.wg_end[0] = gemm16d_wgs(p.M, p.N, v.BM, v.BN);
          hipLaunchKernelGGL(v.fn[mode], dim3((unsigned)ga.wg_end[0]), dim3(NTHREADS), lds, st, ga);
        };
        CK(hipMemsetAsync(cout, 0xff, nout * (f32 ? 4 : 2), st));
        launch(p0);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        const double e = relerr();
        const float t = bench(st, nl, 5, [&](int i) { const Set& z = sets[i % nsets]; launch(prob(z, mode == 0 ? z.y : mode == 1 ? z.dx : z.G)); });
        const int wgs = ((p0.M + v.BM - 1) / v.BM) * ((p0.N + v.BN - 1) / v.BN);
        printf("   %-18s wgs %5d lds %3dK | %8.2f us %7.1f TF/s  x%.2f  relerr %.2e%s\n", v.name, wgs, lds >> 10, t, flops / t * 1e-6,
               t_old > 0 ? t_old / t : 0.0, e, (e > (f32 ? 1e-5 : 3e-3)) ? "  <<<< WRONG" : "");
      }
    }
    if (getenv("G16_ABLATE")) {  // where does a K step spend its time?  (mode 0 on this shape; results are garbage)
      auto prob0 = [&](const Set& z) {
        Gemm16Prob p{};
        p.A = z.x; p.B = z.w; p.C = z.y; p.M = M; p.N = O; p.K = I; p.lda = I; p.ldb = I; p.ldc = O; p.alpha = 1.0f;
        return p;
      };
#define ABL(BM_, BN_, D_, bits)                                                                                                          \
      {                                                                                                                                  \
        auto kern = gemm16d_kernel<__bf16, BM_, BN_, false, false, D_, bits>;                                                            \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));            \
        const float t = bench(st, 30, 5, [&](int i) {                                                                                    \
          Gemm16Group ga{};                                                                                                              \
          ga.n = 1; ga.p[0] = prob0(sets[i % nsets]);                                                                                    \
          ga.wg_end[0] = gemm16d_wgs(M, O, BM_, BN_);                                              \
          hipLaunchKernelGGL(kern, dim3((unsigned)ga.wg_end[0]), dim3(NTHREADS), gemm16d_lds_bytes(BM_, BN_, D_), st, ga);              \
        });                                                                                                                              \
        printf("   ablate %dx%d D%d bits %2d (1 no refill, 2 no mfma, 4 no lds reads, 8 no barrier): %8.2f us\n", BM_, BN_, D_, bits, t);  \
      }
      ABL(128, 128, 3, 0) ABL(128, 128, 3, 1) ABL(128, 128, 3, 2) ABL(128, 128, 3, 4) ABL(128, 128, 3, 6) ABL(128, 128, 3, 7) ABL(128, 128, 3, 8)
      ABL(128, 128, 3, 5) ABL(128, 128, 3, 3) ABL(128, 128, 3, 15)
      ABL(64, 64, 3, 0) ABL(64, 64, 3, 1) ABL(64, 64, 3, 2) ABL(64, 64, 3, 4) ABL(64, 64, 3, 6) ABL(64, 64, 3, 7) ABL(64, 64, 3, 15)
    }
    for (Set& z : sets) { CK(hipFree(z.x)); CK(hipFree(z.g)); CK(hipFree(z.w)); CK(hipFree(z.y)); CK(hipFree(z.dx)); CK(hipFree(z.G)); }
    CK(hipFree(cref)); CK(hipFree(cout));
  }
  return 0;
}
