// kbench.cpp -- kernel-level timing of the LoKr entry points through the C ABI, no Python in the loop.
//
//   benchmarks/kbench [filter]           (built by `make -C lycoris_amd/csrc kbench`)
//
// For every SDXL LoKr (factor 8) row shape: fwd, bwd(dx + dw1) and bwd(dw2) are each captured NLAUNCH times in a
// hipGraph over NSETS rotating buffer sets (total footprint > the 256 MiB Infinity Cache, so the rates are HBM rates,
// not cache rates), replayed REPS times and timed with HIP events on the launch stream.  Prints us / launch, algorithmic
// GB/s (activation bytes moved once) and, for reference, the rate of a plain 16-byte copy kernel on the same chip.
// Development tool: parity is checked by tests/, not here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/lycoris_amd.h"

#define CK(x)                                                                        \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)
#define LYC(x)                                                          \
  do {                                                                  \
    int rc_ = (x);                                                      \
    if (rc_ != 0) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, lyc_last_error());    \
      exit(1);                                                          \
    }                                                                   \
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
__global__ void copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

__global__ void empty_kernel(int* p) {
  if (p && threadIdx.x == 9999) *p = 1;
}

struct Shape {
  const char* tag;
  long M;
  int I, O;
  int count;
};

struct Set {
  void *x, *g, *y, *dx, *ws;
  float *w1, *w2, *dw1, *dw2;
};

static float time_graph(hipStream_t st, hipGraphExec_t ge, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / reps;
}

// KB_EAGER=1: plain launches instead of a hipGraph (rocprofv3 --pmc cannot sample inside graph replays)
static const bool g_eager = getenv("KB_EAGER") != nullptr;

template <typename F>
static float bench(hipStream_t st, int nlaunch, int reps, F&& fn) {
  for (int i = 0; i < 2; ++i) fn(i);  // warm-up, eager
  CK(hipStreamSynchronize(st));
  if (g_eager) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < nlaunch; ++i) fn(i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / nlaunch;
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nlaunch; ++i) fn(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float ms = 1e30f;
  for (int trial = 0; trial < 5; ++trial) {  // best of 5: run-to-run spread of these short graphs is large
    const float t = time_graph(st, ge, reps);
    if (t < ms) ms = t;
  }
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return ms * 1e3f / nlaunch;  // us per launch
}

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const int F = 8;
  std::vector<Shape> shapes = {
      {"attn@1280", 1024, 1280, 1280, 372},   {"ff0@1280", 1024, 1280, 10240, 60}, {"ff2@1280", 1024, 5120, 1280, 60},
      {"attn@640", 4096, 640, 640, 70},       {"ff0@640", 4096, 640, 5120, 10},    {"ff2@640", 4096, 2560, 640, 10},
      {"kv@1280", 77, 2048, 1280, 120},       {"kv@640", 77, 2048, 640, 20},       {"temb", 1, 1280, 1280, 7},
      {"conv320@128", 16384, 2880, 320, 7},   {"conv640@64", 4096, 5760, 640, 6},  {"conv1280@32", 1024, 11520, 1280, 10},
      {"conv2560>1280", 1024, 23040, 1280, 2}, {"sc960>320", 16384, 960, 320, 1},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));

  {  // reference: plain copy rate
    const size_t bytes = 1ull << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemsetAsync(a, 1, bytes, st));
    const float us = bench(st, 4, 5, [&](int) { hipLaunchKernelGGL(copy16, dim3(256 * 8), dim3(256), 0, st, (const uint4*)a, (uint4*)b, bytes / 16); });
    printf("copy16 1 GiB: %.1f us  -> %.0f GB/s (read+write)\n", us, 2.0 * bytes / us * 1e-3);
    CK(hipFree(a));
    CK(hipFree(b));
  }

  {  // floor: an empty 256-block kernel and a 2.6 MB -> 2.6 MB copy in the same graph harness
    const float us0 = bench(st, 64, 5, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, (int*)nullptr); });
    const size_t bytes = 1024 * 1280 * 2;
    std::vector<void*> bufs(128);
    for (auto& b : bufs) CK(hipMalloc(&b, bytes));
    const float us1 = bench(st, 128, 5, [&](int i) {
      hipLaunchKernelGGL(copy16, dim3(640), dim3(256), 0, st, (const uint4*)bufs[i % 128], (uint4*)bufs[(i + 64) % 128], bytes / 16);
    });
    printf("empty kernel: %.2f us/launch;  2.6 MB copy (640 blocks, 1 x 16 B per thread): %.2f us/launch\n", us0, us1);
    for (auto& b : bufs) CK(hipFree(b));
  }

  printf("%-14s %6s %6s %6s | %9s %8s | %9s %8s | %9s %8s | %8s\n", "shape", "M", "I", "O", "fwd us", "GB/s", "bwd-dx us",
         "GB/s", "dw2 us", "GB/s", "sum us");
  double step_us = 0.0, step_bytes = 0.0;
  for (const Shape& s : shapes) {
    if (filter[0] && !strstr(s.tag, filter)) continue;
    const int a = F, b = F, c = s.O / F, d = s.I / F;
    const size_t xb = (size_t)s.M * s.I * 2, yb = (size_t)s.M * s.O * 2;
    const size_t per_set = 2 * xb + 2 * yb;
    int nsets = (int)((600ull << 20) / per_set) + 1;
    if (nsets > 64) nsets = 64;
    if (nsets < 2) nsets = 2;
    std::vector<Set> sets(nsets);
    for (int i = 0; i < nsets; ++i) {
      Set& q = sets[i];
      CK(hipMalloc(&q.x, xb)); CK(hipMalloc(&q.dx, xb)); CK(hipMalloc(&q.g, yb)); CK(hipMalloc(&q.y, yb));
      CK(hipMalloc(&q.w1, a * b * 4)); CK(hipMalloc(&q.dw1, a * b * 4));
      CK(hipMalloc(&q.ws, (size_t)lyc_lokr_bwd_workspace_bytes(s.M, a, b, c, d, LYC_BF16) + 16));
      CK(hipMalloc(&q.w2, (size_t)c * d * 4)); CK(hipMalloc(&q.dw2, (size_t)c * d * 4));
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)q.x, xb / 2, 11u + i, 1.0f);
      hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, (unsigned short*)q.g, yb / 2, 77u + i, 0.05f);
      hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, st, q.w1, (size_t)a * b, 5u + i, 0.3f);
      hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, st, q.w2, (size_t)c * d, 9u + i, 0.05f);
      CK(hipMemsetAsync(q.dw1, 0, a * b * 4, st));
      CK(hipMemsetAsync(q.dw2, 0, (size_t)c * d * 4, st));
    }
    CK(hipStreamSynchronize(st));
    const int nl = nsets * 2 > 32 ? nsets * 2 : 32;
    const float t_f = bench(st, nl, 3, [&](int i) {
      const Set& q = sets[i % nsets];
      LYC(lyc_lokr_linear_fwd(q.x, q.w1, q.w2, nullptr, q.y, s.M, a, b, c, d, 1.0f, LYC_BF16, st));
    });
    float t_b = bench(st, nl, 3, [&](int i) {
      const Set& q = sets[i % nsets];
      LYC(lyc_lokr_linear_bwd(q.g, q.x, q.w1, q.w2, q.dx, q.dw1, nullptr, q.ws, s.M, a, b, c, d, 1.0f, LYC_BF16, st));
    });
    const float t_w = bench(st, nl, 3, [&](int i) {
      const Set& q = sets[i % nsets];
      LYC(lyc_lokr_linear_bwd(q.g, q.x, q.w1, q.w2, nullptr, nullptr, q.dw2, nullptr, s.M, a, b, c, d, 1.0f, LYC_BF16, st));
    });
    const float t_a = bench(st, nl, 3, [&](int i) {  // the whole backward as the autograd op issues it
      const Set& q = sets[i % nsets];
      LYC(lyc_lokr_linear_bwd(q.g, q.x, q.w1, q.w2, q.dx, q.dw1, q.dw2, q.ws, s.M, a, b, c, d, 1.0f, LYC_BF16, st));
    }) ;
    const double bf = xb + yb, bb = 2.0 * xb + yb, bw = xb + yb;
    printf("%-14s %6ld %6d %6d | %9.1f %8.0f | %9.1f %8.0f | %9.1f %8.0f | %8.1f  bwd-all %7.1f\n", s.tag, s.M, s.I, s.O, t_f,
           bf / t_f * 1e-3, t_b, bb / t_b * 1e-3, t_w, bw / t_w * 1e-3, t_f + t_b + t_w, t_a);
    t_b = t_a - t_w;  // count the fused backward in the step sum
    fflush(stdout);
    step_us += (double)s.count * (t_f + t_b + t_w);
    step_bytes += (double)s.count * (bf + bb);
    for (Set& q : sets) {
      CK(hipFree(q.x)); CK(hipFree(q.dx)); CK(hipFree(q.g)); CK(hipFree(q.y));
      CK(hipFree(q.w1)); CK(hipFree(q.dw1)); CK(hipFree(q.w2)); CK(hipFree(q.dw2)); CK(hipFree(q.ws));
    }
  }
  printf("count-weighted sum over listed shapes: %.2f ms, %.2f GB algorithmic (dw2 pass re-reads not counted) -> %.0f GB/s\n",
         step_us * 1e-3, step_bytes * 1e-9, step_bytes / step_us * 1e-3);
  return 0;
}
