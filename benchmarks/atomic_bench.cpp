// atomic_bench.cpp -- fp32 global atomic-add throughput on MI355X for the split-K epilogue patterns (development tool).
//   pattern A: a wave instruction covers 4 rows x 16 consecutive floats (MFMA accumulator layout)
//   pattern B: a wave instruction covers 64 consecutive floats of one row
// Every block adds a full [rows x cols] tile set `per_block` times over the output matrix (like split-K slabs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)

template <int PATTERN>
__global__ void atom_kernel(float* out, int rows, int cols, int tiles_per_block) {
  // the matrix is cut in 16 x 64 element chunks; block b handles chunks b, b + grid, ... (tiles_per_block of them, wrapping)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks_c = cols / 64, nchunks = (rows / 16) * chunks_c;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int ch = (int)(((long)blockIdx.x * tiles_per_block + t) % nchunks);
    const int r0 = (ch / chunks_c) * 16, c0 = (ch % chunks_c) * 64;
    // 16 x 64 chunk = 1024 floats = 4 waves x 4 instructions x 64 lanes
    for (int i = 0; i < 4; ++i) {
      int r, c;
      if (PATTERN == 0) { r = 4 * (lane >> 4) + i; c = 16 * wave + (lane & 15); }   // 4 rows x 16 floats per instruction
      else { r = 4 * wave + i; c = lane; }                                           // 1 row x 64 floats
      __hip_atomic_fetch_add(out + (long)(r0 + r) * cols + c0 + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ void store_kernel(float* out, int rows, int cols, int tiles_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunks_c = cols / 64, nchunks = (rows / 16) * chunks_c;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int ch = (int)(((long)blockIdx.x * tiles_per_block + t) % nchunks);
    const int r0 = (ch / chunks_c) * 16, c0 = (ch % chunks_c) * 64;
    for (int i = 0; i < 4; ++i) out[(long)(r0 + 4 * wave + i) * cols + c0 + lane] = 1.0f;
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { int rows, cols, blocks, tpb; } cfgs[] = {
      {160, 192, 256, 30}, {160, 192, 256, 120}, {160, 192, 1024, 30}, {1280, 192, 256, 240}, {1280, 192, 256, 960}, {1280, 192, 1024, 240},
      {16, 64, 256, 1}, {16, 64, 256, 8},
  };
  for (auto& c : cfgs) {
    float* out; CK(hipMalloc(&out, (size_t)c.rows * c.cols * 4)); CK(hipMemset(out, 0, (size_t)c.rows * c.cols * 4));
    for (int pat = 0; pat < 3; ++pat) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        if (pat == 0) hipLaunchKernelGGL(atom_kernel<0>, dim3(c.blocks), dim3(256), 0, st, out, c.rows, c.cols, c.tpb);
        else if (pat == 1) hipLaunchKernelGGL(atom_kernel<1>, dim3(c.blocks), dim3(256), 0, st, out, c.rows, c.cols, c.tpb);
        else hipLaunchKernelGGL(store_kernel, dim3(c.blocks), dim3(256), 0, st, out, c.rows, c.cols, c.tpb);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      const double n = (double)c.blocks * c.tpb * 1024;
      printf("matrix %4dx%3d blocks %4d chunks/block %3d  %s: %8.1f us  %7.2f G elem/s  (%.1f adds per element)\n", c.rows, c.cols, c.blocks, c.tpb,
             pat == 0 ? "atomic 4x16" : pat == 1 ? "atomic 1x64" : "plain store", best * 1e3, n / best * 1e-6, n / ((double)c.rows * c.cols));
    }
    CK(hipFree(out));
  }
  return 0;
}
