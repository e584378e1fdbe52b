// stbench.cpp -- store-side twin of dmabench.cpp: what the write path delivers for the epilogue patterns of the row kernels.
//   pattern 0: 64 lanes x 16 B contiguous (1 KiB per wave instruction, full 128-byte lines)
//   pattern 1: 16 rows x 64 B  (4 lanes x 16 B per row, row pitch P)
//   pattern 2: 16 rows x 32 B  (4 lanes x 8 B per row)        <- the kron3 / kron4 register epilogue
//   pattern 3: 8 rows x 128 B  (8 lanes x 16 B per row: one full line per row)
// every workgroup owns a [rows x width] tile of a [R x N] bf16 matrix and writes it once; footprint 21 MB (one FFN-up output).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// grid: (R / 128, N / W) ; block 256 = 4 waves x 32 rows; each wave writes its 32 rows x W columns
template <int PAT, int W>
__global__ __launch_bounds__(256) void store_kernel(char* dst, int N, unsigned bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)bytes, 0x00020000);
  const unsigned pitch = (unsigned)N * 2u;
  const unsigned row0 = blockIdx.x * 128u + wave * 32u, col0 = blockIdx.y * (unsigned)W * 2u;  // bytes
  const u32x4 v4 = {(unsigned)lane, 1u, 2u, 3u};
  const u32x2 v2 = {(unsigned)lane, 1u};
  if constexpr (PAT == 0) {  // treat the tile as a linear region (only meaningful as a bandwidth reference)
    const unsigned base = (blockIdx.y * gridDim.x + blockIdx.x) * (128u * W * 2u) + wave * (32u * W * 2u);
#pragma unroll
    for (unsigned o = 0; o < 32u * W * 2u; o += 1024u) __builtin_amdgcn_raw_buffer_store_b128(v4, rs, (int)(base + o + lane * 16u), 0, 0);
  } else if constexpr (PAT == 1) {
#pragma unroll
    for (int r = 0; r < 32; r += 16)
#pragma unroll
      for (unsigned c = 0; c < (unsigned)W * 2u; c += 64u)
        __builtin_amdgcn_raw_buffer_store_b128(v4, rs, (int)((row0 + r + (lane & 15)) * pitch + col0 + c + (lane >> 4) * 16u), 0, 0);
  } else if constexpr (PAT == 2) {
#pragma unroll
    for (int r = 0; r < 32; r += 16)
#pragma unroll
      for (unsigned c = 0; c < (unsigned)W * 2u; c += 32u)
        __builtin_amdgcn_raw_buffer_store_b64(v2, rs, (int)((row0 + r + (lane & 15)) * pitch + col0 + c + (lane >> 4) * 8u), 0, 0);
  } else {
#pragma unroll
    for (int r = 0; r < 32; r += 8)
#pragma unroll
      for (unsigned c = 0; c < (unsigned)W * 2u; c += 128u)
        __builtin_amdgcn_raw_buffer_store_b128(v4, rs, (int)((row0 + r + (lane >> 3)) * pitch + col0 + c + (lane & 7) * 16u), 0, 0);
  }
}

template <int PAT, int W>
float run(hipStream_t st, char** bufs, int nb, int R, int N) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(R / 128, N / W);
  const unsigned bytes = (unsigned)R * N * 2u;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<PAT, W>), grid, dim3(256), 0, st, bufs[i % nb], N, bytes);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  const int reps = 40;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((store_kernel<PAT, W>), grid, dim3(256), 0, st, bufs[i % nb], N, bytes);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int R = 8192, N = 1280;  // the (1024, 1280 -> 10240) forward output as [M * 8, 1280] bf16 rows: 21 MB
  const int nb = 30;             // rotate over 630 MB
  char* bufs[nb];
  for (int i = 0; i < nb; ++i) CK(hipMalloc(&bufs[i], (size_t)R * N * 2 + 4096));
  const double mb = (double)R * N * 2 * 1e-6;
  const char* names[] = {"linear 1 KiB", "16 rows x 64 B", "16 rows x 32 B", "8 rows x 128 B"};
#define GO(PAT, W) { float us = run<PAT, W>(st, bufs, nb, R, N); printf("%-16s tile 128 x %-3d : %7.2f us  %7.1f GB/s (incl. ~2 us launch)\n", names[PAT], W, us, mb / us * 1e3); }
  GO(0, 64) GO(1, 64) GO(2, 64) GO(3, 64) GO(1, 128) GO(2, 128) GO(3, 128) GO(2, 32) GO(1, 32) GO(2, 80) GO(2, 160)
  GO(3, 256) GO(3, 640) GO(3, 1280) GO(2, 1280) GO(1, 1280)
  return 0;
}
