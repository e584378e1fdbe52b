// dmabench.cpp -- what one CU's vector-memory path delivers on gfx950, by operand source and instruction kind (development tool).
//   LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) vs plain 16-byte loads to registers,
//   linear 1 KiB pieces vs 16 rows x 64 bytes pieces, source footprint 2 MiB / 64 MiB / 1 GiB, 1-16 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr;

// each wave streams `iters` x 8 pieces of 1 KiB; piece index advances over the whole footprint so that all waves together
// touch `span` bytes repeatedly.  MODE 0: DMA linear, 1: DMA rows (16 rows x 64 B, row pitch 2560 B), 2: VGPR linear
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const char* src, unsigned span, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned gw = blockIdx.x * nw + wave, tw = gridDim.x * nw;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)span, 0x00020000);
  unsigned voff = MODE == 1 ? (unsigned)(lane >> 2) * 2560u + (unsigned)(lane & 3) * 16u : (unsigned)lane * 16u;
  u32x4 acc = {0, 0, 0, 0};
  unsigned piece = gw * 8u;
  const unsigned npieces = span / (MODE == 1 ? 16u * 2560u : 1024u) * (MODE == 1 ? 40u : 1u);  // rows mode: 40 column steps per 16-row band
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned p = (piece + j) % npieces;
      unsigned so = MODE == 1 ? (p / 40u) * (16u * 2560u) + (p % 40u) * 64u : p * 1024u;
      if constexpr (MODE == 2) {
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)so, 0));
        acc ^= v;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * 8 + j) * 1024), 16, (int)voff, (int)so, 0, 0);
      }
    }
    if constexpr (MODE != 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    piece += tw * 8u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] ^ acc[3];
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  char* src;
  const size_t cap = 1u << 30;
  CK(hipMalloc(&src, cap + 4096));
  CK(hipMemset(src, 1, cap));
  unsigned* sink;
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int m = 0; m < 3; ++m) {
    const void* f = m == 0 ? (const void*)stream_kernel<0> : m == 1 ? (const void*)stream_kernel<1> : (const void*)stream_kernel<2>;
    CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  const unsigned spans[] = {2u << 20, 16u << 20, 128u << 20, 1u << 30};
  const char* names[] = {"DMA linear 1KiB", "DMA 16 rows x 64B", "VGPR linear 16B/lane"};
  for (int mode = 0; mode < 3; ++mode)
    for (unsigned span : spans)
      for (int wpc : {1, 2, 4, 8, 16}) {  // waves per CU: blocks of 256 threads (4 waves) x k, or 1-2 waves in a block
        const int bt = wpc >= 4 ? 256 : 64 * wpc;
        const int blocks = 256 * (wpc >= 4 ? wpc / 4 : 1);
        const int iters = 400;
        const int lds = getenv("DMB_LDS") ? atoi(getenv("DMB_LDS")) : (bt / 64) * 8 * 1024;  // DMB_LDS: a larger allocation (occupancy limit by LDS)
        if (getenv("DMB_QUICK") && (span != (2u << 20) || (wpc != 4 && wpc != 8))) continue;
        auto go = [&]() {
          if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(bt), lds, st, src, span, iters, sink);
          if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(bt), lds, st, src, span, iters, sink);
          if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(bt), lds, st, src, span, iters, sink);
        };
        go();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        go();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)blocks * (bt / 64) * iters * 8 * 1024;
        printf("%-22s span %5u MiB  %2d waves/CU : %8.1f GB/s  (%.1f B/clk/CU at 2.1 GHz)\n", names[mode], span >> 20, wpc, bytes / ms * 1e-6,
               bytes / ms * 1e-6 / 256 / 2.1);
      }
  return 0;
}
