// dmacad.cpp -- completion cadence of LDS-DMA instructions (development tool).  One launch of 256 workgroups x 4 waves; every wave issues N
// buffer_load_dwordx4 ... lds (1 KiB each, distinct LDS slots) back to back and then waits for them in groups of 3 (vmcnt(N - 3), vmcnt(N - 6), ...),
// stamping the shader clock after the issue loop and after every wait.  Printed: the stamps of wave 0 of workgroup 0 (cycles since its first
// instruction).  MODE 0: linear 1 KiB pieces, 1: 16 rows x 64 B (row pitch 2560 B), 2: out-of-range offsets (zero fill, no memory traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int N>
__global__ __launch_bounds__(256) void cad_kernel(const char* src, unsigned span, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t[16] = {0};
  __builtin_amdgcn_sched_barrier(0);
  t[0] = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)span, 0x00020000);
  unsigned voff = MODE == 1 ? (unsigned)(lane >> 2) * 2560u + (unsigned)(lane & 3) * 16u : MODE == 2 ? 0x80000000u : (unsigned)lane * 16u;
  const unsigned base = (blockIdx.x * 4u + wave) * (MODE == 1 ? 16u * 2560u : N * 1024u);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const unsigned so = (base + (MODE == 1 ? j * 64u : j * 1024u)) % (span - 65536u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * N + j) * 1024), 16, (int)voff, (int)(so & ~15u), 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  t[1] = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
#define W(k) if constexpr (N - 3 * (k) >= 0) { wait_vm<(N - 3 * (k) >= 0 ? N - 3 * (k) : 0)>(); __builtin_amdgcn_sched_barrier(0); t[1 + (k)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
  W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10)
  wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);
  t[12] = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int i = 0; i < 16; ++i) out[i] = t[i];
}

int main() {
  char* src;
  const unsigned span = 64u << 20;
  CK(hipMalloc(&src, span + 4096));
  CK(hipMemset(src, 1, span));
  unsigned long long* out;
  CK(hipMalloc(&out, 16 * 8));
  unsigned long long h[16];
#define RUN(MODE, N, BLOCKS)                                                                                                  \
  {                                                                                                                           \
    CK(hipFuncSetAttribute((const void*)cad_kernel<MODE, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));        \
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((cad_kernel<MODE, N>), dim3(BLOCKS), dim3(256), 4 * N * 1024, 0, src, span, out); \
    CK(hipDeviceSynchronize());                                                                                               \
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));                                                                  \
    printf("mode %d N %2d blocks %3d: issued +%llu | groups of 3 landed:", MODE, N, BLOCKS, h[1] - h[0]);                       \
    for (int i = 2; i <= 11; ++i) if (h[i]) printf(" +%llu", h[i] - h[0]);                                                    \
    printf(" | all +%llu\n", h[12] - h[0]);                                                                                   \
  }
  RUN(0, 30, 256) RUN(0, 33, 256) RUN(0, 36, 256) RUN(2, 36, 256) RUN(0, 39, 256) RUN(2, 39, 1)
  return 0;
}
