// trbench.cpp -- LDS read throughput per CU: ds_read_b64_tr_b16 against ds_read_b64 / ds_read_b128 on the address patterns of
// csrc/kron_dw2f.h (row-major 16-bit image, row pitch 320 or 160 bytes, with and without the source-side swizzle).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/trbench.cpp -o benchmarks/trbench && benchmarks/trbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: tr16_b64   1: plain b64 at the same addresses   2: b128 lane-linear (the fast pattern)
template <int MODE, int PITCH, int SWZ>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned addr;
  if (MODE == 2) addr = base + lane * 16;
  else {
    int sh = 0;
    if (SWZ == 1) sh = (g & 1) * 32;
    if (SWZ == 2) sh = ((li >> 3) & 1) * 32;
    addr = base + (4 * g + (li >> 2)) * PITCH + (li & 3) * 8 + sh;
  }
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    u32x2 v[10];
    u32x4 w[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      if (MODE == 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"((j % 5) * 64 + (j / 5) * 16 * PITCH) : "memory");
      if (MODE == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"((j % 5) * 64 + (j / 5) * 16 * PITCH) : "memory");
      if (MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[j]) : "v"(addr), "n"(j * 1024) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 10; ++j) acc += MODE == 2 ? w[j][0] + w[j][3] : v[j][0] + v[j][1];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int PITCH, int SWZ>
int run(const char* name, int wgs_per_cu) {
  unsigned* out;
  const int grid = 256 * wgs_per_cu, iters = 2000;
  CK(hipMalloc(&out, grid * 256 * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PITCH, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, PITCH, SWZ>), dim3(grid), dim3(256), 65536, 0, out, 10);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, PITCH, SWZ>), dim3(grid), dim3(256), 65536, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // per CU: wgs_per_cu * 4 waves * iters * 10 read instructions
  const double reads = (double)wgs_per_cu * 4 * iters * 10;
  const double ns_per_read = ms * 1e6 / reads;
  const int bytes = MODE == 2 ? 1024 : 512;
  printf("%-44s %d WG/CU: %6.2f ns per wave-instruction and CU (%.1f cycles at 2.1 GHz, %5.1f B/clk/CU)\n", name, wgs_per_cu, ns_per_read, ns_per_read * 2.1,
         bytes / (ns_per_read * 2.1));
  CK(hipFree(out));
  return 0;
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 320, 0>("tr16_b64 pitch 320", w);
    run<0, 320, 1>("tr16_b64 pitch 320 swizzle by group", w);
    run<0, 320, 2>("tr16_b64 pitch 320 swizzle by row pair", w);
    run<0, 160, 0>("tr16_b64 pitch 160", w);
    run<0, 128, 0>("tr16_b64 pitch 128", w);
    run<0, 144, 0>("tr16_b64 pitch 144 (padded)", w);
    run<0, 336, 0>("tr16_b64 pitch 336 (padded)", w);
    run<1, 320, 0>("b64 same addresses pitch 320", w);
    run<1, 160, 0>("b64 same addresses pitch 160", w);
    run<2, 0, 0>("b128 lane-linear", w);
  }
  return 0;
}
