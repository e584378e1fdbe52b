// soffset_check.cpp -- is the SGPR offset of a raw buffer load part of the descriptor's range check on gfx950?  (development tool)
// A descriptor of 1024 bytes over a 8192-byte allocation filled with 0x11111111: lane l loads 4 bytes at
//   case A: voffset = 4 l + 2048, soffset = 0       (beyond num_records through the VGPR offset)
//   case B: voffset = 4 l,        soffset = 2048    (beyond num_records through the SGPR offset)
// and the same two through buffer_load ... lds.  Zeros = refused by the range check, 0x11111111 = read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const char* src, unsigned* out, int soff) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[512];
  const int l = threadIdx.x;
  lds[l] = 0xdeadbeefu; lds[64 + l] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 1024, 0x00020000);
  out[l] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * l + 2048, 0, 0);
  out[64 + l] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 4 * l, soff, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 4, 4 * l + 2048, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 64), 4, 4 * l, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[128 + l] = lds[l];
  out[192 + l] = lds[64 + l];
}
int main() {
  char* src; unsigned* out;
  CK(hipMalloc(&src, 8192)); CK(hipMemset(src, 0x11, 8192)); CK(hipMalloc(&out, 256 * 4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 2048);
  CK(hipDeviceSynchronize());
  unsigned h[256];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  printf("to VGPR: beyond via voffset -> %08x   beyond via soffset -> %08x\n", h[5], h[64 + 5]);
  printf("to LDS : beyond via voffset -> %08x   beyond via soffset -> %08x\n", h[128 + 5], h[192 + 5]);
  return 0;
}
