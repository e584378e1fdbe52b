// ia3_kernels.h -- (IA)^3 per-channel scale kernels, gfx950 (HBM-bound streaming).
//
// Reference semantics (rebuild path, lycoris/modules/ia3.py:91-102, 129-144):
//   out-side : y = base + op(x, W * (w*mult) by rows) = base + (base - bias) * (w[o] * mult)
//   in-side  : y = base + op(x * (w[i]*mult), W)       (the dense op itself stays with rocBLAS / MIOpen)
// Tensors are viewed as [outer, C, inner] (Linear: inner = 1; NCHW conv: inner = H*W).
//
//   chan_scale_kernel   : out = a_in * (s0 + w[c]*mult) [- bias[c] * w[c]*mult]
//   chan_reduce_kernel  : dw[c] += mult * sum_{outer, inner} g * (b - bias[c])
#pragma once
#include "tile.h"

namespace lyc {

struct ChanArgs {
  const void* a_in;    // T [outer, C, inner]
  const void* b_in;    // T, second operand of the reduction
  void* out;           // T
  const float* w;      // [C]
  const float* bias;   // [C] or nullptr
  float* dw;           // [C] fp32, accumulated atomically
  long outer;
  long C;
  long inner;
  float s0, mult;
  int vec;             // chan_reduce, inner == 1: the host verified C % VEC == 0 and 16-byte alignment (vector layout)
};

// dst = src[0] + ... + src[n - 1] (n <= 4), fp32 accumulation, ONE rounding: the gradient of a tensor that n sibling projections read
// (round 5: the n dx results of lyc_lokr_linear_bwd_group; autograd's accumulation does n - 1 passes with a rounding each).
struct SumArgs {
  const void* src[4];
  void* dst;
  long total;  // elements, a multiple of the 16-byte vector
  int n;
};
template <typename T>
__global__ __launch_bounds__(NTHREADS) void sum_rows_kernel(SumArgs a) {
  constexpr int VEC = TT<T>::VEC;
  const long nvec = a.total / VEC;
  const long stride = (long)gridDim.x * NTHREADS;
  for (long v = (long)blockIdx.x * NTHREADS + threadIdx.x; v < nvec; v += stride) {
    u32x4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)  // all loads of a vector before the first use; absent sources re-read source 0 (selected out below)
      raw[i] = *reinterpret_cast<const u32x4*>(static_cast<const T*>(a.src[i < a.n ? i : 0]) + v * VEC);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T iv[VEC];
      *reinterpret_cast<u32x4*>(iv) = raw[i];
      const float on = i < a.n ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += on * TT<T>::to_f(iv[e]);
    }
    T ov[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ov[e] = TT<T>::from_f(acc[e]);
    *reinterpret_cast<u32x4*>(static_cast<T*>(a.dst) + v * VEC) = *reinterpret_cast<u32x4*>(ov);
  }
}

// x * s - b * wm with ONE evaluation order in every kernel and layout (the product b * wm rounded, then one fused multiply-add): an
// NCHW tensor and its channels_last twin give the same bits (tests/test_gpu_custom_ops.py), whatever the compiler would contract
__device__ __forceinline__ float chan_affine_f(float x, float s, float b, float wm) { return __fmaf_rn(x, s, -__fmul_rn(b, wm)); }

// out[o, c, i] = a[o, c, i] * (s0 + w[c] * mult) - bias[c] * w[c] * mult
template <typename T>
__global__ __launch_bounds__(NTHREADS) void chan_scale_kernel(ChanArgs a) {
  constexpr int VEC = TT<T>::VEC;
  const T* in = static_cast<const T*>(a.a_in);
  T* out = static_cast<T*>(a.out);
  const long total = a.outer * a.C * a.inner;
  const bool vec_ok = ((a.inner == 1 ? a.C : a.inner) % VEC == 0) && vec_aligned<T>(in, VEC) && vec_aligned<T>(out, VEC);
  const long stride = (long)gridDim.x * NTHREADS;
  if (vec_ok) {
    // UNR independent 16-byte loads per lane in flight before the first use: at 5 MB per launch the kernel lives on
    // memory-level parallelism, not on occupancy (a launch is one or two rounds of workgroups)
    constexpr int UNR = 4;
    const long nvec = total / VEC;
    for (long v0 = (long)blockIdx.x * NTHREADS + threadIdx.x; v0 < nvec; v0 += stride * UNR) {
      u32x4 raw[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long v = v0 + u * stride;
        raw[u] = *reinterpret_cast<const u32x4*>(in + (v < nvec ? v : v0) * VEC);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long v = v0 + u * stride;
        if (v >= nvec) break;
        const long e0 = v * VEC;
        T iv[VEC], ov[VEC];
        *reinterpret_cast<u32x4*>(iv) = raw[u];
        if (a.inner == 1) {
          const long c0 = e0 % a.C;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float wm = a.w[c0 + e] * a.mult;
            const float b = a.bias ? a.bias[c0 + e] : 0.f;
            ov[e] = TT<T>::from_f(chan_affine_f(TT<T>::to_f(iv[e]), a.s0 + wm, b, wm));
          }
        } else {
          const long c = (e0 / a.inner) % a.C;
          const float wm = a.w[c] * a.mult;
          const float b = a.bias ? a.bias[c] : 0.f;
#pragma unroll
          for (int e = 0; e < VEC; ++e) ov[e] = TT<T>::from_f(chan_affine_f(TT<T>::to_f(iv[e]), a.s0 + wm, b, wm));
        }
        *reinterpret_cast<u32x4*>(out + e0) = *reinterpret_cast<u32x4*>(ov);
      }
    }
  } else {
    for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += stride) {
      const long c = (e / a.inner) % a.C;
      const float wm = a.w[c] * a.mult;
      const float b = a.bias ? a.bias[c] : 0.f;
      out[e] = TT<T>::from_f(chan_affine_f(TT<T>::to_f(in[e]), a.s0 + wm, b, wm));
    }
  }
}

// dw[c] += mult * sum_{o, i} a[o, c, i] * (b[o, c, i] - bias[c])
// Linear layout (inner == 1): block (x, y) owns 64 channels x a slab of rows; lanes run along channels
// (coalesced), the 4 waves take interleaved rows, then LDS + one atomic per channel.
// Conv layout (inner > 1): one block per (outer index, channel); the plane is reduced with wave shuffles.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void chan_reduce_kernel(ChanArgs a) {
  const T* A = static_cast<const T*>(a.a_in);
  const T* B = static_cast<const T*>(a.b_in);
  __shared__ float red[NTHREADS];
  const int tid = threadIdx.x;
  if (a.inner == 1 && a.vec) {
    // vector layout (host: C % VEC == 0, aligned): a lane owns VEC consecutive channels (16-byte loads), a wave 64 * VEC,
    // the 4 waves take interleaved rows of the slab; LDS sum over the waves, one atomic per channel.
    constexpr int VEC = TT<T>::VEC;
    __shared__ float redv[NWAVES][64 * VEC];
    const int lane = tid & 63, wave = tid >> 6;
    const long c0 = ((long)blockIdx.x * 64 + lane) * VEC;
    const long rows_per = (a.outer + gridDim.y - 1) / gridDim.y;
    const long rbeg = (long)blockIdx.y * rows_per;
    long rend = rbeg + rows_per;
    if (rend > a.outer) rend = a.outer;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (c0 < a.C) {
      float bv[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) bv[e] = a.bias ? a.bias[c0 + e] : 0.f;
      for (long r = rbeg + wave; r < rend; r += 2 * NWAVES) {  // two rows per iteration: four loads in flight
        const bool two = r + NWAVES < rend;
        T av[2][VEC], gv[2][VEC];
        *reinterpret_cast<u32x4*>(av[0]) = *reinterpret_cast<const u32x4*>(A + r * a.C + c0);
        *reinterpret_cast<u32x4*>(gv[0]) = *reinterpret_cast<const u32x4*>(B + r * a.C + c0);
        *reinterpret_cast<u32x4*>(av[1]) = *reinterpret_cast<const u32x4*>(A + (two ? r + NWAVES : r) * a.C + c0);
        *reinterpret_cast<u32x4*>(gv[1]) = *reinterpret_cast<const u32x4*>(B + (two ? r + NWAVES : r) * a.C + c0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          s[e] = fmaf(TT<T>::to_f(av[0][e]), TT<T>::to_f(gv[0][e]) - bv[e], s[e]);
          if (two) s[e] = fmaf(TT<T>::to_f(av[1][e]), TT<T>::to_f(gv[1][e]) - bv[e], s[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) redv[wave][lane * VEC + e] = s[e];
    __syncthreads();
    for (int i = tid; i < 64 * VEC; i += NTHREADS) {
      const long c = (long)blockIdx.x * 64 * VEC + i;
      if (c < a.C)
        __hip_atomic_fetch_add(a.dw + c, a.mult * (redv[0][i] + redv[1][i] + redv[2][i] + redv[3][i]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (a.inner == 1) {
    const long c = (long)blockIdx.x * 64 + (tid & 63);
    const long rows_per = (a.outer + gridDim.y - 1) / gridDim.y;
    const long rbeg = (long)blockIdx.y * rows_per;
    long rend = rbeg + rows_per;
    if (rend > a.outer) rend = a.outer;
    float s = 0.f;
    if (c < a.C) {
      const float b = a.bias ? a.bias[c] : 0.f;
      for (long r = rbeg + (tid >> 6); r < rend; r += NWAVES)
        s = fmaf(TT<T>::to_f(A[r * a.C + c]), TT<T>::to_f(B[r * a.C + c]) - b, s);
    }
    red[tid] = s;
    __syncthreads();
    if (tid < 64 && c < a.C) {
      const float t = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
      __hip_atomic_fetch_add(a.dw + c, a.mult * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    const long c = blockIdx.x;
    const long o = blockIdx.y;
    const float b = a.bias ? a.bias[c] : 0.f;
    const T* pa = A + (o * a.C + c) * a.inner;
    const T* pb = B + (o * a.C + c) * a.inner;
    float s = 0.f;
    for (long i = tid; i < a.inner; i += NTHREADS) s = fmaf(TT<T>::to_f(pa[i]), TT<T>::to_f(pb[i]) - b, s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0)
      __hip_atomic_fetch_add(a.dw + c, a.mult * (red[0] + red[1] + red[2] + red[3]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- round 6: the whole backward of the per-channel scale in ONE pass (inner == 1, vector layout) -----------------------------------
//   da[r, c]  = g[r, c] * (s0 + w[c] * mult)                      (chan_scale_kernel on g)
//   dw[c]    += mult * sum_r g[r, c] * (a[r, c] - bias[c])        (chan_reduce_kernel)
// The two-launch form reads g twice and pays two launch floors per layer (350 layers x 2 in the SDXL (IA)^3 step: chan_scale 7.9 us,
// chan_reduce 6.0 us per call, profiles/r06_c14_ia3_kernel_stats.csv).  Block (x, y): 64 * VEC channels x a slab of rows; a lane owns VEC
// consecutive channels (16-byte accesses), the 4 waves take interleaved rows, 4 rows per wave in flight (8 loads per lane before the
// first use); LDS sum over the waves, one atomic per channel and workgroup.
// FWD: the forward in the same layout (round 6: chan_scale_kernel pays a 64-bit modulo and VEC gathered factor loads per vector; here
// a lane's channels are fixed, its scale / offset live in registers): out = in * (s0 + w mult) - bias w mult, no reduction, `b_in` unused.
template <typename T, bool FWD = false>
__global__ __launch_bounds__(NTHREADS) void chan_bwd_kernel(ChanArgs a) {
  constexpr int VEC = TT<T>::VEC, RU = 4;
  const T* G = static_cast<const T*>(a.a_in);
  const T* A = static_cast<const T*>(a.b_in);
  T* DA = static_cast<T*>(a.out);
  __shared__ float redv[NWAVES][64 * VEC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long c0 = ((long)blockIdx.x * 64 + lane) * VEC;
  const long rows_per = (a.outer + gridDim.y - 1) / gridDim.y;
  const long rbeg = (long)blockIdx.y * rows_per;
  long rend = rbeg + rows_per;
  if (rend > a.outer) rend = a.outer;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  if (c0 < a.C) {
    float bv[VEC], sc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float wm = a.w[c0 + e] * a.mult;
      bv[e] = a.bias ? a.bias[c0 + e] : 0.f;
      sc[e] = a.s0 + wm;
      if constexpr (FWD) bv[e] = __fmul_rn(bv[e], wm);  // the forward's offset bias[c] * w[c] * mult (chan_affine_f's order)
    }
    for (long r0 = rbeg + wave; r0 < rend; r0 += RU * NWAVES) {
      u32x4 gr[RU], ar[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long r = r0 + u * NWAVES;
        const long rr = r < rend ? r : r0;  // (rows past the slab re-read row r0 and are dropped below)
        gr[u] = *reinterpret_cast<const u32x4*>(G + rr * a.C + c0);
        if constexpr (!FWD) ar[u] = *reinterpret_cast<const u32x4*>(A + rr * a.C + c0);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const long r = r0 + u * NWAVES;
        if (r >= rend) break;
        T gv[VEC], av[VEC], ov[VEC];
        *reinterpret_cast<u32x4*>(gv) = gr[u];
        if constexpr (!FWD) *reinterpret_cast<u32x4*>(av) = ar[u];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float gf = TT<T>::to_f(gv[e]);
          if constexpr (FWD) {
            ov[e] = TT<T>::from_f(__fmaf_rn(gf, sc[e], -bv[e]));
          } else {
            s[e] = fmaf(gf, TT<T>::to_f(av[e]) - bv[e], s[e]);
            ov[e] = TT<T>::from_f(gf * sc[e]);
          }
        }
        if (DA != nullptr) *reinterpret_cast<u32x4*>(DA + r * a.C + c0) = *reinterpret_cast<u32x4*>(ov);
      }
    }
  }
  if (FWD || a.dw == nullptr) return;
#pragma unroll
  for (int e = 0; e < VEC; ++e) redv[wave][lane * VEC + e] = s[e];
  __syncthreads();
  for (int i = tid; i < 64 * VEC; i += NTHREADS) {
    const long c = (long)blockIdx.x * 64 * VEC + i;
    if (c < a.C)
      __hip_atomic_fetch_add(a.dw + c, a.mult * (redv[0][i] + redv[1][i] + redv[2][i] + redv[3][i]), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace lyc
