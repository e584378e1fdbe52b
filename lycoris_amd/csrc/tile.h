// tile.h -- CDNA4 (gfx950) tile primitives shared by every adapter kernel.
//
// Conventions (wave64, 16x16 MFMA tiles everywhere):
//   * activation element type T in {__bf16, _Float16, float}; accumulation is always fp32.
//   * MFMA operand fragments (per lane l, i = l & 15, g = l >> 4):
//       A[i][KV*g .. KV*g+KV-1],  B[KV*g ..][j = l & 15]      (KV = 8 for 16-bit, 1 for fp32)
//     accumulator D: column = l & 15, rows 4*g + {0,1,2,3}.
//   * LDS operand tiles are "K-contiguous": tile[row][k], row stride LDK = BK + pad elements, so a
//     fragment is one 16-byte ds_read_b128 (16-bit types) / one ds_read_b32 (fp32).
//   * fp32 factors and fp32 intermediates that feed a 16-bit MFMA are split into hi + lo parts
//     (x = hi + lo, both representable in T) so the only rounding left in a kernel is the final
//     store of a T-typed output.  For T = float nothing is split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lyc {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Development-only phase stamps (benchmarks/ktrace.cpp builds the kernels with -DLYC_TRACE).  The shader clock is read
// into registers at each LYC_STAMP(i) and written out once by LYC_TRACE_FLUSH(): a store per stamp would sit in the
// vmcnt queue and lengthen every later s_waitcnt vmcnt, i.e. distort exactly what is being measured.
#ifdef LYC_TRACE
__device__ unsigned long long lyc_trace_buf[32];
#ifndef LYC_TRACE_BLOCK
#define LYC_TRACE_BLOCK 0
#endif
#define LYC_TRACE_DECL unsigned long long lyc_t[32] = {0}
#define LYC_STAMP(i)                                   \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    lyc_t[i] = __builtin_readcyclecounter();           \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define LYC_TRACE_FLUSH()                                                                                      \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y == LYC_TRACE_BLOCK) \
      _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) if (lyc_t[i_]) lyc_trace_buf[i_] = lyc_t[i_];          \
  } while (0)
// stamp from a helper function that has no LYC_TRACE_DECL in scope: written straight to memory (perturbs a little)
#define LYC_STAMP_DIRECT(i)                                                                                    \
  do {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    if (threadIdx.x == 0 && blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y == LYC_TRACE_BLOCK) \
      lyc_trace_buf[i] = __builtin_readcyclecounter();                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#else
#define LYC_TRACE_DECL do {} while (0)
#define LYC_STAMP(i) do {} while (0)
#define LYC_STAMP_DIRECT(i) do {} while (0)
#define LYC_TRACE_FLUSH() do {} while (0)
#endif

constexpr int WAVE = 64;
constexpr int NTHREADS = 256;  // every kernel in this library runs 4 waves per workgroup
constexpr int NWAVES = NTHREADS / WAVE;

template <typename T>
struct TT;

template <>
struct TT<__bf16> {
  using frag = bf16x8;
  static constexpr int KV = 8;        // K elements per lane per MFMA
  static constexpr int KSTEP = 32;    // K per MFMA
  static constexpr bool SPLIT = true; // fp32 operands need hi/lo
  static constexpr int VEC = 8;       // elements per 16-byte global access
  static __device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
  static __device__ __forceinline__ __bf16 from_f(float v) { return (__bf16)v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct TT<_Float16> {
  using frag = f16x8;
  static constexpr int KV = 8;
  static constexpr int KSTEP = 32;
  static constexpr bool SPLIT = true;
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
  static __device__ __forceinline__ _Float16 from_f(float v) { return (_Float16)v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct TT<float> {
  using frag = float;
  static constexpr int KV = 1;
  static constexpr int KSTEP = 4;
  static constexpr bool SPLIT = false;
  static constexpr int VEC = 4;
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// LDS row stride (elements) for a K-contiguous tile of BK columns: +16 bytes of padding keeps rows
// 16-byte aligned and spreads ds_read_b128 fragment reads over the bank row.
template <typename T, int BK>
struct TileLD {
  static constexpr int value = BK + 16 / (int)sizeof(T);
};

template <typename T>
__device__ __forceinline__ typename TT<T>::frag load_frag(const T* tile, int ld, int row, int kofs) {
  return *reinterpret_cast<const typename TT<T>::frag*>(tile + row * ld + kofs);
}

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// true when `p` and a row pitch of `ld` elements keep every VEC-aligned column 16-byte aligned
template <typename T>
__host__ __device__ __forceinline__ bool vec_aligned(const void* p, long ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && ((ld * (long)sizeof(T)) % 16 == 0);
}

// hi/lo split of an fp32 value into two T values (T = float: lo unused).
template <typename T>
__device__ __forceinline__ void split_f(float v, T& hi, T& lo) {
  hi = TT<T>::from_f(v);
  lo = TT<T>::from_f(v - TT<T>::to_f(hi));
}

// ---------------------------------------------------------------------------------------------
// Staging: global -> LDS.  All routines are cooperative over the 256 threads of the workgroup,
// zero-fill everything outside [rows_valid) x [k_valid) and never read out of bounds.
// ---------------------------------------------------------------------------------------------

// Row-major activation tile: dst[r][k] = src[(row0 + r) * ld_src + k0 + k],  r < ROWS, k < BK.
// Fast path: 16-byte global loads when the row pitch, base pointer and k0 allow it.
template <typename T, int ROWS, int BK>
__device__ __forceinline__ void stage_rows(T* __restrict__ dst, const T* __restrict__ src, long ld_src, long row0,
                                           long rows_total, long k0, long k_total, bool vec_ok) {
  constexpr int LD = TileLD<T, BK>::value;
  constexpr int VEC = TT<T>::VEC;
  constexpr int VPR = BK / VEC;  // vectors per row
  constexpr int NV = ROWS * VPR;
  const int tid = threadIdx.x;
  if (vec_ok) {
#pragma unroll
    for (int v = tid; v < NV; v += NTHREADS) {
      const int r = v / VPR, kv = (v % VPR) * VEC;
      u32x4 val = {0u, 0u, 0u, 0u};
      const long gr = row0 + r, gk = k0 + kv;
      if (gr < rows_total && gk + VEC <= k_total) {
        val = *reinterpret_cast<const u32x4*>(src + gr * ld_src + gk);
      } else if (gr < rows_total && gk < k_total) {
        T tmp[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) tmp[e] = (gk + e < k_total) ? src[gr * ld_src + gk + e] : TT<T>::from_f(0.f);
        val = *reinterpret_cast<u32x4*>(tmp);
      }
      *reinterpret_cast<u32x4*>(dst + r * LD + kv) = val;
    }
  } else {
    for (int e = tid; e < ROWS * BK; e += NTHREADS) {
      const int r = e / BK, k = e % BK;
      const long gr = row0 + r, gk = k0 + k;
      dst[r * LD + k] = (gr < rows_total && gk < k_total) ? src[gr * ld_src + gk] : TT<T>::from_f(0.f);
    }
  }
}

// Transposed activation tile: the source is row-major [k][n] (n contiguous), the LDS image is
// K-contiguous [n][k]:  dst[n][k] = src[(k0 + k) * ld_src + n0 + n],  n < COLS, k < BK.
// 16-bit fast path: each thread loads a 4(k) x 8(n) block (four 16-byte loads), transposes it in
// registers with v_perm_b32 and writes eight 8-byte rows.  fp32: 4 x 4 blocks.
template <typename T, int COLS, int BK>
__device__ __forceinline__ void stage_cols(T* __restrict__ dst, const T* __restrict__ src, long ld_src, long k0,
                                           long k_total, long n0, long n_total, bool vec_ok) {
  constexpr int LD = TileLD<T, BK>::value;
  const int tid = threadIdx.x;
  if constexpr (sizeof(T) == 2) {
    constexpr int NB = COLS / 8, KB = BK / 4;  // blocks along n, along k
    static_assert(COLS % 8 == 0 && BK % 4 == 0, "tile shape");
    if (vec_ok) {
#pragma unroll
      for (int b = tid; b < NB * KB; b += NTHREADS) {
        const int nb = b % NB, kb = b / NB;  // consecutive threads -> consecutive n (coalesced rows)
        const long gn = n0 + nb * 8;
        u32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long gk = k0 + kb * 4 + j;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (gk < k_total && gn + 8 <= n_total) {
            v = *reinterpret_cast<const u32x4*>(src + gk * ld_src + gn);
          } else if (gk < k_total && gn < n_total) {
            T tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = (gn + e < n_total) ? src[gk * ld_src + gn + e] : TT<T>::from_f(0.f);
            v = *reinterpret_cast<u32x4*>(tmp);
          }
          r[j] = v;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          // columns 2w (low halves) and 2w+1 (high halves) of the four k rows
          u32x2 lo, hi;
          lo[0] = __builtin_amdgcn_perm(r[1][w], r[0][w], 0x05040100u);  // {r0.lo16, r1.lo16}
          lo[1] = __builtin_amdgcn_perm(r[3][w], r[2][w], 0x05040100u);
          hi[0] = __builtin_amdgcn_perm(r[1][w], r[0][w], 0x07060302u);  // {r0.hi16, r1.hi16}
          hi[1] = __builtin_amdgcn_perm(r[3][w], r[2][w], 0x07060302u);
          *reinterpret_cast<u32x2*>(dst + (nb * 8 + 2 * w) * LD + kb * 4) = lo;
          *reinterpret_cast<u32x2*>(dst + (nb * 8 + 2 * w + 1) * LD + kb * 4) = hi;
        }
      }
      return;
    }
  } else {
    constexpr int NB = COLS / 4, KB = BK / 4;
    if (vec_ok) {
#pragma unroll
      for (int b = tid; b < NB * KB; b += NTHREADS) {
        const int nb = b % NB, kb = b / NB;
        const long gn = n0 + nb * 4;
        f32x4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long gk = k0 + kb * 4 + j;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (gk < k_total && gn + 4 <= n_total) {
            v = *reinterpret_cast<const f32x4*>(src + gk * ld_src + gn);
          } else if (gk < k_total && gn < n_total) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (gn + e < n_total) ? src[gk * ld_src + gn + e] : 0.f;
          }
          r[j] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x4 o = {r[0][e], r[1][e], r[2][e], r[3][e]};
          *reinterpret_cast<f32x4*>(dst + (nb * 4 + e) * LD + kb * 4) = o;
        }
      }
      return;
    }
  }
  for (int e = tid; e < COLS * BK; e += NTHREADS) {
    const int n = e % COLS, k = e / COLS;
    const long gn = n0 + n, gk = k0 + k;
    dst[n * LD + k] = (gn < n_total && gk < k_total) ? src[gk * ld_src + gn] : TT<T>::from_f(0.f);
  }
}

// fp32 factor tile with arbitrary strides, split into hi (and lo) K-contiguous LDS tiles:
//   dst[r][k] = scale * src[(row0 + r) * rstride + (k0 + k) * kstride]
// Factors are small and L2-resident, so the generic strided gather is good enough; consecutive
// threads walk whichever of the two dimensions is contiguous in memory.
template <typename T, int ROWS, int BK>
__device__ __forceinline__ void stage_factor(T* __restrict__ dst_hi, T* __restrict__ dst_lo,
                                             const float* __restrict__ src, long rstride, long kstride, long row0,
                                             long rows_total, long k0, long k_total, float scale) {
  constexpr int LD = TileLD<T, BK>::value;
  const int tid = threadIdx.x;
  const bool k_fast = (kstride == 1);
  for (int e = tid; e < ROWS * BK; e += NTHREADS) {
    int r, k;
    if (k_fast) {
      r = e / BK;
      k = e % BK;
    } else {
      r = e % ROWS;
      k = e / ROWS;
    }
    const long gr = row0 + r, gk = k0 + k;
    float v = 0.f;
    if (gr < rows_total && gk < k_total) v = scale * src[gr * rstride + gk * kstride];
    if constexpr (TT<T>::SPLIT) {
      T hi, lo;
      split_f<T>(v, hi, lo);
      dst_hi[r * LD + k] = hi;
      dst_lo[r * LD + k] = lo;
    } else {
      dst_hi[r * LD + k] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA helpers
// ---------------------------------------------------------------------------------------------

// acc[MI][NI] += A(rows a_row0 + 16*mi ..) * B(rows b_row0 + 16*ni ..)^T over one BK-wide LDS tile.
// B_SPLIT: B has hi and lo tiles (two MFMAs per pair).  A is exact T.
template <typename T, int BK, int MI, int NI, bool B_SPLIT>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[MI][NI], const T* a_tile, int a_row0, const T* b_hi,
                                         const T* b_lo, int b_row0) {
  constexpr int LD = TileLD<T, BK>::value;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < BK / TT<T>::KSTEP; ++ks) {
    const int kofs = ks * TT<T>::KSTEP + g * TT<T>::KV;
    typename TT<T>::frag a[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = load_frag<T>(a_tile, LD, a_row0 + 16 * mi + i, kofs);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      typename TT<T>::frag bh = load_frag<T>(b_hi, LD, b_row0 + 16 * ni + i, kofs);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(a[mi], bh, acc[mi][ni]);
      if constexpr (B_SPLIT && TT<T>::SPLIT) {
        typename TT<T>::frag bl = load_frag<T>(b_lo, LD, b_row0 + 16 * ni + i, kofs);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(a[mi], bl, acc[mi][ni]);
      }
    }
  }
}

// Both operands split (fp32 x fp32): hi*hi + lo*hi + hi*lo (lo*lo is below fp32 resolution).
template <typename T, int BK, int MI, int NI>
__device__ __forceinline__ void mma_tile_ss(f32x4 (&acc)[MI][NI], const T* a_hi, const T* a_lo, int a_row0,
                                            const T* b_hi, const T* b_lo, int b_row0) {
  constexpr int LD = TileLD<T, BK>::value;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < BK / TT<T>::KSTEP; ++ks) {
    const int kofs = ks * TT<T>::KSTEP + g * TT<T>::KV;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      typename TT<T>::frag ah = load_frag<T>(a_hi, LD, a_row0 + 16 * mi + i, kofs);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        typename TT<T>::frag bh = load_frag<T>(b_hi, LD, b_row0 + 16 * ni + i, kofs);
        acc[mi][ni] = TT<T>::mma(ah, bh, acc[mi][ni]);
        if constexpr (TT<T>::SPLIT) {
          typename TT<T>::frag al = load_frag<T>(a_lo, LD, a_row0 + 16 * mi + i, kofs);
          typename TT<T>::frag bl = load_frag<T>(b_lo, LD, b_row0 + 16 * ni + i, kofs);
          acc[mi][ni] = TT<T>::mma(al, bh, acc[mi][ni]);
          acc[mi][ni] = TT<T>::mma(ah, bl, acc[mi][ni]);
        }
      }
    }
  }
}

// Spill a wave's accumulators into an fp32 LDS tile out[row][col] (row stride ld_out floats).
template <int MI, int NI>
__device__ __forceinline__ void acc_to_lds(float* out, int ld_out, const f32x4 (&acc)[MI][NI], int row0, int col0,
                                           float alpha) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(row0 + 16 * mi + 4 * g + r) * ld_out + col0 + 16 * ni + c] = alpha * acc[mi][ni][r];
}

// Atomically add a wave's accumulators into a strided fp32 matrix in global memory:
//   dst[(row0 + r) * rs + (col0 + c) * cs] += alpha * acc
template <int MI, int NI>
__device__ __forceinline__ void acc_atomic_add(float* dst, long rs, long cs, long rows_total, long cols_total,
                                               const f32x4 (&acc)[MI][NI], long row0, long col0, float alpha) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gr = row0 + 16 * mi + 4 * g + r, gc = col0 + 16 * ni + c;
        if (gr < rows_total && gc < cols_total)
          __hip_atomic_fetch_add(dst + gr * rs + gc * cs, alpha * acc[mi][ni][r], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
}

// Cooperative coalesced store of an fp32 LDS tile to a row-major global matrix (16 bytes per lane when alignment
// allows).  The destination holds T elements, or fp32 when `dst_f32` (LYC_F32_ROWS: un-rounded rows for col2im).
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void store_tile(void* __restrict__ dst_, long ld_dst, const float* __restrict__ tile,
                                           int ld_tile, long row0, long rows_total, long col0, long cols_total,
                                           bool dst_f32) {
  const int tid = threadIdx.x;
  if (dst_f32 || sizeof(T) == 4) {
    float* dst = static_cast<float*>(dst_);
    const bool vec_ok = vec_aligned<float>(dst, ld_dst);
    constexpr int VPR = COLS / 4;
    for (int v = tid; v < ROWS * VPR; v += NTHREADS) {
      const int r = v / VPR, c = (v % VPR) * 4;
      const long gr = row0 + r, gc = col0 + c;
      if (gr >= rows_total || gc >= cols_total) continue;
      const f32x4 val = *reinterpret_cast<const f32x4*>(tile + r * ld_tile + c);
      if (vec_ok && gc + 4 <= cols_total) {
        *reinterpret_cast<f32x4*>(dst + gr * ld_dst + gc) = val;
      } else {
        for (int e = 0; e < 4 && gc + e < cols_total; ++e) dst[gr * ld_dst + gc + e] = val[e];
      }
    }
    return;
  }
  T* dst = static_cast<T*>(dst_);
  constexpr int VEC = TT<T>::VEC;
  const bool vec_ok = vec_aligned<T>(dst, ld_dst);
  constexpr int VPR = COLS / VEC;
  for (int v = tid; v < ROWS * VPR; v += NTHREADS) {
    const int r = v / VPR, c = (v % VPR) * VEC;
    const long gr = row0 + r, gc = col0 + c;
    if (gr >= rows_total || gc >= cols_total) continue;
    T tmp[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) tmp[e] = TT<T>::from_f(tile[r * ld_tile + c + e]);
    if (vec_ok && gc + VEC <= cols_total) {
      *reinterpret_cast<u32x4*>(dst + gr * ld_dst + gc) = *reinterpret_cast<u32x4*>(tmp);
    } else {
      for (int e = 0; e < VEC && gc + e < cols_total; ++e) dst[gr * ld_dst + gc + e] = tmp[e];
    }
  }
}

}  // namespace lyc
