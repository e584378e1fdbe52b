// kron_fast.h -- fast path of the Kronecker kernel for 16-bit activations (bf16 / fp16), gfx950.
//
// Same math as kron_kernel (lokr_kernels.h); taken when Gin == Gout == G with 16 % G == 0, K % 8 == 0 and x is
// 16-byte aligned (every SDXL / SD1.5 layer with factor in {1,2,4,8,16}).  Differences, all aimed at the HBM roofline:
//
//   * activation fragments go HBM -> registers directly: a row of x3 is consumed by exactly one wave, so staging it
//     through LDS buys no reuse; each lane issues all 16-byte loads of a K chunk (<= 160) before anything waits;
//   * the w2 tile of a chunk is converted fp32 -> hi/lo once, with float4 loads and 8-byte LDS writes (4x4 register
//     transposes for the w2^T orientation of the backward pass);
//   * stage 2 (the G x G mix with w1) runs on the matrix cores without leaving registers: the accumulator layout of a
//     16x16 MFMA tile (lane = column, 4 consecutive rows per lane) IS the B-operand layout of v_mfma_f32_16x16x16, so
//       Y[(m,p), n] = sum_(m',u) (I (x) w1)[(m,p),(m',u)] * S1[(m',u), n]
//     is three 16x16x16 MFMAs per tile (w1 hi/lo x S1 hi/lo), no LDS round trip, no VALU mix;
//   * outputs leave through an LDS image so every row segment is written with 16-byte stores (128 B per row);
//   * backward: the w1 gradient uses the block trick D[(m',u),(m'',p)] += S1[(m',u),:] . xref[(m'',p),:] with the
//     diagonal blocks summed at the end (one MFMA per 16 rows instead of one per m).
#pragma once
#include "lokr_kernels.h"

namespace lyc {

template <typename T>
struct Mma16;
template <>
struct Mma16<__bf16> {
  typedef __attribute__((ext_vector_type(4))) short frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma16<_Float16> {
  typedef __attribute__((ext_vector_type(4))) _Float16 frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  }
};

constexpr int KF_RT = 128;        // stage-1 rows per workgroup
constexpr int KF_TQ = 64;         // output columns (n) per workgroup
constexpr int KF_KC = 96;         // K chunk held in LDS (w2 tile) / registers (x fragments)
constexpr int KF_KS = KF_KC / 32; // MFMA k-steps per chunk
constexpr int KF_LDB = KF_KC + 8; // LDS row stride of the w2 tiles (elements)
constexpr int KF_LDY = KF_TQ + 8; // LDS row stride of the epilogue images (elements)

__host__ __device__ constexpr int kron_fast_lds_bytes(bool with_dw1) {
  const int stage = 2 * KF_TQ * KF_LDB * 2;
  const int epi = (with_dw1 ? 3 : 1) * KF_RT * KF_LDY * 2;
  return (stage > epi ? stage : epi) + (with_dw1 ? NWAVES * 256 * 4 : 0);
}

template <typename T>
__device__ __forceinline__ void kf_split4(const f32x4& v, T (&hi)[4], T (&lo)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) split_f<T>(v[e], hi[e], lo[e]);
}

// w2 chunk -> registers -> Bh/Bl[n][k] (n < TQ, k < KC).  Element (n, k) lives at w2[n * s2n + k * s2k]; one of the
// strides is 1.  Loading and converting are separate so that every global load of a chunk is in flight at once and the
// next chunk can be fetched while the matrix cores work on the current one.  The geometry is static (KC = 96).
constexpr int KF_NRAW = 8;  // float4 registers per thread: 6 (row mode) or 2 x 4 (transposed mode)

enum { KF_W2_ROWS = 0, KF_W2_COLS = 1, KF_W2_SCALAR = 2 };
static_assert((KF_TQ * (KF_KC / 4)) % NTHREADS == 0 && KF_TQ * (KF_KC / 4) / NTHREADS <= KF_NRAW, "row-mode geometry");
static_assert((KF_TQ / 4) * (KF_KC / 4) <= 2 * NTHREADS, "transposed-mode geometry");

__device__ __forceinline__ int kf_w2_mode(const float* w2, long s2n, long s2k) {
  const bool aligned = (reinterpret_cast<uintptr_t>(w2) & 15u) == 0;
  if (s2k == 1 && aligned && (s2n % 4 == 0)) return KF_W2_ROWS;
  if (s2n == 1 && aligned && (s2k % 4 == 0)) return KF_W2_COLS;
  return KF_W2_SCALAR;
}

__device__ __forceinline__ void kf_load_w2(f32x4 (&raw)[KF_NRAW], int mode, const float* __restrict__ w2, long s2n,
                                           long s2k, long n0, long N, long k0, long K) {
  const int tid = threadIdx.x;
  if (mode == KF_W2_ROWS) {
    constexpr int QPR = KF_KC / 4;  // float4 per row
#pragma unroll
    for (int it = 0; it < KF_TQ * QPR / NTHREADS; ++it) {
      const int e = tid + NTHREADS * it;
      const int n = e / QPR, c = e % QPR;
      const long gn = n0 + n, gk = k0 + 4 * c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gn < N && gk < K) {
        if (gk + 4 <= K) {
          v = *reinterpret_cast<const f32x4*>(w2 + gn * s2n + gk);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (gk + j < K) ? w2[gn * s2n + gk + j] : 0.f;
        }
      }
      raw[it] = v;
    }
  } else if (mode == KF_W2_COLS) {
    constexpr int NQ = KF_TQ / 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int b = tid + NTHREADS * it;
      const int nq = b % NQ, kq = b / NQ;
      const long gn = n0 + 4 * nq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long gk = k0 + 4 * kq + j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kq < KF_KC / 4 && gk < K && gn < N) {
          if (gn + 4 <= N) {
            v = *reinterpret_cast<const f32x4*>(w2 + gk * s2k + gn);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (gn + e < N) ? w2[gk * s2k + gn + e] : 0.f;
          }
        }
        raw[4 * it + j] = v;
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void kf_store_w2(T* __restrict__ Bh, T* __restrict__ Bl, const f32x4 (&raw)[KF_NRAW],
                                            int mode, const float* __restrict__ w2, long s2n, long s2k, long n0,
                                            long N, long k0, long K) {
  const int tid = threadIdx.x;
  if (mode == KF_W2_ROWS) {
    constexpr int QPR = KF_KC / 4;
#pragma unroll
    for (int it = 0; it < KF_TQ * QPR / NTHREADS; ++it) {
      const int e = tid + NTHREADS * it;
      const int n = e / QPR, c = e % QPR;
      T h[4], l[4];
      kf_split4<T>(raw[it], h, l);
      *reinterpret_cast<u32x2*>(Bh + n * KF_LDB + 4 * c) = *reinterpret_cast<u32x2*>(h);
      *reinterpret_cast<u32x2*>(Bl + n * KF_LDB + 4 * c) = *reinterpret_cast<u32x2*>(l);
    }
  } else if (mode == KF_W2_COLS) {
    constexpr int NQ = KF_TQ / 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int b = tid + NTHREADS * it;
      const int nq = b % NQ, kq = b / NQ;
      if (kq < KF_KC / 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 col = {raw[4 * it][e], raw[4 * it + 1][e], raw[4 * it + 2][e], raw[4 * it + 3][e]};
          T h[4], l[4];
          kf_split4<T>(col, h, l);
          *reinterpret_cast<u32x2*>(Bh + (4 * nq + e) * KF_LDB + 4 * kq) = *reinterpret_cast<u32x2*>(h);
          *reinterpret_cast<u32x2*>(Bl + (4 * nq + e) * KF_LDB + 4 * kq) = *reinterpret_cast<u32x2*>(l);
        }
      }
    }
  } else {  // unaligned / odd strides: element-wise, straight from global
    for (int e = tid; e < KF_TQ * KF_KC; e += NTHREADS) {
      int n, k;
      if (s2k == 1) {
        n = e / KF_KC;
        k = e % KF_KC;
      } else {
        n = e % KF_TQ;
        k = e / KF_TQ;
      }
      const long gn = n0 + n, gk = k0 + k;
      const float v = (gn < N && gk < K) ? w2[gn * s2n + gk * s2k] : 0.f;
      T h, l;
      split_f<T>(v, h, l);
      Bh[n * KF_LDB + k] = h;
      Bl[n * KF_LDB + k] = l;
    }
  }
}

template <typename T, bool WITH_DW1>
__global__ __launch_bounds__(NTHREADS) void kron_fast_kernel(KronArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = 2, NI = KF_TQ / 16;
  T* Bh = reinterpret_cast<T*>(smem);
  T* Bl = Bh + KF_TQ * KF_LDB;
  T* Ys = reinterpret_cast<T*>(smem);           // epilogue images alias the w2 tiles
  T* S1h = Ys + KF_RT * KF_LDY;
  T* S1l = S1h + KF_RT * KF_LDY;
  float* red = reinterpret_cast<float*>(smem + (2 * KF_TQ * KF_LDB * 2 > 3 * KF_RT * KF_LDY * 2 ? 2 * KF_TQ * KF_LDB * 2
                                                                                                   : 3 * KF_RT * KF_LDY * 2));
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;

  const T* x = static_cast<const T*>(a.x);
  T* y = static_cast<T*>(a.y);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int G = a.Gin, K = a.K, N = a.N;
  const int lg = 31 - __builtin_clz((unsigned)G);  // G is a power of two (16 % G == 0)
  const int TM = KF_RT >> lg;
  const long m0 = (long)blockIdx.x * TM;
  const long n0 = (long)blockIdx.y * KF_TQ;
  const long row0 = m0 << lg;
  long rows_end = row0 + KF_RT;
  if (rows_end > (a.M << lg)) rows_end = a.M << lg;

  // stage-2 operand: (I (x) w1) restricted to one 16x16 block; lane (i = li, g) holds k = 4g .. 4g+3
  F4 a2h, a2l;
  {
    const int mi_ = li >> lg, po = li & (G - 1);
    T h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = 4 * g + j;
      const float v = ((kk >> lg) == mi_) ? a.w1[po * a.s1o + (kk & (G - 1)) * a.s1i] : 0.f;
      split_f<T>(v, h[j], l[j]);
    }
    a2h = *reinterpret_cast<F4*>(h);
    a2l = *reinterpret_cast<F4*>(l);
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  using F8v = F8;
  auto load_a = [&](F8v (&dst)[MI][KF_KS], long k0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long gr = row0 + wave * 32 + mi * 16 + li;
#pragma unroll
      for (int ks = 0; ks < KF_KS; ++ks) {
        u32x4 v = {0u, 0u, 0u, 0u};
        const long gk = k0 + ks * 32 + 8 * g;
        if (gr < rows_end && gk < K) v = *reinterpret_cast<const u32x4*>(x + gr * K + gk);  // K % 8 == 0
        dst[mi][ks] = *reinterpret_cast<F8v*>(&v);
      }
    }
  };
  const int w2mode = kf_w2_mode(a.w2, a.s2n, a.s2k);
  F8 af[MI][KF_KS];
  f32x4 raw[KF_NRAW];
  load_a(af, 0);
  kf_load_w2(raw, w2mode, a.w2, a.s2n, a.s2k, n0, N, 0, K);
  for (long k0 = 0; k0 < K; k0 += KF_KC) {
    const long krem = K - k0;
    const int nks = krem >= KF_KC ? KF_KS : (int)((krem + 31) / 32);
    if (k0 > 0) __syncthreads();  // previous chunk's fragment reads are done
    kf_store_w2<T>(Bh, Bl, raw, w2mode, a.w2, a.s2n, a.s2k, n0, N, k0, K);
    __syncthreads();
    const bool more = k0 + KF_KC < K;
    // the next chunk's w2 values travel while the matrix cores run this chunk ...
    if (more) kf_load_w2(raw, w2mode, a.w2, a.s2n, a.s2k, n0, N, k0 + KF_KC, K);
#pragma unroll
    for (int ks = 0; ks < KF_KS; ++ks) {
      if (ks < nks) {
        const int kofs = ks * 32 + 8 * g;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const F8 bh = *reinterpret_cast<const F8*>(Bh + (16 * ni + li) * KF_LDB + kofs);
          const F8 bl = *reinterpret_cast<const F8*>(Bl + (16 * ni + li) * KF_LDB + kofs);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            acc[mi][ni] = TT<T>::mma(af[mi][ks], bh, acc[mi][ni]);
            acc[mi][ni] = TT<T>::mma(af[mi][ks], bl, acc[mi][ni]);
          }
        }
      }
      // ... and each activation fragment is re-fetched in place as soon as its MFMAs have consumed it
      if (more) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long gr = row0 + wave * 32 + mi * 16 + li;
          const long gk = k0 + KF_KC + ks * 32 + 8 * g;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (gr < rows_end && gk < K) v = *reinterpret_cast<const u32x4*>(x + gr * K + gk);
          af[mi][ks] = *reinterpret_cast<F8*>(&v);
        }
      }
    }
  }
  __syncthreads();  // the epilogue images alias the w2 tiles

  // ---- stage 2 on the matrix cores, images to LDS ----
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      T h[4], l[4];
      kf_split4<T>(acc[mi][ni], h, l);
      const F4 sh = *reinterpret_cast<F4*>(h), sl = *reinterpret_cast<F4*>(l);
      f32x4 yv = zero4();
      yv = Mma16<T>::mma(a2h, sh, yv);
      yv = Mma16<T>::mma(a2l, sh, yv);
      yv = Mma16<T>::mma(a2h, sl, yv);
      const int rbase = wave * 32 + mi * 16 + 4 * g, col = 16 * ni + li;
      if (a.out_f32) {  // LYC_F32_ROWS: un-rounded rows straight from the accumulators
        float* outf = static_cast<float*>(a.y);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long R = row0 + rbase + r;
          if (R < rows_end && n0 + col < N) outf[(R >> lg) * ((long)G * N) + (R & (G - 1)) * (long)N + n0 + col] = a.alpha * yv[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Ys[(rbase + r) * KF_LDY + col] = TT<T>::from_f(a.alpha * yv[r]);
        if constexpr (WITH_DW1) {
          S1h[(rbase + r) * KF_LDY + col] = h[r];
          S1l[(rbase + r) * KF_LDY + col] = l[r];
        }
      }
    }
  __syncthreads();

  // ---- coalesced stores: each wave writes its own 32 rows (row = (m, po)), 128 B per row ----
  if (!a.out_f32) {
    const long ldy = (long)G * N;
    const bool y_vec = vec_aligned<T>(y, N);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane + 64 * it;
      const int rl = wave * 32 + c / 8, cc = (c % 8) * 8;
      const long R = row0 + rl;
      const long gn = n0 + cc;
      if (R >= rows_end || gn >= N) continue;
      const long gm = R >> lg;
      const int po = (int)(R & (G - 1));
      const T* src = Ys + rl * KF_LDY + cc;
      T* dst = y + gm * ldy + (long)po * N + gn;
      if (y_vec && gn + 8 <= N) {
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
      } else {
        for (int e = 0; e < 8 && gn + e < N; ++e) dst[e] = src[e];
      }
    }
  }

  // ---- w1 gradient (backward mode): D[(m',ui),(m'',po)] += S1[(m',ui), n] * xref[(m'',po), n], diagonal blocks ----
  if constexpr (WITH_DW1) {
    const T* xr = static_cast<const T*>(a.xref);
    const long ldr = (long)G * N;
    const bool r_vec = vec_aligned<T>(xr, N);
    f32x4 c = zero4();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rl = wave * 32 + mi * 16 + li;
      const long R = row0 + rl;
      const bool row_ok = R < rows_end;
      const long gm = R >> lg;
      const int po = (int)(R & (G - 1));
#pragma unroll
      for (int ks = 0; ks < KF_TQ / 32; ++ks) {
        const int kofs = ks * 32 + 8 * g;
        const F8 ah = *reinterpret_cast<const F8*>(S1h + rl * KF_LDY + kofs);
        const F8 al = *reinterpret_cast<const F8*>(S1l + rl * KF_LDY + kofs);
        u32x4 bv = {0u, 0u, 0u, 0u};
        const long gc = n0 + kofs;
        if (row_ok && gc < N) {
          const T* src = xr + gm * ldr + (long)po * N + gc;
          if (r_vec && gc + 8 <= N) {
            bv = *reinterpret_cast<const u32x4*>(src);
          } else {
            T tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = (gc + e < N) ? src[e] : TT<T>::from_f(0.f);
            bv = *reinterpret_cast<u32x4*>(tmp);
          }
        }
        const F8 bf = *reinterpret_cast<F8*>(&bv);
        c = TT<T>::mma(ah, bf, c);
        c = TT<T>::mma(al, bf, c);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * g + r) * 16 + li] = c[r];
    __syncthreads();
    {
      const int i = tid >> 4, j = tid & 15;  // i = (m', ui), j = (m'', po)
      if ((i >> lg) == (j >> lg)) {
        const float s = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
        __hip_atomic_fetch_add(a.dw1 + (long)(j & (G - 1)) * a.s1o + (long)(i & (G - 1)) * a.s1i, a.alpha * s, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace lyc
