// lokr_kernels.h -- Kronecker-factored adapter kernels (LoKr), gfx950.
//
// Math (reference: lycoris/functional/lokr.py:11-20 make_kron, :154-247 factored bypass;
// lycoris/modules/lokr.py:358-381 get_weight, :543-566 forward):
//   dW[(po, n), (ui, k)] = w1[po, ui] * w2[n, k] * alpha        (kron, out = po*N + n, in = ui*K + k)
//   y[m, po*N + n] = alpha * sum_ui w1[po, ui] * ( sum_k w2[n, k] * x[m, ui*K + k] )
// dW is never materialised.  Two kernels:
//
//   kron_kernel : stage 1  S1[(m, ui), n] = sum_k x3[(m, ui), k] * w2[n, k]    (MFMA, w2 split hi/lo)
//                 stage 2  y[m, po, n]    = alpha * sum_ui w1[po, ui] * S1[(m, ui), n]   (fp32 VALU from LDS)
//                 optional dW1[po, ui]   += alpha * sum_{m, n} S1[(m, ui), n] * xref[(m, po), n]   (MFMA)
//     forward : x := x, w1, w2.   backward-dx : x := g, w1 := w1^T, w2 := w2^T (strides), xref := x,
//     which yields dx and the w1 gradient in one pass over g.
//
//   kron_dw2_kernel : dW2[i, j] += alpha * sum_{(m, s)} Q[(m, s), i] * ( sum_t W[s, t] * P[(m, t), j] )
//     a "TN" weight-gradient GEMM over the M * Gs flat rows with the small w1 mix applied on the fly to
//     the P operand; split over row chunks with fp32 atomics.
#pragma once
#include "tile.h"

namespace lyc {

// Pixel-row gather of the stage-1 operand: the implicit-GEMM form of Conv2d on NHWC row matrices.  The stage-1 rows
// are (destination pixel, group); for tap (i, j) of the kh x kw window the operand row comes from the source pixel
//   mode 1 (forward,  dst = output pixel): hs = hd*sh - ph + i*dh                       (ws likewise)
//   mode 2 (backward, dst = input pixel) : hs = (hd + ph - i*dh) / sh, if divisible     (the transposed conv)
// rows that fall outside the source image (padding) read as zero.  The K dimension runs over (tap, K per tap).
struct KronGather {
  int mode;            // 0 = no gather (nn.Linear / 1x1 rows)
  int taps, kw;        // kh * kw, kw
  int Hs, Ws, Hd, Wd;  // source / destination spatial sizes
  int sh, sw, ph, pw, dh, dw;
  long s2t;            // w2 element offset per tap (per-tap segments)
  int flat;            // 1: the K index is the flat (tap, k) index and w2 is addressed by it (s2n, s2k); needs
                       //    source pixel = base + offset[tap]: forward with any stride, backward with stride 1
};

struct KronArgs {
  const void* x;     // [M, Gin * K]
  void* y;           // [M, Gout * N]
  const float* w1;   // element (po, ui) at po * s1o + ui * s1i
  const float* w2;   // element (n, k)  at n * s2n + k * s2k
  const void* w2p;   // optional (kron3 plain rows, PL instantiations): pre-packed hi / lo planes of this role (kron_conv.h)
  float* dw1;        // optional, same addressing as w1 (accumulated atomically)
  float* dw1_ws;     // optional (kron3 only): per-workgroup dw1 partials [grid.y * grid.x][G * G] instead of atomics
  const void* xref;  // optional [M, Gout * N] (needed iff dw1 != nullptr)
  const void* base;  // optional [M, Gout * N] (kron3 plain-row forward only): y = base + alpha * (...), rounded once
  long M;
  int Gin, K, Gout, N;
  long s1o, s1i, s2n, s2k;
  float alpha;
  int out_f32;       // write y as fp32 rows (LYC_F32_ROWS)
  KronGather gat;    // kron3 only
};

template <typename T>
struct KronCfg {
  static constexpr int BK = (sizeof(T) == 2) ? 32 : 16;
  static constexpr int RT = 128;  // stage-1 rows per workgroup (= TM * Gin)
  static constexpr int W1_LDS_MAX = 1024;
};

template <typename T, int TQ>
__global__ __launch_bounds__(NTHREADS) void kron_kernel(KronArgs a) {
  using C = KronCfg<T>;
  constexpr int BK = C::BK, RT = C::RT;
  constexpr int LD = TileLD<T, BK>::value;
  constexpr int NI = TQ / 16;
  constexpr int LDH = TQ + 4;
  constexpr int STAGE_BYTES = (RT * LD + 2 * TQ * LD) * (int)sizeof(T);
  constexpr int EPI_BYTES = RT * LDH * 4;
  constexpr int UNION_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[UNION_BYTES + C::W1_LDS_MAX * 4 + 16 * 16 * 4 * NWAVES];
  T* As = reinterpret_cast<T*>(smem);
  T* Bh = As + RT * LD;
  T* Bl = Bh + TQ * LD;
  float* Hs = reinterpret_cast<float*>(smem);
  float* w1s = reinterpret_cast<float*>(smem + UNION_BYTES);
  float* red = w1s + C::W1_LDS_MAX;

  const T* x = static_cast<const T*>(a.x);
  T* y = static_cast<T*>(a.y);
  const int tid = threadIdx.x, wave = tid >> 6;
  const int Gin = a.Gin, Gout = a.Gout, K = a.K, N = a.N;
  const int TM = RT / Gin;  // host guarantees 1 <= Gin <= RT
  const long m0 = (long)blockIdx.x * TM;
  const long n0 = (long)blockIdx.y * TQ;
  const long row0 = m0 * Gin;
  long rows_end = row0 + (long)TM * Gin;
  if (rows_end > a.M * Gin) rows_end = a.M * Gin;
  const bool x_vec = vec_aligned<T>(x, K);
  const bool w1_in_lds = Gin * Gout <= C::W1_LDS_MAX;

  if (w1_in_lds)
    for (int e = tid; e < Gin * Gout; e += NTHREADS) {
      const int po = e / Gin, ui = e % Gin;
      w1s[e] = a.w1[po * a.s1o + ui * a.s1i];
    }

  // ---- stage 1: S1 tile [RT x TQ] over K ----
  f32x4 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  for (long k0 = 0; k0 < K; k0 += BK) {
    stage_rows<T, RT, BK>(As, x, K, row0, rows_end, k0, K, x_vec);
    stage_factor<T, TQ, BK>(Bh, Bl, a.w2, a.s2n, a.s2k, n0, N, k0, K, 1.0f);
    __syncthreads();
    mma_tile<T, BK, 2, NI, true>(acc, As, wave * 32, Bh, Bl, 0);
    __syncthreads();
  }
  acc_to_lds<2, NI>(Hs, LDH, acc, wave * 32, 0, 1.0f);
  __syncthreads();

  // ---- stage 2: y[m, po, n] = alpha * sum_ui w1[po, ui] * S1[(m, ui), n] ----
  {
    constexpr int NV = TT<T>::VEC;
    constexpr int NG = TQ / NV;
    constexpr int NWORK = NTHREADS / NG;
    const int ng = tid % NG, wk = tid / NG;
    const long ldy = (long)Gout * N;
    const bool y_vec = vec_aligned<T>(y, N);
    const long gn = n0 + ng * NV;
    if (gn < N) {
      for (int pair = wk; pair < TM * Gout; pair += NWORK) {
        const int m = pair / Gout, po = pair % Gout;
        const long gm = m0 + m;
        if (gm >= a.M) break;
        float o[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) o[e] = 0.f;
        const float* hrow = Hs + (m * Gin) * LDH + ng * NV;
        for (int ui = 0; ui < Gin; ++ui) {
          const float w = w1_in_lds ? w1s[po * Gin + ui] : a.w1[po * a.s1o + ui * a.s1i];
#pragma unroll
          for (int e4 = 0; e4 < NV / 4; ++e4) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(hrow + ui * LDH + 4 * e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * e4 + e] = fmaf(w, h[e], o[4 * e4 + e]);
          }
        }
        if (a.out_f32) {
          float* dstf = static_cast<float*>(a.y) + gm * ldy + (long)po * N + gn;
          for (int e = 0; e < NV && gn + e < N; ++e) dstf[e] = a.alpha * o[e];
          continue;
        }
        T ov[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) ov[e] = TT<T>::from_f(a.alpha * o[e]);
        T* dst = y + gm * ldy + (long)po * N + gn;
        if (y_vec && gn + NV <= N) {
          *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<u32x4*>(ov);
        } else {
          for (int e = 0; e < NV && gn + e < N; ++e) dst[e] = ov[e];
        }
      }
    }
  }

  // ---- optional: dW1[po, ui] += alpha * sum_{m, n} S1[(m, ui), n] * xref[(m, po), n] ----
  if (a.dw1 != nullptr) {
    const T* xr = static_cast<const T*>(a.xref);
    const long ldr = (long)Gout * N;
    const bool r_vec = vec_aligned<T>(xr, N);
    const int lane = tid & 63, li = lane & 15, g = lane >> 4;
    constexpr int KV = TT<T>::KV, KSTEP = TT<T>::KSTEP;
    for (int ui0 = 0; ui0 < Gin; ui0 += 16) {
      for (int po0 = 0; po0 < Gout; po0 += 16) {
        f32x4 c = zero4();
        for (int m = wave; m < TM; m += NWAVES) {
          const long gm = m0 + m;
          if (gm >= a.M) break;
          for (int ks = 0; ks < TQ / KSTEP; ++ks) {
            const int kofs = ks * KSTEP + g * KV;
            // A fragment: S1 rows (m, ui0 + li) from the fp32 LDS tile, split on the fly
            float av[KV];
            const bool a_ok = (ui0 + li) < Gin;
#pragma unroll
            for (int e = 0; e < KV; ++e) av[e] = a_ok ? Hs[(m * Gin + ui0 + li) * LDH + kofs + e] : 0.f;
            // B fragment: xref rows (m, po0 + li), columns n0 + kofs .. (exact T values)
            typename TT<T>::frag bf;
            {
              T bv[KV];
              const bool b_ok = (po0 + li) < Gout;
              const long gc = n0 + kofs;
              const T* src = xr + gm * ldr + (long)(po0 + li) * N + gc;
              bool done = false;
              if constexpr (KV > 1) {
                if (b_ok && r_vec && gc + KV <= N) {
                  *reinterpret_cast<u32x4*>(bv) = *reinterpret_cast<const u32x4*>(src);
                  done = true;
                }
              }
              if (!done) {
#pragma unroll
                for (int e = 0; e < KV; ++e) bv[e] = (b_ok && gc + e < N) ? src[e] : TT<T>::from_f(0.f);
              }
              bf = *reinterpret_cast<typename TT<T>::frag*>(bv);
            }
            T ah[KV], al[KV];
#pragma unroll
            for (int e = 0; e < KV; ++e) {
              if constexpr (TT<T>::SPLIT) {
                split_f<T>(av[e], ah[e], al[e]);
              } else {
                ah[e] = av[e];
              }
            }
            c = TT<T>::mma(*reinterpret_cast<typename TT<T>::frag*>(ah), bf, c);
            if constexpr (TT<T>::SPLIT) c = TT<T>::mma(*reinterpret_cast<typename TT<T>::frag*>(al), bf, c);
          }
        }
        // cross-wave reduction of the 16x16 tile, then one atomic per element
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * g + r) * 16 + li] = c[r];
        __syncthreads();
        {
          const int ui = tid >> 4, po = tid & 15;  // tile row = ui, column = po
          const float s = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
          if (ui0 + ui < Gin && po0 + po < Gout)
            __hip_atomic_fetch_add(a.dw1 + (long)(po0 + po) * a.s1o + (long)(ui0 + ui) * a.s1i, a.alpha * s,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dW2 kernel
// ---------------------------------------------------------------------------------------------
struct KronDw2Args {
  const void* Q;    // [M, Gs * I]   exact operand, contributes the output rows i
  const void* P;    // [M, Gt * J]   operand that is mixed with W, contributes the output columns j
  const float* W;   // element (s, t) at s * ws + t * wt
  float* out;       // element (i, j) at i * os + j * oj   (accumulated atomically)
  long M;
  int Gs, I, Gt, J;
  long ws, wt, os, oj;
  long rows_per_block;  // flat (m, s) rows handled by one workgroup (multiple of BK)
  float alpha;
};

template <typename T, int MI>
__global__ __launch_bounds__(NTHREADS) void kron_dw2_kernel(KronDw2Args a) {
  constexpr int BK = 32;            // flat rows per step
  constexpr int TI = 16 * MI;       // output rows per workgroup
  constexpr int TJ = 64;            // output columns per workgroup (16 per wave)
  constexpr int LD = TileLD<T, BK>::value;
  __shared__ __attribute__((aligned(16))) char smem[(TI * LD + 2 * TJ * LD) * sizeof(T)];
  T* Qs = reinterpret_cast<T*>(smem);
  T* Zh = Qs + TI * LD;
  T* Zl = Zh + TJ * LD;

  const T* Q = static_cast<const T*>(a.Q);
  const T* P = static_cast<const T*>(a.P);
  const int tid = threadIdx.x, wave = tid >> 6;
  const long i0 = (long)blockIdx.x * TI;
  const long j0 = (long)blockIdx.y * TJ;
  const long rows_total = a.M * a.Gs;
  const long rbeg = (long)blockIdx.z * a.rows_per_block;
  long rend = rbeg + a.rows_per_block;
  if (rend > rows_total) rend = rows_total;
  const bool q_vec = vec_aligned<T>(Q, a.I);
  const bool p_vec = vec_aligned<T>(P, a.J);
  constexpr int NV = TT<T>::VEC;
  constexpr int JG = TJ / NV;             // column groups per row
  constexpr int RPP = NTHREADS / JG;      // rows per pass of the mix

  f32x4 acc[MI][1];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) acc[mi][0] = zero4();

  for (long r0 = rbeg; r0 < rend; r0 += BK) {
    // Q^T tile: Qs[i][r] = Q3[(r0 + r), i0 + i]   (Q3 = Q viewed as [M * Gs, I])
    stage_cols<T, TI, BK>(Qs, Q, a.I, r0, rend, i0, a.I, q_vec);
    // mixed P tile: Z[r][j] = sum_t W[s(r), t] * P[(m(r), t), j0 + j]  ->  Zh/Zl[j][r]
    for (int rr = tid / JG; rr < BK; rr += RPP) {
      const int jg = tid % JG;
      const long r = r0 + rr;
      const long gj = j0 + jg * NV;
      float z[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) z[e] = 0.f;
      if (r < rend && gj < a.J) {
        const long m = r / a.Gs;
        const int s = (int)(r % a.Gs);
        const T* prow = P + m * ((long)a.Gt * a.J) + gj;
        for (int t = 0; t < a.Gt; ++t) {
          const float w = a.W[s * a.ws + t * a.wt];
          T pv[NV];
          if (p_vec && gj + NV <= a.J) {
            *reinterpret_cast<u32x4*>(pv) = *reinterpret_cast<const u32x4*>(prow + (long)t * a.J);
          } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) pv[e] = (gj + e < a.J) ? prow[(long)t * a.J + e] : TT<T>::from_f(0.f);
          }
#pragma unroll
          for (int e = 0; e < NV; ++e) z[e] = fmaf(w, TT<T>::to_f(pv[e]), z[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        if constexpr (TT<T>::SPLIT) {
          T hi, lo;
          split_f<T>(z[e], hi, lo);
          Zh[(jg * NV + e) * LD + rr] = hi;
          Zl[(jg * NV + e) * LD + rr] = lo;
        } else {
          Zh[(jg * NV + e) * LD + rr] = z[e];
        }
      }
    }
    __syncthreads();
    mma_tile<T, BK, MI, 1, true>(acc, Qs, 0, Zh, Zl, wave * 16);
    __syncthreads();
  }
  acc_atomic_add<MI, 1>(a.out, a.os, a.oj, a.I, a.J, acc, i0, j0 + wave * 16, a.alpha);
}

}  // namespace lyc
