// dense_kernels.h -- dense MFMA GEMMs used by LoHa (whose Hadamard product of two rank-r matrices is full rank,
// so the adapter is a dense contraction; reference lycoris/functional/loha.py:10-30) and its helpers.
//
//   gemm_nt_kernel : out[M, N]   = alpha * A[M, K] * (Bh + Bl)[N, K]^T      A exact T, B as hi/lo T planes
//   gemm_tn_kernel : out32[I, J] (+)= alpha * sum_k A[k, I] * B[k, J]       both exact T (weight-gradient form)
//   loha_rebuild_kernel     : dW = ((w1a w1b) * (w2a w2b)) * s  -> hi/lo planes in both orientations
//   loha_factor_grad_kernel : HadaWeight.backward on G = g^T x (fp32): d_w1a, d_w1b, d_w2a, d_w2b
#pragma once
#include "tile.h"

namespace lyc {

struct GemmArgs {
  const void* A;
  const void* Bh;
  const void* Bl;   // may alias Bh's type; ignored for T = float
  void* out;
  long M, N, K;
  long lda, ldb, ldo;
  float alpha;
  int atomic;       // gemm_tn: 1 = atomicAdd into out (split over gridDim.z), 0 = plain store
  long chunk;       // gemm_tn: contraction rows per z-slice
  int out_f32;      // gemm_nt: write fp32 rows instead of T (LYC_F32_ROWS)
};

template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(GemmArgs a) {
  constexpr int BK = (sizeof(T) == 2) ? 32 : 16;
  constexpr int TM = 128, TN = 128;
  constexpr int LD = TileLD<T, BK>::value;
  constexpr int LDO = TN + 16 / (int)sizeof(T);
  constexpr int STAGE = (TM + 2 * TN) * LD * (int)sizeof(T);
  constexpr int EPI = TM * LDO * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char smem[STAGE > EPI ? STAGE : EPI];
  T* As = reinterpret_cast<T*>(smem);
  T* Bh = As + TM * LD;
  T* Bl = Bh + TN * LD;
  T* Os = reinterpret_cast<T*>(smem);
  const T* A = static_cast<const T*>(a.A);
  const T* gBh = static_cast<const T*>(a.Bh);
  const T* gBl = static_cast<const T*>(a.Bl);
  T* out = static_cast<T*>(a.out);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const long m0 = (long)blockIdx.x * TM, n0 = (long)blockIdx.y * TN;
  const bool a_vec = vec_aligned<T>(A, a.lda);
  const bool b_vec = vec_aligned<T>(gBh, a.ldb) && (!TT<T>::SPLIT || vec_aligned<T>(gBl, a.ldb));
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
  for (long k0 = 0; k0 < a.K; k0 += BK) {
    stage_rows<T, TM, BK>(As, A, a.lda, m0, a.M, k0, a.K, a_vec);
    stage_rows<T, TN, BK>(Bh, gBh, a.ldb, n0, a.N, k0, a.K, b_vec);
    if constexpr (TT<T>::SPLIT) stage_rows<T, TN, BK>(Bl, gBl, a.ldb, n0, a.N, k0, a.K, b_vec);
    __syncthreads();
    mma_tile<T, BK, 4, 4, true>(acc, As, wr * 64, Bh, Bl, wc * 64);
    __syncthreads();
  }
  if (a.out_f32) {  // un-rounded rows straight from the accumulators (conv backward path)
    float* outf = static_cast<float*>(a.out);
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long gr = m0 + wr * 64 + 16 * mi + 4 * g + r, gc = n0 + wc * 64 + 16 * ni + c;
          if (gr < a.M && gc < a.N) outf[gr * a.ldo + gc] = a.alpha * acc[mi][ni][r];
        }
    return;
  }
  {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Os[(wr * 64 + 16 * mi + 4 * g + r) * LDO + wc * 64 + 16 * ni + c] = TT<T>::from_f(a.alpha * acc[mi][ni][r]);
  }
  __syncthreads();
  constexpr int VEC = TT<T>::VEC, VPR = TN / VEC;
  const bool o_vec = vec_aligned<T>(out, a.ldo);
  for (int v = tid; v < TM * VPR; v += NTHREADS) {
    const int r = v / VPR, cc = (v % VPR) * VEC;
    const long gr = m0 + r, gc = n0 + cc;
    if (gr >= a.M || gc >= a.N) continue;
    if (o_vec && gc + VEC <= a.N) {
      *reinterpret_cast<u32x4*>(out + gr * a.ldo + gc) = *reinterpret_cast<const u32x4*>(Os + r * LDO + cc);
    } else {
      for (int e = 0; e < VEC && gc + e < a.N; ++e) out[gr * a.ldo + gc + e] = Os[r * LDO + cc + e];
    }
  }
}

// out32[i, j] (+)= alpha * sum_k A[k, i] * B[k, j]     A:[K, M(cols)] (lda), B:[K, N(cols)] (ldb)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(GemmArgs a) {
  constexpr int BK = 32;
  constexpr int TI = 128, TJ = 128;
  constexpr int LD = TileLD<T, BK>::value;
  __shared__ __attribute__((aligned(16))) char smem[(TI + TJ) * LD * sizeof(T)];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + TI * LD;
  const T* A = static_cast<const T*>(a.A);
  const T* B = static_cast<const T*>(a.Bh);
  float* out = static_cast<float*>(a.out);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const long i0 = (long)blockIdx.x * TI, j0 = (long)blockIdx.y * TJ;
  const long kbeg = (long)blockIdx.z * a.chunk;
  long kend = kbeg + a.chunk;
  if (kend > a.K) kend = a.K;
  const bool a_vec = vec_aligned<T>(A, a.lda), b_vec = vec_aligned<T>(B, a.ldb);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    stage_cols<T, TI, BK>(As, A, a.lda, k0, kend, i0, a.M, a_vec);
    stage_cols<T, TJ, BK>(Bs, B, a.ldb, k0, kend, j0, a.N, b_vec);
    __syncthreads();
    mma_tile<T, BK, 4, 4, false>(acc, As, wr * 64, Bs, Bs, wc * 64);
    __syncthreads();
  }
  if (a.atomic) {
    acc_atomic_add<4, 4>(out, a.ldo, 1, a.M, a.N, acc, i0 + wr * 64, j0 + wc * 64, a.alpha);
  } else {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long gi = i0 + wr * 64 + 16 * mi + 4 * g + r, gj = j0 + wc * 64 + 16 * ni + c;
          if (gi < a.M && gj < a.N) out[gi * a.ldo + gj] = a.alpha * acc[mi][ni][r];
        }
  }
}

// ---------------------------------------------------------------------------------------------
struct LohaArgs {
  const float *w1a, *w1b, *w2a, *w2b;  // w*a:[O, R]  w*b:[R, I]
  void *Wn_h, *Wn_l;                   // dW  planes [O, ldn]  (K-contiguous along I)  -- forward operand
  void *Wt_h, *Wt_l;                   // dW^T planes [I, ldt] (K-contiguous along O)  -- backward-dx operand
  const float* G;                      // [O, I] fp32 = g^T x   (factor-grad kernel)
  float *d_w1a, *d_w1b, *d_w2a, *d_w2b;
  long O, I;
  int R;
  long ldn, ldt;
  float scale;
};

constexpr int LOHA_T = 64;   // tile edge
constexpr int LOHA_RC = 32;  // rank chunk held in LDS

// Each thread owns a 4x4 micro-tile (rows 4*ty.., cols 4*tx..) of a 64x64 tile; factor slices sit in LDS.
__device__ __forceinline__ void loha_products(float (&p1)[4][4], float (&p2)[4][4], const LohaArgs& a, long o0,
                                              long i0, float* sA1, float* sA2, float* sB1, float* sB2) {
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) p1[r][c] = p2[r][c] = 0.f;
  for (int r0 = 0; r0 < a.R; r0 += LOHA_RC) {
    __syncthreads();
    for (int e = tid; e < LOHA_T * LOHA_RC; e += NTHREADS) {
      const int o = e / LOHA_RC, rr = e % LOHA_RC;  // a-factors: [o][rr], rr contiguous in memory
      const bool ok = (o0 + o < a.O) && (r0 + rr < a.R);
      sA1[o * (LOHA_RC + 1) + rr] = ok ? a.w1a[(o0 + o) * a.R + r0 + rr] : 0.f;
      sA2[o * (LOHA_RC + 1) + rr] = ok ? a.w2a[(o0 + o) * a.R + r0 + rr] : 0.f;
      const int rb = e / LOHA_T, i = e % LOHA_T;    // b-factors: [rr][i], i contiguous in memory
      const bool okb = (r0 + rb < a.R) && (i0 + i < a.I);
      sB1[rb * LOHA_T + i] = okb ? a.w1b[(long)(r0 + rb) * a.I + i0 + i] : 0.f;
      sB2[rb * LOHA_T + i] = okb ? a.w2b[(long)(r0 + rb) * a.I + i0 + i] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = 0; rr < LOHA_RC; ++rr) {
      float a1[4], a2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a1[r] = sA1[(4 * ty + r) * (LOHA_RC + 1) + rr];
        a2[r] = sA2[(4 * ty + r) * (LOHA_RC + 1) + rr];
      }
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB1 + rr * LOHA_T + 4 * tx);
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(sB2 + rr * LOHA_T + 4 * tx);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          p1[r][c] = fmaf(a1[r], b1[c], p1[r][c]);
          p2[r][c] = fmaf(a2[r], b2[c], p2[r][c]);
        }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void loha_rebuild_kernel(LohaArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * LOHA_T * (LOHA_RC + 1) + 2 * LOHA_RC * LOHA_T];
  float* sA1 = sm;
  float* sA2 = sA1 + LOHA_T * (LOHA_RC + 1);
  float* sB1 = sA2 + LOHA_T * (LOHA_RC + 1);
  float* sB2 = sB1 + LOHA_RC * LOHA_T;
  const long o0 = (long)blockIdx.x * LOHA_T, i0 = (long)blockIdx.y * LOHA_T;
  float p1[4][4], p2[4][4];
  loha_products(p1, p2, a, o0, i0, sA1, sA2, sB1, sB2);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  T* nh = static_cast<T*>(a.Wn_h);
  T* nl = static_cast<T*>(a.Wn_l);
  T* th = static_cast<T*>(a.Wt_h);
  T* tl = static_cast<T*>(a.Wt_l);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long o = o0 + 4 * ty + r, i = i0 + 4 * tx + c;
      if (o >= a.O || i >= a.I) continue;
      const float v = p1[r][c] * p2[r][c] * a.scale;
      T hi, lo;
      split_f<T>(v, hi, lo);
      nh[o * a.ldn + i] = hi;
      th[i * a.ldt + o] = hi;
      if constexpr (TT<T>::SPLIT) {
        nl[o * a.ldn + i] = lo;
        tl[i * a.ldt + o] = lo;
      }
    }
}

// G:[O, I] fp32 (= g^T x).  With W1 = w1a w1b, W2 = w2a w2b, s = scale:
//   T1 = s * G * W2, T2 = s * G * W1;  d_w1a += T1 w1b^T, d_w1b += w1a^T T1, d_w2a += T2 w2b^T, d_w2b += w2a^T T2
__global__ __launch_bounds__(NTHREADS) void loha_factor_grad_kernel(LohaArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * LOHA_T * (LOHA_RC + 1) + 2 * LOHA_RC * LOHA_T +
                                                   2 * LOHA_T * (LOHA_T + 1)];
  float* sA1 = sm;
  float* sA2 = sA1 + LOHA_T * (LOHA_RC + 1);
  float* sB1 = sA2 + LOHA_T * (LOHA_RC + 1);
  float* sB2 = sB1 + LOHA_RC * LOHA_T;
  float* sT1 = sB2 + LOHA_RC * LOHA_T;
  float* sT2 = sT1 + LOHA_T * (LOHA_T + 1);
  constexpr int LT = LOHA_T + 1;
  const long o0 = (long)blockIdx.x * LOHA_T, i0 = (long)blockIdx.y * LOHA_T;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  float p1[4][4], p2[4][4];
  loha_products(p1, p2, a, o0, i0, sA1, sA2, sB1, sB2);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long o = o0 + 4 * ty + r, i = i0 + 4 * tx + c;
      const float gv = (o < a.O && i < a.I) ? a.G[o * a.I + i] * a.scale : 0.f;
      sT1[(4 * ty + r) * LT + 4 * tx + c] = gv * p2[r][c];
      sT2[(4 * ty + r) * LT + 4 * tx + c] = gv * p1[r][c];
    }
  // contractions against the factor slices, one rank chunk at a time
  for (int r0 = 0; r0 < a.R; r0 += LOHA_RC) {
    __syncthreads();
    for (int e = tid; e < LOHA_T * LOHA_RC; e += NTHREADS) {
      const int o = e / LOHA_RC, rr = e % LOHA_RC;
      const bool ok = (o0 + o < a.O) && (r0 + rr < a.R);
      sA1[o * (LOHA_RC + 1) + rr] = ok ? a.w1a[(o0 + o) * a.R + r0 + rr] : 0.f;
      sA2[o * (LOHA_RC + 1) + rr] = ok ? a.w2a[(o0 + o) * a.R + r0 + rr] : 0.f;
      const int rb = e / LOHA_T, i = e % LOHA_T;
      const bool okb = (r0 + rb < a.R) && (i0 + i < a.I);
      sB1[rb * LOHA_T + i] = okb ? a.w1b[(long)(r0 + rb) * a.I + i0 + i] : 0.f;
      sB2[rb * LOHA_T + i] = okb ? a.w2b[(long)(r0 + rb) * a.I + i0 + i] : 0.f;
    }
    __syncthreads();
    const int q = tid & 63;  // row (o) or column (i) index inside the tile
    constexpr int NRR = LOHA_RC / NWAVES;
    float da1[NRR], da2[NRR];
#pragma unroll
    for (int k = 0; k < NRR; ++k) {
      const int rr = (tid >> 6) + NWAVES * k;
      float a1 = 0.f, a2 = 0.f, db1 = 0.f, db2 = 0.f;
      for (int j = 0; j < LOHA_T; ++j) {
        // d_wXa[o=q, rr] : sum over i=j of T[q][j] * b[rr][j];   d_wXb[rr, i=q] : sum over o=j of a[j][rr] * T[j][q]
        a1 = fmaf(sT1[q * LT + j], sB1[rr * LOHA_T + j], a1);
        a2 = fmaf(sT2[q * LT + j], sB2[rr * LOHA_T + j], a2);
        db1 = fmaf(sA1[j * (LOHA_RC + 1) + rr], sT1[j * LT + q], db1);
        db2 = fmaf(sA2[j * (LOHA_RC + 1) + rr], sT2[j * LT + q], db2);
      }
      da1[k] = a1;
      da2[k] = a2;
      if (r0 + rr < a.R && i0 + q < a.I) {  // lane = q: 64 consecutive floats per atomic instruction
        __hip_atomic_fetch_add(a.d_w1b + (long)(r0 + rr) * a.I + i0 + q, db1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.d_w2b + (long)(r0 + rr) * a.I + i0 + q, db2, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // The [O, R] gradients are contiguous along rr: with lane = o one atomic instruction touched 64 different cache
    // lines, and the same-line serialisation of fp32 atomics made this kernel 30 % of the LoHa step.  Transpose the
    // partial sums through the (now dead) a-factor tiles and issue the atomics with lane = rr: two lines per instruction.
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NRR; ++k) {
      const int rr = (tid >> 6) + NWAVES * k;
      sA1[q * (LOHA_RC + 1) + rr] = da1[k];
      sA2[q * (LOHA_RC + 1) + rr] = da2[k];
    }
    __syncthreads();
    {
      const int rl = tid & (LOHA_RC - 1), og = tid / LOHA_RC;
      if (r0 + rl < a.R)
        for (int o = og; o < LOHA_T; o += NTHREADS / LOHA_RC)
          if (o0 + o < a.O) {
            __hip_atomic_fetch_add(a.d_w1a + (o0 + o) * a.R + r0 + rl, sA1[o * (LOHA_RC + 1) + rl], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.d_w2a + (o0 + o) * a.R + r0 + rl, sA2[o * (LOHA_RC + 1) + rl], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          }
    }
  }
}

}  // namespace lyc
