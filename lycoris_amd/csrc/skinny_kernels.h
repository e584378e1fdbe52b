// skinny_kernels.h -- rank-r building blocks (LoCon and the low-rank legs of the other adapters), gfx950.
//
// A rank-r adapter is "reduce to r channels, then expand" (reference: lycoris/functional/locon.py:64-85,
// lycoris/modules/locon.py:286-304): every product has one huge dimension (M rows of activations), one
// model dimension (I or O) and one tiny dimension (r).  Three kernels cover forward and backward:
//
//   skinny_nt_kernel : out32[M, Nn]  += alpha * A[M, K]   * B32[Nn, K]^T    A exact T, B fp32 factor (hi/lo),
//                                                                             split over K, fp32 atomics
//   expand_nt_kernel : out [M, Nn]    = alpha * A32[M, Kr] * B32[Nn, Kr]^T   both fp32 (hi/lo x hi/lo), T output
//                                                                             through LDS with 16-byte stores
//   skinny_tn_kernel : out32[I, Nn]  += alpha * A[Kd, I]^T * B32[Kd, Nn]     A exact T (transposed staging),
//                                                                             B fp32, split over Kd, atomics
//   forward  : t = skinny_nt(x, down) ; y = expand_nt(t, up)
//   backward : dt = skinny_nt(g, up^T) ; d_up = skinny_tn(g, t) ; dx = expand_nt(dt, down^T) ; d_down = skinny_tn(x, dt)^T
#pragma once
#include "tile.h"

namespace lyc {

struct SkinnyArgs {
  const void* A;     // activations (T) or fp32 matrix (expand_nt)
  const float* B;    // fp32 factor, element (n, k) at n * bn + k * bk
  void* out;
  long M;            // rows of A (NT kernels) / contraction length Kd (TN kernel)
  long K;            // NT: contraction length; TN: number of A columns (= output rows I)
  int Nn;            // size of the skinny dimension on the B side
  long lda;          // row pitch of A (elements)
  long bn, bk;
  long os, oj;       // output element (row, col) at row * os + col * oj
  long chunk;        // contraction elements per workgroup (multiple of the K tile)
  float alpha;
  int out_f32;       // expand_nt: write fp32 rows instead of T (LYC_F32_ROWS)
};

template <typename T>
struct SkinnyCfg {
  static constexpr int BK = (sizeof(T) == 2) ? 64 : 16;
};

// out32[m, n] += alpha * sum_k A[m, k] * B[n, k]
template <typename T, int NI>
__global__ __launch_bounds__(NTHREADS) void skinny_nt_kernel(SkinnyArgs a) {
  constexpr int BK = SkinnyCfg<T>::BK;
  constexpr int TM = 64, TN = 16 * NI;
  constexpr int LD = TileLD<T, BK>::value;
  __shared__ __attribute__((aligned(16))) char smem[(TM * LD + 2 * TN * LD) * sizeof(T)];
  T* As = reinterpret_cast<T*>(smem);
  T* Bh = As + TM * LD;
  T* Bl = Bh + TN * LD;
  const T* A = static_cast<const T*>(a.A);
  const int wave = threadIdx.x >> 6;
  const long m0 = (long)blockIdx.x * TM;
  const long n0 = (long)blockIdx.z * TN;
  const long kbeg = (long)blockIdx.y * a.chunk;
  long kend = kbeg + a.chunk;
  if (kend > a.K) kend = a.K;
  const bool a_vec = vec_aligned<T>(A, a.lda);
  f32x4 acc[1][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) acc[0][ni] = zero4();
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    stage_rows<T, TM, BK>(As, A, a.lda, m0, a.M, k0, kend, a_vec);
    stage_factor<T, TN, BK>(Bh, Bl, a.B, a.bn, a.bk, n0, a.Nn, k0, kend, 1.0f);
    __syncthreads();
    mma_tile<T, BK, 1, NI, true>(acc, As, wave * 16, Bh, Bl, 0);
    __syncthreads();
  }
  acc_atomic_add<1, NI>(static_cast<float*>(a.out), a.os, a.oj, a.M, a.Nn, acc, m0 + wave * 16, n0, a.alpha);
}

// out[m, n] = alpha * sum_k A32[m, k] * B32[n, k]      (k < K small)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void expand_nt_kernel(SkinnyArgs a) {
  constexpr int BK = (sizeof(T) == 2) ? 32 : 16;
  constexpr int TM = 64, TN = 128;   // wave w: rows 16w.., all 128 columns (NI = 8)
  constexpr int NI = TN / 16;
  constexpr int LD = TileLD<T, BK>::value;
  constexpr int LDO = TN + 4;
  constexpr int STAGE_BYTES = 2 * (TM + TN) * LD * (int)sizeof(T);
  constexpr int EPI_BYTES = TM * LDO * 4;
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
  T* Ah = reinterpret_cast<T*>(smem);
  T* Al = Ah + TM * LD;
  T* Bh = Al + TM * LD;
  T* Bl = Bh + TN * LD;
  float* Os = reinterpret_cast<float*>(smem);
  const float* A = static_cast<const float*>(a.A);
  const int wave = threadIdx.x >> 6;
  const long m0 = (long)blockIdx.x * TM;
  const long n0 = (long)blockIdx.y * TN;
  f32x4 acc[1][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) acc[0][ni] = zero4();
  for (long k0 = 0; k0 < a.K; k0 += BK) {
    stage_factor<T, TM, BK>(Ah, Al, A, a.lda, 1, m0, a.M, k0, a.K, 1.0f);
    stage_factor<T, TN, BK>(Bh, Bl, a.B, a.bn, a.bk, n0, a.Nn, k0, a.K, 1.0f);
    __syncthreads();
    mma_tile_ss<T, BK, 1, NI>(acc, Ah, Al, wave * 16, Bh, Bl, 0);
    __syncthreads();
  }
  acc_to_lds<1, NI>(Os, LDO, acc, wave * 16, 0, a.alpha);
  __syncthreads();
  store_tile<T, TM, TN>(a.out, a.os, Os, LDO, m0, a.M, n0, a.Nn, a.out_f32 != 0);
}

// out32[i, n] += alpha * sum_k A[k, i] * B[k, n]     (A row-major [Kd, I], contraction over its rows)
template <typename T, int NI>
__global__ __launch_bounds__(NTHREADS) void skinny_tn_kernel(SkinnyArgs a) {
  constexpr int BK = 32;
  constexpr int TI = 64, TN = 16 * NI;
  constexpr int LD = TileLD<T, BK>::value;
  __shared__ __attribute__((aligned(16))) char smem[(TI * LD + 2 * TN * LD) * sizeof(T)];
  T* As = reinterpret_cast<T*>(smem);
  T* Bh = As + TI * LD;
  T* Bl = Bh + TN * LD;
  const T* A = static_cast<const T*>(a.A);
  const int wave = threadIdx.x >> 6;
  const long i0 = (long)blockIdx.x * TI;
  const long n0 = (long)blockIdx.z * TN;
  const long kbeg = (long)blockIdx.y * a.chunk;
  long kend = kbeg + a.chunk;
  if (kend > a.M) kend = a.M;
  const bool a_vec = vec_aligned<T>(A, a.lda);
  f32x4 acc[1][NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) acc[0][ni] = zero4();
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    stage_cols<T, TI, BK>(As, A, a.lda, k0, kend, i0, a.K, a_vec);
    stage_factor<T, TN, BK>(Bh, Bl, a.B, a.bn, a.bk, n0, a.Nn, k0, kend, 1.0f);
    __syncthreads();
    mma_tile<T, BK, 1, NI, true>(acc, As, wave * 16, Bh, Bl, 0);
    __syncthreads();
  }
  acc_atomic_add<1, NI>(static_cast<float*>(a.out), a.os, a.oj, a.K, a.Nn, acc, i0 + wave * 16, n0, a.alpha);
}

}  // namespace lyc
