// dense_kernels.h -- dense MFMA GEMMs used by LoHa (whose Hadamard product of two rank-r matrices is full rank,
// so the adapter is a dense contraction; reference lycoris/functional/loha.py:10-30) and its helpers.
//
//   gemm_nt_kernel : out[M, N]   = alpha * A[M, K] * (Bh + Bl)[N, K]^T      A exact T, B as hi/lo T planes
//   gemm_tn_kernel : out32[I, J] (+)= alpha * sum_k A[k, I] * B[k, J]       both exact T (weight-gradient form)
//   (the LoHa-specific rebuild / factor-gradient kernels live in loha_mfma.h; LohaArgs is declared here)
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

// BSPLIT = false: B is ONE plane of T (LoHa's dW in the activation type): no lo tile, one MFMA per pair
template <typename T, bool BSPLIT = true>
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
  const bool b_vec = vec_aligned<T>(gBh, a.ldb) && (!(TT<T>::SPLIT && BSPLIT) || vec_aligned<T>(gBl, a.ldb));
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
  for (long k0 = 0; k0 < a.K; k0 += BK) {
    stage_rows<T, TM, BK>(As, A, a.lda, m0, a.M, k0, a.K, a_vec);
    stage_rows<T, TN, BK>(Bh, gBh, a.ldb, n0, a.N, k0, a.K, b_vec);
    if constexpr (TT<T>::SPLIT && BSPLIT) stage_rows<T, TN, BK>(Bl, gBl, a.ldb, n0, a.N, k0, a.K, b_vec);
    __syncthreads();
    mma_tile<T, BK, 4, 4, BSPLIT>(acc, As, wr * 64, Bh, Bl, wc * 64);
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

}  // namespace lyc
