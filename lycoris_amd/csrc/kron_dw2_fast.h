// kron_dw2_fast.h -- fast path of the LoKr w2-gradient kernel for 16-bit activations, gfx950.
//
//   dW2[i, j] += alpha * sum_{r = (m, s)} Q[r, i] * Z[r, j],     Z[(m, s), j] = sum_t W[s, t] * P[(m, t), j]
//
// Taken when Gs == Gt == G with 16 % G == 0 and both operands allow 16-byte loads.  Per step of 32 flat rows:
//   * Q and P tiles are staged TRANSPOSED (K-contiguous [col][row]) through registers: 4x8 blocks, v_perm_b32
//     transposes, 8-byte LDS writes; the loads of step k+1 are issued before the MFMAs of step k;
//   * the G x G mix runs on the matrix cores: Z block = (I (x) W)(hi + lo) * P block with v_mfma_f32_16x16x16
//     (P is exact T, W is split), result in the accumulator layout (lane = column j, 4 consecutive rows);
//   * that layout is fed straight into the main v_mfma_f32_16x16x32 as the B operand by permuting the K index
//     (element e < 4 of lane group g <-> row 4g+e of row-block 0, e >= 4 <-> row 4g+e-4 of row-block 1); the A operand
//     (Q^T tile) is read from LDS with the same permutation (two 8-byte reads).  Z never touches LDS.
//   * split over row slabs with fp32 atomics (the output tile is in [i][j] order, j contiguous in dW2 for the
//     orientation the launcher picks, so atomics are coalesced).
#pragma once
#include "kron_fast.h"

namespace lyc {

constexpr int DW_BK = 32;
constexpr int DW_LD = DW_BK + 8;
constexpr int DW_TI = 128;  // output rows per workgroup (2 waves x 64)

// transposed staging split in a load phase (global -> registers) and a store phase (registers -> LDS [col][row])
template <typename T, int COLS>
__device__ __forceinline__ void dw_cols_load(u32x4 (&raw)[4], int b, const T* __restrict__ src, long ld_src, long k0,
                                             long k_end, long n0, long n_total) {
  constexpr int NB = COLS / 8;
  const int nb = b % NB, kb = b / NB;
  const long gn = n0 + nb * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long gk = k0 + kb * 4 + j;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gk < k_end && gn < n_total) {
      if (gn + 8 <= n_total) {
        v = *reinterpret_cast<const u32x4*>(src + gk * ld_src + gn);
      } else {
        T tmp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[e] = (gn + e < n_total) ? src[gk * ld_src + gn + e] : TT<T>::from_f(0.f);
        v = *reinterpret_cast<u32x4*>(tmp);
      }
    }
    raw[j] = v;
  }
}

template <typename T, int COLS>
__device__ __forceinline__ void dw_cols_store(T* __restrict__ dst, const u32x4 (&r)[4], int b) {
  constexpr int NB = COLS / 8;
  const int nb = b % NB, kb = b / NB;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    u32x2 lo, hi;
    lo[0] = __builtin_amdgcn_perm(r[1][w], r[0][w], 0x05040100u);
    lo[1] = __builtin_amdgcn_perm(r[3][w], r[2][w], 0x05040100u);
    hi[0] = __builtin_amdgcn_perm(r[1][w], r[0][w], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(r[3][w], r[2][w], 0x07060302u);
    *reinterpret_cast<u32x2*>(dst + (nb * 8 + 2 * w) * DW_LD + kb * 4) = lo;
    *reinterpret_cast<u32x2*>(dst + (nb * 8 + 2 * w + 1) * DW_LD + kb * 4) = hi;
  }
}

template <typename T, int NJ>  // workgroup tile: 128 (i) x 32*NJ (j); waves 2 (i) x 2 (j), wave tile 64 x 16*NJ
__global__ __launch_bounds__(NTHREADS) void kron_dw2_fast_kernel(KronDw2Args a) {
  constexpr int MI = 4;
  constexpr int TJ = 32 * NJ;
  constexpr int QB = (DW_TI / 8) * (DW_BK / 4);  // 4x8 blocks in the Q tile
  constexpr int PB = (TJ / 8) * (DW_BK / 4);
  constexpr int NIT = (QB + PB + NTHREADS - 1) / NTHREADS;
  __shared__ __attribute__((aligned(16))) T smem[(DW_TI + TJ) * DW_LD];
  T* Qs = smem;
  T* Ps = smem + DW_TI * DW_LD;
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;

  const T* Q = static_cast<const T*>(a.Q);
  const T* P = static_cast<const T*>(a.P);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wi = wave >> 1, wj = wave & 1;
  const int G = a.Gs;
  const int lg = 31 - __builtin_clz((unsigned)G);
  const long i0 = (long)blockIdx.x * DW_TI;
  const long j0 = (long)blockIdx.y * TJ;
  const long rows_total = a.M << lg;
  const long rbeg = (long)blockIdx.z * a.rows_per_block;
  long rend = rbeg + a.rows_per_block;
  if (rend > rows_total) rend = rows_total;

  // mix operand (I (x) W) for one 16x16 block: lane (i = li, g) holds k = 4g .. 4g+3
  F4 a2h, a2l;
  {
    const int mi_ = li >> lg, s_ = li & (G - 1);
    T h[4], l[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kk = 4 * g + jj;
      const float v = ((kk >> lg) == mi_) ? a.W[s_ * a.ws + (kk & (G - 1)) * a.wt] : 0.f;
      split_f<T>(v, h[jj], l[jj]);
    }
    a2h = *reinterpret_cast<F4*>(h);
    a2l = *reinterpret_cast<F4*>(l);
  }

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = zero4();

  u32x4 raw[NIT][4];
  auto load_step = [&](long r0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int b = tid + NTHREADS * it;
      if (b < QB) {
        dw_cols_load<T, DW_TI>(raw[it], b, Q, a.I, r0, rend, i0, a.I);
      } else if (b < QB + PB) {
        dw_cols_load<T, TJ>(raw[it], b - QB, P, a.J, r0, rend, j0, a.J);
      }
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int b = tid + NTHREADS * it;
      if (b < QB) {
        dw_cols_store<T, DW_TI>(Qs, raw[it], b);
      } else if (b < QB + PB) {
        dw_cols_store<T, TJ>(Ps, raw[it], b - QB);
      }
    }
  };

  load_step(rbeg);
  for (long r0 = rbeg; r0 < rend; r0 += DW_BK) {
    if (r0 > rbeg) __syncthreads();
    store_step();
    __syncthreads();
    if (r0 + DW_BK < rend) load_step(r0 + DW_BK);

    // mix: Z blocks of this wave's NJ column blocks, both 16-row blocks, kept as hi/lo B fragments
    F8 zh[NJ], zl[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      T hh[8], ll[8];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const F4 pf = *reinterpret_cast<const F4*>(Ps + (wj * 16 * NJ + 16 * nj + li) * DW_LD + 16 * rb + 4 * g);
        f32x4 z = zero4();
        z = Mma16<T>::mma(a2h, pf, z);
        z = Mma16<T>::mma(a2l, pf, z);
#pragma unroll
        for (int e = 0; e < 4; ++e) split_f<T>(z[e], hh[4 * rb + e], ll[4 * rb + e]);
      }
      zh[nj] = *reinterpret_cast<F8*>(hh);
      zl[nj] = *reinterpret_cast<F8*>(ll);
    }
    // main: acc[i, j] += Q^T[i, r] * Z[r, j] with the same K permutation on the A side
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const T* qrow = Qs + (wi * 64 + 16 * mi + li) * DW_LD + 4 * g;
      u32x2 a0 = *reinterpret_cast<const u32x2*>(qrow);
      u32x2 a1 = *reinterpret_cast<const u32x2*>(qrow + 16);
      u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
      const F8 af = *reinterpret_cast<F8*>(&av);
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        acc[mi][nj] = TT<T>::mma(af, zh[nj], acc[mi][nj]);
        acc[mi][nj] = TT<T>::mma(af, zl[nj], acc[mi][nj]);
      }
    }
  }
  acc_atomic_add<MI, NJ>(a.out, a.os, a.oj, a.I, a.J, acc, i0 + wi * 64, j0 + wj * 16 * NJ, a.alpha);
}

}  // namespace lyc
