// loha16.h -- LoHa rebuild and HadaWeight.backward on the 16-bit matrix cores, gfx950.
//
// Reference: HadaWeight.forward / backward (lycoris/functional/loha.py:10-30):
//   dW = (w1a w1b) * (w2a w2b) * s                                                      (rebuild)
//   T1 = s G * (w2a w2b), T2 = s G * (w1a w1b)        G = g^T x, fp32 [O, I]
//   d_w1a += T1 w1b^T, d_w1b += w1a^T T1, d_w2a += T2 w2b^T, d_w2b += w2a^T T2                  (factor gradients)
//
// loha_mfma.h does these rank-r x 64 x 64 products on the exact-fp32 matrix core (v_mfma_f32_16x16x4_f32: 1/16 of the
// 16-bit MFMA rate).  Measured (profiles/r01_v7_bench_loha_kernel_stats.csv): 22 % of the LoHa step in the factor-gradient
// kernel, 9 % in the rebuild.  Here every fp32 operand is a hi + lo pair of 16-bit values and a product is three
// v_mfma_f32_16x16x32 (hi*hi + lo*hi + hi*lo; lo*lo is below fp32 resolution): ~5x less matrix-pipe time at the same
// fp32-level accuracy (the products are exact, the accumulation is fp32).  Ranks <= 32 (one K step); larger ranks keep
// the fp32 kernels.
//
// Operand images in LDS are K-contiguous tiles (tile.h): the SAME factor is staged in both orientations where two
// products need different contraction dims (stage_factor takes arbitrary strides; the factors are L2-resident):
//   P   = a b           A = a  [o][k = r]          B = b  as [i][k = r]
//   d_a = T b^T         A = T  [o][k = i]          B = b  as [r][k = i]
//   d_b = a^T T         A = a  as [r][k = o]       B = T  as [i][k = o]   (T written to LDS a second time, transposed)
#pragma once
#include "dense_kernels.h"

namespace lyc {

constexpr int L16_T = 64;                    // tile edge
constexpr int L16_R = 32;                    // rank capacity (one MFMA K step)
constexpr int L16_LDR = L16_R + 8;           // pitch of [*][k = r] tiles (TileLD<T, 32>)
constexpr int L16_LDT = L16_T + 8;           // pitch of [*][k = 64] tiles (TileLD<T, 64>)

// dW tile -> ONE plane in the activation type (rounded once: the reference's diff_weight.to(base_weight.dtype))
template <typename T>
__global__ __launch_bounds__(NTHREADS) void loha_rebuild16_kernel(LohaArgs a) {
  static_assert(sizeof(T) == 2, "16-bit activations");
  constexpr int PL = L16_T * L16_LDR;  // elements per [64][k = r] plane
  __shared__ __attribute__((aligned(16))) T sm[8 * PL];
  T *a1h = sm, *a1l = sm + PL, *a2h = sm + 2 * PL, *a2l = sm + 3 * PL;
  T *b1h = sm + 4 * PL, *b1l = sm + 5 * PL, *b2h = sm + 6 * PL, *b2l = sm + 7 * PL;
  constexpr int LDO = L16_T + 4;
  float* Os = reinterpret_cast<float*>(sm);  // [64][LDO] fp32 (17 KiB of the 40 KiB), after the products
  const long o0 = (long)blockIdx.x * L16_T, i0 = (long)blockIdx.y * L16_T;
  const int wave = threadIdx.x >> 6;
  stage_factor<T, L16_T, L16_R>(a1h, a1l, a.w1a, a.R, 1, o0, a.O, 0, a.R, 1.0f);
  stage_factor<T, L16_T, L16_R>(a2h, a2l, a.w2a, a.R, 1, o0, a.O, 0, a.R, 1.0f);
  stage_factor<T, L16_T, L16_R>(b1h, b1l, a.w1b, 1, a.I, i0, a.I, 0, a.R, 1.0f);  // rows = i, k = r
  stage_factor<T, L16_T, L16_R>(b2h, b2l, a.w2b, 1, a.I, i0, a.I, 0, a.R, 1.0f);
  __syncthreads();
  f32x4 p1[1][4], p2[1][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) p1[0][c] = p2[0][c] = zero4();
  mma_tile_ss<T, L16_R, 1, 4>(p1, a1h, a1l, wave * 16, b1h, b1l, 0);
  mma_tile_ss<T, L16_R, 1, 4>(p2, a2h, a2l, wave * 16, b2h, b2l, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) p1[0][c][q] *= p2[0][c][q];
  __syncthreads();  // every wave is done reading the factor tiles
  acc_to_lds<1, 4>(Os, LDO, p1, wave * 16, 0, a.scale);
  __syncthreads();
  store_tile<T, L16_T, L16_T>(a.Wn_h, a.ldn, Os, LDO, o0, a.O, i0, a.I, false);
}

constexpr int loha16_grad_lds_bytes() {
  // a [o][r] x2 factors x hi/lo, b as [i][r] x2x2, b as [r][i] x2x2, a as [r][o] x2x2, T / T^T x2 (T1, T2) x hi/lo
  return (4 * L16_T * L16_LDR + 4 * L16_T * L16_LDR + 4 * L16_R * L16_LDT + 4 * L16_R * L16_LDT + 4 * L16_T * L16_LDT) * 2;
}

// One workgroup: row tiles ob*NO .. +NO-1, column tiles jb*nt .. +nt-1.  The w*a gradients of a row tile accumulate in
// registers over the nt column tiles, the w*b gradients of a column tile over the NO row tiles (fewer fp32 atomics).
template <typename T, int NO>
__global__ __launch_bounds__(NTHREADS) void loha_factor_grad16_kernel(LohaArgs a, LohaGradGeom gm) {
  static_assert(sizeof(T) == 2, "16-bit matrix cores");
  extern __shared__ __attribute__((aligned(16))) char l16_smem[];
  constexpr int PA = L16_T * L16_LDR, PB = L16_R * L16_LDT, PT = L16_T * L16_LDT;
  T* sA = reinterpret_cast<T*>(l16_smem);   // a1h a1l a2h a2l   [o][r]
  T* sBi = sA + 4 * PA;                      // b1h b1l b2h b2l   [i][r]
  T* sBr = sBi + 4 * PA;                     // b1h b1l b2h b2l   [r][i]
  T* sAr = sBr + 4 * PB;                     // a1h a1l a2h a2l   [r][o]
  T* sT = sAr + 4 * PB;                      // T1h T1l T2h T2l   [o][i], then [i][o]
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const long ob = (long)blockIdx.x * NO, jb = (long)blockIdx.y * gm.nt;
  const long tiles_j = (a.I + L16_T - 1) / L16_T;

  f32x4 da1[NO][1][2], da2[NO][1][2];  // d_w*a of row tile os: rows 16 wave + 4 g + q, columns r = 16 rt + li
#pragma unroll
  for (int os = 0; os < NO; ++os)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) da1[os][0][rt] = da2[os][0][rt] = zero4();

  for (long jt = jb; jt < jb + gm.nt && jt < tiles_j; ++jt) {
    const long i0 = jt * L16_T;
    f32x4 db1[2][1], db2[2][1];  // d_w*b of this column tile: rows r = 16 rt + 4 g + q, column i = 16 wave + li
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) db1[rt][0] = db2[rt][0] = zero4();
    __syncthreads();  // previous column tile's readers of sBi / sBr are done
    stage_factor<T, L16_T, L16_R>(sBi, sBi + PA, a.w1b, 1, a.I, i0, a.I, 0, a.R, 1.0f);
    stage_factor<T, L16_T, L16_R>(sBi + 2 * PA, sBi + 3 * PA, a.w2b, 1, a.I, i0, a.I, 0, a.R, 1.0f);
    stage_factor<T, L16_R, L16_T>(sBr, sBr + PB, a.w1b, a.I, 1, 0, a.R, i0, a.I, 1.0f);          // rows = r, k = i
    stage_factor<T, L16_R, L16_T>(sBr + 2 * PB, sBr + 3 * PB, a.w2b, a.I, 1, 0, a.R, i0, a.I, 1.0f);
#pragma unroll
    for (int os = 0; os < NO; ++os) {
      const long o0 = (ob + os) * L16_T;
      if (o0 >= a.O) break;
      __syncthreads();  // previous row tile's readers of sA / sAr / sT are done (and the b tiles above are written)
      stage_factor<T, L16_T, L16_R>(sA, sA + PA, a.w1a, a.R, 1, o0, a.O, 0, a.R, 1.0f);
      stage_factor<T, L16_T, L16_R>(sA + 2 * PA, sA + 3 * PA, a.w2a, a.R, 1, o0, a.O, 0, a.R, 1.0f);
      stage_factor<T, L16_R, L16_T>(sAr, sAr + PB, a.w1a, 1, a.R, 0, a.R, o0, a.O, 1.0f);       // rows = r, k = o
      stage_factor<T, L16_R, L16_T>(sAr + 2 * PB, sAr + 3 * PB, a.w2a, 1, a.R, 0, a.R, o0, a.O, 1.0f);
      __syncthreads();
      // ---- P1, P2: wave rows o = 16 wave + 4 g + q, columns i = 16 c + li
      f32x4 p1[1][4], p2[1][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) p1[0][c] = p2[0][c] = zero4();
      mma_tile_ss<T, L16_R, 1, 4>(p1, sA, sA + PA, wave * 16, sBi, sBi + PA, 0);
      mma_tile_ss<T, L16_R, 1, 4>(p2, sA + 2 * PA, sA + 3 * PA, wave * 16, sBi + 2 * PA, sBi + 3 * PA, 0);
      // ---- T1 = s G * P2, T2 = s G * P1 (fp32 in registers), written to LDS as hi/lo [o][i]
      float t1[4][4], t2[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ol = 16 * wave + 4 * g + q, il = 16 * c + li;
          const bool ok = (o0 + ol < a.O) && (i0 + il < a.I);
          const float gv = a.G[ok ? (o0 + ol) * a.I + i0 + il : 0];
          const float gs = ok ? gv * a.scale : 0.f;
          t1[c][q] = gs * p2[0][c][q];
          t2[c][q] = gs * p1[0][c][q];
          T h, l;
          split_f<T>(t1[c][q], h, l);
          sT[ol * L16_LDT + il] = h;
          sT[PT + ol * L16_LDT + il] = l;
          split_f<T>(t2[c][q], h, l);
          sT[2 * PT + ol * L16_LDT + il] = h;
          sT[3 * PT + ol * L16_LDT + il] = l;
        }
      __syncthreads();
      // d_w*a[o, r] += sum_i T[o, i] b[r, i]
      mma_tile_ss<T, L16_T, 1, 2>(da1[os], sT, sT + PT, wave * 16, sBr, sBr + PB, 0);
      mma_tile_ss<T, L16_T, 1, 2>(da2[os], sT + 2 * PT, sT + 3 * PT, wave * 16, sBr + 2 * PB, sBr + 3 * PB, 0);
      __syncthreads();  // everybody is done with the [o][i] image: overwrite it with [i][o]
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ol = 16 * wave + 4 * g + q, il = 16 * c + li;
          T h, l;
          split_f<T>(t1[c][q], h, l);
          sT[il * L16_LDT + ol] = h;
          sT[PT + il * L16_LDT + ol] = l;
          split_f<T>(t2[c][q], h, l);
          sT[2 * PT + il * L16_LDT + ol] = h;
          sT[3 * PT + il * L16_LDT + ol] = l;
        }
      __syncthreads();
      // d_w*b[r, i] += sum_o a[o, r] T[o, i]:  A = a as [r][k = o] (rows 0..31), B = T as [i][k = o] (rows 16 wave ..)
      mma_tile_ss<T, L16_T, 2, 1>(db1, sAr, sAr + PB, 0, sT, sT + PT, wave * 16);
      mma_tile_ss<T, L16_T, 2, 1>(db2, sAr + 2 * PB, sAr + 3 * PB, 0, sT + 2 * PT, sT + 3 * PT, wave * 16);
    }
    // d_w*b of this column tile, summed over the NO row tiles
    acc_atomic_add<2, 1>(a.d_w1b, a.I, 1, a.R, a.I, db1, 0, i0 + 16 * wave, 1.0f);
    acc_atomic_add<2, 1>(a.d_w2b, a.I, 1, a.R, a.I, db2, 0, i0 + 16 * wave, 1.0f);
  }
#pragma unroll
  for (int os = 0; os < NO; ++os) {
    const long o0 = (ob + os) * L16_T;
    if (o0 >= a.O) break;
    acc_atomic_add<1, 2>(a.d_w1a, a.R, 1, a.O, a.R, da1[os], o0 + 16 * wave, 0, 1.0f);
    acc_atomic_add<1, 2>(a.d_w2a, a.R, 1, a.O, a.R, da2[os], o0 + 16 * wave, 0, 1.0f);
  }
}

}  // namespace lyc
