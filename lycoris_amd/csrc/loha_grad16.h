// loha_grad16.h -- HadaWeight.backward on the 16-bit matrix cores (round 6).
//
// Reference: lycoris/functional/loha.py:18-30.  With G = g^T x (fp32, [O, I]) and P1 = w1a w1b, P2 = w2a w2b:
//     T1 = s G * P2,  T2 = s G * P1
//     d_w1a += T1 w1b^T   d_w1b += w1a^T T1   d_w2a += T2 w2b^T   d_w2b += w2a^T T2
// loha_mfma.h does all of it with v_mfma_f32_16x16x4_f32 (exact products, 1/16 of the 16-bit matrix rate, one ds_read_b32 per operand
// value): 192 MFMAs = 6.1 k matrix cycles per 64 x 64 tile and wave, LDS-issue bound on top -- 16.3 ms of the SDXL LoHa step
// (profiles/r06_c9_loha_kernel_stats.csv: loha_factor_grad_group_kernel<2, true>).  Here every fp32 operand is split x = hi + lo into two
// bf16 values (17 significant bits; tile.h split_f) and a product is three v_mfma_f32_16x16x32_bf16 (hi hi + lo hi + hi lo; the dropped
// lo lo term is 2^-18 relative): 72 MFMAs = 1.2 k matrix cycles per tile and wave, 2^-16 relative per product against the 1e-4 bound
// of the fp32 factor gradients.  bf16, not the activation type: fp16's lo part would underflow for |G| ~ 1e-4.
//
// One 64 x 64 tile of G, wave w, lane (li, g):
//   rebuild  P[i][o] = sum_r b[r][i] a[o][r]      A = b^T: ds_read_b64_tr_b16 out of the [r][i] image (paired column order, gemm16d.h),
//                                                 B = a  : ds_read_b128 out of the [o][r] image;  the lane ends up with P1, P2 of
//                                                 row o = 16 w + li, columns 32 q + 8 g .. + 7  (q = 0, 1) -- the layout G is loaded in
//                                                 (two 16-byte loads per q) and the layout of an MFMA A operand with K = i:
//   d_w*a    D[o][r] = sum_i T[o][i] b[r][i]      A = T straight from the registers, B = b: ds_read_b128 (K = i contiguous)
//   T images [o][i] (hi, lo; bf16) -> LDS, barrier
//   d_w*b    D[r][i] = sum_o a[o][r] T[o][i]      A = a^T, B = T: both ds_read_b64_tr_b16 (K = o strided); wave w owns columns 16 w ..
// A workgroup owns NO row tiles x nt column tiles (loha_mfma.h's plan): the a-side sums stay in registers over the nt column tiles, the
// b-side sums over the NO row tiles; both leave through an fp32 LDS image so that every atomic instruction of a wave covers 256
// CONSECUTIVE bytes (two cache lines; the direct form touched four 64-byte runs per instruction).
// Taken for R <= 32, R % 4 == 0, I % 8 == 0, 16-byte aligned factors and G (loha_grad16_ok); everything else stays on loha_mfma.h.
#pragma once
#include "loha_mfma.h"

namespace lyc {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int LG_AP = 80;                  // bytes per row of an a image [64 o][32 r] (64 + 16: conflict-free ds_read_b128 over 16 rows)
constexpr int LG_BP = 144;                 // bytes per row of a b image [32 r][64 i] (128 + 16)
constexpr int LG_TP = 144;                 // bytes per row of a T image [64 o][64 i]
constexpr int LG_A_BYTES = LOHA_T * LG_AP;   // 5120
constexpr int LG_B_BYTES = LOHA_RC * LG_BP;  // 4608
constexpr int LG_T_BYTES = LOHA_T * LG_TP;   // 9216
constexpr int LG_OFF_B = 4 * LG_A_BYTES, LG_OFF_T = LG_OFF_B + 4 * LG_B_BYTES;
constexpr int loha_grad16_lds_bytes() { return LG_OFF_T + 4 * LG_T_BYTES; }  // 75 776: two workgroups per CU
static_assert(2 * LOHA_T * LOHA_RC * 4 <= 4 * LG_T_BYTES && 2 * LOHA_RC * (LOHA_T + 4) * 4 <= 4 * LG_T_BYTES, "emit images fit the T region");

inline bool loha_grad16_ok(const float* w1a, const float* w1b, const float* w2a, const float* w2b, const float* G, long I, int r) {
  return r <= LOHA_RC && (r % 4) == 0 && (I % 8) == 0 &&
         (((reinterpret_cast<uintptr_t>(w1a) | reinterpret_cast<uintptr_t>(w2a) | reinterpret_cast<uintptr_t>(w1b) |
            reinterpret_cast<uintptr_t>(w2b) | reinterpret_cast<uintptr_t>(G)) & 15u) == 0);
}

typedef __attribute__((address_space(3))) bf16x4* lg_lds_bf4;
__device__ __forceinline__ bf16x8 lg_read_tr(const char* p0, const char* p1) {
  // k rows 8 g .. + 3 (p0) and + 4 .. + 7 (p1) of this lane's column
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lg_lds_bf4)(p0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lg_lds_bf4)(p1));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ f32x4 lg_mma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
}
__device__ __forceinline__ void lg_split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    __bf16 a, b;
    split_f<__bf16>(v[k], a, b);
    h[k] = a;
    l[k] = b;
  }
}
__device__ __forceinline__ void lg_put4(char* img_h, char* img_l, int off, const f32x4 v) {
  __bf16 h[4] __attribute__((aligned(8))), l[4] __attribute__((aligned(8)));
#pragma unroll
  for (int q = 0; q < 4; ++q) split_f<__bf16>(v[q], h[q], l[q]);
  *reinterpret_cast<u32x2*>(img_h + off) = *reinterpret_cast<const u32x2*>(h);
  *reinterpret_cast<u32x2*>(img_l + off) = *reinterpret_cast<const u32x2*>(l);
}

// `bx`, `by`: this workgroup's block of row tiles / column tiles; `sm`: loha_grad16_lds_bytes()
// ABL != 0: ablation builds of benchmarks/lg16bench.cpp (results are garbage): 1 = no atomics, 2 = no G loads, 4 = no d_w*b part,
// 8 = no rebuild, 16 = no factor staging after the first tile
template <int NO, int ABL = 0>
__device__ __forceinline__ void loha_factor_grad16_body(const LohaArgs& a, const int nt, char* sm, const int bx, const int by) {
  // images: A1h A1l A2h A2l | B1h B1l B2h B2l | T1h T1l T2h T2l
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4, wave = tid >> 6;
  const long ob = (long)bx * NO, jb = (long)by * nt;
  const long tiles_j = (a.I + LOHA_T - 1) / LOHA_T;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 da1[NO][2], da2[NO][2];  // d_w*a of row tile os: rows o = 16 wave + 4 g + q, column r = 16 rt + li
#pragma unroll
  for (int os = 0; os < NO; ++os)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) da1[os][rt] = da2[os][rt] = z4;

  // fragment addresses of this lane
  const char* a_kc = sm + (16 * wave + li) * LG_AP + g * 16;                                  // a as MFMA B operand (K = r): + p * LG_A_BYTES
  const char* b_tr = sm + LG_OFF_B + (8 * g + (li >> 2)) * LG_BP + (li & 3) * 16;             // b^T, paired columns: + p, + q * 64 + e * 8
  const char* b_kc = sm + LG_OFF_B + li * LG_BP + g * 16;                                     // b as B operand (K = i): + p, + rt * 16 rows, + q * 64
  const char* a_tr = sm + (8 * g + (li >> 2)) * LG_AP + (li & 3) * 8;                         // a^T (K = o): + p, + kk * 32 rows, + rt * 32 bytes
  const char* t_tr = sm + LG_OFF_T + (8 * g + (li >> 2)) * LG_TP + (16 * wave + 4 * (li & 3)) * 2;  // T (K = o): + p, + kk * 32 rows
  char* t_wr = sm + LG_OFF_T + (16 * wave + li) * LG_TP + g * 16;                             // + p, + q * 64

  for (long jt = jb; jt < jb + nt && jt < tiles_j; ++jt) {
    const long i0 = jt * LOHA_T;
    f32x4 db1[2] = {z4, z4}, db2[2] = {z4, z4};  // d_w*b of this column tile: rows r = 16 rt + 4 g + q, column i = 16 wave + li
#pragma unroll
    for (int os = 0; os < NO; ++os) {
      const long o0 = (ob + os) * LOHA_T;
      if (o0 >= a.O) break;
      // ---- G of this lane: row o, columns 32 q + 8 g .. + 7 (issued first: the longest latency of the tile)
      const long o = o0 + 16 * wave + li;
      f32x4 gv[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long i = i0 + 32 * q + 8 * g;
        const bool ok = o < a.O && i < a.I;
        const float* src = a.G + (ok ? o * a.I + i : 0);
        if constexpr (ABL & 2) {
          gv[q][0] = gv[q][1] = f32x4{1.f, 2.f, 3.f, 4.f};
        } else {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
          gv[q][0] = ok ? v0 : z4;
          gv[q][1] = ok ? v1 : z4;
        }
      }
      // ---- factors -> hi / lo images (the b side once per column tile)
      f32x4 va[2][2], vb[2][2];
      const bool stage = !(ABL & 16) || (jt == jb && os == 0);
      if (stage)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int e = tid + NTHREADS * it;
        const int oo = e >> 3, c4 = (e & 7) * 4;
        const bool ok = (o0 + oo < a.O) && (c4 < a.R);
        const long idx = ok ? (o0 + oo) * a.R + c4 : 0;
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.w1a + idx), x2 = *reinterpret_cast<const f32x4*>(a.w2a + idx);
        va[0][it] = ok ? x1 : z4;
        va[1][it] = ok ? x2 : z4;
        if (os == 0) {
          const int rb = e >> 4, i4 = (e & 15) * 4;
          const bool okb = (rb < a.R) && (i0 + i4 < a.I);
          const long idb = okb ? (long)rb * a.I + i0 + i4 : 0;
          const f32x4 y1 = *reinterpret_cast<const f32x4*>(a.w1b + idb), y2 = *reinterpret_cast<const f32x4*>(a.w2b + idb);
          vb[0][it] = okb ? y1 : z4;
          vb[1][it] = okb ? y2 : z4;
        }
      }
      __syncthreads();  // the previous tile's readers of the images are done
      if (stage)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int e = tid + NTHREADS * it;
        const int oo = e >> 3, c4 = (e & 7) * 4, rb = e >> 4, i4 = (e & 15) * 4;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          lg_put4(sm + (2 * f) * LG_A_BYTES, sm + (2 * f + 1) * LG_A_BYTES, oo * LG_AP + c4 * 2, va[f][it]);
          if (os == 0)
            lg_put4(sm + LG_OFF_B + (2 * f) * LG_B_BYTES, sm + LG_OFF_B + (2 * f + 1) * LG_B_BYTES, rb * LG_BP + i4 * 2, vb[f][it]);
        }
      }
      __syncthreads();
      // ---- rebuild: P1, P2 of (row o, columns 32 q + 8 g + 4 e + j)
      bf16x8 af[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) af[p] = *reinterpret_cast<const bf16x8*>(a_kc + p * LG_A_BYTES);
      bf16x8 t1h[2], t1l[2], t2h[2], t2l[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v1[8], v2[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          bf16x8 bf[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const char* s = b_tr + p * LG_B_BYTES + q * 64 + e * 8;
            bf[p] = lg_read_tr(s, s + 4 * LG_BP);
          }
          f32x4 p1 = f32x4{1.f, 1.f, 1.f, 1.f}, p2 = p1;
          if constexpr (!(ABL & 8)) {
            p1 = lg_mma3(bf[0], bf[1], af[0], af[1], z4);
            p2 = lg_mma3(bf[2], bf[3], af[2], af[3], z4);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gs = gv[q][e][j] * a.scale;
            v1[4 * e + j] = gs * p2[j];  // T1 = s G * P2
            v2[4 * e + j] = gs * p1[j];  // T2 = s G * P1
          }
        }
        lg_split8(v1, t1h[q], t1l[q]);
        lg_split8(v2, t2h[q], t2l[q]);
        *reinterpret_cast<bf16x8*>(t_wr + 0 * LG_T_BYTES + q * 64) = t1h[q];
        *reinterpret_cast<bf16x8*>(t_wr + 1 * LG_T_BYTES + q * 64) = t1l[q];
        *reinterpret_cast<bf16x8*>(t_wr + 2 * LG_T_BYTES + q * 64) = t2h[q];
        *reinterpret_cast<bf16x8*>(t_wr + 3 * LG_T_BYTES + q * 64) = t2l[q];
      }
      // ---- d_w*a[o, r] += sum_i T[o, i] b[r, i]
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const char* s = b_kc + rt * 16 * LG_BP + q * 64;
          const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(s), b1l = *reinterpret_cast<const bf16x8*>(s + LG_B_BYTES);
          const bf16x8 b2h = *reinterpret_cast<const bf16x8*>(s + 2 * LG_B_BYTES), b2l = *reinterpret_cast<const bf16x8*>(s + 3 * LG_B_BYTES);
          da1[os][rt] = lg_mma3(t1h[q], t1l[q], b1h, b1l, da1[os][rt]);
          da2[os][rt] = lg_mma3(t2h[q], t2l[q], b2h, b2l, da2[os][rt]);
        }
      __syncthreads();  // T images visible
      // ---- d_w*b[r, i] += sum_o a[o, r] T[o, i]
      if constexpr (!(ABL & 4))
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const char* ts = t_tr + kk * 32 * LG_TP;
        const bf16x8 x1h = lg_read_tr(ts, ts + 4 * LG_TP), x1l = lg_read_tr(ts + LG_T_BYTES, ts + LG_T_BYTES + 4 * LG_TP);
        const bf16x8 x2h = lg_read_tr(ts + 2 * LG_T_BYTES, ts + 2 * LG_T_BYTES + 4 * LG_TP),
                     x2l = lg_read_tr(ts + 3 * LG_T_BYTES, ts + 3 * LG_T_BYTES + 4 * LG_TP);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const char* as = a_tr + kk * 32 * LG_AP + rt * 32;
          const bf16x8 a1h = lg_read_tr(as, as + 4 * LG_AP), a1l = lg_read_tr(as + LG_A_BYTES, as + LG_A_BYTES + 4 * LG_AP);
          const bf16x8 a2h = lg_read_tr(as + 2 * LG_A_BYTES, as + 2 * LG_A_BYTES + 4 * LG_AP),
                       a2l = lg_read_tr(as + 3 * LG_A_BYTES, as + 3 * LG_A_BYTES + 4 * LG_AP);
          db1[rt] = lg_mma3(a1h, a1l, x1h, x1l, db1[rt]);
          db2[rt] = lg_mma3(a2h, a2l, x2h, x2l, db2[rt]);
        }
      }
    }
    // ---- d_w*b of this column tile (summed over the NO row tiles): fp32 image [2][32 r][64 i + 4] in the T region, then whole rows
    __syncthreads();
    {
      float* img = reinterpret_cast<float*>(sm + LG_OFF_T);
      constexpr int P = LOHA_T + 4;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          img[(16 * rt + 4 * g + q) * P + 16 * wave + li] = db1[rt][q];
          img[(LOHA_RC + 16 * rt + 4 * g + q) * P + 16 * wave + li] = db2[rt][q];
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2 * LOHA_RC * LOHA_T / NTHREADS; ++k) {
        const int e = tid + NTHREADS * k;
        const int f = e / (LOHA_RC * LOHA_T), r = (e / LOHA_T) % LOHA_RC, i = e % LOHA_T;
        if (r < a.R && i0 + i < a.I && (!(ABL & 1) || img[(f * LOHA_RC + r) * P + i] == 1234.5f))
          __hip_atomic_fetch_add((f ? a.d_w2b : a.d_w1b) + (long)r * a.I + i0 + i, img[(f * LOHA_RC + r) * P + i], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // ---- d_w*a of the NO row tiles (summed over the nt column tiles): fp32 image [2][64 o][32 r] in the T region
#pragma unroll
  for (int os = 0; os < NO; ++os) {
    const long o0 = (ob + os) * LOHA_T;
    if (o0 >= a.O) break;
    float* img = reinterpret_cast<float*>(sm + LG_OFF_T);
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        img[(16 * wave + 4 * g + q) * LOHA_RC + 16 * rt + li] = da1[os][rt][q];
        img[(LOHA_T + 16 * wave + 4 * g + q) * LOHA_RC + 16 * rt + li] = da2[os][rt][q];
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2 * LOHA_T * LOHA_RC / NTHREADS; ++k) {
      const int e = tid + NTHREADS * k;
      const int f = e / (LOHA_T * LOHA_RC), oo = (e / LOHA_RC) % LOHA_T, r = e % LOHA_RC;
      if (o0 + oo < a.O && r < a.R && (!(ABL & 1) || img[(f * LOHA_T + oo) * LOHA_RC + r] == 1234.5f))
        __hip_atomic_fetch_add((f ? a.d_w2a : a.d_w1a) + (o0 + oo) * a.R + r, img[(f * LOHA_T + oo) * LOHA_RC + r], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int NO, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void loha_factor_grad16_kernel(LohaArgs a, LohaGradGeom gm) {
  extern __shared__ __attribute__((aligned(16))) char lg_smem[];
  loha_factor_grad16_body<NO, ABL>(a, gm.nt, lg_smem, (int)blockIdx.x, (int)blockIdx.y);
}

template <int NO>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void loha_factor_grad16_group_kernel(LohaGradGroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char lg_smem[];
  const int b = (int)blockIdx.x;
  int q = 0;
  while (q + 1 < ga.n && b >= ga.wg_end[q]) ++q;
  const int bl = b - (q ? ga.wg_end[q - 1] : 0);
  const LohaGradItem& it = ga.p[q];
  LohaArgs a{};
  a.w1a = it.w1a; a.w1b = it.w1b; a.w2a = it.w2a; a.w2b = it.w2b; a.G = it.G;
  a.d_w1a = it.d_w1a; a.d_w1b = it.d_w1b; a.d_w2a = it.d_w2a; a.d_w2b = it.d_w2b;
  a.O = it.O; a.I = it.I; a.R = it.R; a.scale = it.scale;
  loha_factor_grad16_body<NO>(a, it.nt, lg_smem, bl % it.gx, bl / it.gx);
}

}  // namespace lyc
