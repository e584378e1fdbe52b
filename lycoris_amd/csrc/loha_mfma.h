// loha_mfma.h -- the LoHa-specific kernels on the fp32 matrix core, gfx950.
//
// Reference: HadaWeight.forward / backward (lycoris/functional/loha.py:10-30):
//   dW = (w1a w1b) * (w2a w2b) * s                                                      (rebuild)
//   T1 = s G * (w2a w2b), T2 = s G * (w1a w1b)        G = g^T x, fp32 [O, I]
//   d_w1a += T1 w1b^T, d_w1b += w1a^T T1, d_w2a += T2 w2b^T, d_w2b += w2a^T T2                  (factor gradients)
// All of it is rank-r x 64 x 64 products on fp32 data.  v_mfma_f32_16x16x4_f32 takes one fp32 value per lane for
// (row, k) / (k, column): exact products, no hi/lo split, operands read from row-major LDS tiles as they are.  The
// first version of these kernels did the products on the VALU out of LDS (LDS-issue bound, ~36 k cycles per 64x64 tile);
// here a tile costs 192 MFMAs per wave (~6 k cycles).  What is left is the fp32 atomics of the factor gradients (they
// are paid per touched cache line, see lowrank.h): a workgroup therefore owns NO x NT tiles -- the w*a gradients of its
// rows accumulate in registers over the NT column tiles, the w*b gradients of a column tile over the NO row tiles -- and
// every wave emits whole 64-byte runs.
#pragma once
#include "dense_kernels.h"
#include "gemm16d.h"

namespace lyc {

constexpr int LH_AP = LOHA_RC + 4;  // pitch of the a-factor tiles [64 o][32 r]  (multiple of 4: 16-byte LDS writes)
constexpr int LH_BP = LOHA_T + 4;   // pitch of the b-factor tiles [32 r][64 i]
constexpr int LH_TP = LOHA_T + 4;   // pitch of the T tiles [64 o][64 i]

__device__ __forceinline__ float lh_mma(float a, float b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  return 0.f;
}

// Factor staging.  Every load of a tile is ISSUED before the first LDS write: the first version looped "load one element,
// wait, write" -- 16 dependent L2 round trips per tile, which was most of the time of a tile (profiles/r01_v7: the rebuild
// of a 640 x 640 layer took 10 us).  Whole float4s when the rows allow it (R % 4 == 0 / I % 4 == 0, 16-byte aligned base).
__device__ __forceinline__ bool lh_vec_ok(const LohaArgs& a) {
  return (a.R % 4 == 0) && (a.I % 4 == 0) &&
         (((reinterpret_cast<uintptr_t>(a.w1a) | reinterpret_cast<uintptr_t>(a.w2a) | reinterpret_cast<uintptr_t>(a.w1b) |
            reinterpret_cast<uintptr_t>(a.w2b)) & 15u) == 0);
}

struct LhRaw {
  f32x4 a1[2], a2[2], b1[2], b2[2];
};

// issue: a-factor rows o0 .. o0+63 x rank chunk r0 .. r0+31 (512 float4, two per thread and factor), b-factor rank chunk x
// columns i0 .. i0+63 (512 float4)
__device__ __forceinline__ void lh_load_vec(const LohaArgs& a, long o0, long i0, int r0, LhRaw& w) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int o = e >> 3, c4 = (e & 7) * 4;
    const bool ok = (o0 + o < a.O) && (r0 + c4 < a.R);
    const long idx = ok ? (o0 + o) * a.R + r0 + c4 : 0;
    w.a1[it] = *reinterpret_cast<const f32x4*>(a.w1a + idx);
    w.a2[it] = *reinterpret_cast<const f32x4*>(a.w2a + idx);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int rb = e >> 4, c4 = (e & 15) * 4;
    const bool ok = (r0 + rb < a.R) && (i0 + c4 < a.I);
    const long idx = ok ? (long)(r0 + rb) * a.I + i0 + c4 : 0;
    w.b1[it] = *reinterpret_cast<const f32x4*>(a.w1b + idx);
    w.b2[it] = *reinterpret_cast<const f32x4*>(a.w2b + idx);
  }
}
__device__ __forceinline__ void lh_store_vec(const LohaArgs& a, long o0, long i0, int r0, const LhRaw& w, float* sA1,
                                             float* sA2, float* sB1, float* sB2) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int o = e >> 3, c4 = (e & 7) * 4;
    const bool ok = (o0 + o < a.O) && (r0 + c4 < a.R);
    *reinterpret_cast<f32x4*>(sA1 + o * LH_AP + c4) = ok ? w.a1[it] : z;
    *reinterpret_cast<f32x4*>(sA2 + o * LH_AP + c4) = ok ? w.a2[it] : z;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int rb = e >> 4, c4 = (e & 15) * 4;
    const bool ok = (r0 + rb < a.R) && (i0 + c4 < a.I);
    *reinterpret_cast<f32x4*>(sB1 + rb * LH_BP + c4) = ok ? w.b1[it] : z;
    *reinterpret_cast<f32x4*>(sB2 + rb * LH_BP + c4) = ok ? w.b2[it] : z;
  }
}

// element-wise form (odd ranks / widths / alignment); also with all loads of a factor pair in flight before the writes
__device__ __forceinline__ void lh_stage_scalar(const LohaArgs& a, long o0, long i0, int r0, float* sA1, float* sA2,
                                                float* sB1, float* sB2) {
  constexpr int N = LOHA_T * LOHA_RC / NTHREADS;  // 8 elements per thread and tile
  float v1[N], v2[N];
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int o = e / LOHA_RC, rr = e % LOHA_RC;
    const bool ok = (o0 + o < a.O) && (r0 + rr < a.R);
    const long idx = ok ? (o0 + o) * a.R + r0 + rr : 0;
    v1[it] = a.w1a[idx];
    v2[it] = a.w2a[idx];
  }
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int o = e / LOHA_RC, rr = e % LOHA_RC;
    const bool ok = (o0 + o < a.O) && (r0 + rr < a.R);
    sA1[o * LH_AP + rr] = ok ? v1[it] : 0.f;
    sA2[o * LH_AP + rr] = ok ? v2[it] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int rb = e / LOHA_T, i = e % LOHA_T;
    const bool ok = (r0 + rb < a.R) && (i0 + i < a.I);
    const long idx = ok ? (long)(r0 + rb) * a.I + i0 + i : 0;
    v1[it] = a.w1b[idx];
    v2[it] = a.w2b[idx];
  }
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int e = threadIdx.x + NTHREADS * it;
    const int rb = e / LOHA_T, i = e % LOHA_T;
    const bool ok = (r0 + rb < a.R) && (i0 + i < a.I);
    sB1[rb * LH_BP + i] = ok ? v1[it] : 0.f;
    sB2[rb * LH_BP + i] = ok ? v2[it] : 0.f;
  }
}

// both factor pairs of tile (o0, i0), rank chunk r0 -> LDS
__device__ __forceinline__ void lh_stage(const LohaArgs& a, bool vec, long o0, long i0, int r0, float* sA1, float* sA2,
                                         float* sB1, float* sB2) {
  if (vec) {
    LhRaw w;
    lh_load_vec(a, o0, i0, r0, w);
    lh_store_vec(a, o0, i0, r0, w, sA1, sA2, sB1, sB2);
  } else {
    lh_stage_scalar(a, o0, i0, r0, sA1, sA2, sB1, sB2);
  }
}

// dW operand plane(s).  Computed transposed (A = b-factor^T, B = a-factor^T) so that a lane ends up with 4 consecutive i
// of one row o: 8-byte stores of the K-contiguous plane.  16-bit activations: ONE plane, dW rounded once to the activation
// type -- exactly the reference's `diff_weight.to(base_weight.dtype)` (modules/loha.py:310); LO adds the residual plane
// (hi + lo = the fp32 value) for callers that want the un-rounded operand.  WT: also the transposed plane (fp32 path).
template <typename T, bool WT, bool LO = false>
__global__ __launch_bounds__(NTHREADS) void loha_rebuild_mfma_kernel(LohaArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * LOHA_T * LH_AP + 2 * LOHA_RC * LH_BP];
  float* sA1 = sm;
  float* sA2 = sA1 + LOHA_T * LH_AP;
  float* sB1 = sA2 + LOHA_T * LH_AP;
  float* sB2 = sB1 + LOHA_RC * LH_BP;
  const long o0 = (long)blockIdx.x * LOHA_T, i0 = (long)blockIdx.y * LOHA_T;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  f32x4 p1[4], p2[4];  // [ti]: rows i = 16 ti + 4 g + q, column o = 16 wave + li
#pragma unroll
  for (int t = 0; t < 4; ++t) p1[t] = p2[t] = zero4();
  const bool vec_in = lh_vec_ok(a);
  for (int r0 = 0; r0 < a.R; r0 += LOHA_RC) {
    if (r0) __syncthreads();
    lh_stage(a, vec_in, o0, i0, r0, sA1, sA2, sB1, sB2);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < LOHA_RC / 4; ++ks) {
      const float b1 = sA1[(16 * wave + li) * LH_AP + 4 * ks + g], b2 = sA2[(16 * wave + li) * LH_AP + 4 * ks + g];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        lh_mma(sB1[(4 * ks + g) * LH_BP + 16 * t + li], b1, p1[t]);
        lh_mma(sB2[(4 * ks + g) * LH_BP + 16 * t + li], b2, p2[t]);
      }
    }
  }
  T* nh = static_cast<T*>(a.Wn_h);
  T* nl = static_cast<T*>(a.Wn_l);
  const long o = o0 + 16 * wave + li;
  const bool vec = (a.ldn % 4) == 0 && (reinterpret_cast<uintptr_t>(nh) & 7u) == 0 &&
                   (!(TT<T>::SPLIT && LO) || (reinterpret_cast<uintptr_t>(nl) & 7u) == 0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long i = i0 + 16 * t + 4 * g;
    T hi[4] __attribute__((aligned(16))), lo[4] __attribute__((aligned(16)));
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = p1[t][q] * p2[t][q] * a.scale;
      split_f<T>(v[q], hi[q], lo[q]);
    }
    if (o < a.O && i < a.I) {
      bool done = false;
      if constexpr (sizeof(T) == 2) {
        if (vec && i + 4 <= a.I) {
          *reinterpret_cast<u32x2*>(nh + o * a.ldn + i) = *reinterpret_cast<const u32x2*>(hi);
          if constexpr (LO) *reinterpret_cast<u32x2*>(nl + o * a.ldn + i) = *reinterpret_cast<const u32x2*>(lo);
          done = true;
        }
      }
      if (!done) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i + q < a.I) {
            nh[o * a.ldn + i + q] = hi[q];
            if constexpr (TT<T>::SPLIT && LO) nl[o * a.ldn + i + q] = lo[q];
          }
      }
      if constexpr (WT) {
        T* th = static_cast<T*>(a.Wt_h);
        T* tl = static_cast<T*>(a.Wt_l);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i + q < a.I) {
            th[(i + q) * a.ldt + o] = hi[q];
            if constexpr (TT<T>::SPLIT && LO) tl[(i + q) * a.ldt + o] = lo[q];
          }
      }
    }
  }
}

// ---- round 6: the 16-bit operand plane on the 16-bit matrix cores ---------------------------------------------------------------------
// loha_rebuild_mfma_kernel above forms W1 = w1a w1b and W2 = w2a w2b with v_mfma_f32_16x16x4_f32 (exact products, 1/16 of the 16-bit
// matrix rate, one ds_read_b32 per operand value): 10.4 us per layer, 8.2 ms of the SDXL step for 5 GB of plane writes
// (profiles/r06_c9_loha_kernel_stats.csv).  The rank is 32 = ONE K step of v_mfma_f32_16x16x32: with the fp32 factors split into hi + lo
// parts (x = hi + lo up to 2^-17 relative; tile.h split_f) a 16 x 16 block of W is three MFMAs (hi hi + lo hi + hi lo; the dropped lo lo
// term is 2^-18 relative) -- 2^-16 relative in W where the plane is then rounded to T at 2^-9 (bf16) / 2^-12 (fp16).
//   * a-side factors [64 o][32 r] -> LDS as T hi / lo images, K-contiguous rows: ds_read_b128 fragments;
//   * b-side factors [32 r][64 i] -> LDS as they lie in memory (k-major): ds_read_b64_tr_b16 fragments (gemm16d.h's scheme, paired
//     column order), so that a lane ends up with 8 consecutive i of one row o: 16-byte stores of the K-contiguous plane.
// Taken for T in {bf16, fp16}, R <= 32, R % 4 == 0, I % 8 == 0, 16-byte aligned factors and plane; everything else stays above.
inline bool loha_rebuild16_ok(const LohaArgs& a) {
  return a.R <= LOHA_RC && (a.R % 4) == 0 && (a.I % 8) == 0 && (a.ldn % 8) == 0 &&
         (((reinterpret_cast<uintptr_t>(a.w1a) | reinterpret_cast<uintptr_t>(a.w2a) | reinterpret_cast<uintptr_t>(a.w1b) |
            reinterpret_cast<uintptr_t>(a.w2b) | reinterpret_cast<uintptr_t>(a.Wn_h)) & 15u) == 0);
}
template <typename T>
__device__ __forceinline__ void loha_rebuild16_body(const LohaArgs& a, char* sm, const int bx, const int by) {
  // images: A1h A1l A2h A2l [64 o][32 r] T (4 KiB each), B1h B1l B2h B2l [32 r][64 i] T (4 KiB each)
  using F8 = typename TT<T>::frag;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4, wave = tid >> 6;
  const long o0 = (long)bx * LOHA_T, i0 = (long)by * LOHA_T;
  // ---- stage: every load issued before the first LDS write ---------------------------------------------------------------------
  f32x4 va[2][2], vb[2][2];  // [factor 1 / 2][iteration]
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + NTHREADS * it;
    const int o = e >> 3, c4 = (e & 7) * 4;            // a: row o, ranks c4 .. c4 + 3
    const bool ok = (o0 + o < a.O) && (c4 < a.R);
    const long idx = ok ? (o0 + o) * a.R + c4 : 0;
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.w1a + idx), x2 = *reinterpret_cast<const f32x4*>(a.w2a + idx);
    va[0][it] = ok ? x1 : z;
    va[1][it] = ok ? x2 : z;
    const int rb = e >> 4, i4 = (e & 15) * 4;          // b: rank rb, columns i4 .. i4 + 3
    const bool okb = (rb < a.R) && (i0 + i4 < a.I);
    const long idb = okb ? (long)rb * a.I + i0 + i4 : 0;
    const f32x4 y1 = *reinterpret_cast<const f32x4*>(a.w1b + idb), y2 = *reinterpret_cast<const f32x4*>(a.w2b + idb);
    vb[0][it] = okb ? y1 : z;
    vb[1][it] = okb ? y2 : z;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + NTHREADS * it;
    const int o = e >> 3, c4 = (e & 7) * 4, rb = e >> 4, i4 = (e & 15) * 4;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      T h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split_f<T>(va[f][it][q], h[q], l[q]);
      *reinterpret_cast<u32x2*>(sm + (2 * f) * 4096 + o * 64 + c4 * 2) = *reinterpret_cast<const u32x2*>(h);
      *reinterpret_cast<u32x2*>(sm + (2 * f + 1) * 4096 + o * 64 + c4 * 2) = *reinterpret_cast<const u32x2*>(l);
#pragma unroll
      for (int q = 0; q < 4; ++q) split_f<T>(vb[f][it][q], h[q], l[q]);
      *reinterpret_cast<u32x2*>(sm + (4 + 2 * f) * 4096 + rb * 128 + i4 * 2) = *reinterpret_cast<const u32x2*>(h);
      *reinterpret_cast<u32x2*>(sm + (5 + 2 * f) * 4096 + rb * 128 + i4 * 2) = *reinterpret_cast<const u32x2*>(l);
    }
  }
  __syncthreads();
  // ---- a side (MFMA B operand): lane (o = 16 wave + li, g) holds ranks 8 g .. 8 g + 7 of A1h A1l A2h A2l ------------------------------
  F8 af[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) af[p] = *reinterpret_cast<const F8*>(sm + p * 4096 + (16 * wave + li) * 64 + g * 16);
  // ---- b side (MFMA A operand), transposed reads: lane t of a 16-lane group supplies (rank 8 g + (t >> 2) [+ 4], 4 columns); paired
  //      column order: tile 2 q + e holds columns 32 q + 8 (t >> 2) + 4 e + (t & 3), so lane g owns 8 consecutive columns of the pair
  const unsigned rdb = (unsigned)(size_t)(k4_lds_ptr)sm + 4 * 4096 + (unsigned)((8 * g + (li >> 2)) * 128 + (li & 3) * 16);
  T* plane = static_cast<T*>(a.Wn_h);
  const long o = o0 + 16 * wave + li;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    u32x2 r[2][4][2];  // [e][B1h B1l B2h B2l][k half]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        r[e][p][0] = g16d_read_tr(rdb, p * 4096 + q * 64 + e * 8);
        r[e][p][1] = g16d_read_tr(rdb, p * 4096 + q * 64 + e * 8 + 4 * 128);
      }
    g16d_lgkm<0>();
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        asm volatile("" : "+v"(r[e][p][0]));
        asm volatile("" : "+v"(r[e][p][1]));
      }
    __builtin_amdgcn_sched_barrier(0);
    T ov[8];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      F8 bf[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) bf[p] = __builtin_bit_cast(F8, u32x4{r[e][p][0][0], r[e][p][0][1], r[e][p][1][0], r[e][p][1][1]});
      f32x4 w1 = zero4(), w2 = zero4();
      w1 = TT<T>::mma(bf[0], af[0], w1);
      w2 = TT<T>::mma(bf[2], af[2], w2);
      w1 = TT<T>::mma(bf[1], af[0], w1);
      w2 = TT<T>::mma(bf[3], af[2], w2);
      w1 = TT<T>::mma(bf[0], af[1], w1);
      w2 = TT<T>::mma(bf[2], af[3], w2);
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[4 * e + j] = TT<T>::from_f(w1[j] * w2[j] * a.scale);
    }
    const long n = i0 + 32 * q + 8 * g;
    if (o < a.O && n < a.I) *reinterpret_cast<u32x4*>(plane + o * a.ldn + n) = *reinterpret_cast<const u32x4*>(ov);
  }
}
template <typename T>
__global__ __launch_bounds__(NTHREADS) void loha_rebuild16_kernel(LohaArgs a) {
  __shared__ __attribute__((aligned(16))) char sm[8 * 4096];
  loha_rebuild16_body<T>(a, sm, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- round 6: the operand planes of MANY layers in one launch (the once-per-optimizer-step refresh of the plane cache) -----------------
// The factors change once per optimizer step, not per layer call: csrc/torch_ops.cpp keeps the plane of every layer whose four factors
// are leaf parameters (5 GB for the SDXL preset, of 288) and refreshes all of them with this kernel when a step has passed -- one
// throughput-bound grid per 48 layers instead of 788 latency-bound launches of 7 us inside the forward pass.
// Grouped form: a workgroup owns 128 rows x LRG_NCT column tiles.  The a-side factors of its rows are staged once (their fragments stay in
// registers), the b-side tile of column tile t + 1 is fetched into registers while tile t is computed and lands in the other half of a
// double-buffered LDS image: one barrier per tile, every b fragment read serves two row strips.  The first version ran the per-layer
// body once per 64 x 64 tile (32 KB of fp32 factors fetched per 8 KB written, two barriers, nothing in flight): 3.6 ms per refresh of the
// SDXL preset's 5 GB (profiles/r06_c22_loha_kernel_stats.csv) -- SLOWER than the 788 per-layer launches it replaced.
// Same MFMA sequence per element as loha_rebuild16_body: the planes are bit-identical.
constexpr int LRG_NCT = 4;
template <typename T>
__device__ __forceinline__ void loha_rebuild16_strip_body(const LohaArgs& a, char* sm, const int bx, const int by0) {
  // images: A1h A1l A2h A2l [128 o][32 r] T (8 KiB each) | 2 x { B1h B1l B2h B2l [32 r][64 i] T (4 KiB each) }
  using F8 = typename TT<T>::frag;
  constexpr int OFF_B = 4 * 8192, BUF = 4 * 4096;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4, wave = tid >> 6;
  const long o0 = (long)bx * 128;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 vb[2][2];  // [factor 1 / 2][iteration]
  auto load_b = [&](long i0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + NTHREADS * it;
      const int rb = e >> 4, i4 = (e & 15) * 4;
      const bool okb = (rb < a.R) && (i0 + i4 < a.I);
      const long idb = okb ? (long)rb * a.I + i0 + i4 : 0;
      const f32x4 y1 = *reinterpret_cast<const f32x4*>(a.w1b + idb), y2 = *reinterpret_cast<const f32x4*>(a.w2b + idb);
      vb[0][it] = okb ? y1 : z;
      vb[1][it] = okb ? y2 : z;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + NTHREADS * it;
      const int rb = e >> 4, i4 = (e & 15) * 4;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        T h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_f<T>(vb[f][it][q], h[q], l[q]);
        *reinterpret_cast<u32x2*>(sm + OFF_B + buf * BUF + (2 * f) * 4096 + rb * 128 + i4 * 2) = *reinterpret_cast<const u32x2*>(h);
        *reinterpret_cast<u32x2*>(sm + OFF_B + buf * BUF + (2 * f + 1) * 4096 + rb * 128 + i4 * 2) = *reinterpret_cast<const u32x2*>(l);
      }
    }
  };
  // ---- a side: 128 rows, staged once ---------------------------------------------------------------------------------------------
  {
    f32x4 va[2][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + NTHREADS * it;
      const int o = e >> 3, c4 = (e & 7) * 4;
      const bool ok = (o0 + o < a.O) && (c4 < a.R);
      const long idx = ok ? (o0 + o) * a.R + c4 : 0;
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.w1a + idx), x2 = *reinterpret_cast<const f32x4*>(a.w2a + idx);
      va[0][it] = ok ? x1 : z;
      va[1][it] = ok ? x2 : z;
    }
    load_b((long)by0 * LOHA_T);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + NTHREADS * it;
      const int o = e >> 3, c4 = (e & 7) * 4;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        T h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_f<T>(va[f][it][q], h[q], l[q]);
        *reinterpret_cast<u32x2*>(sm + (2 * f) * 8192 + o * 64 + c4 * 2) = *reinterpret_cast<const u32x2*>(h);
        *reinterpret_cast<u32x2*>(sm + (2 * f + 1) * 8192 + o * 64 + c4 * 2) = *reinterpret_cast<const u32x2*>(l);
      }
    }
    store_b(0);
  }
  __syncthreads();
  F8 af[2][4];  // [row strip][A1h A1l A2h A2l]: rows 64 s + 16 wave + li, ranks 8 g .. 8 g + 7
#pragma unroll
  for (int sI = 0; sI < 2; ++sI)
#pragma unroll
    for (int p = 0; p < 4; ++p) af[sI][p] = *reinterpret_cast<const F8*>(sm + p * 8192 + (64 * sI + 16 * wave + li) * 64 + g * 16);
  const unsigned rdb0 = (unsigned)(size_t)(k4_lds_ptr)sm + OFF_B + (unsigned)((8 * g + (li >> 2)) * 128 + (li & 3) * 16);
  T* plane = static_cast<T*>(a.Wn_h);
  const long tiles_j = (a.I + LOHA_T - 1) / LOHA_T;
  for (int ct = 0; ct < LRG_NCT; ++ct) {
    const long jt = by0 + ct;
    if (jt >= tiles_j) break;
    const long i0 = jt * LOHA_T;
    const bool more = ct + 1 < LRG_NCT && jt + 1 < tiles_j;
    if (more) load_b(i0 + LOHA_T);  // in flight behind the MFMAs of this tile
    const unsigned rdb = rdb0 + (unsigned)((ct & 1) * BUF);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x2 r[2][4][2];  // [e][B1h B1l B2h B2l][k half]
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          r[e][p][0] = g16d_read_tr(rdb, p * 4096 + q * 64 + e * 8);
          r[e][p][1] = g16d_read_tr(rdb, p * 4096 + q * 64 + e * 8 + 4 * 128);
        }
      g16d_lgkm<0>();
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          asm volatile("" : "+v"(r[e][p][0]));
          asm volatile("" : "+v"(r[e][p][1]));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sI = 0; sI < 2; ++sI) {
        T ov[8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          F8 bf[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) bf[p] = __builtin_bit_cast(F8, u32x4{r[e][p][0][0], r[e][p][0][1], r[e][p][1][0], r[e][p][1][1]});
          f32x4 w1 = zero4(), w2 = zero4();
          w1 = TT<T>::mma(bf[0], af[sI][0], w1);
          w2 = TT<T>::mma(bf[2], af[sI][2], w2);
          w1 = TT<T>::mma(bf[1], af[sI][0], w1);
          w2 = TT<T>::mma(bf[3], af[sI][2], w2);
          w1 = TT<T>::mma(bf[0], af[sI][1], w1);
          w2 = TT<T>::mma(bf[2], af[sI][3], w2);
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[4 * e + j] = TT<T>::from_f(w1[j] * w2[j] * a.scale);
        }
        const long o = o0 + 64 * sI + 16 * wave + li, n = i0 + 32 * q + 8 * g;
        if (o < a.O && n < a.I) *reinterpret_cast<u32x4*>(plane + o * a.ldn + n) = *reinterpret_cast<const u32x4*>(ov);
      }
    }
    if (more) store_b((ct + 1) & 1);
    __syncthreads();  // the next tile's image is complete; this tile's readers are done before it is overwritten two tiles on
  }
}

constexpr int LRG_MAX = 48;
struct LohaRebuildItem {
  const float *w1a, *w1b, *w2a, *w2b;
  void* plane;
  int O, I, R, ldn, gx;  // gx: 128-row blocks (block index = by * gx + bx; by: strip of LRG_NCT column tiles)
  float scale;
};
struct LohaRebuildGroupArgs {
  int n;
  int wg_end[LRG_MAX];
  LohaRebuildItem p[LRG_MAX];
};
static_assert(sizeof(LohaRebuildGroupArgs) <= 3840, "kernel arguments are limited to 4 KiB");
template <typename T>
__global__ __launch_bounds__(NTHREADS) void loha_rebuild16_group_kernel(LohaRebuildGroupArgs ga) {
  __shared__ __attribute__((aligned(16))) char sm[4 * 8192 + 2 * 4 * 4096];
  const int b = (int)blockIdx.x;
  int q = 0;
  while (q + 1 < ga.n && b >= ga.wg_end[q]) ++q;
  const int bl = b - (q ? ga.wg_end[q - 1] : 0);
  const LohaRebuildItem& it = ga.p[q];
  LohaArgs a{};
  a.w1a = it.w1a; a.w1b = it.w1b; a.w2a = it.w2a; a.w2b = it.w2b; a.Wn_h = it.plane;
  a.O = it.O; a.I = it.I; a.R = it.R; a.ldn = it.ldn; a.scale = it.scale;
  loha_rebuild16_strip_body<T>(a, sm, bl % it.gx, (bl / it.gx) * LRG_NCT);
}

struct LohaGradGeom {
  int nt;  // column tiles per workgroup (runtime); row tiles per workgroup = template NO
};

// One workgroup: row tiles ob*NO .. +NO-1, column tiles jb*nt .. +nt-1 (R <= 32: a single rank chunk stays resident;
// larger ranks run with NO = nt = 1 and loop over the chunks).
constexpr int loha_grad_lds_floats() { return 2 * LOHA_T * LH_AP + 2 * LOHA_RC * LH_BP + 2 * LOHA_T * LH_TP; }

// `bx`, `by`: this workgroup's block of row tiles / column tiles; `sm`: loha_grad_lds_floats() floats
// FAST: rank <= 32 (one chunk) and float4 staging (lh_vec_ok) are guaranteed by the host: the chunk loops and the element-wise
// staging path disappear at compile time, and with them ~60 live registers (280 -> 2 waves per SIMD without spills).
template <int NO, bool FAST>
__device__ __forceinline__ void loha_factor_grad_body(const LohaArgs& a, const LohaGradGeom& gm, float* sm, const int bx,
                                                      const int by) {
  float* sA1 = sm;
  float* sA2 = sA1 + LOHA_T * LH_AP;
  float* sB1 = sA2 + LOHA_T * LH_AP;
  float* sB2 = sB1 + LOHA_RC * LH_BP;
  float* sT1 = sB2 + LOHA_RC * LH_BP;
  float* sT2 = sT1 + LOHA_T * LH_TP;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const long ob = (long)bx * NO;  // first row tile
  const long jb = (long)by * gm.nt;
  const long tiles_j = (a.I + LOHA_T - 1) / LOHA_T;
  const int nchunk = FAST ? 1 : (a.R + LOHA_RC - 1) / LOHA_RC;
  const bool vec_in = FAST ? true : lh_vec_ok(a);

  f32x4 da1[NO][2], da2[NO][2];  // d_w*a of row tile os: rows o = 16 wave + 4 g + q, column r = 16 rt + li
#pragma unroll
  for (int os = 0; os < NO; ++os)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) da1[os][rt] = da2[os][rt] = zero4();

  auto emit_a = [&](int os, int r0) {
    const long o = (ob + os) * LOHA_T + 16 * wave + 4 * g;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + 16 * rt + li;
        if (o + q < a.O && r < a.R) {
          __hip_atomic_fetch_add(a.d_w1a + (o + q) * a.R + r, da1[os][rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(a.d_w2a + (o + q) * a.R + r, da2[os][rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        da1[os][rt][q] = 0.f;
        da2[os][rt][q] = 0.f;
      }
  };

  for (long jt = jb; jt < jb + gm.nt && jt < tiles_j; ++jt) {
    const long i0 = jt * LOHA_T;
    f32x4 db1[2], db2[2];  // d_w*b of this column tile: rows r = 16 rt + 4 g + q, column i = 16 wave + li
#pragma unroll
    for (int os = 0; os < NO; ++os) {
      const long o0 = (ob + os) * LOHA_T;
      if (o0 >= a.O) break;
      // ---- products over all rank chunks: wave rows o = 16 wave + 4 g + q, columns i = 16 c + li
      f32x4 p1[4], p2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) p1[c] = p2[c] = zero4();
      for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();  // previous users of the factor tiles / T tiles are done
        lh_stage(a, vec_in, o0, i0, ch * LOHA_RC, sA1, sA2, sB1, sB2);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < LOHA_RC / 4; ++ks) {
          const float a1 = sA1[(16 * wave + li) * LH_AP + 4 * ks + g], a2 = sA2[(16 * wave + li) * LH_AP + 4 * ks + g];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            lh_mma(a1, sB1[(4 * ks + g) * LH_BP + 16 * c + li], p1[c]);
            lh_mma(a2, sB2[(4 * ks + g) * LH_BP + 16 * c + li], p2[c]);
          }
        }
      }
      // ---- T1 = s G * P2, T2 = s G * P1 -> LDS [o][i]
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ol = 16 * wave + 4 * g + q, il = 16 * c + li;
          const bool ok = (o0 + ol < a.O) && (i0 + il < a.I);
          const float gv = a.G[ok ? (o0 + ol) * a.I + i0 + il : 0];
          const float gs = ok ? gv * a.scale : 0.f;
          sT1[ol * LH_TP + il] = gs * p2[c][q];
          sT2[ol * LH_TP + il] = gs * p1[c][q];
        }
      // ---- contractions, one rank chunk at a time (the last staged chunk is still resident)
      for (int ch = nchunk - 1; ch >= 0; --ch) {
        if (ch != nchunk - 1) {
          __syncthreads();
          lh_stage(a, vec_in, o0, i0, ch * LOHA_RC, sA1, sA2, sB1, sB2);
        }
        __syncthreads();  // T tiles (and re-staged factors) visible
        if (os == 0 || nchunk > 1) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) db1[rt] = db2[rt] = zero4();
        }
#pragma unroll 4
        for (int ks = 0; ks < LOHA_T / 4; ++ks) {
          // d_w*a[o, r] += sum_i T[o, i] b[r, i]:  A = T (row o = 16 wave + li, k = i), B = b^T (k = i, column r)
          const float t1 = sT1[(16 * wave + li) * LH_TP + 4 * ks + g], t2 = sT2[(16 * wave + li) * LH_TP + 4 * ks + g];
          // d_w*b[r, i] += sum_o a[o, r] T[o, i]:  A = a^T (row r, k = o), B = T (k = o, column i = 16 wave + li)
          const float u1 = sT1[(4 * ks + g) * LH_TP + 16 * wave + li], u2 = sT2[(4 * ks + g) * LH_TP + 16 * wave + li];
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            lh_mma(t1, sB1[(16 * rt + li) * LH_BP + 4 * ks + g], da1[os][rt]);
            lh_mma(t2, sB2[(16 * rt + li) * LH_BP + 4 * ks + g], da2[os][rt]);
            lh_mma(sA1[(4 * ks + g) * LH_AP + 16 * rt + li], u1, db1[rt]);
            lh_mma(sA2[(4 * ks + g) * LH_AP + 16 * rt + li], u2, db2[rt]);
          }
        }
        if (nchunk > 1) {  // several chunks: nothing can stay in registers, emit per chunk (NO == nt == 1 here)
          emit_a(os, ch * LOHA_RC);
          const long i = i0 + 16 * wave + li;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = ch * LOHA_RC + 16 * rt + 4 * g + q;
              if (r < a.R && i < a.I) {
                __hip_atomic_fetch_add(a.d_w1b + (long)r * a.I + i, db1[rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(a.d_w2b + (long)r * a.I + i, db2[rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
        }
      }
    }
    if (nchunk == 1) {  // d_w*b of this column tile, summed over the NO row tiles
      const long i = i0 + 16 * wave + li;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * rt + 4 * g + q;
          if (r < a.R && i < a.I) {
            __hip_atomic_fetch_add(a.d_w1b + (long)r * a.I + i, db1[rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(a.d_w2b + (long)r * a.I + i, db2[rt][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
    }
  }
  if (nchunk == 1) {
#pragma unroll
    for (int os = 0; os < NO; ++os) emit_a(os, 0);
  }
}

// Two waves per SIMD (<= 256 registers) for the lean instantiations with one or two row tiles per workgroup: the general form
// needs 280 - 512 registers (the NO = 4 block even spills), and with one wave per SIMD nothing hides the LDS / global round
// trips between the MFMA bursts of a tile -- measured: every instantiation ran at ~1/5 of its MFMA bound.
#define LYC_LH_OCC(NO, FAST) __attribute__((amdgpu_waves_per_eu(((FAST) && (NO) <= 2) ? 2 : 1, 2)))
template <int NO, bool FAST = false>
__global__ __launch_bounds__(NTHREADS) LYC_LH_OCC(NO, FAST) void loha_factor_grad_mfma_kernel(LohaArgs a, LohaGradGeom gm) {
  __shared__ __attribute__((aligned(16))) float sm[loha_grad_lds_floats()];
  loha_factor_grad_body<NO, FAST>(a, gm, sm, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- grouped launch: HadaWeight.backward of up to LHG_MAX layers in ONE grid ---------------------------------------------
// The per-layer launch needs ~400+ workgroups for parallelism, i.e. one 64 x 64 tile each, and then pays 2 fp32 atomics per
// element of G (the a-side gradients of a tile are shared with the other column tiles, the b-side with the other row tiles):
// 4.4 G atomics per SDXL step, ~15 ms at the measured 300 G/s.  With the factor gradients deferred (they feed the optimizer
// only) a batch of layers supplies the parallelism, so a workgroup can own an NO x nt BLOCK of tiles and keep the a-side sums in
// registers over nt column tiles and the b-side sums over NO row tiles: 1/nt + 1/NO atomics per element instead of 2.
constexpr int LHG_MAX = 24;
struct LohaGradItem {
  const float *w1a, *w1b, *w2a, *w2b, *G;
  float *d_w1a, *d_w1b, *d_w2a, *d_w2b;
  int O, I, R, nt, gx;  // gx: workgroups along the row-tile axis (block index = by * gx + bx)
  float scale;
};
struct LohaGradGroupArgs {
  int n;
  int wg_end[LHG_MAX];
  LohaGradItem p[LHG_MAX];
};
static_assert(sizeof(LohaGradGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

template <int NO, bool FAST = false>
__global__ __launch_bounds__(NTHREADS) LYC_LH_OCC(NO, FAST) void loha_factor_grad_group_kernel(LohaGradGroupArgs ga) {
  __shared__ __attribute__((aligned(16))) float sm[loha_grad_lds_floats()];
  const int b = (int)blockIdx.x;
  int q = 0;
  while (q + 1 < ga.n && b >= ga.wg_end[q]) ++q;
  const int bl = b - (q ? ga.wg_end[q - 1] : 0);
  const LohaGradItem& it = ga.p[q];
  LohaArgs a{};
  a.w1a = it.w1a; a.w1b = it.w1b; a.w2a = it.w2a; a.w2b = it.w2b; a.G = it.G;
  a.d_w1a = it.d_w1a; a.d_w1b = it.d_w1b; a.d_w2a = it.d_w2a; a.d_w2b = it.d_w2b;
  a.O = it.O; a.I = it.I; a.R = it.R; a.scale = it.scale;
  LohaGradGeom gm{it.nt};
  loha_factor_grad_body<NO, FAST>(a, gm, sm, bl % it.gx, bl / it.gx);
}

}  // namespace lyc
