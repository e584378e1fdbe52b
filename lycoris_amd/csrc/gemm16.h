// gemm16.h -- the dense 16-bit GEMMs of LoHa, hand-written for gfx950 (round 3: replaces the rocBLAS calls of rounds 1-2).
//
// LoHa's dW = (w1a w1b) * (w2a w2b) is full rank, so its activation path is three plain dense contractions on the rebuilt
// operand plane (reference lycoris/modules/loha.py:301-322, functional/loha.py:10-30):
//     y  = x  dW^T      [M, O] = [M, I] x [O, I]^T      A K-contiguous, B K-contiguous   ("NT")
//     dx = g  dW        [M, I] = [M, O] x [O, I]        A K-contiguous, B K-strided      ("NN")
//     G  = g^T x        [O, I] = [M, O]^T x [M, I]      A K-strided,   B K-strided      ("TN", fp32 out, up to 24 layers per launch)
// One kernel template: 64 x 128 (or 128 x 128) output tile, BK = 64, four waves as 2 x 2, v_mfma_f32_16x16x32, fp32
// accumulation.  Both operand tiles live K-contiguous in LDS ([row][64 + 8]: 144-byte pitch, ds_read_b128 fragments);
// a K-contiguous operand is staged with 16-byte loads, a K-strided one as 4 (k) x 8 (row) register blocks transposed with
// v_perm_b32 (tile.h stage_cols' scheme).  Software pipeline: the global loads of K tile t + 2 are issued before the MFMAs of
// tile t (two register stages) and written to the other LDS buffer a tile later -- one barrier per K tile.  All loads go through buffer
// descriptors: rows / columns beyond the matrix return zeros, no edge branches.  Output through an LDS image: 16-byte
// coalesced stores of T or fp32.
#pragma once
#include "tile.h"

namespace lyc {

constexpr int G16_BK = 64;
constexpr int G16_TM = 64;          // rows of the output tile the launchers use (the template also builds with 128)
constexpr int G16_LD = G16_BK + 8;  // LDS pitch (elements): 144 bytes

struct Gemm16Prob {
  const void* A;   // KC: [M, K] (lda)   KS: [K, M] (lda)
  const void* B;   // KC: [N, K] (ldb)   KS: [K, N] (ldb)
  void* C;         // [M, N] (ldc), T or fp32
  int M, N, K;
  int lda, ldb, ldc;
  float alpha;
};
constexpr int G16_MAX = 24;
struct Gemm16Group {
  int n;
  int out_f32;
  int wg_end[G16_MAX];
  Gemm16Prob p[G16_MAX];
};
static_assert(sizeof(Gemm16Group) <= 3584, "kernel arguments are limited to 4 KiB");

// can gemm16_kernel take this problem?  (16-byte vectors along the contiguous dimension of each operand, 32-bit offsets)
inline bool gemm16_ok(const Gemm16Prob& p, bool a_ks, bool b_ks) {
  const bool al = ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15u) == 0 && (p.lda % 8) == 0 && (p.ldb % 8) == 0;
  const bool kc = (a_ks && b_ks) || (p.K % 8) == 0;  // a K-contiguous operand is read in 16-byte vectors along K
  const bool ks = (!a_ks || (p.M % 8) == 0) && (!b_ks || (p.N % 8) == 0);
  const long ae = a_ks ? (long)p.K * p.lda : (long)p.M * p.lda, be = b_ks ? (long)p.K * p.ldb : (long)p.N * p.ldb;
  return al && kc && ks && ae < (1L << 30) && be < (1L << 30);
}

template <int TM>
__host__ __device__ constexpr int gemm16_lds_bytes() {
  const int stage = 2 * (TM + 128) * G16_LD * 2;   // two buffers, A + B tiles
  const int epi = TM * (128 + 4) * 4;               // fp32 output image
  return stage > epi ? stage : epi;
}

// LDS image [row][9 units of 16 bytes] (8 data units + 1 pad).  The data unit u of row r lives at unit u ^ ((r >> 4) & 7): the
// transposing store of a K-strided operand writes, per instruction, the SAME unit of rows 8 apart -- 8 * 144 bytes = 0 mod the
// 128-byte bank period, a 16-way conflict on every ds_write_b64 without the swizzle (first version: the NN / TN forms ran 2.2x
// slower than NT); with it only the two blocks of one 16-row group still alias (2-way: free for a 64-bit store).  The 16 rows
// of a fragment share one XOR value, so the conflict-free pattern of the fragment reads is unchanged.
__device__ __forceinline__ int g16_unit(int row, int u) { return u ^ ((row >> 4) & 7); }

// ---- staging -------------------------------------------------------------------------------------------------------------
// ROWS x 64 operand tile.  KC: element (r, k) at (r0 + r) * ld + k0 + k.  KS: at (k0 + k) * ld + r0 + r.
template <int ROWS, bool KS>
struct G16Stage {
  static constexpr int NV = KS ? (ROWS >= 128 ? 4 : 4) : ROWS / 32;  // 16-byte registers per thread
  u32x4 v[KS ? 4 : ROWS / 32];
};

template <typename T, int ROWS, bool KS>
__device__ __forceinline__ void g16_load(G16Stage<ROWS, KS>& s, const __amdgpu_buffer_rsrc_t& rs, int ld, int r0, int rows_total, int k0,
                                         int k_total) {
  const int tid = threadIdx.x;
  const int OOR = 0x7ffffff0;
  if constexpr (!KS) {
#pragma unroll
    for (int j = 0; j < ROWS / 32; ++j) {
      const int vi = tid + NTHREADS * j;
      const int r = vi >> 3, kv = (vi & 7) * 8;
      const bool ok = (r0 + r) < rows_total && (k0 + kv) < k_total;  // K % 8 == 0: a vector is all in or all out
      const int off = ok ? ((r0 + r) * ld + k0 + kv) * 2 : OOR;
      s.v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  } else {
    // blocks of 4 k rows x 8 r columns: ROWS / 8 blocks along r (consecutive threads: coalesced k rows), 16 along k
    constexpr int RB = ROWS / 8;
    const bool active = tid < RB * 16;
    const int rb = tid % RB, kb = tid / RB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * kb + j, r = r0 + 8 * rb;
      const bool ok = active && k < k_total && r < rows_total;  // rows_total % 8 == 0
      const int off = ok ? (k * ld + r) * 2 : OOR;
      s.v[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  }
}

template <typename T, int ROWS, bool KS>
__device__ __forceinline__ void g16_store(const G16Stage<ROWS, KS>& s, T* tile) {
  const int tid = threadIdx.x;
  if constexpr (!KS) {
#pragma unroll
    for (int j = 0; j < ROWS / 32; ++j) {
      const int vi = tid + NTHREADS * j;
      const int r = vi >> 3;
      *reinterpret_cast<u32x4*>(tile + r * G16_LD + 8 * g16_unit(r, vi & 7)) = s.v[j];
    }
  } else {
    constexpr int RB = ROWS / 8;
    if (tid < RB * 16) {
      const int rb = tid % RB, kb = tid / RB;
      // rows 8 rb .. 8 rb + 7 share (row >> 4): one swizzled unit for the whole block; 4 k values = half a unit
      T* dst = tile + (8 * rb) * G16_LD + 8 * g16_unit(8 * rb, kb >> 1) + 4 * (kb & 1);
#pragma unroll
      for (int w = 0; w < 4; ++w) {  // columns 2w (low halves) and 2w + 1 (high halves) of the four k rows
        u32x2 lo, hi;
        lo[0] = __builtin_amdgcn_perm(s.v[1][w], s.v[0][w], 0x05040100u);
        lo[1] = __builtin_amdgcn_perm(s.v[3][w], s.v[2][w], 0x05040100u);
        hi[0] = __builtin_amdgcn_perm(s.v[1][w], s.v[0][w], 0x07060302u);
        hi[1] = __builtin_amdgcn_perm(s.v[3][w], s.v[2][w], 0x07060302u);
        *reinterpret_cast<u32x2*>(dst + (2 * w) * G16_LD) = lo;
        *reinterpret_cast<u32x2*>(dst + (2 * w + 1) * G16_LD) = hi;
      }
    }
  }
}

template <typename T, int TM, bool A_KS, bool B_KS>
__device__ __forceinline__ void gemm16_body(const Gemm16Prob& p, int out_f32, char* smem, int tile) {
  constexpr int TN = 128, MI = TM / 32, NI = 4;
  using F8 = typename TT<T>::frag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_m = (p.M + TM - 1) / TM;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * TM, n0 = tn * TN;
  T* As[2] = {reinterpret_cast<T*>(smem), reinterpret_cast<T*>(smem) + (TM + TN) * G16_LD};
  T* Bs[2] = {As[0] + TM * G16_LD, As[1] + TM * G16_LD};

  const long a_elems = A_KS ? (long)p.K * p.lda : (long)p.M * p.lda;
  const long b_elems = B_KS ? (long)p.K * p.ldb : (long)p.N * p.ldb;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)(a_elems * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)(b_elems * 2), 0x00020000);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  // Software pipeline, prefetch distance TWO K tiles: the per-layer problems of this workload put ~1 workgroup on a CU, whose
  // time is then the serial chain of its K loop -- with the loads of tile t + 1 issued only one tile of MFMAs (~500 cycles)
  // ahead, every iteration waited out most of an L2 / HBM round trip (first version: 2.5x slower than the vendor library).
  // Two register stages: tile t + 2 is requested into the stage that tile t has just left for LDS.
  G16Stage<TM, A_KS> sa0, sa1;
  G16Stage<TN, B_KS> sb0, sb1;
  const int nk = (p.K + G16_BK - 1) / G16_BK;
  g16_load<T, TM, A_KS>(sa0, ra, p.lda, m0, p.M, 0, p.K);
  g16_load<T, TN, B_KS>(sb0, rb, p.ldb, n0, p.N, 0, p.K);
  // (loads are UNCONDITIONAL: a tile beyond K is all out of range and comes back as zeros without memory traffic; a load under
  // `if (kt + 2 < nk)` would make the stage a PHI of loaded / not loaded and hipcc then waits vmcnt(0) at the join)
  g16_load<T, TM, A_KS>(sa1, ra, p.lda, m0, p.M, G16_BK, p.K);
  g16_load<T, TN, B_KS>(sb1, rb, p.ldb, n0, p.N, G16_BK, p.K);
  g16_store<T, TM, A_KS>(sa0, As[0]);
  g16_store<T, TN, B_KS>(sb0, Bs[0]);
  __syncthreads();
  auto compute = [&](int cur) {
    const T* at = As[cur];
    const T* bt = Bs[cur];
    const int ar0 = wr * (TM / 2) + li, br0 = wc * 64 + li;  // tile-local rows of this lane's fragments (+ 16 mi / 16 ni)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      F8 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = *reinterpret_cast<const F8*>(at + (ar0 + 16 * mi) * G16_LD + 8 * g16_unit(ar0 + 16 * mi, 4 * ks + g));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const F8*>(bt + (br0 + 16 * ni) * G16_LD + 8 * g16_unit(br0 + 16 * ni, 4 * ks + g));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT<T>::mma(af[mi], bf[ni], acc[mi][ni]);
    }
  };
  // even tiles travel through stage 0 and LDS buffer 0, odd tiles through stage 1 and buffer 1 (two tiles per trip: static
  // register indexing)
  for (int kt = 0; kt < nk; kt += 2) {
    // tile kt is in buffer 0; tile kt + 1 is in flight in stage 1; stage 0 is free
    g16_load<T, TM, A_KS>(sa0, ra, p.lda, m0, p.M, (kt + 2) * G16_BK, p.K);
    g16_load<T, TN, B_KS>(sb0, rb, p.ldb, n0, p.N, (kt + 2) * G16_BK, p.K);
    compute(0);
    g16_store<T, TM, A_KS>(sa1, As[1]);
    g16_store<T, TN, B_KS>(sb1, Bs[1]);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // tile kt + 1 is in buffer 1; tile kt + 2 is in flight in stage 0; stage 1 is free
    g16_load<T, TM, A_KS>(sa1, ra, p.lda, m0, p.M, (kt + 3) * G16_BK, p.K);
    g16_load<T, TN, B_KS>(sb1, rb, p.ldb, n0, p.N, (kt + 3) * G16_BK, p.K);
    compute(1);
    g16_store<T, TM, A_KS>(sa0, As[0]);
    g16_store<T, TN, B_KS>(sb0, Bs[0]);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS image -> coalesced stores --------------------------------------------------------------
  // (the last barrier of the loop has passed: nobody reads the operand tiles any more)
  if (out_f32) {
    constexpr int LDO = TN + 4;
    float* Os = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Os[(wr * (TM / 2) + 16 * mi + 4 * g + r) * LDO + wc * 64 + 16 * ni + li] = p.alpha * acc[mi][ni][r];
    __syncthreads();
    float* C = static_cast<float*>(p.C);
    const bool vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
    for (int v = tid; v < TM * (TN / 4); v += NTHREADS) {
      const int r = v / (TN / 4), c = (v % (TN / 4)) * 4;
      const long gr = m0 + r, gc = n0 + c;
      if (gr >= p.M || gc >= p.N) continue;
      const f32x4 val = *reinterpret_cast<const f32x4*>(Os + r * LDO + c);
      if (vec && gc + 4 <= p.N) {
        *reinterpret_cast<f32x4*>(C + gr * p.ldc + gc) = val;
      } else {
        for (int e = 0; e < 4 && gc + e < p.N; ++e) C[gr * p.ldc + gc + e] = val[e];
      }
    }
  } else {
    constexpr int LDO = TN + 8;
    T* Os = reinterpret_cast<T*>(smem);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Os[(wr * (TM / 2) + 16 * mi + 4 * g + r) * LDO + wc * 64 + 16 * ni + li] = TT<T>::from_f(p.alpha * acc[mi][ni][r]);
    __syncthreads();
    T* C = static_cast<T*>(p.C);
    const bool vec = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
    for (int v = tid; v < TM * (TN / 8); v += NTHREADS) {
      const int r = v / (TN / 8), c = (v % (TN / 8)) * 8;
      const long gr = m0 + r, gc = n0 + c;
      if (gr >= p.M || gc >= p.N) continue;
      if (vec && gc + 8 <= p.N) {
        *reinterpret_cast<u32x4*>(C + gr * p.ldc + gc) = *reinterpret_cast<const u32x4*>(Os + r * LDO + c);
      } else {
        for (int e = 0; e < 8 && gc + e < p.N; ++e) C[gr * p.ldc + gc + e] = Os[r * LDO + c + e];
      }
    }
  }
}

template <typename T, int TM, bool A_KS, bool B_KS>
__global__ __launch_bounds__(NTHREADS, 2) void gemm16_kernel(Gemm16Group ga) {  // two workgroups per CU (72 KB of LDS each)
  extern __shared__ __attribute__((aligned(16))) char g16_smem[];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;
  const int b0 = p ? ga.wg_end[p - 1] : 0;
  gemm16_body<T, TM, A_KS, B_KS>(ga.p[p], ga.out_f32, g16_smem, b - b0);
}

}  // namespace lyc
