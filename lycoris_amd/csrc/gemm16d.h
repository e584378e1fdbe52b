// gemm16d.h -- the dense 16-bit GEMMs of LoHa on kron4's skeleton (round 6: replaces gemm16.h's register-staged tiles wherever the
// operands are 16-byte aligned).
//
// LoHa's dW = (w1a w1b) * (w2a w2b) is full rank, so its activation path is three plain dense contractions on the rebuilt operand
// plane (reference lycoris/modules/loha.py:301-322, functional/loha.py:10-30):
//     y  = x  dW^T      [M, O] = [M, I] x [O, I]^T      A K-contiguous, B K-contiguous   (mode 0)
//     dx = g  dW        [M, I] = [M, O] x [O, I]        A K-contiguous, B K-strided      (mode 1)
//     G  = g^T x        [O, I] = [M, O]^T x [M, I]      A K-strided,   B K-strided      (mode 2, fp32 out, many layers per launch)
// gemm16.h (round 3) staged both operands through registers (buffer_load -> VGPR -> ds_write, one __syncthreads per K tile, 64 x 128
// tiles only): at the library's speed, 0.076 of the MFMA peak on the SDXL step.  What this file changes:
//
//   * BOTH operands go HBM -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) into a ring of D K-step slots with
//     counted `s_waitcnt vmcnt(N)` and raw `s_barrier` -- no staging registers, no ds_write, no drain per step (kron4.h's scheme);
//   * a K-CONTIGUOUS operand lies in LDS as [row][64 k] (128-byte rows); the eight 16-byte chunks of a row are stored XOR-permuted --
//     applied on the SOURCE address of the DMA, whose LDS image is lane-linear -- so that the 16 rows of a fragment read fall on 16
//     different 16-byte slots of the 256-byte bank row (conflict-free ds_read_b128);
//   * a K-STRIDED operand lies in LDS as it lies in memory, [k][BM or BN columns], and the MFMA fragments are read TRANSPOSED out of
//     that image with ds_read_b64_tr_b16 (two per fragment) -- no v_perm transposes, no second image;
//   * the MFMA roles are swapped (D[n][m] = sum_k B[n][k] A[m][k]) and the n tiles are formed in PAIRS (row i of tile 2q + e is column
//     32 q + 8 (i >> 2) + 4 e + (i & 3)): a lane then owns 8 CONSECUTIVE output columns of one row -- 16-byte stores of T (two of fp32)
//     straight from the accumulators, no LDS round trip, no barrier in the epilogue;
//   * tile shapes 128 x 128, 128 x 64 and 64 x 64 (four waves as 2 x 2), so that an attention-sized problem (1024 x 1280 x 1280) is 160
//     or 320 workgroups instead of 80, and an XCD-contiguous tile order: the eight L2s each see one contiguous range of tiles (row
//     tiles fastest: the tiles of a range share their B panel and all of A).
//
// Taken when: T in {bf16, fp16}; base pointers and leading dimensions 16-byte aligned; K % 64 == 0 for a K-contiguous operand (every
// real layer: 320 ... 10240, 9 x 320 ...), M % 8 == 0 / N % 8 == 0 for a K-strided A / B and N % 8 == 0 for the stores; every operand
// < 2 GiB.  Everything else stays on gemm16.h (ragged test shapes).
#pragma once
#include <type_traits>
#include <utility>

#include "gemm16.h"
#include "kron4.h"

namespace lyc {

constexpr int G16D_BK = 64;

__host__ __device__ constexpr int gemm16d_stage_bytes(int BM, int BN) { return (BM + BN) * 128; }
__host__ __device__ constexpr int gemm16d_lds_bytes(int BM, int BN, int D) { return D * gemm16d_stage_bytes(BM, BN); }

// ---- LDS reads as inline asm: the compiler makes every LDS read it knows about wait for ALL LDS-DMA in flight (it cannot tell which
//      slot a DMA writes), which would drain the ring on every K step (kron_dw2f.h); the waits are ours -----------------------------
__device__ __forceinline__ u32x4 g16d_read128(unsigned lds_addr, int imm_off) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_off) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 g16d_read_tr(unsigned lds_addr, int imm_off) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_off) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void g16d_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <typename F, int... Q>
__device__ __forceinline__ void g16d_for_each(F&& f, std::integer_sequence<int, Q...>) {
  (f(std::integral_constant<int, Q>{}), ...);
}

// XOR applied to the 16-byte chunk index of a row (K-contiguous image, 8 chunks per row)
//   natural row order (lane i reads row 16 t + i):                     (r >> 1) & 7
//   paired order (lane i reads row 32 q + 8 (i >> 2) + 4 e + (i & 3)):  ((r >> 1) & 1) | (((r >> 3) & 3) << 1)
// -- with the row's parity (128-byte rows in a 256-byte bank period) both give 16 distinct slots over the 16 rows of a fragment.
template <bool PAIRED>
__device__ __forceinline__ int g16d_swz_kc(int r) {
  return PAIRED ? (((r >> 1) & 1) | (((r >> 3) & 3) << 1)) : ((r >> 1) & 7);
}
// ... and of a k row (K-strided image, CH = columns / 8 chunks per k row): the four k rows a 16-lane group of a transposed read touches
// must fall on different bank ranges -- 32 bytes each in natural column order, 64 bytes (four 8-byte quads at 16-byte stride) paired
template <bool PAIRED, int CH>
__device__ __forceinline__ int g16d_swz_ks(int kr) {
  const int rho = kr & 3;
  if (CH >= 16) return PAIRED ? (rho << 2) : (rho << 1);
  return PAIRED ? ((rho >> 1) << 2) : ((rho >> 1) << 1);  // 128-byte k rows: the row's parity already separates two of the four
}

// One operand of one workgroup tile: DMA addressing (issue) and fragment addressing (read).
//   ROWS   : tile extent of this operand (BM or BN);  KS: K-strided in memory
//   PAIRED : the n operand (paired tile order), else the m operand (natural order)
template <typename T, int ROWS, bool KS, bool PAIRED>
struct G16DOperand {
  static constexpr int PIECES = ROWS / 8;                          // 1 KiB DMA pieces per K step
  static constexpr int PPW = PIECES / NWAVES;                      // per wave
  static constexpr int WEXT = ROWS / 2, NT = WEXT / 16;            // extent and MFMA tiles per wave (waves as 2 x 2)
  static constexpr int CH = ROWS / 8;                              // KS: 16-byte chunks per k row
  static constexpr int PITCH = ROWS * 2;                           // KS: bytes per k row
  static constexpr int NADDR = KS ? (PAIRED ? NT / 2 : NT) : 2;    // fragment base addresses per lane
  static_assert(PIECES % NWAVES == 0 && (!PAIRED || NT % 2 == 0), "tile shape");

  __amdgpu_buffer_rsrc_t rs;
  unsigned voff[PPW];
  unsigned kstep;          // bytes the source advances per K step (soffset = ks * kstep)
  unsigned rd[NADDR];      // LDS byte offsets (stage-relative) of this lane's fragment reads

  // base: the operand's first element; ld: leading dimension (elements); r0: first row (KC) / column (KS) of the tile; ext: rows (KC) /
  // columns (KS) of the whole operand; kdim: K; stage_off: byte offset of this operand inside a stage; wsel: which half this wave reads
  __device__ __forceinline__ void setup(const void* base, int ld, int r0, int ext, int kdim, int stage_off, int wsel) {
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (!KS) {
      // descriptor rebased to the tile's first row; rows beyond the operand are beyond the buffer (-> zeros in LDS)
      const size_t skip = (size_t)r0 * (size_t)ld * 2u;
      const size_t total = (size_t)ext * (size_t)ld * 2u;
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base)) + (skip < total ? skip : total), 0,
                                             (int)(skip < total ? total - skip : 0), K4_RSRC_FLAGS);
      kstep = G16D_BK * 2;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int p = wave + NWAVES * j;
        const int r = 8 * p + (lane >> 3), s = lane & 7;
        const int c = s ^ g16d_swz_kc<PAIRED>(r);
        voff[j] = (unsigned)r * (unsigned)ld * 2u + (unsigned)c * 16u;
      }
      // natural: rows wsel * WEXT + 16 t + i (the swizzle of row 16 t + i is that of row i); paired: rows wsel * WEXT + 32 q + 4 e +
      // 8 (i >> 2) + (i & 3) (q moves bit 5 up, e bit 2: neither enters the swizzle)
      const int r = wsel * WEXT + (PAIRED ? 8 * (li >> 2) + (li & 3) : li);
      const int x = g16d_swz_kc<PAIRED>(r);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) rd[kk] = (unsigned)(stage_off + r * 128 + (((4 * kk + g) ^ x) << 4));
    } else {
      // [K, ext] row-major; tile columns r0 .. r0 + ROWS; k rows beyond K are beyond the buffer; columns beyond ext are masked per lane
      const size_t total = (size_t)kdim * (size_t)ld * 2u;
      const size_t skip = (size_t)r0 * 2u;
      rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base)) + (skip < total ? skip : total), 0,
                                             (int)(skip < total ? total - skip : 0), K4_RSRC_FLAGS);
      kstep = (unsigned)G16D_BK * (unsigned)ld * 2u;
      constexpr int RP = 1024 / PITCH;  // k rows per piece: 4 (ROWS 128) or 8 (ROWS 64)
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int p = wave + NWAVES * j;
        const int kr = RP * p + lane / CH, s = lane % CH;
        const int c = s ^ g16d_swz_ks<PAIRED, CH>(kr);
        voff[j] = (r0 + 8 * c < ext) ? (unsigned)kr * (unsigned)ld * 2u + (unsigned)c * 16u : K4_OOB;
      }
      // transposed reads: lane t of a 16-lane group supplies the address of 8 bytes = (k row 8 g + (t >> 2) [+ 4 h + 32 kk], 4 columns)
      const int rho = li >> 2, quad = li & 3;
      const int x = g16d_swz_ks<PAIRED, CH>(rho);
      const int krow = 8 * g + rho;
      if constexpr (!PAIRED) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = wsel * WEXT + 16 * t + 4 * quad;
          rd[t] = (unsigned)(stage_off + krow * PITCH + (((col >> 3) ^ x) << 4) + (col & 7) * 2);
        }
      } else {
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
          const int col = wsel * WEXT + 32 * q + 8 * quad;  // + 4 e: the other half of the same chunk
          rd[q] = (unsigned)(stage_off + krow * PITCH + (((col >> 3) ^ x) << 4));
        }
      }
    }
  }

  // piece J of this wave for K step ks (beyond K: the K-strided source is out of bounds -> zeros, a K-contiguous one reads on into the
  // next row of a slot nobody reads: the refill is UNCONDITIONAL, so every K step issues the same number of DMA operations)
  template <int J>
  __device__ __forceinline__ void issue_piece(char* smem, int slot_off, int stage_off, int ks) const {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char* dst = smem + slot_off + stage_off + (wave + NWAVES * J) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (k4_lds_ptr)dst, 16, (int)voff[J], (int)((unsigned)ks * kstep), 0, 0);
  }
  template <int... J>
  __device__ __forceinline__ void issue(char* smem, int slot_off, int stage_off, int ks, std::integer_sequence<int, J...>) const {
    (issue_piece<J>(smem, slot_off, stage_off, ks), ...);
  }
  __device__ __forceinline__ void issue(char* smem, int slot_off, int stage_off, int ks) const {
    issue(smem, slot_off, stage_off, ks, std::make_integer_sequence<int, PPW>{});
  }

  // The fragments of K half KK (32 deep) of the stage at LDS byte address `sb`: lane (i, g) receives k = 32 KK + 8 g .. + 7 of its row.
  // Issued as asm (above); the results may be touched only behind tie(), which follows the s_waitcnt that retires them.
  static constexpr int NRAW = KS ? 2 * NT : NT;
  using RawT = typename std::conditional<KS, u32x2, u32x4>::type;
  template <int KK, int... TI>
  __device__ __forceinline__ void read(unsigned sb, RawT (&f)[NRAW], std::integer_sequence<int, TI...>) const {
    if constexpr (!KS) {
      ((f[TI] = g16d_read128(sb + rd[KK], PAIRED ? (TI >> 1) * 4096 + (TI & 1) * 512 : TI * 2048)), ...);
    } else {
      ((f[2 * TI] = g16d_read_tr(sb + rd[PAIRED ? TI >> 1 : TI], KK * 32 * PITCH + (PAIRED ? (TI & 1) * 8 : 0)),
        f[2 * TI + 1] = g16d_read_tr(sb + rd[PAIRED ? TI >> 1 : TI], KK * 32 * PITCH + (PAIRED ? (TI & 1) * 8 : 0) + 4 * PITCH)),
       ...);
    }
  }
  template <int KK>
  __device__ __forceinline__ void read(unsigned sb, RawT (&f)[NRAW]) const {
    read<KK>(sb, f, std::make_integer_sequence<int, NT>{});
  }
  static __device__ __forceinline__ void tie(RawT (&f)[NRAW]) {
#pragma unroll
    for (int t = 0; t < NRAW; ++t) asm volatile("" : "+v"(f[t]));
  }
  using F8 = typename TT<T>::frag;
  static __device__ __forceinline__ F8 get(const RawT (&f)[NRAW], int t) {
    if constexpr (!KS) {
      return __builtin_bit_cast(F8, f[t]);
    } else {
      return __builtin_bit_cast(F8, u32x4{f[2 * t][0], f[2 * t][1], f[2 * t + 1][0], f[2 * t + 1][1]});
    }
  }
};

// tile -> (tm, tn): consecutive workgroup ids go round-robin over the 8 XCDs (one L2 each), so XCD x takes the workgroups x, x + 8, ...
// Round 6 (first version): XCD x owned a contiguous range of the column-major tile list -- every XCD read ALL of A and 1 / 8 of B
// (measured fabric reads of a 1024 x 1280 x 1280 problem: 36 - 50 MB for 6 MB of operands, profiles/r06_interim_pmc_traffic_loha_*.txt).
// Now the 8 XCDs form a gm x gn grid over the output (gm * gn = 8) and XCD (xm, xn) owns the block of tm_per x tn_per tiles at
// (xm * tm_per, xn * tn_per), row tiles fastest inside it: its L2 holds M / gm rows of A and N / gn rows of B.  gm minimises the
// first-touch traffic 8 * (M / gm + N / gn) * K = (M * gn + N * gm) * K (ties: the smaller gm).  -DG16D_MAP1D: the first version (A/B).
struct G16DMap {
  int lgm, tm_per, tn_per;  // gm = 1 << lgm, gn = 8 >> lgm
};
__host__ __device__ inline G16DMap gemm16d_map(int tiles_m, int tiles_n, long M, long N) {
  G16DMap best{0, tiles_m, (tiles_n + 7) >> 3};
#ifndef G16D_MAP1D
  long cost = M * 8 + N;
#pragma unroll
  for (int lg = 1; lg <= 3; ++lg) {  // (shifts only: this runs in every workgroup's prologue)
    const int gm = 1 << lg, gn = 8 >> lg;
    const long c = M * gn + N * gm;
    if (gm <= tiles_m && gn <= tiles_n && c < cost) {
      cost = c;
      best = G16DMap{lg, (tiles_m + gm - 1) >> lg, (tiles_n + gn - 1) >> (3 - lg)};
    }
  }
#endif
  return best;
}
// workgroups of a problem (a multiple of 8; the surplus ones return)
__host__ __device__ inline int gemm16d_wgs(long M, long N, int bm, int bn) {
  const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (int)((N + bn - 1) / bn);
  const G16DMap mp = gemm16d_map(tiles_m, tiles_n, M, N);
  return 8 * mp.tm_per * mp.tn_per;
}

// ABL != 0: ablation builds of benchmarks/g16bench.cpp (results are garbage): 1 = no refill DMA in the loop, 2 = no MFMAs, 4 = no LDS
// fragment reads, 8 = no barrier
template <typename T, int BM, int BN, bool A_KS, bool B_KS, int D, int ABL = 0>
__device__ __forceinline__ void gemm16d_body(const Gemm16Prob& p, int out_f32, char* smem, int b) {
  using OA = G16DOperand<T, BM, A_KS, false>;
  using OB = G16DOperand<T, BN, B_KS, true>;
  constexpr int MI = OA::NT, NI = OB::NT;
  constexpr int STAGE = gemm16d_stage_bytes(BM, BN), OFF_B = BM * 128;
  constexpr int C = OA::PPW + OB::PPW;  // DMA operations per wave and K step
  static_assert(D >= 2 && D <= 7 && (D - 1) * C <= 63, "ring depth");

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const G16DMap mp = gemm16d_map(tiles_m, tiles_n, p.M, p.N);
  const int xcd = b & 7, idx = b >> 3;
  const int ln = idx / mp.tm_per, lm = idx - ln * mp.tm_per;
  const int tm = (xcd & ((1 << mp.lgm) - 1)) * mp.tm_per + lm, tn = (xcd >> mp.lgm) * mp.tn_per + ln;
  if (ln >= mp.tn_per || tm >= tiles_m || tn >= tiles_n) return;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  OA oa;
  OB ob;
  oa.setup(p.A, p.lda, m0, p.M, p.K, 0, wr);
  ob.setup(p.B, p.ldb, n0, p.N, p.K, OFF_B, wc);

  // Ring of D slots, two-phase register pipeline.  A K step is two 32-deep halves; the fragments of a half are read from LDS while the
  // MFMAs of the half before run, the workgroup barrier sits in the MIDDLE of the step:
  //
  //     read half 1 (ks)            | MFMA half 0 (ks)                    [its fragments were read during the step before]
  //     wait: half 1 landed, step ks + 1 landed (counted vmcnt) -- s_barrier: slot ks is free, slot ks + 1 is visible
  //     read half 0 (ks + 1)        | MFMA half 1 (ks)  + the refill of slot ks with step ks + D, one DMA behind every STRIDE-th MFMA
  //
  // so the LDS pipe (a 128 x 128 step reads 64 KiB = 512 cycles at 128 B / clk) and the matrix pipe (32 MFMAs per wave = 512 cycles)
  // run side by side instead of one after the other.  Measured history (profiles/r06_c2_g16bench_first_version.log, r06_c4_g16bench_two_phase_pipeline.log): all refills of a step
  // issued up front, reads, then MFMAs: ~1270 cycles per K step; refills interleaved: the same (the LDS reads were the exposed part).
  // The refill is UNCONDITIONAL (beyond K it fetches zeros / dead rows), so "all but the newest D - 2 groups have landed" is one
  // counted wait, the same in every step.
  const int nk = (p.K + G16D_BK - 1) / G16D_BK;
  for (int s = 0; s < D; ++s) {
    oa.issue(smem, s * STAGE, 0, s);
    ob.issue(smem, s * STAGE, OFF_B, s);
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  constexpr int NH = MI * NI;          // MFMAs per wave and K half
  constexpr int STRIDE = NH / C;       // one refill DMA behind every STRIDE-th MFMA of the second half
  static_assert(STRIDE >= 1 && STRIDE * C <= NH, "refill interleave");
  constexpr int R1 = OA::NRAW + OB::NRAW;  // LDS reads of one K half (lgkmcnt is a 4-bit counter: at most 15 are named)
  const unsigned sbase = (unsigned)(size_t)(k4_lds_ptr)smem;
  typename OA::RawT a0[OA::NRAW], a1[OA::NRAW];
  typename OB::RawT b0[OB::NRAW], b1[OB::NRAW];
  k4_wait_vm<(D - 1) * C>();
  __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain the DMA queue (vmcnt(0))
  asm volatile("" ::: "memory");
  ob.template read<0>(sbase, b0);
  oa.template read<0>(sbase, a0);
  int slot = 0;
  for (int ks = 0; ks < nk; ++ks) {
    const unsigned sb = sbase + (unsigned)(slot * STAGE);
    const int fill_off = slot * STAGE, ksf = ks + D;
    const int next = slot + 1 == D ? 0 : slot + 1;
    if constexpr ((ABL & 4) == 0) {
      ob.template read<1>(sb, b1);
      oa.template read<1>(sb, a1);
    }
    g16d_lgkm<(R1 > 15 ? 15 : R1)>();
    OB::tie(b0);
    OA::tie(a0);
    __builtin_amdgcn_sched_barrier(0);
    g16d_for_each([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      if constexpr ((ABL & 2) == 0) acc[Q % MI][Q / MI] = TT<T>::mma(OB::get(b0, Q / MI), OA::get(a0, Q % MI), acc[Q % MI][Q / MI]);
    }, std::make_integer_sequence<int, NH>{});
    __builtin_amdgcn_sched_barrier(0);
    g16d_lgkm<0>();
    OB::tie(b1);
    OA::tie(a1);
    if constexpr ((ABL & 1) == 0) k4_wait_vm<(D - 2) * C>();
    if constexpr ((ABL & 8) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned sn = sbase + (unsigned)(next * STAGE);
    if constexpr ((ABL & 4) == 0) {
      ob.template read<0>(sn, b0);
      oa.template read<0>(sn, a0);
    }
    __builtin_amdgcn_sched_barrier(0);
    g16d_for_each([&](auto qc) {
      constexpr int Q = decltype(qc)::value;
      if constexpr ((ABL & 2) == 0) acc[Q % MI][Q / MI] = TT<T>::mma(OB::get(b1, Q / MI), OA::get(a1, Q % MI), acc[Q % MI][Q / MI]);
      if constexpr ((Q + 1) % STRIDE == 0 && (Q + 1) / STRIDE <= C && (ABL & 1) == 0) {
        constexpr int P = (Q + 1) / STRIDE - 1;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P < OA::PPW) oa.template issue_piece<P>(smem, fill_off, 0, ksf);
        else ob.template issue_piece<P - OA::PPW>(smem, fill_off, OFF_B, ksf);
        __builtin_amdgcn_sched_barrier(0);
      }
    }, std::make_integer_sequence<int, NH>{});
    slot = next;
  }
  // the fragment reads of the step behind the last one and the surplus refills (zeros / dead rows) have landed before anything else
  // is put into those registers / before the workgroup leaves
  g16d_lgkm<0>();
  OB::tie(b0);
  OA::tie(a0);
  k4_wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: straight from the accumulators.  acc[mi][2q + e] of lane (li, g): row m0 + wr WM + 16 mi + li, columns
  //      n0 + wc WN + 32 q + 8 g + 4 e + {0..3} -- a tile pair is 8 consecutive columns --------------------------------------------
  const size_t cbytes = (size_t)p.M * (size_t)p.ldc * (out_f32 ? 4u : 2u);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)cbytes, K4_RSRC_FLAGS);
  const float alpha = p.alpha;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wr * (BM / 2) + 16 * mi + li;
#pragma unroll
    for (int q = 0; q < NI / 2; ++q) {
      const int n = n0 + wc * (BN / 2) + 32 * q + 8 * g;
      const bool ok = m < p.M && n < p.N;  // N % 8 == 0: a group of 8 columns is all in or all out
      const f32x4 lo = acc[mi][2 * q], hi = acc[mi][2 * q + 1];
      if (out_f32) {
        const unsigned off = ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 4u : K4_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{alpha * lo[0], alpha * lo[1], alpha * lo[2], alpha * lo[3]}), rc,
                                               (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{alpha * hi[0], alpha * hi[1], alpha * hi[2], alpha * hi[3]}), rc,
                                               (int)off, 16, 0);
      } else {
        T o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = TT<T>::from_f(alpha * lo[e]);
          o[4 + e] = TT<T>::from_f(alpha * hi[e]);
        }
        const unsigned off = ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)n) * 2u : K4_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<u32x4*>(o), rc, (int)off, 0, 0);
      }
    }
  }
}

template <int BM, int BN, int D>
__host__ __device__ constexpr int gemm16d_occupancy() { return 2 * gemm16d_lds_bytes(BM, BN, D) <= 160 * 1024 ? 2 : 1; }

template <typename T, int BM, int BN, bool A_KS, bool B_KS, int D, int ABL = 0>
__global__ __launch_bounds__(NTHREADS, (gemm16d_occupancy<BM, BN, D>())) void gemm16d_kernel(Gemm16Group ga) {
  extern __shared__ __attribute__((aligned(1024))) char g16d_smem[];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;
  const int b0 = p ? ga.wg_end[p - 1] : 0;  // a multiple of 8: every problem's range is gemm16d_wgs(M, N, BM, BN)
  gemm16d_body<T, BM, BN, A_KS, B_KS, D, ABL>(ga.p[p], ga.out_f32, g16d_smem, b - b0);
}

// can the DMA kernel take this problem?  (mode bits as gemm16.h: A_KS / B_KS)
inline bool gemm16d_ok(const Gemm16Prob& p, bool a_ks, bool b_ks, bool out_f32) {
  const bool al = ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C)) & 15u) == 0 &&
                  (p.lda % 8) == 0 && (p.ldb % 8) == 0 && (p.ldc % (out_f32 ? 4 : 8)) == 0;
  const bool kc = (a_ks && b_ks) || (p.K % G16D_BK) == 0;
  const bool ks = (!a_ks || (p.M % 8) == 0) && (!b_ks || (p.N % 8) == 0) && (p.N % 8) == 0;
  const long ae = a_ks ? (long)p.K * p.lda : (long)p.M * p.lda, be = b_ks ? (long)p.K * p.ldb : (long)p.N * p.ldb;
  const long ce = (long)p.M * p.ldc * (out_f32 ? 2 : 1);
  return al && kc && ks && p.M >= 1 && p.N >= 1 && p.K >= 1 && ae < (1L << 30) && be < (1L << 30) && ce < (1L << 30);
}

}  // namespace lyc
