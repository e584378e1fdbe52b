// lowrank4.h -- rank-r (LoCon / LoRA) "reduce to r channels, expand again" launch on kron4's skeleton, gfx950.  Round 6.
//
//   mid[m, n]  = alpha1 * sum_k A[m, k] * F1(n, k)                 stage 1, v_mfma_f32_16x16x32 (F1 split into hi + lo on the fly)
//   out[m, c]  = alpha2 * sum_n mid[m, n] * F2(c, n)               stage 2, v_mfma_f32_16x16x16 (both operands hi / lo: 3 MFMAs)
//
// The math, the tile ownership (a workgroup = 16 MI rows x one slice of the output columns, every slice repeats stage 1) and
// the accumulator layouts are bneck_kernel's (lowrank.h; reference: lycoris/functional/locon.py:64-85, modules/locon.py:286-332).
// What changed is how the operands travel (shader-clock stamps of benchmarks/lctrace, round 6 call 26: a bneck workgroup of the 1024 x 1280 -> 1280
// layer lives 5.7 us, 4.0 of them in stage 1 with THREE 6 KiB steps in flight per wave, 1.35 in a stage 2 of five serial
// tiles per wave):
//
//   * every operand goes HBM / L2 -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) into rings that are
//     PRIVATE to a wave: no staging registers, no ds_write, and no barrier in either stage -- a wave waits with counted vmcnt
//     for ITS OWN pieces only.  A k step of 32 is (MI + 2) KiB per wave (MI pieces of 16 rows x 64 bytes of A, two pieces of
//     16 x 16 fp32 of F1); the ring holds D steps, so a K = 1280 layer has its whole stage-1 input in flight at once;
//   * the F2 tiles of the wave's share of the output columns are requested FIRST, before stage 1 (they are the oldest entries
//     of the vmcnt queue: a counted stage-1 wait covers them), and a slice is planned so that they all fit the prologue: stage 2
//     issues no load, so its stores never stand between a counted wait and the load it waits for (loads and stores share vmcnt
//     on gfx9 and may retire out of order with respect to each other);
//   * output columns are formed in PAIRS of MFMA tiles (kron4.h "NP"): which F2 row feeds which MFMA row is chosen by the DMA
//     source address, lane (m, g) then owns 8 consecutive columns of its row -> one 16-byte store per row and pair;
//   * the transposed factor layouts of the backward pass (F1 = up^T, F2 = down^T) are DMA'd as they lie in memory
//     ([k][n] rows) and read back with ds_read_b32; the ROW placement inside a piece is permuted on the DMA source side so that
//     the four lane groups of a read instruction hit four different 64-byte bank quarters.
//
// Taken when (capi.hip: bneck4_plan): T in {bf16, fp16}, R <= 16, R % 4 == 0, K1 % 8 == 0, N2 % 8 == 0, dense factors,
// 16-byte aligned dense rows, every tensor < 2 GiB.  Everything else stays on bneck_kernel.
#pragma once
#include "kron4.h"
#include "lowrank.h"

#ifndef B4_ABL
#define B4_ABL 0  // ablation builds of benchmarks/lcbench.cpp (results are garbage): 1 = no A DMA, 2 = no F1 DMA, 4 = no stage-1 reads / MFMAs
#endif

namespace lyc {

struct Bneck4Args {
  const void* A;    // [M, K1] T, row pitch lda (elements, % 8 == 0), 16-byte aligned
  const float* F1;  // FT = false: [R, K1] (row n, k contiguous);  FT = true: [K1, R] (row k, n contiguous)
  const float* F2;  // FT = false: [N2, R] (row c, n contiguous);  FT = true: [R, N2] (row n, c contiguous)
  float* mid;       // [M, R] fp32 or nullptr
  void* out;        // [M, N2] T, row pitch ldo (elements, % 8 == 0), 16-byte aligned; nullptr = only mid wanted
  unsigned a_bytes, f1_bytes, f2_bytes, out_bytes;  // extents of the four tensors (buffer descriptors: out of range = zeros / dropped)
  int lda, ldo;
  int M, K1, KS;    // KS = ceil(K1 / 32)
  int R, N2;
  float alpha1, alpha2;
  int D, D2;        // ring depths per wave: k steps of (MI + 2) KiB, column pairs of 2 KiB (all pairs of the slice: D2 >= pairs per wave)
};

// nsum: problems per workgroup (1, or the n of bneck4_sum_kernel)
__host__ __device__ inline int bneck4_lds_bytes(int NW, int MI, int D, int D2, int nsum = 1) {
  return NW * (D * (MI + 2) * 1024 + nsum * D2 * 2048) + NW * MI * 1024 + nsum * MI * 2048;
}

// Host-side tile plan (capi.hip, benchmarks/lcbench.cpp).  Measured on the SDXL / SD1.5 shapes (profiles/r06_c30_lcbench_library_plan.log, r06_c34_lcbench_sibling_sum.log):
//   * 8 waves of 16 rows below 8192 rows (one wave's DMA issue overlaps another wave's fragment reads and MFMAs: 7.5 -> 6.3 us on the
//     1024 x 1280 -> 1280 layer), 4 waves of 32 rows above (many row tiles: the factor traffic per row halves);
//   * as many column slices as keep the grid within one round of 256 workgroups, and at least as many as make a wave's share of the
//     column pairs fit its prologue (D2 <= 32 / NW pairs);
//   * the stage-1 ring as deep as the LDS allows (two workgroups per CU once the grid has more than one round), at most 8 steps.
struct Bneck4Plan {
  int nw, mi, ns, D, D2, lds;
};
inline bool bneck4_make_plan(long M, int K1, int N2, bool has_out, int nprob, Bneck4Plan& p, int nsum = 1) {
  auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
  // the fused sibling sum pays from a few hundred rows on: below, the separate summation pass costs 2 us and the n problems fill more of
  // the chip as n workgroups (M = 77: 8.4 against 7.3 us, profiles/r06_c34_lcbench_sibling_sum.log)
  if (nsum > 1 && (M < 256 || nsum > 3)) return false;  // (four problems: 25.6 against 17.3 us on the 1024 x 1280 layer -- five slices repeat four reduce stages)
  p.mi = M >= 8192 ? 2 : 1;
  p.nw = M >= 8192 ? 4 : 8;
  const long rows = cdiv(M, 16 * p.mi) * nprob;
  const long npairs = cdiv(N2, 32);
  const int d2max = 32 / p.nw;
  long ns = 1;
  if (has_out) {
    ns = 256 / rows;
    if (ns > cdiv(npairs, p.nw)) ns = cdiv(npairs, p.nw);
    if (ns < 1) ns = 1;
    if (cdiv(cdiv(npairs, ns), p.nw) > d2max) ns = cdiv(npairs, (long)p.nw * d2max);
    if (ns > 65535) return false;
  }
  p.ns = (int)ns;
  p.D2 = has_out ? (int)cdiv(cdiv(npairs, ns), p.nw) : 0;
  const int steps = (int)cdiv(cdiv(K1, 32), p.nw);
  const int budget = rows * ns > 256 ? 78 * 1024 : 158 * 1024;
  int D = steps < 8 ? steps : 8;
  while (D > 2 && bneck4_lds_bytes(p.nw, p.mi, D, p.D2, nsum) > budget) --D;
  if (D < 1) D = 1;
  // several problems per workgroup (bneck4_sum_kernel): their column pairs all wait in the LDS -- more column slices until they fit
  while (nsum > 1 && has_out && bneck4_lds_bytes(p.nw, p.mi, D, p.D2, nsum) > 160 * 1024 && p.D2 > 1) {
    ++ns;
    p.ns = (int)ns;
    p.D2 = (int)cdiv(cdiv(npairs, ns), p.nw);
  }
  p.D = D;
  p.lds = bneck4_lds_bytes(p.nw, p.mi, p.D, p.D2, nsum);
  return p.lds <= 160 * 1024;
}

// chunk permutation of a [16 rows][4 x 16 bytes] piece (kron4.h): stored chunk = logical chunk ^ phi(row >> 2), phi = (0, 3, 2, 1)
__device__ __forceinline__ int b4_swz(int row, int chunk) { return chunk ^ ((0 - (row >> 2)) & 3); }

// `probs[0 .. n)`: n problems of ONE shape (M, K1, R, N2, lda, ldo, D, D2 of probs[0] hold for all) run by the same workgroup one after
// the other.  n == 1 is the plain launch.  n > 1 (bneck4_sum_kernel): the gradient of a tensor that n sibling projections read --
// dx = sum_z dt_z . down_z -- as ONE expand stage over the n `mid` tiles: every problem keeps its own reduce stage and its own `mid`
// (the factor gradients need it), the stage-2 results are added in fp32 registers and stored once to `out_sum`: no dx_z reaches HBM
// and the summation pass (lyc_sum_rows: 4.7 us per set, 140 sets per SDXL LoCon step) is gone (kron4_sum_kernel's idea, kron4.h).
template <typename T, int NW, int MI, bool FT, int NMAX>
__device__ __forceinline__ void bneck4_body(const Bneck4Args* probs, const int n, void* out, const int bx, const int by, const int nby) {
  extern __shared__ __attribute__((aligned(1024))) char b4_smem[];
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;
  constexpr int SB = (MI + 2) * 1024;  // bytes of one k step in a wave's ring
  constexpr int C1 = MI + 2;           // DMA operations per k step
  constexpr int RP = 20;               // row pitch of mid in LDS (floats)
  const Bneck4Args& a = probs[0];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = a.D, D2 = a.D2, K1 = a.K1, KS = a.KS, R = a.R, N2 = a.N2;
  char* const ring1 = b4_smem + wave * (D * SB);
  char* const ring2 = b4_smem + NW * (D * SB) + wave * (n * D2 * 2048);  // [problem][pair]
  float* const red = reinterpret_cast<float*>(b4_smem + NW * (D * SB + n * D2 * 2048));
  float* const mids = red + NW * MI * 256;                                // [problem][16 MI][RP] (512 floats per problem and MI)
  const int m0 = bx * (16 * MI);
  LYC_TRACE_DECL;
  LYC_STAMP(0);

  // ---- stage-2 operands: the wave's column pairs of every problem, requested first -----------------------------------------------
  const int npairs_all = (N2 + 31) >> 5;
  const int tper = (npairs_all + nby - 1) / nby;
  const int pbeg = by * tper;
  const int pend = pbeg + tper < npairs_all ? pbeg + tper : npairs_all;
  const int np2 = (out != nullptr && pend > pbeg + wave) ? (pend - pbeg - wave + NW - 1) / NW : 0;  // pairs of this wave (<= D2)
  LYC_STAMP(10);
  {
    // pair p = columns [32 p, 32 p + 32): MFMA row i of its tile e holds column 32 p + 8 (i >> 2) + 4 e + (i & 3)
    unsigned v2[2];
    if constexpr (!FT) {  // piece = [16 MFMA rows i][16 n fp32], chunk-permuted; source row = the column, 4 n per chunk
      const int i = lane >> 2, c = b4_swz(i, lane & 3);
#pragma unroll
      for (int e = 0; e < 2; ++e) v2[e] = 4 * c < R ? (unsigned)(8 * (i >> 2) + 4 * e + (i & 3)) * (unsigned)R * 4u + (unsigned)c * 16u : K4_OOB;
    } else {              // piece = [16 n rows, placed at rho(n)][16 columns fp32]; chunk j = 4 consecutive source columns
      const int pos = lane >> 2, nn = (pos & ~3) | ((pos ^ (pos >> 2)) & 3), j = lane & 3;
#pragma unroll
      for (int e = 0; e < 2; ++e) v2[e] = (unsigned)nn * (unsigned)N2 * 4u + (unsigned)(8 * j + 4 * e) * 4u;  // n >= R: beyond f2_bytes
    }
    for (int z = 0; z < n; ++z) {
      const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(probs[z].F2), 0, (int)a.f2_bytes, K4_RSRC_FLAGS);
      for (int q = 0; q < np2; ++q) {
        const int p = pbeg + wave + NW * q;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned vo = v2[e];
          if constexpr (FT) {
            if (32 * p + 8 * (lane & 3) + 4 * e >= N2) vo = K4_OOB;  // N2 % 4 == 0: a chunk is all in or all out (the next row must not leak in)
          }
          // (the hardware's range check covers the VGPR offset only, not the SGPR offset: whatever must be refused by the descriptor
          //  -- !FT: columns >= N2 -- has to be part of the former)
          const unsigned so = FT ? (unsigned)p * 128u : 0u;
          if constexpr (!FT) {
            if (vo != K4_OOB) vo += (unsigned)p * 32u * (unsigned)R * 4u;  // columns >= N2: beyond f2_bytes
          }
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (k4_lds_ptr)(ring2 + (z * D2 + q) * 2048 + e * 1024), 16, (int)vo, (int)so, 0, 0);
        }
      }
    }
  }
  LYC_STAMP(11);

  // ---- stage 1: this wave's k steps s = wave, wave + NW, ... of every problem ---------------------------------------------------------
  const unsigned lda2 = (unsigned)a.lda * 2u;
  const int klast = K1 - 32 * (KS - 1);  // columns of the last k step: 8, 16, 24 or 32
  unsigned vA[MI], vAl[MI], v1[2], v1l[2];
  {
    const int xr = lane >> 2, xc = b4_swz(xr, lane & 3);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      vA[mi] = (unsigned)(mi * 16 + xr) * lda2 + (unsigned)xc * 16u;  // rows >= M: beyond a_bytes
      vAl[mi] = 8 * xc + 8 <= klast ? vA[mi] : K4_OOB;                 // beyond K1: zeros, not the next row
    }
    if constexpr (!FT) {  // piece pi = [16 n][k = 16 pi .. 16 pi + 15], chunk-permuted
      const int nn = lane >> 2, c = b4_swz(nn, lane & 3);
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        v1[pi] = (unsigned)nn * (unsigned)K1 * 4u + (unsigned)(16 * pi + 4 * c) * 4u;  // n >= R: beyond f1_bytes
        v1l[pi] = 16 * pi + 4 * c + 4 <= klast ? v1[pi] : K4_OOB;
      }
    } else {  // piece pi = [16 k rows r, placed at pos(pi, r)][16 n]; pos = (r & ~3) | ((r & 3) ^ ((r >> 3) & 1) ^ 2 pi)
      const int pos = lane >> 2, c = lane & 3;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int r = (pos & ~3) | ((pos & 3) ^ ((pos >> 3) & 1) ^ (2 * pi));
        v1[pi] = 4 * c < R ? (unsigned)(16 * pi + r) * (unsigned)R * 4u + (unsigned)c * 16u : K4_OOB;  // rows k >= K1: beyond f1_bytes
        v1l[pi] = v1[pi];
      }
    }
  }
  const int nst = wave < KS ? (KS - wave + NW - 1) / NW : 0;  // k steps of this wave
  const int npro = nst < D ? nst : D;
  __amdgpu_buffer_rsrc_t rsA, rs1;
  auto bind = [&](int z) {
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(probs[z].A)) + (size_t)(unsigned)m0 * lda2, 0,
                                            (int)(a.a_bytes - (unsigned)m0 * lda2), K4_RSRC_FLAGS);
    rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(probs[z].F1), 0, (int)a.f1_bytes, K4_RSRC_FLAGS);
  };
  auto issue1 = [&](int j, int slot) {
    const int s = wave + NW * j;
    const bool last = s == KS - 1;
    char* dst = ring1 + slot * SB;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (k4_lds_ptr)(dst + mi * 1024), 16, (B4_ABL & 1) ? (int)K4_OOB : (int)(last ? vAl[mi] : vA[mi]), s * 64, 0, 0);
    // FT: the rows k >= K1 of the last step must be refused by the descriptor -> the step's offset travels in the VGPR offset (the
    // range check does not see the SGPR offset); !FT: the lane mask v1l covers the tail, the step's offset may stay scalar
    const unsigned so = FT ? 0u : (unsigned)s * 128u;
    const unsigned vstep = FT ? (unsigned)s * 32u * (unsigned)R * 4u : 0u;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      unsigned vo = last ? v1l[pi] : v1[pi];
      if (FT && vo != K4_OOB) vo += vstep;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (k4_lds_ptr)(dst + (MI + pi) * 1024), 16, (B4_ABL & 2) ? (int)K4_OOB : (int)vo, (int)so, 0, 0);
    }
  };
  bind(0);
  for (int j = 0; j < npro; ++j) issue1(j, j);
  LYC_STAMP(1);

  const unsigned rdA = (unsigned)(li * 4 + b4_swz(li, g)) * 16u;
  constexpr int TR = (MI * 256 + NW * 64 - 1) / (NW * 64);  // elements of a mid tile per thread (2 for 4 waves x 32 rows)
  float keep[NMAX][TR];  // this thread's elements of every problem's mid (stored to HBM at the very end: a store in the middle of the
                     // kernel would stand in the vmcnt queue between a counted wait and the DMA group it waits for)
  for (int z = 0; z < n; ++z) {
    f32x4 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = zero4();
    int slot = 0;
    for (int j = 0; j < nst; ++j) {
      int newest = j + D - 1;
      if (newest > nst - 1) newest = nst - 1;
      k4_wait_groups<C1>(newest - j);  // at most the steps after j outstanding (the stage-2 pieces are older than all of them)
      asm volatile("" ::: "memory");
      if (j == 0) LYC_STAMP(12);
      const char* sp = ring1 + slot * SB;
      if constexpr ((B4_ABL & 4) != 0) { slot = slot + 1 == D ? 0 : slot + 1; if (j + D < nst) issue1(j + D, slot); continue; }
      F8 af[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const F8*>(sp + mi * 1024 + rdA);
      f32x4 f0, f1;
      if constexpr (!FT) {  // lane (n = li, k = 8 g .. 8 g + 7): piece g >> 1, chunks 2 (g & 1), 2 (g & 1) + 1
        const char* fp = sp + (MI + (g >> 1)) * 1024 + li * 64;
        f0 = *reinterpret_cast<const f32x4*>(fp + b4_swz(li, 2 * (g & 1)) * 16);
        f1 = *reinterpret_cast<const f32x4*>(fp + b4_swz(li, 2 * (g & 1) + 1) * 16);
      } else {  // row r = 8 (g & 1) + e of piece g >> 1 lies at pos = 8 (g & 1) + (e & ~3) + ((e & 3) ^ g); word li
        const char* fp = sp + (MI + (g >> 1)) * 1024 + (g & 1) * 512 + li * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f0[e] = *reinterpret_cast<const float*>(fp + (e ^ g) * 64);
          f1[e] = *reinterpret_cast<const float*>(fp + (4 + (e ^ g)) * 64);
        }
      }
      F8 bh, bl;
      lr_split8<T>(f0, f1, bh, bl);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi] = TT<T>::mma(af[mi], bh, acc[mi]);
        acc[mi] = TT<T>::mma(af[mi], bl, acc[mi]);
      }
      if (j + D < nst) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of this slot have returned before its next DMA can land
        issue1(j + D, slot);
      }
      slot = slot + 1 == D ? 0 : slot + 1;
    }
    if (z + 1 < n) {  // the next problem's first steps travel while this one is summed (this wave is past every read of its ring)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bind(z + 1);
      for (int j = 0; j < npro; ++j) issue1(j, j);
    } else {
      k4_wait_vm<0>();  // (a wave without k steps has not waited for its stage-2 pieces yet)
    }
    if (z == 0) LYC_STAMP(2);
    // ---- cross-wave sum -> mid (LDS; HBM at the end) ---------------------------------------------------------------------------------
    // raw barriers: __syncthreads() would wait vmcnt(0), i.e. for the DMAs just issued
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) *reinterpret_cast<f32x4*>(red + (wave * MI + mi) * 256 + lane * 4) = acc[mi];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
      const int e = tid + tr * (NW * 64);
      keep[z][tr] = 0.f;
      if (e >= MI * 256) continue;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * MI * 256 + e];
      s *= probs[z].alpha1;
      const int t = e >> 8, l = (e >> 2) & 63, q = e & 3;  // accumulator element: column l & 15, row 4 (l >> 4) + q
      mids[z * (MI * 512) + (16 * t + 4 * (l >> 4) + q) * RP + (l & 15)] = s;
      keep[z][tr] = s;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  LYC_STAMP(3);

  // ---- stage 2: out^T tile = sum_z F2_z tile . mid_z^T, pairs of column tiles (the pairs were requested before stage 1 and waited
  //      for at its end: no load, no wait in this loop) ------------------------------------------------------------------------------------
  if (out != nullptr) {
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)a.out_bytes, K4_RSRC_FLAGS);
    unsigned rofs[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + 16 * mi + li;
      rofs[mi] = m < a.M ? (unsigned)m * (unsigned)a.ldo * 2u : K4_OOB;
    }
    for (int q = 0; q < np2; ++q) {
      const int p = pbeg + wave + NW * q;
      f32x4 ys[MI][2];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) ys[mi][0] = ys[mi][1] = zero4();
      for (int z = 0; z < n; ++z) {
        const char* tp = ring2 + (z * D2 + q) * 2048;
        F4 ah[2], al[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f32x4 fv;
          if constexpr (!FT) {  // lane (i = li, n = 4 g .. 4 g + 3)
            fv = *reinterpret_cast<const f32x4*>(tp + e * 1024 + (li * 4 + b4_swz(li, g)) * 16);
          } else {              // row n = 4 g + ee lies at rho(n) = 4 g + (ee ^ g); word li
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) fv[ee] = *reinterpret_cast<const float*>(tp + e * 1024 + (4 * g + (ee ^ g)) * 64 + li * 4);
          }
          lr_split4<T>(fv, ah[e], al[e]);
        }
        const float a2 = probs[z].alpha2;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          F4 mh, ml;  // B operand: lane (m = li, n = 4 g .. 4 g + 3)
          lr_split4<T>(*reinterpret_cast<const f32x4*>(mids + z * (MI * 512) + (16 * mi + li) * RP + 4 * g), mh, ml);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x4 y = Mma16<T>::mma(ah[e], mh, zero4());
            y = Mma16<T>::mma(al[e], mh, y);
            y = Mma16<T>::mma(ah[e], ml, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) ys[mi][e][k] += a2 * y[k];
          }
        }
      }
      const int gn = 32 * p + 8 * g;  // this lane's 8 columns (N2 % 8 == 0: all in or all out)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        T o[8] __attribute__((aligned(16)));
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < 4; ++k) o[4 * e + k] = TT<T>::from_f(ys[mi][e][k]);
        const unsigned off = gn < N2 ? rofs[mi] + (unsigned)gn * 2u : K4_OOB;  // out-of-bounds stores are dropped
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(o), rsO, (int)off, 0, 0);
      }
    }
  }
  // ---- mid -> HBM (the backward pass / the factor gradients read it) -------------------------------------------------------------------
  if (by == 0) {
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
      const int e = tid + tr * (NW * 64);
      const int t = e >> 8, l = (e >> 2) & 63, q = e & 3;
      const int m = 16 * t + 4 * (l >> 4) + q, nn = l & 15;
      if (e < MI * 256 && m0 + m < a.M && nn < R)
        for (int z = 0; z < n; ++z)
          if (probs[z].mid != nullptr) probs[z].mid[(long)(m0 + m) * R + nn] = keep[z][tr];
    }
  }
  LYC_STAMP(4);
  LYC_TRACE_FLUSH();
}

template <typename T, int NW, int MI, bool FT>
__global__ __launch_bounds__(NW * 64) void bneck4_kernel(Bneck4Args a) {
  bneck4_body<T, NW, MI, FT, 1>(&a, 1, a.out, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// Several problems of one shape in one launch (sibling projections: bneck_group_kernel's role): blockIdx.z selects the problem.
constexpr int BNECK4_GROUP_MAX = 4;
struct Bneck4GroupArgs {
  int n;
  Bneck4Args p[BNECK4_GROUP_MAX];
};
template <typename T, int NW, int MI, bool FT>
__global__ __launch_bounds__(NW * 64) void bneck4_group_kernel(Bneck4GroupArgs ga) {
  bneck4_body<T, NW, MI, FT, 1>(&ga.p[blockIdx.z], 1, ga.p[blockIdx.z].out, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
// The n problems in ONE workgroup, their stage-2 results summed (see bneck4_body): `out_sum` [M, N2] replaces the n `out` tensors.
template <typename T, int NW, int MI, bool FT>
__global__ __launch_bounds__(NW * 64) void bneck4_sum_kernel(Bneck4GroupArgs ga, void* out_sum) {
  bneck4_body<T, NW, MI, FT, BNECK4_GROUP_MAX>(ga.p, ga.n, out_sum, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

}  // namespace lyc
