// kron_dw2s.h -- LoKr w2-gradient kernel for 16-bit activations, gfx950 ("s" = streaming waves).
//
//   dW2[i, j] += alpha * sum_{r = (m, s)} Q[r, i] * Z[r, j],     Z[(m, s), j] = sum_t W[s, t] * P[(m, t), j]
//
// A "TN" GEMM whose K dimension is the M * G flat rows (thousands) and whose output is the small w2 matrix.  Measured
// on MI355X (benchmarks/atomic_bench.cpp): fp32 atomics retire at ~300 elements/ns chip-wide (whole lines; fewer when scattered), so the old design (big
// output tiles x ~100 row slabs = millions of atomics) spent 3/4 of its time in the atomic units.  This version:
//   * small output tiles (16 MI x 16 NJ) and FEW row slabs: the split factor is chosen by the host so that the atomic
//     traffic stays below ~0.4 M elements per launch; with one slab the tile is added with plain loads/stores;
//   * the 4 waves of a workgroup split the ROWS of the slab (32-row steps, round robin) and each keeps a full copy of the
//     output tile in registers: no barrier in the main loop, no shared staging -- each wave transposes its own 32-row
//     Q / P blocks through a private LDS tile (4 x 8 register blocks, v_perm_b32, 8-byte writes);
//   * the G x G mix runs on the matrix cores (v_mfma_f32_16x16x16, (I (x) W) hi/lo x P) and its accumulator layout is
//     fed straight back as the B operand of the main v_mfma_f32_16x16x32 with a permuted K index (see below);
//   * one cross-wave reduction through LDS at the end.
// The extra z-slice of the grid reduces the per-workgroup w1-gradient partials written by kron3_kernel (fixed order,
// no same-address atomic storm: 320 one-line atomics cost ~10 us on this chip).
#pragma once
#include "kron3.h"

namespace lyc {

struct KronDw2sArgs {
  const void* Q;    // [M, G * I]   exact operand, contributes the output rows i
  const void* P;    // [M, G * J]   operand that is mixed with W, contributes the output columns j
  const float* W;   // element (s, t) at s * ws + t * wt
  float* out;       // element (i, j) at i * os + j   (j contiguous)
  long M;
  int G, I, J;
  long ws, wt, os;
  long rows_per_block;  // flat (m, s) rows handled by one workgroup (multiple of 32)
  int nsplit;           // row slabs
  int tiles_i, tiles_j; // output tiles; the grid is 1-D: round_up(tiles_i * tiles_j * nsplit, 8) (+ dw1_red reducers)
  float alpha;
  int force_atomic;     // != 0: `out` / `dw1` may be shared with another problem of the same launch (grouped launches)
  // w1-gradient partial reduction (runs in grid slice z == nsplit; nullptr = nothing to do)
  const float* dw1_ws;  // [dw1_nblk][dw1_n] partials, already in dw1 memory order
  float* dw1;
  int dw1_nblk, dw1_n, dw1_red;  // partial blocks, elements per partial, reducer workgroups
  // implicit Conv2d (GATHER kernels): the J columns are (tap, Jt columns per tap); the P row of destination (= output)
  // pixel row r and tap t is the source (= input) pixel row given by gat (mode 1); P rows hold Jt elements per group
  KronGather gat;
  int Jt;
};

constexpr int DS_LD = 36;  // LDS row pitch (elements) of the transposed [col][32 rows] tiles: 18 dwords

// per wave: two private transposed tiles [16 (MI + NJ) cols][DS_LD] (double buffer); afterwards the same memory holds the
// cross-wave reduction image [4 waves][<= 8 tiles][256] fp32
template <int MI, int NJ>
__host__ __device__ constexpr int kron_dw2s_lds_bytes() {
  const int stage = NWAVES * 2 * 16 * (MI + NJ) * DS_LD * 2;
  const int tiles = MI * NJ < 8 ? MI * NJ : 8;
  const int red = NWAVES * tiles * 256 * 4;
  return stage > red ? stage : red;
}

__device__ __forceinline__ void dw1_reduce_role(const KronDw2sArgs& a, int r, float* lds) {
  // reducer r of dw1_red sums partial blocks r, r + dw1_red, ...; 256 threads = (256 / n) lanes per element
  const int n = a.dw1_n;  // G * G <= 256, a power of two
  const int tid = threadIdx.x;
  const int e = tid % n, part = tid / n, nparts = NTHREADS / n;
  float s = 0.f;
  for (int b = r + a.dw1_red * part; b < a.dw1_nblk; b += a.dw1_red * nparts) s += a.dw1_ws[(long)b * n + e];
  lds[tid] = s;
  __syncthreads();
  if (tid < n) {
    float t = 0.f;
    for (int p = 0; p < nparts; ++p) t += lds[p * n + tid];
    if (a.dw1_red == 1 && !a.force_atomic)
      a.dw1[tid] += t;
    else
      __hip_atomic_fetch_add(a.dw1 + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// U = 32-row steps per prefetch group: the loads of two groups (2 * U * NB * 4 x 16 bytes per lane) are in flight while a
// group is processed -- with ~1 wave per SIMD (small problems) only instruction-level parallelism hides the HBM latency.
// `b_`: index of this workgroup in the launch; `smem`: kron_dw2s_lds_bytes<MI, NJ>() bytes, 16-byte aligned.
// (A device function: round 2 ran it and kron3_body as two roles of ONE grid -- the whole LoKr backward of a layer in a
// single launch.  Measured on the SDXL step: 19.3 ms against 17.5 ms for the two launches back to back, twice, A/B in one
// process (profiles/r02_fused_bwd_ab.txt): the two kernels are not idle-latency-bound but share the per-CU load path, and
// the merged kernel runs at the register count of the larger role.  Removed again; the split stays for such experiments.)
template <typename T, int MI, int NJ, int U, bool GATHER>
__device__ __forceinline__ void kron_dw2s_body(const KronDw2sArgs& a, char* smem, const int b_) {
  constexpr int TI = 16 * MI, TJ = 16 * NJ, NC = TI + TJ;
  constexpr int NB = (NC + 63) / 64;  // 4 x 8 blocks per lane per 32-row step
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the row bookkeeping in SGPRs
  LYC_TRACE_DECL;
  LYC_STAMP(0);
  // Workgroup -> (output tile, row slab).  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only),
  // and each XCD has its own L2.  The work items are put in an order in which neighbours share operand data and every XCD
  // is dealt one CONTIGUOUS eighth of that order:
  //   >= 8 slabs: slab-major -- all tiles of a slab re-read the slab's Q / P rows from one L2;
  //   <  8 slabs: Q-column-major -- the tiles of a column group of Q (the big operand) sit on one XCD.
  // Measured without it (rocprofv3 FETCH_SIZE): 8x the algorithmic bytes crossing the fabric.
  const int ntile = a.tiles_i * a.tiles_j;
  const int nwork = ntile * a.nsplit;
  const int per = (nwork + 7) >> 3;
  if (b_ >= per * 8) {  // w1-gradient reducer workgroups ride at the end of the grid
    const int r = b_ - per * 8;
    if (a.dw1_ws != nullptr && r < a.dw1_red) dw1_reduce_role(a, r, reinterpret_cast<float*>(smem));
    return;
  }
  const int o = (b_ & 7) * per + (b_ >> 3);
  if (o >= nwork) return;  // padding of the last XCD's share
  int slab, tile_x, tile_y;
  if (a.nsplit >= 8) {
    slab = o / ntile;
    const int tile = o - slab * ntile;
    tile_y = tile / a.tiles_i;
    tile_x = tile - tile_y * a.tiles_i;
  } else {
    tile_x = o / (a.tiles_j * a.nsplit);
    const int rem = o - tile_x * (a.tiles_j * a.nsplit);
    tile_y = rem / a.nsplit;
    slab = rem - tile_y * a.nsplit;
  }

  const T* Q = static_cast<const T*>(a.Q);
  const T* P = static_cast<const T*>(a.P);
  const int G = a.G;
  const int lg = 31 - __builtin_clz((unsigned)G);
  const long i0 = (long)tile_x * TI, j0 = (long)tile_y * TJ;
  const long rows_total = a.M << lg;
  const long rbeg = (long)slab * a.rows_per_block;
  long rend = rbeg + a.rows_per_block;
  if (rend > rows_total) rend = rows_total;
  T* tile0 = reinterpret_cast<T*>(smem) + wave * 2 * NC * DS_LD;  // this wave's private [NC cols][32 rows] tiles
  T* tile1 = tile0 + NC * DS_LD;

  // this lane's 4 x 8 blocks: block b < TI: Q columns 8*(b % (TI/8)), rows 4*(b / (TI/8)); else P likewise.
  // Column groups beyond I / J read column 0 instead: they only feed output rows / columns that are never stored.
  const T* bptr[NB];  // address of (row 0 + brow, column) of the block's operand
  long bld[NB];       // row pitch (elements)
  int bcol[NB], brow[NB], btap[NB];
  bool bgat[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int b = lane + 64 * it;
    const bool isq = b < TI;
    const int bb = isq ? b : (b < NC ? b - TI : 0);
    const int ncg = isq ? TI / 8 : TJ / 8;
    const int cg = bb % ncg, rg = bb / ncg;
    const long gc = (isq ? i0 : j0) + 8 * cg;
    const long ntot = isq ? a.I : a.J;
    long csrc = gc < ntot ? gc : 0;  // I, J % 8 == 0: a block column group is all in or all out
    bld[it] = ntot;
    bgat[it] = false;
    btap[it] = 0;
    if constexpr (GATHER) {
      if (!isq) {  // P column (tap, v): rows are gathered per tap, Jt elements per (pixel, group) row
        btap[it] = (int)(csrc / a.Jt);
        csrc -= (long)btap[it] * a.Jt;
        bld[it] = a.Jt;
        bgat[it] = true;
      }
    }
    bcol[it] = (isq ? 0 : TI) + 8 * cg;
    brow[it] = 4 * rg;
    bptr[it] = (isq ? Q : P) + csrc + (GATHER ? 0 : (long)brow[it] * bld[it]);
  }

  // one 32-row step: global -> registers.  Steps are requested in increasing row order (rbeg + 32 wave, + 128, ...), so
  // the non-gather path keeps one running byte pointer per block and advances it by a precomputed increment: steps that
  // lie completely inside the slab (all but possibly the last) cost 2 VALU adds per 16-byte load and no selects.
  const char* pcur[NB];
  long pinc[NB], prow[NB];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    prow[it] = bld[it] * (long)sizeof(T);
    pinc[it] = 32 * NWAVES * prow[it];
    pcur[it] = reinterpret_cast<const char*>(bptr[it]) + (rbeg + 32 * wave) * prow[it];
  }
  auto load_step = [&](u32x4 (&raw)[NB][4], long r0) {
    if constexpr (!GATHER) {
      if (r0 + 32 <= rend) {
#pragma unroll
        for (int it = 0; it < NB; ++it) {
#pragma unroll
          for (int j = 0; j < 4; ++j) raw[it][j] = *reinterpret_cast<const u32x4*>(pcur[it] + j * prow[it]);
          pcur[it] += pinc[it];
        }
        return;
      }
#pragma unroll
      for (int it = 0; it < NB; ++it) pcur[it] += pinc[it];
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      long rbase = r0 + brow[it];  // first of the block's 4 flat rows; the gather maps it to another pixel's rows
      bool gok = true;
      const T* src = bptr[it];
      if constexpr (GATHER) {
        if (bgat[it]) {  // the 4 rows share one pixel (G % 4 == 0, rbase % 4 == 0)
          const long pix = (rbase < rend ? rbase : rbeg) >> lg;
          const int hw = a.gat.Hd * a.gat.Wd;
          const int pb = (int)(pix / hw);
          const int rem = (int)(pix - (long)pb * hw);
          const int hd = rem / a.gat.Wd, wd = rem - hd * a.gat.Wd;
          const int ti = btap[it] / a.gat.kw, tj = btap[it] - ti * a.gat.kw;
          const int hs = hd * a.gat.sh - a.gat.ph + ti * a.gat.dh, ws = wd * a.gat.sw - a.gat.pw + tj * a.gat.dw;
          gok = hs >= 0 && hs < a.gat.Hs && ws >= 0 && ws < a.gat.Ws;
          const long spix = gok ? ((long)pb * a.gat.Hs + hs) * a.gat.Ws + ws : 0;
          rbase = (spix << lg) + (rbase & (G - 1));
        }
      } else {
        src -= (long)brow[it] * bld[it];  // bptr already includes the block's row offset
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = gok && (r0 + brow[it] + j) < rend;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (ok ? rbase + j : 0) * bld[it]);
        const u32x4 z = {0u, 0u, 0u, 0u};
        raw[it][j] = ok ? v : z;
      }
    }
  };
  auto store_step = [&](const u32x4 (&raw)[NB][4], T* tile) {
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      if (lane + 64 * it < NC) {
        T* dst = tile + bcol[it] * DS_LD + brow[it];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          u32x2 lo, hi;
          lo[0] = __builtin_amdgcn_perm(raw[it][1][w], raw[it][0][w], 0x05040100u);
          lo[1] = __builtin_amdgcn_perm(raw[it][3][w], raw[it][2][w], 0x05040100u);
          hi[0] = __builtin_amdgcn_perm(raw[it][1][w], raw[it][0][w], 0x07060302u);
          hi[1] = __builtin_amdgcn_perm(raw[it][3][w], raw[it][2][w], 0x07060302u);
          *reinterpret_cast<u32x2*>(dst + (2 * w) * DS_LD) = lo;
          *reinterpret_cast<u32x2*>(dst + (2 * w + 1) * DS_LD) = hi;
        }
      }
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = zero4();

  // this wave's 32-row steps are rbeg + 32 * (wave + NWAVES * t); a group is U consecutive steps of the wave
  constexpr long STEP = 32 * NWAVES, GROUP = STEP * U;
  u32x4 rawA[U][NB][4], rawB[U][NB][4];
  auto load_group = [&](u32x4 (&raw)[U][NB][4], long r0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r0 + STEP * u < rend) load_step(raw[u], r0 + STEP * u);
  };
  long r0 = rbeg + 32 * wave;
  LYC_STAMP(1);
  // The four W values are requested BEFORE the row groups: vector-memory results return in issue order, so behind the
  // two preloaded groups they would only become usable once every preloaded row had arrived, and the first step's MFMAs
  // (which need W and only the first 32 rows) would wait for all of it.
  float wraw[4];
  {
    const int s_ = li & (G - 1);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) wraw[jj] = a.W[s_ * a.ws + ((4 * g + jj) & (G - 1)) * a.wt];
  }
  LR3_FENCE();
  load_group(rawA, r0);
  load_group(rawB, r0 + GROUP);
  LYC_STAMP(2);

  // mix operand (I (x) W) for one 16x16 block: lane (i = li, g) holds k = 4g .. 4g+3
  F4 a2h, a2l;
  {
    const int mi_ = li >> lg;
    T h[4], l[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kk = 4 * g + jj;
      split_f<T>(((kk >> lg) == mi_) ? wraw[jj] : 0.f, h[jj], l[jj]);
    }
    a2h = *reinterpret_cast<F4*>(h);
    a2l = *reinterpret_cast<F4*>(l);
  }

  auto compute_step = [&](const T* tile) {
    // mix: Z blocks of the NJ column blocks, both 16-row halves, kept as hi/lo B fragments of the main MFMA.
    // K index permutation of the main MFMA: element e < 4 of lane group g <-> row 4g+e, e >= 4 <-> row 16 + 4g + e - 4
    // (the accumulator layout of the two mix results); the A operand below is read with the same permutation.
    F8 zh[NJ], zl[NJ];
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      T hh[8], ll[8];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const F4 pf = *reinterpret_cast<const F4*>(tile + (TI + 16 * nj + li) * DS_LD + 16 * rb + 4 * g);
        f32x4 z = zero4();
        z = Mma16<T>::mma(a2h, pf, z);
        z = Mma16<T>::mma(a2l, pf, z);
#pragma unroll
        for (int e = 0; e < 4; ++e) split_f<T>(z[e], hh[4 * rb + e], ll[4 * rb + e]);
      }
      zh[nj] = *reinterpret_cast<F8*>(hh);
      zl[nj] = *reinterpret_cast<F8*>(ll);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const T* qrow = tile + (16 * mi + li) * DS_LD + 4 * g;
      const u32x2 a0 = *reinterpret_cast<const u32x2*>(qrow);
      const u32x2 a1 = *reinterpret_cast<const u32x2*>(qrow + 16);
      const u32x4 av = {a0[0], a0[1], a1[0], a1[1]};
      const F8 af = *reinterpret_cast<const F8*>(&av);
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = TT<T>::mma(af, zh[nj], acc[mi][nj]);
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = TT<T>::mma(af, zl[nj], acc[mi][nj]);
    }
  };
  // One group: the transposed image of step u+1 is written (other tile) before the MFMAs of step u are issued, so the
  // VALU transposes / LDS writes of the next step overlap the matrix work of this one.  `first` of the next group comes
  // from the other register set.
  T* cur = tile0;  // tile holding the transposed image of the step about to be computed
  T* nxt = tile1;
  auto process_group = [&](const u32x4 (&raw)[U][NB][4], const u32x4 (&next0)[NB][4], long r0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + STEP * u >= rend) break;
      if (r0 + STEP * (u + 1) < rend) {
        if (u + 1 < U) store_step(raw[u + 1 < U ? u + 1 : 0], nxt);
        else store_step(next0, nxt);
      }
      compute_step(cur);
      T* t = cur;
      cur = nxt;
      nxt = t;
    }
  };
  LYC_STAMP(3);
  if (r0 < rend) store_step(rawA[0], cur);
  while (r0 < rend) {
    process_group(rawA, rawB[0], r0);
    load_group(rawA, r0 + 2 * GROUP);
    r0 += GROUP;
    if (r0 >= rend) break;
    process_group(rawB, rawA[0], r0);
    load_group(rawB, r0 + 2 * GROUP);
    r0 += GROUP;
  }

  // cross-wave reduction through LDS: every wave publishes its tiles, wave w sums and stores tiles t = w (mod 4)
  float* red = reinterpret_cast<float*>(smem);
  constexpr int NT = MI * NJ, TB = NT < 8 ? NT : 8;  // tiles per batch (LDS budget); NT need not be a multiple of TB
  const bool plain = a.nsplit == 1 && !a.force_atomic;
  LYC_STAMP(4);
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TB) {
    __syncthreads();  // the staging tiles (or the previous batch) are dead
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      if (t0 + t < NT) {
        const int mi = (t0 + t) / NJ, nj = (t0 + t) % NJ;
        *reinterpret_cast<f32x4*>(red + (wave * TB + t) * 256 + lane * 4) = acc[mi][nj];
      }
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < (TB + NWAVES - 1) / NWAVES; ++tt) {
      const int t = tt * NWAVES + wave;  // this wave's tile of the batch
      if (t >= TB || t0 + t >= NT) continue;
      const int mi = (t0 + t) / NJ, nj = (t0 + t) % NJ;
      f32x4 s = zero4();
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) s += *reinterpret_cast<const f32x4*>(red + (w * TB + t) * 256 + lane * 4);
      const long gj = j0 + 16 * nj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long gi = i0 + 16 * mi + 4 * g + r;
        if (gi < a.I && gj < a.J) {
          float* dst = a.out + gi * a.os + gj;
          if (plain)
            *dst += a.alpha * s[r];
          else
            __hip_atomic_fetch_add(dst, a.alpha * s[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  LYC_STAMP(6);
  LYC_TRACE_FLUSH();
}

template <typename T, int MI, int NJ, int U, bool GATHER>
__global__ __launch_bounds__(NTHREADS) void kron_dw2s_kernel(KronDw2sArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[kron_dw2s_lds_bytes<MI, NJ>()];
  kron_dw2s_body<T, MI, NJ, U, GATHER>(a, smem, (int)blockIdx.x);
}

// ---- grouped launch: the weight gradients of up to DW2G_MAX layers in ONE grid -----------------------------------------
// dW1 / dW2 of a layer are consumed by the optimizer only, so nothing in the backward pass waits for them: the host keeps
// (g, x) of the finished layers alive (288 GB of HBM: a few hundred MB) and hands a batch of them to this kernel.  A single
// layer's launch is ~500 workgroups of ~10 k cycles each -- one resident round whose time is the serial chain of one
// workgroup plus the launch overhead; a batch is thousands of workgroups, so the chains of co-resident workgroups overlap,
// the split-K factor per problem (and with it the atomic traffic) drops, and 24 launches become one.
// The problem descriptors travel by value in the kernel arguments (no device-side table to keep coherent, legal inside
// hipGraph capture).  Every problem's workgroup count is a multiple of 8, so `index % 8` is still the XCD.
constexpr int DW2G_MAX = 24;
struct KronDw2sItem {
  const void* Q;
  const void* P;
  const float* W;
  float* out;
  const float* dw1_ws;
  float* dw1;
  long M;
  long rows_per_block;
  int G, I, J, nsplit, tiles_i, tiles_j, dw1_nblk, dw1_n, dw1_red;
  int ws, wt, os;
  float alpha;
  int force_atomic;
};
struct KronDw2sGroupArgs {
  int n;
  int wg_end[DW2G_MAX];  // exclusive prefix of the workgroup counts
  KronDw2sItem p[DW2G_MAX];
};
static_assert(sizeof(KronDw2sGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

// The same for the implicit Conv2d form (GATHER kernels): the item also carries the window geometry, so fewer fit into the
// 4 KiB of kernel arguments.  Round 3: the 38 3x3 convs of an SDXL step used to run one ~70-100 us launch each.
constexpr int DW2GC_MAX = 14;
struct KronDw2sConvItem {
  KronDw2sItem it;
  KronGather gat;
  int Jt;
};
struct KronDw2sConvGroupArgs {
  int n;
  int wg_end[DW2GC_MAX];
  KronDw2sConvItem p[DW2GC_MAX];
};
static_assert(sizeof(KronDw2sConvGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

template <typename T, int MI, int NJ, int U>
__global__ __launch_bounds__(NTHREADS) void kron_dw2s_conv_group_kernel(KronDw2sConvGroupArgs ga) {
  __shared__ __attribute__((aligned(16))) char smem[kron_dw2s_lds_bytes<MI, NJ>()];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;
  const int b0 = p ? ga.wg_end[p - 1] : 0;
  const KronDw2sItem& it = ga.p[p].it;
  KronDw2sArgs a{};
  a.Q = it.Q; a.P = it.P; a.W = it.W; a.out = it.out; a.M = it.M; a.G = it.G; a.I = it.I; a.J = it.J;
  a.ws = it.ws; a.wt = it.wt; a.os = it.os; a.rows_per_block = it.rows_per_block; a.nsplit = it.nsplit;
  a.tiles_i = it.tiles_i; a.tiles_j = it.tiles_j; a.alpha = it.alpha; a.force_atomic = it.force_atomic;
  a.dw1_ws = it.dw1_ws; a.dw1 = it.dw1; a.dw1_nblk = it.dw1_nblk; a.dw1_n = it.dw1_n; a.dw1_red = it.dw1_red;
  a.gat = ga.p[p].gat;
  a.Jt = ga.p[p].Jt;
  kron_dw2s_body<T, MI, NJ, U, true>(a, smem, b - b0);
}

template <typename T, int MI, int NJ, int U>
__global__ __launch_bounds__(NTHREADS) void kron_dw2s_group_kernel(KronDw2sGroupArgs ga) {
  __shared__ __attribute__((aligned(16))) char smem[kron_dw2s_lds_bytes<MI, NJ>()];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;  // uniform: scalar loads from the kernel-argument segment
  const int b0 = p ? ga.wg_end[p - 1] : 0;
  const KronDw2sItem& it = ga.p[p];
  KronDw2sArgs a{};
  a.Q = it.Q; a.P = it.P; a.W = it.W; a.out = it.out; a.M = it.M; a.G = it.G; a.I = it.I; a.J = it.J;
  a.ws = it.ws; a.wt = it.wt; a.os = it.os; a.rows_per_block = it.rows_per_block; a.nsplit = it.nsplit;
  a.tiles_i = it.tiles_i; a.tiles_j = it.tiles_j; a.alpha = it.alpha; a.force_atomic = it.force_atomic;
  a.dw1_ws = it.dw1_ws; a.dw1 = it.dw1; a.dw1_nblk = it.dw1_nblk; a.dw1_n = it.dw1_n; a.dw1_red = it.dw1_red;
  kron_dw2s_body<T, MI, NJ, U, false>(a, smem, b - b0);
}

// stand-alone reduction of the w1-gradient partials (used when the caller asks for dw1 but not dw2)
__global__ __launch_bounds__(NTHREADS) void kron_dw1_reduce_kernel(KronDw2sArgs a) {
  __shared__ float lds[NTHREADS];
  if ((int)blockIdx.x < a.dw1_red) dw1_reduce_role(a, (int)blockIdx.x, lds);
}

}  // namespace lyc
