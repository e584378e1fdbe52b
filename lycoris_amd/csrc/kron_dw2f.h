// kron_dw2f.h -- LoKr w2 gradient on a FULL-WIDTH output tile, 16-bit activations, gfx950 (round 4; "f" = full tile).
//
//   dW2[i, j] += alpha * sum_{r = (m, s)} Q[r, i] * Z[r, j],     Z[(m, s), j] = sum_t W[s, t] * P[(m, t), j]
//
// kron_dw2s.h (rounds 1-3) keeps an 80 x 32 output tile per WAVE and splits the rows over the waves: every slab of g / x rows is read
// once per tile, 1.96x the algorithmic bytes over the SDXL mix (profiles/pmc_traffic.json, unchanged since round 2), and its
// operand transposes run on the VALU (v_perm_b32 + 8-byte LDS writes), which is what bounds it.  Here ONE workgroup owns up to
// 160 x 160 outputs (25 600 fp32 accumulators = 100 per lane) for a slab of rows:
//
//   * the 4 waves form a WR x WC grid over the tile (x WK row-split groups when the tile is small): 160 x 160 -> 2 x 2 waves of 80 x 80,
//     80 x 80 -> one wave each on its own 32-row step, 160 x 80 -> 2 x 1 x 2; every wave keeps 5 x 5 MFMA tiles;
//   * a "super step" = WK steps of 32 rows: its Q and P rows go HBM -> LDS by LDS-DMA as they lie in memory (row-major, whole
//     320-byte row segments), into a ring of D slots with counted vmcnt waits and raw s_barrier (kron4.h's scheme) -- every row of
//     a slab is read exactly ONCE per workgroup tile;
//   * the MFMA operands are k-major (rows are the contraction index): ds_read_b64_tr_b16 reads the 4 x 16 blocks TRANSPOSED
//     straight out of the row-major image -- no v_perm, no second LDS image, no VALU transposes;
//   * the G x G mix runs on the matrix cores as before ((I (x) W) hi/lo x P block), its accumulators (lane = column, 4 rows) are the
//     hi/lo B operand of the main v_mfma_f32_16x16x32 with the k permutation {rows 4g..4g+3 of block 0, rows 4g..4g+3 of block 1}, and
//     the A operand (Q) is read with the same permutation (two transposed reads per 16-column tile);
//   * few slabs per layer (the host aims at ~8): the tile is added with fp32 atomics (one slab and exclusive output: plain adds).
// The w1-gradient partial reduction rides at the end of each problem's workgroup range as in kron_dw2s.h.
// Reference math: the autograd products of lycoris/modules/lokr.py:543-566 (make_kron backward w.r.t. w2).
#pragma once
#include "kron4.h"
#include "kron_dw2s.h"

namespace lyc {

struct KronDw2fItem {
  const void* Q;        // [rows_total, I]  exact operand, output rows i      (g as [M * G, c] rows)
  const void* P;        // [rows_total, J]  mixed with W, output columns j     (x as [M * G, d] rows)
  const float* W;       // element (s, t) at s * ws + t * wt
  float* out;           // element (i, j) at i * os + j
  const float* dw1_ws;  // w1-gradient partials of the dx launch (nullptr: none)
  float* dw1;
  int rows_total, I, J, lg;
  int ws, wt, os;
  int tiles_i, tiles_j, nslab, rows_per_slab;  // rows_per_slab % (32 * WK) == 0
  int dw1_nblk, dw1_n, dw1_red;
  float alpha;
  int plain;            // 1: one slab and nobody else adds into `out` during this launch -> load / add / store instead of atomics
  // Conv2d form (taps > 1 or a strided / padded 1x1): P is a VIRTUAL [rows_total, taps * dtap] matrix whose row (destination pixel,
  // t) and column (tap, v) is x_rows[(source pixel of the tap) * G + t, v], zero outside the image (lycoris modules/lokr.py: the
  // F.conv2d of the rebuilt weight; the gather of csrc/kron_dw2s.h).  J = taps * dtap, out element (i, tap, v) at i * os + tap * dtap + v.
  unsigned p_bytes;     // bytes of x_rows (the P descriptor covers the whole source matrix)
  int Hs, Ws, Hd, Wd, dtap;
  unsigned char taps, kw, sh, sw, ph, pw, dh, dw;  // taps == 0: plain rows (nn.Linear)
};
constexpr int DW2F_MAX = 24;
struct KronDw2fGroupArgs {
  int n;
  int wg_end[DW2F_MAX];  // exclusive prefix of the workgroup counts (tiles * slabs + reducers, per problem)
  KronDw2fItem p[DW2F_MAX];
};
static_assert(sizeof(KronDw2fGroupArgs) <= 4000, "kernel arguments are limited to 4 KiB");

__host__ __device__ constexpr int kron_dw2f_slot_bytes(int TI, int TJ, int WK) { return WK * 32 * 16 * (TI + TJ) * 2; }
__host__ __device__ constexpr int kron_dw2f_lds_bytes(int TI, int TJ, int WK, int D) { return D * kron_dw2f_slot_bytes(TI, TJ, WK) + 1024; }

// 4 x 16 block of a row-major 16-bit LDS image, transposed: lane (c = l & 15, g = l >> 4) receives image[row0 + 4 g + e][col0 + c],
// e = 0 .. 3.  ds_read_b64_tr_b16: within each group of 16 lanes, lane t supplies the address of the 8 bytes (row t >> 2, columns
// 4 (t & 3) .. + 3) and receives column t of the 4 x 16 block the group loaded (cdna_hip_programming.md T10).
//
// The reads are issued as inline asm: the compiler makes every LDS read it knows about wait for ALL LDS-DMA operations in flight
// (s_waitcnt vmcnt(0): it cannot tell which slot a DMA writes), which would drain the ring on every step -- measured: 2 us per
// 32-row step instead of 0.5.  With asm the waits are ours: k2f_lgkm<N>() below, tied to the registers it releases.
__device__ __forceinline__ u32x2 k2f_tr_issue(unsigned lds_addr, int imm_off) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(imm_off) : "memory");
  return v;
}
// wait until at most N LDS reads of this wave are outstanding (they return in order); the "+v" ties make the consumers of the
// registers depend on the wait
template <int N>
__device__ __forceinline__ void k2f_lgkm(u32x2& a, u32x2& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void k2f_lgkm10(u32x2 (&r)[5][2]) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1]), "+v"(r[3][0]), "+v"(r[3][1]),
                 "+v"(r[4][0]), "+v"(r[4][1])
               : "n"(N));
}

// TI x TJ: workgroup tile in units of 16; WR x WC x WK = 4 waves; D: ring depth (super steps).  `o_`: index of this workgroup within
// the problem's range (a multiple of 8 workgroups for the tile work, then the w1-gradient reducers).
template <typename T, int TI, int TJ, int WR, int WC, int WK, int D, bool CONV = false>
__device__ __forceinline__ void kron_dw2f_body(const KronDw2fItem& it, const int o_) {
  static_assert(WR * WC * WK == NWAVES && TI % WR == 0 && TJ % WC == 0, "wave grid");
  extern __shared__ __attribute__((aligned(1024))) char k2f_smem[];
  using F8 = typename TT<T>::frag;
  constexpr int SI = TI / WR, SJ = TJ / WC;                 // MFMA tiles per wave
  static_assert(SI == 5 && SJ == 5, "every wave keeps 5 x 5 MFMA tiles (k2f_lgkm10)");
  constexpr int QB = 32 * TI * 32, PB = 32 * TJ * 32;       // bytes of one 32-row step of Q / P in LDS (row pitch 32 TI / 32 TJ bytes)
  constexpr int SLOT = WK * (QB + PB);
  constexpr int NQ = WK * TI, NP = WK * TJ;                 // 1 KiB pieces of the Q / P parts of a slot
  constexpr int PQW = (NQ + NWAVES - 1) / NWAVES, PJW = (NP + NWAVES - 1) / NWAVES, PPW = PQW + PJW;
  static_assert(SLOT == (NQ + NP) * 1024, "slot size");
  constexpr int QP = TI * 32, PP = TJ * 32;                 // LDS row pitches (bytes)
  constexpr int QCH = 2 * TI, PCH = 2 * TJ;                 // 16-byte chunks per row
  // Row pitch 320 bytes (10 tiles): source-side swizzle -- the DMA puts chunk c of row r at chunk c ^ (2 * ((r >> 2) & 1)), i.e. odd
  // 16-lane groups see neighbouring 16-column tiles exchanged.  Measured (benchmarks/trbench.cpp, 2 workgroups per CU):
  // ds_read_b64_tr_b16 at pitch 320: 4.0 cycles per wave instruction and CU unswizzled, 3.0 with this swizzle, 4.0 with a swizzle by
  // row pairs; pitch 160 (5 tiles): 2.9 as it is (same as plain ds_read_b64 on the same addresses).
  constexpr bool QSW = (TI % 2) == 0, PSW = (TJ % 2) == 0;
  constexpr int TRASH = D * SLOT;                           // 1 KiB behind the ring for the dummy pieces
  static_assert((SI * SJ * (WK - 1) * (NWAVES / WK)) * 1024 <= D * SLOT, "the cross-wave reduction reuses the ring");

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwork = it.tiles_i * it.tiles_j * it.nslab;
  const int per = (nwork + 7) >> 3;
  if (o_ >= per * 8) {  // w1-gradient reducer workgroups ride at the end of the problem's range
    const int r = o_ - per * 8;
    if (it.dw1_ws != nullptr && r < it.dw1_red) {
      KronDw2sArgs a{};
      a.dw1_ws = it.dw1_ws; a.dw1 = it.dw1; a.dw1_nblk = it.dw1_nblk; a.dw1_n = it.dw1_n; a.dw1_red = it.dw1_red;
      a.force_atomic = it.plain ? 0 : 1;
      dw1_reduce_role(a, r, reinterpret_cast<float*>(k2f_smem));
    }
    return;
  }
  // Workgroup b of a launch runs on XCD b % 8 (observed dispatch order; used for speed only) and every XCD has its own L2: the
  // problem's range starts at a multiple of 8, XCD x is dealt the contiguous eighth [x * per, (x + 1) * per) of the work order
  const int o = (o_ & 7) * per + (o_ >> 3);
  if (o >= nwork) return;  // padding of the last XCD's share
  // slab-major: the tiles of one slab run next to each other (they share Q / P rows through L2)
  const int slab = o / (it.tiles_i * it.tiles_j);
  const int tl = o - slab * (it.tiles_i * it.tiles_j);
  const int ty = tl / it.tiles_i, tx = tl - ty * it.tiles_i;
  const int i0 = tx * 16 * TI, j0 = ty * 16 * TJ;
  const int lg = it.lg, G = 1 << lg;
  const int I = it.I, J = it.J;
  const int rbeg = slab * it.rows_per_slab;
  int rend = rbeg + it.rows_per_slab;
  if (rend > it.rows_total) rend = it.rows_total;
  const int nsuper = (rend - rbeg + 32 * WK - 1) / (32 * WK);
  // wave roles
  const int wk = wave % WK, wrc = wave / WK, wr = wrc / WC, wc = wrc % WC;

  // mix operand (I (x) W) for one 16 x 16 block: lane (i = li, g) holds k = 4g .. 4g+3 (kron_dw2s.h).  Loaded BEFORE the first DMA:
  // the loads are then the oldest entries of the vmcnt queue and the wait for them does not drain the prologue.
  float wraw[4];
  {
    const int s_ = li & (G - 1);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) wraw[jj] = it.W[s_ * it.ws + ((4 * g + jj) & (G - 1)) * it.wt];
  }

  // ---- DMA: the slot image is [wk][Q rows 32 x 32 TI bytes][P rows 32 x 32 TJ bytes]; piece p = 1 KiB of it, lane -> 16-byte chunk
  // Q / P descriptors cover rows [0, rend) of the matrices: rows of the slab's last super step beyond `rend` (and whole super steps
  // beyond the slab: the schedule below is static) read as zeros
  const unsigned qpitch = (unsigned)I * 2u, ppitch = (unsigned)J * 2u;
  const unsigned ppitch_src = CONV ? (unsigned)it.dtap * 2u : ppitch;  // Conv2d: a row of x_rows holds ONE pixel-group's dtap channels
  const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(it.Q), 0, (int)((unsigned)rend * qpitch), K4_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(it.P), 0,
                                                                       CONV ? (int)it.p_bytes : (int)((unsigned)rend * ppitch), K4_RSRC_FLAGS);
  // Pieces (1 KiB of the slot image = 64 lanes x 16 bytes): the NQ = WK * TI pieces of the Q parts and the NP = WK * TJ pieces of the P
  // parts are dealt to the waves separately (piece wave + 4 j of each kind), so whether operation j of a wave is a Q or a P piece is
  // known at compile time.  Every wave issues exactly PPW = ceil(NQ / 4) + ceil(NP / 4) operations per super step -- surplus ones go
  // to the trash KiB, out of bounds -- so the vmcnt bookkeeping is the same constant in all waves.
  unsigned voff[PPW];  // source offset of this lane's chunk for super step 0 (bytes), or out of bounds
  // Conv2d: per P piece the destination pixel of the lane's row (image, y, x: advanced by 32 WK / G pixels per super step), the tap's
  // source displacement and the byte offset of (t, v) within a source pixel
  int cb[CONV ? PJW : 1], cy[CONV ? PJW : 1], cx[CONV ? PJW : 1], coy[CONV ? PJW : 1], cox[CONV ? PJW : 1], crow[CONV ? PJW : 1];
  unsigned ccol[CONV ? PJW : 1];
#pragma unroll
  for (int jx = 0; jx < PPW; ++jx) {
    const bool q = jx < PQW;                              // compile time after unrolling
    const int idx = wave + NWAVES * (q ? jx : jx - PQW);  // piece among the Q (P) pieces
    const int per = q ? TI : TJ;
    const int k = idx / per, c = idx - k * per;           // step of the super step, piece within the step's Q (P) part
    const int chunk = c * 64 + lane;                      // 16-byte chunk within the Q (P) part
    const int row = chunk / (q ? QCH : PCH);
    int col = chunk - row * (q ? QCH : PCH);
    if (q ? QSW : PSW) col ^= 2 * ((row >> 2) & 1);
    const int gcol = (q ? i0 : j0) + col * 8;            // first element column of the chunk
    const bool ok = idx < (q ? NQ : NP) && gcol < (q ? I : J);  // I, J % 8 == 0: a chunk is all in or all out
    voff[jx] = ok ? (unsigned)(rbeg + k * 32 + row) * (q ? qpitch : ppitch) + (unsigned)gcol * 2u : K4_OOB;
    if constexpr (CONV) {
      if (!q) {
        const int e = jx - PQW;
        const int r0 = rbeg + k * 32 + row;
        const int dpix = r0 >> lg, t = r0 & (G - 1);
        const int hw = it.Hd * it.Wd;
        cb[e] = dpix / hw;
        const int rem = dpix - cb[e] * hw;
        cy[e] = rem / it.Wd;
        cx[e] = rem - cy[e] * it.Wd;
        const int tap = ok ? gcol / it.dtap : 0, v = gcol - tap * it.dtap;  // dtap % 8 == 0: a chunk lies within one tap
        const int ky = tap / it.kw, kx = tap - ky * it.kw;
        coy[e] = ky * it.dh - it.ph;
        cox[e] = kx * it.dw - it.pw;
        ccol[e] = (unsigned)(t * it.dtap + v) * 2u;
        crow[e] = ok ? r0 : 0x40000000;  // beyond every rend: the piece stays out of bounds
      }
    }
  }
  // Called with s = 0, 1, 2, ... in this order (the Conv2d state advances).
  auto issue = [&](int s, int slot) {
#pragma unroll
    for (int jx = 0; jx < PPW; ++jx) {
      const bool q = jx < PQW;
      const int idx = wave + NWAVES * (q ? jx : jx - PQW);
      const int per = q ? TI : TJ;
      const int k = idx / per, c = idx - k * per;
      // the row advance goes into the per-lane offset (not into soffset): the out-of-bounds test of rows >= rend then does not
      // depend on how the descriptor's range check treats the scalar offset; K4_OOB + advance may wrap for matrices close to
      // 2 GiB, which is why the host keeps them below 1 GiB (dw2f_ok)
      unsigned v = voff[jx] + (unsigned)s * (unsigned)(32 * WK) * (q ? qpitch : ppitch);
      if constexpr (CONV) {
        if (!q) {
          const int e = jx - PQW;
          const int ys = cy[e] * it.sh + coy[e], xs = cx[e] * it.sw + cox[e];
          const bool in = crow[e] < rend && ys >= 0 && ys < it.Hs && xs >= 0 && xs < it.Ws;
          v = in ? (unsigned)(((cb[e] * it.Hs + ys) * it.Ws + xs) << lg) * ppitch_src + ccol[e] : K4_OOB;
          crow[e] += 32 * WK;
          cx[e] += (32 * WK) >> lg;
          while (cx[e] >= it.Wd) {
            cx[e] -= it.Wd;
            if (++cy[e] >= it.Hd) { cy[e] = 0; ++cb[e]; }
          }
        }
      }
      char* dst = k2f_smem + (idx < (q ? NQ : NP) ? slot * SLOT + k * (QB + PB) + (q ? 0 : QB) + c * 1024 : TRASH);
      if (q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsq, (k4_lds_ptr)dst, 16, (int)v, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (k4_lds_ptr)dst, 16, (int)v, 0, 0, 0);
    }
  };
  constexpr int C = PPW;  // DMA operations per wave and super step
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s, s);

  F8 a2;  // lane (i = li, g): k = (hi, 4g + e), (lo, 4g + e) of the block-diagonal (I (x) W)
  {
    const int mi_ = li >> lg;
    T hl[8];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kk = 4 * g + jj;
      split_f<T>(((kk >> lg) == mi_) ? wraw[jj] : 0.f, hl[jj], hl[4 + jj]);
    }
    a2 = *reinterpret_cast<F8*>(hl);
  }
  f32x4 acc[SI][SJ];
#pragma unroll
  for (int a = 0; a < SI; ++a)
#pragma unroll
    for (int c = 0; c < SJ; ++c) acc[a][c] = zero4();

  // per-lane read addresses (LDS byte offsets) of tile 0 of this wave, for even / odd absolute tile index (the swizzle)
  const unsigned sbase = (unsigned)(size_t)(k4_lds_ptr)k2f_smem;
  const int t4 = li >> 2, t3 = li & 3;
  const unsigned qlane = (unsigned)((4 * g + t4) * QP + t3 * 8 + wr * SI * 32 + wk * (QB + PB));
  const unsigned plane = (unsigned)((4 * g + t4) * PP + t3 * 8 + wc * SJ * 32 + wk * (QB + PB) + QB);
  const int qsh = QSW ? (g & 1) * 32 : 0, psh = PSW ? (g & 1) * 32 : 0;
  // absolute tile A = wr * SI + a: odd groups read tile A ^ 1, i.e. +32 bytes when A is even and -32 when it is odd
  const int qpar = (wr * SI) & 1, ppar = (wc * SJ) & 1;
  const unsigned qad[2] = {sbase + qlane + (unsigned)(qpar ? -qsh : qsh), sbase + qlane + (unsigned)(qpar ? qsh : -qsh)};
  const unsigned pad[2] = {sbase + plane + (unsigned)(ppar ? -psh : psh), sbase + plane + (unsigned)(ppar ? psh : -psh)};

  int slot = 0, prev = 0;
#ifdef LYC_TRACE  // benchmarks (-DLYC_TRACE): shader-clock sums of the phases of a step in workgroup LYC_TRACE_BLOCK.  s_memtime returns
  // through lgkmcnt: a stamp also waits for the LDS reads in flight (tB: issue + latency of all 20 reads, tC then ~0)
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tprev = __builtin_readcyclecounter();
#define K2F_STAMP(acc_)                                   \
  do {                                                    \
    __builtin_amdgcn_sched_barrier(0);                    \
    const unsigned long long t_ = __builtin_readcyclecounter(); \
    acc_ += t_ - tprev;                                   \
    tprev = t_;                                           \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#else
#define K2F_STAMP(acc_) do {} while (0)
#endif
  for (int s = 0; s < nsuper; ++s) {
    if (s == 0) k4_wait_vm<(D - 1) * C>();
    else k4_wait_vm<(D - 2) * C>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    K2F_STAMP(tA);
    const unsigned so = (unsigned)(slot * SLOT);
    u32x2 pr[SJ][2], qr[SI][2];
#pragma unroll
    for (int c = 0; c < SJ; ++c) {
      pr[c][0] = k2f_tr_issue(pad[c & 1] + so, c * 32);
      pr[c][1] = k2f_tr_issue(pad[c & 1] + so, 16 * PP + c * 32);
    }
#pragma unroll
    for (int a = 0; a < SI; ++a) {
      qr[a][0] = k2f_tr_issue(qad[a & 1] + so, a * 32);
      qr[a][1] = k2f_tr_issue(qad[a & 1] + so, 16 * QP + a * 32);
    }
    // the refill goes out behind the LDS reads: a wave that stalls in the issue of its DMA operations (the queue of the texture
    // path is short) stalls while its reads are in flight, not in front of them
    __builtin_amdgcn_sched_barrier(0);
    if (s >= 1) issue(s - 1 + D, prev);  // every wave is past its reads of `prev`; steps beyond the slab fetch nothing
    K2F_STAMP(tB);
    __builtin_amdgcn_sched_barrier(0);
    k2f_lgkm10<2 * SI>(pr);
    K2F_STAMP(tC);
    // mix: Z blocks of this wave's SJ column tiles, both 16-row halves -> hi / lo B fragments of the main MFMA.  ONE
    // v_mfma_16x16x32 per block: k = (hi | lo part of W, t) against the P rows twice, i.e. W_hi P + W_lo P in a single pass.
    f32x4 z[SJ][2];
#pragma unroll
    for (int c = 0; c < SJ; ++c)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const u32x4 bv = {pr[c][rb][0], pr[c][rb][1], pr[c][rb][0], pr[c][rb][1]};
        z[c][rb] = TT<T>::mma(a2, *reinterpret_cast<const F8*>(&bv), zero4());
      }
    F8 zh[SJ], zl[SJ];
#pragma unroll
    for (int c = 0; c < SJ; ++c) {
      T hh[8], ll[8];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) split_f<T>(z[c][rb][e], hh[4 * rb + e], ll[4 * rb + e]);
      zh[c] = *reinterpret_cast<F8*>(hh);
      zl[c] = *reinterpret_cast<F8*>(ll);
    }
    K2F_STAMP(tD);
#pragma unroll
    for (int a = 0; a < SI; ++a) {
      __builtin_amdgcn_sched_barrier(0);
      if (a == 0) k2f_lgkm<8>(qr[0][0], qr[0][1]);
      else if (a == 1) k2f_lgkm<6>(qr[1][0], qr[1][1]);
      else if (a == 2) k2f_lgkm<4>(qr[2][0], qr[2][1]);
      else if (a == 3) k2f_lgkm<2>(qr[3][0], qr[3][1]);
      else k2f_lgkm<0>(qr[4][0], qr[4][1]);
      const u32x4 av = {qr[a][0][0], qr[a][0][1], qr[a][1][0], qr[a][1][1]};
      const F8 af = *reinterpret_cast<const F8*>(&av);
#pragma unroll
      for (int c = 0; c < SJ; ++c) acc[a][c] = TT<T>::mma(af, zh[c], acc[a][c]);
#pragma unroll
      for (int c = 0; c < SJ; ++c) acc[a][c] = TT<T>::mma(af, zl[c], acc[a][c]);
    }
    K2F_STAMP(tE);
    prev = slot;
    slot = slot + 1 == D ? 0 : slot + 1;
  }
#ifdef LYC_TRACE
  if (blockIdx.x == LYC_TRACE_BLOCK && threadIdx.x == 0) {
    lyc_trace_buf[0] = tA; lyc_trace_buf[1] = tB; lyc_trace_buf[2] = tC; lyc_trace_buf[3] = tD; lyc_trace_buf[4] = tE;
    lyc_trace_buf[5] = (unsigned long long)nsuper;
  }
#endif
  k4_wait_vm<0>();  // the zero-filling operations of the steps beyond the slab must not outlive the workgroup's LDS allocation

#ifdef LYC_TUNE
  if (it.plain & 2) return;  // development builds: time the kernel without its output path
#endif
  // ---- WK > 1: the waves of a row-split group add their tiles up through LDS (the ring is free now) -- tile t of the 25 belongs to
  // wave t % WK of the group, the others park theirs in buf[group][t][source slot] (1 KiB each, lane-linear 16 bytes: conflict-free).
  // One atomic per output element and SLAB instead of one per wave.
  constexpr int NT = SI * SJ;
  f32x4* const red = reinterpret_cast<f32x4*>(k2f_smem);
  if constexpr (WK > 1) {
    __builtin_amdgcn_s_barrier();  // every wave's DMA has landed (each waited for its own) and every wave is past its last reads
    asm volatile("" ::: "memory");
#pragma unroll
    for (int a = 0; a < SI; ++a)
#pragma unroll
      for (int c = 0; c < SJ; ++c) {
        const int t = a * SJ + c, owner = t % WK;
        if (wk != owner) {
          const int src = (wk - owner - 1 + WK) % WK;  // 0 .. WK - 2
          red[((wrc * NT + t) * (WK - 1) + src) * 64 + lane] = acc[a][c];
        }
      }
    __syncthreads();
  }

  // ---- the tile: lane (column j = li, rows 4g + r) of MFMA tile (a, c) ----------------------------------------------------------------
  float* const out = it.out;
  const int os = it.os;
  const float alpha = it.alpha;
  const bool plain = it.plain != 0;
#pragma unroll
  for (int a = 0; a < SI; ++a)
#pragma unroll
    for (int c = 0; c < SJ; ++c) {
      const int t = a * SJ + c;
      if (WK > 1 && wk != t % WK) continue;
      f32x4 v = acc[a][c];
      if constexpr (WK > 1) {
#pragma unroll
        for (int src = 0; src < WK - 1; ++src) v += red[((wrc * NT + t) * (WK - 1) + src) * 64 + lane];
      }
      const int gj = j0 + 16 * (wc * SJ + c) + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + 16 * (wr * SI + a) + 4 * g + r;
        if (gi < I && gj < J) {
          float* dst = out + (long)gi * os + gj;
          if (plain) *dst += alpha * v[r];
          else __hip_atomic_fetch_add(dst, alpha * v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
}

// (a) up to DW2F_MAX problems in the kernel arguments
// __launch_bounds__(256, 2): two workgroups per CU -- and, with at most 256 registers per lane, the compiler keeps the MFMA
// accumulators in VGPRs (no v_accvgpr_read in front of the hi / lo split: 70 of 275 VALU instructions per step, PMC-counted)
template <typename T, int TI, int TJ, int WR, int WC, int WK, int D, bool CONV = false>
__global__ __launch_bounds__(NTHREADS, 2) void kron_dw2f_group_kernel(KronDw2fGroupArgs ga) {
  const int b = (int)blockIdx.x;
  int pi = 0;
  while (pi + 1 < ga.n && b >= ga.wg_end[pi]) ++pi;  // uniform: scalar loads from the kernel-argument segment
  const int b0 = pi ? ga.wg_end[pi - 1] : 0;
  const KronDw2fItem it = ga.p[pi];
  kron_dw2f_body<T, TI, TJ, WR, WC, WK, D, CONV>(it, b - b0);
}

// (b) any number of problems in a device table (written by kron_dw2f_table_write_kernel launches in front of this one): ONE launch
// per tile class for a whole backward pass -- no launch tails between groups of 24 layers, two workgroups per CU throughout
template <typename T, int TI, int TJ, int WR, int WC, int WK, int D>
__global__ __launch_bounds__(NTHREADS, 2) void kron_dw2f_table_kernel(const KronDw2fItem* __restrict__ items, const int* __restrict__ wg_end, int n) {
  const int b = (int)blockIdx.x;
  int lo = 0, hi = n - 1;  // first problem whose (exclusive) end is beyond b; uniform -> scalar loads
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b >= wg_end[mid]) lo = mid + 1;
    else hi = mid;
  }
  const int b0 = lo ? wg_end[lo - 1] : 0;
  const KronDw2fItem it = items[lo];
  kron_dw2f_body<T, TI, TJ, WR, WC, WK, D>(it, b - b0);
}

// copies the problems of one kernel-argument block into the device table (capture-safe: the arguments are part of the launch)
__global__ __launch_bounds__(64) void kron_dw2f_table_write_kernel(KronDw2fGroupArgs ga, KronDw2fItem* items, int* wg_end, int first, int wg_base) {
  const int t = threadIdx.x;
  if (t < ga.n) {
    items[first + t] = ga.p[t];
    wg_end[first + t] = wg_base + ga.wg_end[t];
  }
}

}  // namespace lyc
