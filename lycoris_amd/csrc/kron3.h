// kron3.h -- Kronecker (LoKr) row kernel for 16-bit activations, gfx950.  Third generation: built around the measured
// cost structure of the SDXL shapes (benchmarks/kbench.cpp, benchmarks/ktrace.cpp): launches are 2-10 us long, so the
// serial latency chain of one workgroup and the fp32 atomics decide the time, not peak bandwidth.
//
//   S1[(m,u), n]   = sum_k x3[(m,u), k] * w2[n, k]                      stage 1, v_mfma_f32_16x16x32 (w2 = hi + lo)
//   y [(m,p), n]   = alpha * sum_u w1[p,u] * S1[(m,u), n]                stage 2, v_mfma_f32_16x16x16 in registers
//   dW1[p,u]      += alpha * sum_{m,n} S1[(m,u), n] * xref[(m,p), n]     backward only, in registers
//
// Taken when Gin == Gout == G, 16 % G == 0, K % 8 == 0, x 16-byte aligned (every SDXL / SD1.5 layer, factor 1..16).
//
//   * x fragments go HBM -> registers (a row of x3 is used by exactly one wave); each fragment register is re-loaded
//     for the next K chunk right after its last MFMA, so the loads of chunk c+1 fly under the MFMAs of chunk c;
//   * the w2 chunk is converted fp32 -> hi/lo once per workgroup into a DOUBLE-buffered LDS tile [n][k] (row pitch
//     = 16 mod 32 elements: conflict-free ds_read_b128 fragments), one barrier per chunk;
//   * the w1 values are fetched at kernel start but first touched in the epilogue (no early round trip);
//   * stage 2 is computed transposed: D[n, (m,p)] = S1^T (accumulator registers used as the A operand) x (I (x) w1)^T,
//     which leaves each lane with 4 CONSECUTIVE n of one output row -> 8-byte global stores straight from registers,
//     no LDS image, no barrier;
//   * the w1 gradient also stays in registers: S1 hi/lo are transposed by an MFMA with the identity and fed back as the
//     A operand against 8-byte xref fragments; one 64-lane atomic instruction per workgroup (same-line fp32 atomics
//     serialise at ~16 ns each on this chip, so the count of atomic instructions is what matters).
#pragma once
#include <type_traits>

#include "lokr_kernels.h"

namespace lyc {

// compiler + scheduler fence (no instruction): keeps the global loads on either side in source order
#define LR3_FENCE()                      \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

template <typename T>
struct Mma16;
template <>
struct Mma16<__bf16> {
  typedef __attribute__((ext_vector_type(4))) short frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma16<_Float16> {
  typedef __attribute__((ext_vector_type(4))) _Float16 frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
  }
};

constexpr int K3_RT = 128;          // stage-1 rows per workgroup (4 waves x 32)
// K chunk per LDS tile (template parameter KC, a multiple of 32):
//   160 for the narrow tile (NI = 2, small latency-bound problems: the SDXL / SD1.5 factor dims are multiples of 160 or
//       fit in one chunk, so most launches have a single segment and a single barrier);
//    96 for the wide tile (NI = 4, throughput-bound problems: fewer live registers -> 3-4 waves per SIMD).
// LDS row pitch of the w2 tiles = KC + 16 elements (8 mod 16 dwords: conflict-free ds_read_b128 fragments).
__host__ __device__ constexpr int kron3_kc(int NI) { return NI <= 2 ? 160 : 96; }

// one hi/lo tile pair per buffer; a second buffer only when there is more than one segment
__host__ __device__ constexpr int kron3_lds_bytes(int NI, int nbuf) {
  const int b = nbuf * 2 * 16 * NI * (kron3_kc(NI) + 16) * 2;
  return b > 4096 ? b : 4096;  // the w1-gradient reduction scratch needs 4 KiB
}

// GM == 3: x goes HBM -> registers -> a small per-wave LDS tile -> fragments.  A fragment-direct load gives the four lanes
// of a quad four different rows (four cache lines): the address unit then needs ~64 cycles per wave instruction.  Here
// four consecutive lanes read 64 contiguous bytes of one row (one 32-column k-step), 16 rows per instruction: ~16 cycles.
// Per wave: two [32 rows][K3_XP] tiles (one per k-step parity), written and read by the same wave only (no barrier).
constexpr int K3_XP = 40;  // row pitch (elements) of the x tiles: 80 bytes
__host__ __device__ constexpr int kron3_xs_bytes() { return 4 * 2 * 32 * K3_XP * 2; }

enum { K3_W2_ROWS = 0, K3_W2_COLS = 1, K3_W2_SCALAR = 2 };

// The vector modes load whole float4s only: ROWS needs K % 4 == 0 (the fast path has K % 8 == 0), COLS needs N % 4 == 0.
__device__ __forceinline__ int k3_w2_mode(const float* w2, long s2n, long s2k, int N, int K) {
  const bool aligned = (reinterpret_cast<uintptr_t>(w2) & 15u) == 0;
  if (s2k == 1 && aligned && (s2n % 4 == 0) && (K % 4 == 0)) return K3_W2_ROWS;
  if (s2n == 1 && aligned && (s2k % 4 == 0) && (N % 4 == 0)) return K3_W2_COLS;
  return K3_W2_SCALAR;
}

template <typename T>
__device__ __forceinline__ void k3_split4(const f32x4& v, T (&hi)[4], T (&lo)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) split_f<T>(v[e], hi[e], lo[e]);
}

// w2 chunk: global -> registers -> LDS, in 4 (n) x 4 (k) blocks so that both orientations run the same instruction
// stream.  Element (n, k) lives at w2[n * s2n + k * s2k], one of the strides is 1:
//   ROWS (k contiguous): the four float4 of a block are its n rows (read along k), written as they are;
//   COLS (n contiguous): the four float4 are its k rows (read along n), transposed in registers when written.
// Consecutive threads walk the contiguous dimension.  SCALAR (odd strides / alignment) goes element-wise at store time.
template <int TQ, int KC>
struct K3Raw {
  static constexpr int NQ = TQ / 4, KQ = KC / 4;
  static constexpr int N_BLK = (NQ * KQ + NTHREADS - 1) / NTHREADS;  // blocks per thread
  static constexpr int NRAW = 4 * N_BLK;
};

template <int TQ, int KC>
__device__ __forceinline__ void k3_load_w2(f32x4 (&raw)[(K3Raw<TQ, KC>::NRAW)], int mode, const float* __restrict__ w2,
                                           long s2n, long s2k, long n0, int N, long k0, int K, const void* safe16) {
  using R = K3Raw<TQ, KC>;
  // SCALAR mode (unaligned / odd strides) does not use `raw`, but an early return here would make `raw` a PHI of
  // "loaded" and "undefined": the compiler then copies all of it at the join and waits vmcnt(0) right there, before the
  // caller's x loads are even issued.  So the loads are issued in every mode, in SCALAR mode from `safe16` (any valid
  // 16-byte aligned, >= 16-byte buffer: the activations) and ignored.
  const bool scalar = mode == K3_W2_SCALAR;
  const float* dflt = scalar ? static_cast<const float*>(safe16) : w2;
  const bool rows = (mode == K3_W2_ROWS);
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < R::N_BLK; ++it) {
    const int b = tid + NTHREADS * it;
    const int nq = rows ? b / R::KQ : b % R::NQ;
    const int kq = rows ? b % R::KQ : b / R::NQ;
    const long gn = n0 + 4 * nq, gk = k0 + 4 * kq;
    const bool blk_ok = !scalar && (b < R::NQ * R::KQ) && gn < N && gk < K;
    // ROWS: float4 j = row gn + j, columns gk .. gk+3 (K % 4 == 0: whole);  COLS: float4 j = k row gk + j, n gn .. gn+3
    // unconditional loads from a clamped (always valid) address, zeroed by a select: no exec-mask branches, so all
    // loads of a chunk issue back to back
    const long jstride = rows ? s2n : s2k;
    const long jlimit = rows ? (long)N - gn : (long)K - gk;
    const float* base = blk_ok ? w2 + gn * s2n + gk * s2k : dflt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = blk_ok && j < jlimit;
      raw[4 * it + j] = *reinterpret_cast<const f32x4*>(ok ? base + j * jstride : dflt);  // zeroed in k3_store_w2
    }
  }
}

template <typename T, int TQ, int KC>
__device__ __forceinline__ void k3_store_w2(T* __restrict__ Bh, T* __restrict__ Bl,
                                            const f32x4 (&raw)[(K3Raw<TQ, KC>::NRAW)], int mode,
                                            const float* __restrict__ w2, long s2n, long s2k, long n0, int N, long k0,
                                            int K) {
  using R = K3Raw<TQ, KC>;
  constexpr int K3_LDB = KC + 16, K3_KC = KC;
  const int tid = threadIdx.x;
  if (mode != K3_W2_SCALAR) {
    const bool rows = (mode == K3_W2_ROWS);
#pragma unroll
    for (int it = 0; it < R::N_BLK; ++it) {
      const int b = tid + NTHREADS * it;
      const int nq = rows ? b / R::KQ : b % R::NQ;
      const int kq = rows ? b % R::KQ : b / R::NQ;
      if (b < R::NQ * R::KQ) {
        // the loads were unconditional (clamped addresses): zero what lies outside [N) x [K) here, at the point of use
        const long gn = n0 + 4 * nq, gk = k0 + 4 * kq;
        const bool blk_ok = gn < N && gk < K;
        const long jlimit = rows ? (long)N - gn : (long)K - gk;
        f32x4 rz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          rz[j] = (blk_ok && j < jlimit) ? raw[4 * it + j] : z;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // LDS row n = 4 nq + e, columns 4 kq .. 4 kq + 3
          const f32x4 t = {rz[0][e], rz[1][e], rz[2][e], rz[3][e]};
          const f32x4 r = rz[e];
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = rows ? r[q] : t[q];
          T h[4], l[4];
          k3_split4<T>(v, h, l);
          *reinterpret_cast<u32x2*>(Bh + (4 * nq + e) * K3_LDB + 4 * kq) = *reinterpret_cast<u32x2*>(h);
          *reinterpret_cast<u32x2*>(Bl + (4 * nq + e) * K3_LDB + 4 * kq) = *reinterpret_cast<u32x2*>(l);
        }
      }
    }
  } else {  // unaligned / odd strides: element-wise, straight from global
    for (int e = tid; e < TQ * K3_KC; e += NTHREADS) {
      int n, k;
      if (s2k == 1) {
        n = e / K3_KC;
        k = e % K3_KC;
      } else {
        n = e % TQ;
        k = e / TQ;
      }
      const long gn = n0 + n, gk = k0 + k;
      const float v = (gn < N && gk < K) ? w2[gn * s2n + gk * s2k] : 0.f;
      T h, l;
      split_f<T>(v, h, l);
      Bh[n * K3_LDB + k] = h;
      Bl[n * K3_LDB + k] = l;
    }
  }
}

// Stage 1 of the factored kernels: acc[mi][ni] (16x16 tiles, wave tile 32 rows x 16 NI columns) =
//   sum over segments of  x3[row, k] * w2[n0 + n, k]      (x exact T from HBM, w2 fp32 -> hi/lo through LDS)
// for the workgroup's 128 stage-1 rows starting at row0 (rows >= rows_end read as zero).  Shared by the LoKr kernel
// (kron3_body) and the LoCon kernel (locon3.h: N = rank).  `smem`: kron3_lds_bytes(NI, segments > 1 ? 2 : 1) bytes.
// GM: 0 = plain rows, 1 = pixel-row gather with one K segment sequence per tap (any geometry), 2 = gather over the
// FLAT K index (tap, k) -- the source pixel of tap t is base + offset[t] (forward with any stride, backward with stride
// 1), so segments are full K3_KC chunks however short the per-tap rows are (C = 320 convs have 40 columns per tap).
// `after_loads()` is invoked once, right behind the issue of the first chunk's w2 and x loads: the place for loads the
// caller needs only after stage 1 (vector-memory results return in issue order, so they must not precede the w2 loads).
// PL (round 3): the w2 operand comes from PRE-PACKED hi / lo planes (kron_conv.h: fragment-major units of 2 KiB per (n tile, k
// step), written once per optimizer step by lyc_lokr_pack_w2 / _pack_group) streamed into the LDS tile by LDS-DMA: no fp32 loads,
// no conversion (~1900 of a workgroup's ~10000 cycles, profiles/r02_ktrace_kron3_phases.log), no ds_write, lane-linear
// conflict-free fragment reads.  Plain-row modes only (GM = 0 / 3); the chunk / double-buffer structure is unchanged.
template <typename T, int NI, int GM, bool PL, typename Hook>
__device__ __forceinline__ void k3_stage1(const KronArgs& a, char* smem, long row0, long rows_end, long n0,
                                          f32x4 (&acc)[2][NI], Hook&& after_loads) {
  static_assert(!PL || GM == 0 || GM == 3, "packed planes: plain-row kernels only");
  constexpr bool GATHER = GM == 1 || GM == 2;
  constexpr bool FLAT = GM == 2;
  constexpr bool XS = GM == 3;  // plain rows, x through the per-wave LDS stage (quad-coalesced loads)
  constexpr int MI = 2, TQ = 16 * NI;
  constexpr int K3_KC = kron3_kc(NI), K3_KS = K3_KC / 32, K3_LDB = K3_KC + 16;
  constexpr int PLANE = TQ * K3_LDB;  // elements per hi or lo tile
  T* Bbase = reinterpret_cast<T*>(smem);
  using F8 = typename TT<T>::frag;
  using RW = K3Raw<TQ, K3_KC>;
  const T* x = static_cast<const T*>(a.x);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int G = a.Gin, K = a.K, N = a.N;
  const int lg = 31 - __builtin_clz((unsigned)G);

  // this lane's two stage-1 rows (dst pixel, group).  With a gather the operand row of tap t comes from another pixel.
  const long gr0 = row0 + wave * 32 + li, gr1 = gr0 + 16;
  const bool ok0 = gr0 < rows_end, ok1 = gr1 < rows_end;
  const int taps = (GATHER && !FLAT) ? a.gat.taps : 1;              // segment sequences
  const int Kloop = FLAT ? a.gat.taps * K : K;                        // K extent of one sequence
  int pb[2], ph_[2], pw_[2];  // (image, h, w) of the two destination pixels
  if constexpr (GATHER) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const long pix = ((r ? (ok1 ? gr1 : row0) : (ok0 ? gr0 : row0)) >> lg);
      const int hw = a.gat.Hd * a.gat.Wd;
      pb[r] = (int)(pix / hw);
      const int rem = (int)(pix - (long)pb[r] * hw);
      ph_[r] = rem / a.gat.Wd;
      pw_[r] = rem - ph_[r] * a.gat.Wd;
    }
  }
  // FLAT: per row, the flat source row index of tap offset 0 and the bit mask of the taps that fall inside the image;
  // per tap, the source-row offset (LDS table).  Fragment 8g..8g+7 of a k-step never straddles taps (K % 8 == 0).
  __shared__ long k3_tapoff[FLAT ? 64 : 1];
  long frow[2] = {0, 0};
  unsigned long long fmask[2] = {0ull, 0ull};
  if constexpr (FLAT) {
    const int kh = a.gat.taps / a.gat.kw;
    if (tid < a.gat.taps) {
      const int i = tid / a.gat.kw, j = tid - i * a.gat.kw;
      const long off = (long)i * a.gat.dh * a.gat.Ws + (long)j * a.gat.dw;
      k3_tapoff[tid] = (a.gat.mode == 1 ? off : -off) << lg;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const long gr = r ? gr1 : gr0;
      const int h0 = a.gat.mode == 1 ? ph_[r] * a.gat.sh - a.gat.ph : ph_[r] + a.gat.ph;
      const int w0 = a.gat.mode == 1 ? pw_[r] * a.gat.sw - a.gat.pw : pw_[r] + a.gat.pw;
      const int sgn = a.gat.mode == 1 ? 1 : -1;
      unsigned long long wm = 0ull;
      for (int j = 0; j < a.gat.kw; ++j) {
        const int ws = w0 + sgn * j * a.gat.dw;
        if (ws >= 0 && ws < a.gat.Ws) wm |= 1ull << j;
      }
      unsigned long long m = 0ull;
      for (int i = 0; i < kh; ++i) {
        const int hs = h0 + sgn * i * a.gat.dh;
        if (hs >= 0 && hs < a.gat.Hs) m |= wm << (i * a.gat.kw);
      }
      fmask[r] = (r ? ok1 : ok0) ? m : 0ull;
      frow[r] = ((((long)pb[r] * a.gat.Hs + h0) * a.gat.Ws + w0) << lg) + (gr & (G - 1));
    }
    __syncthreads();  // the tap table
  }
  // FLAT fragment of lane row r for the k-step starting at flat index kk (uniform)
  auto load_frag_flat = [&](int r, long kk) -> F8 {
    int tap = (int)(kk / K);
    int rem = (int)(kk - (long)tap * K) + 8 * g;
#pragma unroll
    for (int w = 0; w < 3; ++w)
      if (rem >= K) {
        rem -= K;
        ++tap;
      }
    const bool ok = tap < a.gat.taps && ((fmask[r] >> tap) & 1ull);
    const long srow = frow[r] + k3_tapoff[ok ? tap : 0];
    const u32x4 v = *reinterpret_cast<const u32x4*>(ok ? x + srow * K + rem : x);
    const u32x4 z = {0u, 0u, 0u, 0u};
    const u32x4 o = ok ? v : z;
    return *reinterpret_cast<const F8*>(&o);
  };
  // row base pointer (WITHOUT the lane's 8g column offset: load_frag_x adds it only where the fragment lies inside the row)
  // and validity of lane row r for tap t
  auto row_src = [&](int r, int t, const T*& p, bool& ok) {
    const long gr = r ? gr1 : gr0;
    const bool rok = r ? ok1 : ok0;
    if constexpr (!GATHER) {
      ok = rok;
      p = x + (rok ? gr : row0) * K;
      return;
    }
    const int i = t / a.gat.kw, j = t - i * a.gat.kw;
    int hs, ws;
    bool v;
    if (a.gat.mode == 1) {
      hs = ph_[r] * a.gat.sh - a.gat.ph + i * a.gat.dh;
      ws = pw_[r] * a.gat.sw - a.gat.pw + j * a.gat.dw;
      v = hs >= 0 && hs < a.gat.Hs && ws >= 0 && ws < a.gat.Ws;
    } else {
      const int hn = ph_[r] + a.gat.ph - i * a.gat.dh, wn = pw_[r] + a.gat.pw - j * a.gat.dw;
      hs = hn / a.gat.sh;
      ws = wn / a.gat.sw;
      v = hn >= 0 && wn >= 0 && hs * a.gat.sh == hn && ws * a.gat.sw == wn && hs < a.gat.Hs && ws < a.gat.Ws;
    }
    ok = rok && v;
    const long spix = ok ? ((long)pb[r] * a.gat.Hs + hs) * a.gat.Ws + ws : 0;
    p = x + ((spix << lg) + (gr & (G - 1))) * K;
  };
  // The load is unconditional (masked afterwards), so its address must be valid for EVERY lane: a lane whose fragment lies
  // beyond the row (kk + 8g >= K) reads the row's first 16 bytes.  (Until the end of round 2 it read `row + 8g`, which for
  // K < 32 is past the row and, on the matrix's last rows, up to 48 bytes past the END of the activation tensor: a rare
  // memory access fault when the tensor filled its allocation exactly -- found by benchmarks/stress_grouped.py.)
  auto load_frag_x = [&](bool ok, const T* p, long kk) -> F8 {
    const bool k_ok = kk + 8 * g < K;  // K % 8 == 0: a fragment is all in or all out
    const u32x4 v = *reinterpret_cast<const u32x4*>(p + (k_ok ? kk + 8 * g : 0));
    const u32x4 z = {0u, 0u, 0u, 0u};
    const u32x4 r = (ok && k_ok) ? v : z;
    return *reinterpret_cast<const F8*>(&r);
  };

  const int w2mode = k3_w2_mode(a.w2, a.s2n, a.s2k, N, Kloop);
  F8 af[MI][K3_KS];
  f32x4 raw[RW::NRAW];
  // The w2 loads go out FIRST: vector-memory results return in issue order, and the w2 tile is on the critical path
  // (convert -> LDS -> barrier -> first MFMA) while the x fragments are only needed at the MFMAs.  Issued behind the ten
  // x loads (HBM) the L2-resident w2 data could not be touched before all of x had arrived.
  // PL: unit (n tile nt, k step ks) of the planes lives at ((nt * ksteps + ks) * 2 + {hi, lo}) KiB; a chunk's image in LDS is
  // [ni][ks][hi | lo][64 lanes][16 B] (piece p = (ni * K3_KS + ks) * 2 + half at p KiB), filled by 1 KiB LDS-DMA pieces
  const char* planes = static_cast<const char*>(a.w2p);
  const int pl_ksteps = (K + 31) >> 5;
  const int pl_ntiles = (N + 15) >> 4;
  auto issue_planes = [&](long k0c, char* dst) {
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks0 = (int)(k0c >> 5);
    for (int p = wv; p < 2 * NI * K3_KS; p += NWAVES) {
      const int unit = p >> 1, half = p & 1;
      const int ni = unit / K3_KS, kk = unit - ni * K3_KS;
      long nt = n0 / 16 + ni;
      if (nt >= pl_ntiles) nt = pl_ntiles - 1;  // beyond N: a valid duplicate, its columns are never stored
      if (ks0 + kk < pl_ksteps) {
        const char* src = planes + ((nt * pl_ksteps + ks0 + kk) * 2 + half) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
      }
    }
  };
  if constexpr (PL) issue_planes(0, smem);
  else k3_load_w2<TQ, K3_KC>(raw, w2mode, a.w2, a.s2n, a.s2k, n0, N, 0, Kloop, a.x);
  LR3_FENCE();
  // XS: this lane's piece of a k-step: rows (lane >> 2) and (lane >> 2) + 16 of the wave's 32, columns 8 (lane & 3) .. + 7
  const int xs_r = lane >> 2, xs_c = 8 * (lane & 3);
  const T* xs_p[2] = {x, x};
  bool xs_v[2] = {false, false};
  T* xs_tile = nullptr;
  if constexpr (XS) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const long gr = row0 + wave * 32 + xs_r + 16 * p;
      xs_v[p] = gr < rows_end;
      xs_p[p] = x + (xs_v[p] ? gr : row0) * K + xs_c;
    }
    xs_tile = reinterpret_cast<T*>(smem + kron3_lds_bytes(NI, Kloop > K3_KC ? 2 : 1)) + wave * 2 * 32 * K3_XP;
  }
  auto load_raw_x = [&](int p, long kk) -> F8 {  // unconditional, from a clamped address; masked when written to LDS
    const u32x4 v = *reinterpret_cast<const u32x4*>(xs_p[p] + (kk + xs_c < K ? kk : 0));
    return *reinterpret_cast<const F8*>(&v);
  };
  if constexpr (XS) {
#pragma unroll
    for (int ks = 0; ks < K3_KS; ++ks) {
      af[0][ks] = load_raw_x(0, ks * 32);
      af[1][ks] = load_raw_x(1, ks * 32);
    }
  } else if constexpr (FLAT) {
#pragma unroll
    for (int ks = 0; ks < K3_KS; ++ks) {
      af[0][ks] = load_frag_flat(0, ks * 32);
      af[1][ks] = load_frag_flat(1, ks * 32);
    }
  } else {
    const T *p0, *p1;
    bool v0, v1;
    row_src(0, 0, p0, v0);
    row_src(1, 0, p1, v1);
#pragma unroll
    for (int ks = 0; ks < K3_KS; ++ks) {
      af[0][ks] = load_frag_x(v0, p0, ks * 32);
      af[1][ks] = load_frag_x(v1, p1, ks * 32);
    }
  }
  after_loads();
  LR3_FENCE();  // ... and the x loads before the first use of the w2 data (the scheduler would hoist the conversions)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  LYC_STAMP_DIRECT(1);  // all loads of the first chunk issued
  if constexpr (PL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
  else k3_store_w2<T, TQ, K3_KC>(Bbase, Bbase + PLANE, raw, w2mode, a.w2, a.s2n, a.s2k, n0, N, 0, Kloop);
  LYC_STAMP_DIRECT(2);  // w2 data arrived, converted, written to LDS
  __syncthreads();
  LYC_STAMP_DIRECT(3);  // first barrier passed

  // one chunk: NKS k-steps of MFMAs from (af, LDS tile); with MORE the fragment registers of each k-step are re-loaded
  // for the next chunk as soon as they are free.  Straight-line code (no per-k-step branches), so the compiler hoists the
  // ds_read_b128 fragment reads over the MFMAs.
  auto chunk = [&](auto nks_tag, auto more_tag, const T* Bh, const T* Bl, const T* np0, bool nv0, const T* np1,
                   bool nv1, long knext, long kcur) {
    constexpr int NKS = decltype(nks_tag)::value;
    constexpr bool MORE = decltype(more_tag)::value;
#pragma unroll
    for (int ks = 0; ks < K3_KS; ++ks) {
      F8 a0 = af[0][ks], a1 = af[1][ks];
      if constexpr (XS) {
        if (ks < NKS) {  // raw pieces -> this wave's tile -> fragments (LDS operations of one wave execute in order)
          T* xt = xs_tile + (ks & 1) * 32 * K3_XP;
          const bool kok = kcur + ks * 32 + xs_c < K;
          const u32x4 z = {0u, 0u, 0u, 0u};
          u32x4 r0v = *reinterpret_cast<const u32x4*>(&af[0][ks]), r1v = *reinterpret_cast<const u32x4*>(&af[1][ks]);
          *reinterpret_cast<u32x4*>(xt + xs_r * K3_XP + xs_c) = (xs_v[0] && kok) ? r0v : z;
          *reinterpret_cast<u32x4*>(xt + (xs_r + 16) * K3_XP + xs_c) = (xs_v[1] && kok) ? r1v : z;
          a0 = *reinterpret_cast<const F8*>(xt + li * K3_XP + 8 * g);
          a1 = *reinterpret_cast<const F8*>(xt + (16 + li) * K3_XP + 8 * g);
        }
      }
      if (ks < NKS) {
        const int kofs = ks * 32 + 8 * g;
        F8 bh[NI], bl[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (PL) {  // Bh = this chunk's image: lane-linear units
            const char* up = reinterpret_cast<const char*>(Bh) + ((ni * K3_KS + ks) * 2) * 1024 + lane * 16;
            bh[ni] = *reinterpret_cast<const F8*>(up);
            bl[ni] = *reinterpret_cast<const F8*>(up + 1024);
          } else {
            bh[ni] = *reinterpret_cast<const F8*>(Bh + (16 * ni + li) * K3_LDB + kofs);
            bl[ni] = *reinterpret_cast<const F8*>(Bl + (16 * ni + li) * K3_LDB + kofs);
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[0][ni] = TT<T>::mma(a0, bh[ni], acc[0][ni]);
          acc[1][ni] = TT<T>::mma(a1, bh[ni], acc[1][ni]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[0][ni] = TT<T>::mma(a0, bl[ni], acc[0][ni]);
          acc[1][ni] = TT<T>::mma(a1, bl[ni], acc[1][ni]);
        }
      }
      if constexpr (MORE) {  // the fragment registers of this k-step are free again: fetch the next segment's
        if constexpr (XS) {
          af[0][ks] = load_raw_x(0, knext + ks * 32);
          af[1][ks] = load_raw_x(1, knext + ks * 32);
        } else if constexpr (FLAT) {
          af[0][ks] = load_frag_flat(0, knext + ks * 32);
          af[1][ks] = load_frag_flat(1, knext + ks * 32);
        } else {
          af[0][ks] = load_frag_x(nv0, np0, knext + ks * 32);
          af[1][ks] = load_frag_x(nv1, np1, knext + ks * 32);
        }
      }
    }
  };
  auto run_chunk = [&](int nks, auto more_tag, const T* Bh, const T* Bl, const T* np0, bool nv0, const T* np1, bool nv1,
                       long knext, long kcur) {
    switch (nks) {
      case 1: chunk(std::integral_constant<int, 1>{}, more_tag, Bh, Bl, np0, nv0, np1, nv1, knext, kcur); break;
      case 2: chunk(std::integral_constant<int, 2>{}, more_tag, Bh, Bl, np0, nv0, np1, nv1, knext, kcur); break;
      case 3: chunk(std::integral_constant<int, 3>{}, more_tag, Bh, Bl, np0, nv0, np1, nv1, knext, kcur); break;
      case 4: chunk(std::integral_constant<int, (K3_KS < 4 ? K3_KS : 4)>{}, more_tag, Bh, Bl, np0, nv0, np1, nv1, knext, kcur); break;
      default: chunk(std::integral_constant<int, K3_KS>{}, more_tag, Bh, Bl, np0, nv0, np1, nv1, knext, kcur); break;
    }
  };
  static_assert(K3_KS == 5 || K3_KS == 3, "run_chunk dispatch assumes 3 or 5 k-steps per chunk");

  // segments: (tap, chunk of K3_KC within the K columns of the tap); one LDS w2 tile per segment, double buffered
  const int cpt = (Kloop + K3_KC - 1) / K3_KC;
  const int nseg = taps * cpt;
  int buf = 0, tap = 0;
  long k0 = 0;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const long krem = Kloop - k0;
    const int nks = krem >= K3_KC ? K3_KS : (int)((krem + 31) / 32);
    const T* Bh = Bbase + buf * 2 * PLANE;
    const T* Bl = Bh + PLANE;
    if (sgm + 1 < nseg) {
      int ntap = tap;
      long nk0 = k0 + K3_KC;
      if (nk0 >= Kloop) {
        nk0 = 0;
        ++ntap;
      }
      const T *np0 = nullptr, *np1 = nullptr;
      bool nv0 = false, nv1 = false;
      if constexpr (!FLAT) {
        row_src(0, ntap, np0, nv0);
        row_src(1, ntap, np1, nv1);
      }
      const float* w2n = (GATHER && !FLAT) ? a.w2 + (long)ntap * a.gat.s2t : a.w2;
      T* Nh = Bbase + (buf ^ 1) * 2 * PLANE;
      if constexpr (PL) issue_planes(nk0, reinterpret_cast<char*>(Nh));  // lands under this chunk's MFMAs
      else k3_load_w2<TQ, K3_KC>(raw, w2mode, w2n, a.s2n, a.s2k, n0, N, nk0, Kloop, a.x);
      run_chunk(nks, std::true_type{}, Bh, Bl, np0, nv0, np1, nv1, nk0, k0);
      if constexpr (PL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else k3_store_w2<T, TQ, K3_KC>(Nh, Nh + PLANE, raw, w2mode, w2n, a.s2n, a.s2k, n0, N, nk0, Kloop);
      __syncthreads();
      buf ^= 1;
      tap = ntap;
      k0 = nk0;
    } else {
      run_chunk(nks, std::false_type{}, Bh, Bl, nullptr, false, nullptr, false, 0, k0);
    }
  }
}

// The body is a device function so that the fused backward launch (kron_bwd_fused_kernel) can run it as one role.
// `bx`, `by`: tile coordinates (M tile, N tile).  `smem`: kron3_lds_bytes(NI, K > K3_KC ? 2 : 1) bytes, 16-byte aligned.
template <typename T, int NI, bool WITH_DW1, int GM, bool BASE = false, bool PL = false>
__device__ __forceinline__ void kron3_body(const KronArgs& a, char* smem, int bx, int by, int nbx) {
  static_assert(!(WITH_DW1 && BASE), "base + delta is a forward epilogue");
  constexpr int MI = 2, TQ = 16 * NI;
  using F4 = typename Mma16<T>::frag;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int G = a.Gin, N = a.N;
  const int lg = 31 - __builtin_clz((unsigned)G);  // G is a power of two (16 % G == 0)
  const int TM = K3_RT >> lg;
  const long n0 = (long)by * TQ;
  const long row0 = ((long)bx * TM) << lg;
  long rows_end = row0 + K3_RT;
  if (rows_end > (a.M << lg)) rows_end = a.M << lg;
  LYC_TRACE_DECL;
  LYC_STAMP(0);

  // (I (x) w1) block operand, raw fp32: lane (j = li, g) holds k = 4g .. 4g+3 of column j; converted in the epilogue
  float w1raw[4];
  {
    const int mi_ = li >> lg, po = li & (G - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = 4 * g + j;
      const float v = a.w1[po * a.s1o + (kk & (G - 1)) * a.s1i];  // always in range; selected, not branched
      w1raw[j] = ((kk >> lg) == mi_) ? v : 0.f;
    }
  }

  // backward: the xref fragments of the w1 gradient (B[k = n][j = row li], 8 bytes per (mi, ni)) are fetched while stage 1
  // runs instead of inside the epilogue (they were ~1 us of exposed latency there); masked at use
  // (plain-row kernels only: the gather variants have no registers to spare -- 2 -> 1 waves per SIMD with it).
  // forward, BASE instantiation: the same registers prefetch the frozen layer's output for the fused `base + delta`
  // epilogue (a template parameter, not a runtime test: unconditional dummy loads in the plain forward cost 5 %, measured;
  // a conditional load makes the registers a PHI of loaded / undefined and the compiler waits vmcnt(0) before stage 1).
  constexpr bool XPRE = (WITH_DW1 || BASE) && (GM == 0 || GM == 3);
  u32x2 xrv[MI][NI];
  const T* pre = WITH_DW1 ? static_cast<const T*>(a.xref) : static_cast<const T*>(a.base);
  const bool xr_vec = (WITH_DW1 || BASE) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(pre) & 7u) == 0);
  auto prefetch_xref = [&]() {
    if constexpr (XPRE) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const long R = row0 + wave * 32 + mi * 16 + li;
        const long rofs = (R >> lg) * ((long)G * N) + (R & (G - 1)) * (long)N;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const long gn = n0 + 16 * ni + 4 * g;
          const bool ok = xr_vec && R < rows_end && gn < N;
          xrv[mi][ni] = *reinterpret_cast<const u32x2*>(pre + (ok ? rofs + gn : 0));
        }
      }
    }
  };
  f32x4 acc[MI][NI];
  k3_stage1<T, NI, GM, PL>(a, smem, row0, rows_end, n0, acc, prefetch_xref);
  LYC_STAMP(4);

  // ---- epilogue, all in registers ----
  F4 a2h, a2l;
  {
    T h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_f<T>(w1raw[j], h[j], l[j]);
    a2h = *reinterpret_cast<F4*>(h);
    a2l = *reinterpret_cast<F4*>(l);
  }
  F4 ident;  // identity as a B operand: B[k = 4g+e][j = li]
  {
    T idv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) idv[e] = TT<T>::from_f((4 * g + e) == li ? 1.f : 0.f);
    ident = *reinterpret_cast<F4*>(idv);
  }
  const long ldy = (long)G * N;
  const bool y_vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & (a.out_f32 ? 15u : 7u)) == 0);
  f32x4 cdw = zero4();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const long R = row0 + wave * 32 + mi * 16 + li;  // output row (m, po) of this lane
    const bool row_ok = R < rows_end;
    const long rofs = (R >> lg) * ldy + (R & (G - 1)) * (long)N;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long gn = n0 + 16 * ni + 4 * g;  // first of this lane's 4 output columns
      T h[4], l[4];
      k3_split4<T>(acc[mi][ni], h, l);
      const F4 sh = *reinterpret_cast<F4*>(h), sl = *reinterpret_cast<F4*>(l);
      f32x4 yv = zero4();
      yv = Mma16<T>::mma(sh, a2h, yv);
      yv = Mma16<T>::mma(sl, a2h, yv);
      yv = Mma16<T>::mma(sh, a2l, yv);
      if (row_ok && gn < N) {
        if (a.out_f32) {
          float* dst = static_cast<float*>(a.y) + rofs + gn;
          const f32x4 o = {a.alpha * yv[0], a.alpha * yv[1], a.alpha * yv[2], a.alpha * yv[3]};
          if (y_vec && gn + 4 <= N) {
            *reinterpret_cast<f32x4*>(dst) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gn + e < N) dst[e] = o[e];
          }
        } else {
          T* dst = static_cast<T*>(a.y) + rofs + gn;
          T o[4];
          float bb[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (BASE) {
            {  // fused `base + delta`: fp32 add, one rounding
              const T* bp = static_cast<const T*>(a.base) + rofs + gn;
              if (XPRE && xr_vec) {  // prefetched; gn + 4 <= N here (N % 4 == 0)
                T bt[4];
                *reinterpret_cast<u32x2*>(bt) = xrv[mi][ni];
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[e] = TT<T>::to_f(bt[e]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (gn + e < N) bb[e] = TT<T>::to_f(bp[e]);
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(a.alpha * yv[e] + bb[e]);
          if (y_vec && gn + 4 <= N) {
            *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gn + e < N) dst[e] = o[e];
          }
        }
      }
      if constexpr (WITH_DW1) {
        // S1 (hi, lo) transposed through the matrix core: lane (li = row, 4g+e = n) -- exact, the values are T
        const f32x4 th = Mma16<T>::mma(sh, ident, zero4());
        const f32x4 tl = Mma16<T>::mma(sl, ident, zero4());
        T thv[4], tlv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          thv[e] = TT<T>::from_f(th[e]);
          tlv[e] = TT<T>::from_f(tl[e]);
        }
        // xref fragment: B[k = n][j = row li]
        const T* xr = static_cast<const T*>(a.xref) + rofs + gn;
        T bv[4];
        if (xr_vec) {  // gn + 4 <= N or gn >= N; prefetched from a clamped address, masked here
          const bool ok = row_ok && gn < N;
          const u32x2 z = {0u, 0u};
          u32x2 v;
          if constexpr (XPRE) v = xrv[mi][ni];
          else v = *reinterpret_cast<const u32x2*>(ok ? xr : static_cast<const T*>(a.xref));
          *reinterpret_cast<u32x2*>(bv) = ok ? v : z;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = (row_ok && gn + e < N) ? xr[e] : TT<T>::from_f(0.f);
        }
        const F4 bf = *reinterpret_cast<F4*>(bv);
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(thv), bf, cdw);
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(tlv), bf, cdw);
      }
    }
  }
  LYC_STAMP(5);
  LYC_TRACE_FLUSH();

  if constexpr (WITH_DW1) {
    // cdw: D[i = (m', u)][j = (m'', po)], lane (col j = li, rows 4g+r); only the diagonal blocks m' == m'' count.
    // Cross-wave sum through LDS (the w2 tiles are dead: every wave is past the last chunk barrier or K fit in one
    // chunk and everybody passed the first barrier... other waves may still READ the last tile -> barrier first).
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * g + r) * 16 + li] = cdw[r];
    __syncthreads();
    if (tid < G * G) {
      const int u = tid >> lg, po = tid & (G - 1);
      float s = 0.f;
      for (int b = 0; b < (16 >> lg); ++b) {
        const int e = ((b << lg) + u) * 16 + (b << lg) + po;
        s += red[e] + red[256 + e] + red[512 + e] + red[768 + e];
      }
      const long e = (long)po * a.s1o + (long)u * a.s1i;  // position in dw1 memory order
      if (a.dw1_ws != nullptr)  // partial of this workgroup; summed in fixed order by kron_dw2s_kernel's reducer slice
        a.dw1_ws[((long)by * nbx + bx) * (G * G) + e] = a.alpha * s;
      else
        __hip_atomic_fetch_add(a.dw1 + e, a.alpha * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    LYC_STAMP(6);
  }
}

template <typename T, int NI, bool WITH_DW1, int GM, bool BASE = false, bool PL = false>
__global__ __launch_bounds__(NTHREADS) void kron3_kernel(KronArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char k3_smem[];
  kron3_body<T, NI, WITH_DW1, GM, BASE, PL>(a, k3_smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

}  // namespace lyc
