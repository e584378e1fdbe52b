// lowrank.h -- rank-r (LoCon / LoRA) kernels for 16-bit activations, gfx950.
//
// Reference math (lycoris/functional/locon.py:64-85, lycoris/modules/locon.py:286-332): with down:[r, I], up:[O, r]
//   forward   t  = x . down^T              y  = alpha * t . up^T
//   backward  dt = alpha * g . up          dx = dt . down          d_up += alpha * g^T . t      d_down += dt^T . x
// Every product has one long dimension (the M activation rows), one model dimension and the tiny rank, so the cost is
// reading x / g once and writing y / dx once.  Two kernels, three launches per layer and step:
//
//   bneck_kernel  "reduce to r channels, expand again" in ONE launch (forward, and backward with the factor roles
//       swapped):  mid = alpha1 * A . F1^T ,  out = alpha2 * mid . F2^T.
//       A workgroup owns 16 MI rows.  Stage 1: its waves split K in 64-element steps; the A fragments go HBM ->
//       registers (a row is used by exactly one workgroup, 32 bytes per lane and step, a ring of D steps in flight),
//       the fp32 factor F1 is split into hi + lo on the fly (v_mfma_f32_16x16x32 twice per fragment); one cross-wave
//       reduction through LDS leaves `mid` (fp32) in LDS and in HBM (the backward pass needs it).  Stage 2: the waves
//       split the N2 output columns; computed transposed (F2 is the A operand, mid the B operand of
//       v_mfma_f32_16x16x16, both hi/lo: 3 MFMAs per tile) so that a lane ends up with 4 consecutive columns of one
//       output row -> 8-byte stores straight from the accumulators.
//   lowrank_tn_kernel  both factor gradients in one launch:  out[c, n] += alpha * sum_m Act[m, c] * Mid[m, n].
//       The contraction runs over the ROWS of two row-major matrices, which a 16-bit MFMA could only take after a
//       transpose through LDS.  v_mfma_f32_16x16x4_f32 takes ONE value per lane for (row-of-A = column c, k = row m):
//       exactly the natural layout (16 lanes walk the columns, 4 lane groups walk 4 rows), fp32 x fp32 so neither hi/lo
//       split nor LDS is needed, and at 2 M C r flops the fp32 MFMA rate is far from being the bound.  Each WAVE owns
//       a (16 CV column tile, row slab) work item: no barrier, no shared memory; slabs meet through fp32 atomics
//       (plain read-modify-write when there is a single slab).
#pragma once
#include "kron3.h"

namespace lyc {

// Keeps the global loads of a software-pipelined ring where the source puts them.  Without it LLVM merges the
// preheader loads with the re-loads at the end of the loop body and rotates them to the top of the loop ("load D steps,
// wait, compute D steps"): every iteration then exposes a full memory latency.  A compiler-level fence only -- it emits no
// instruction and does not touch the hardware counters.  The sched_barrier keeps the machine scheduler from undoing it
// (it would hoist all MFMAs and cluster the re-loads at the end of the loop body, with a vmcnt(0) at the back edge).
#define LR_LOAD_FENCE()                  \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
// Pins the first use of a loaded register to this point of the program: LLVM otherwise software-pipelines the cheap
// conversions of the NEXT iteration's values to the end of the loop body, i.e. waits for every load just issued.
#define LR_USE(x) asm volatile("" : "+v"(x))

struct BneckArgs {
  const void* A;   // [M, K1] activations (T), row pitch lda
  long lda;
  long M;
  int K1;
  const float* F1;  // element (n, k) at n * f1n + k * f1k,  n < R, k < K1
  long f1n, f1k;
  int R;
  float* mid;       // [M, R] fp32, row pitch R; nullptr = not needed
  const float* F2;  // element (n, k) at n * f2n + k * f2k,  n < N2, k < R
  long f2n, f2k;
  int N2;
  void* out;        // [M, N2], row pitch ldo; T, or fp32 when out_f32
  long ldo;
  int out_f32;
  float alpha1, alpha2;
  int nsplit;       // gridDim.y: each y slice repeats stage 1 and expands its share of the N2 column tiles
  // implicit Conv2d forward (GAT kernels): A holds the NHWC pixel rows [B * Hs * Ws, Ck]; the K1 = taps * Ck flat index
  // is (tap, channel); the A row of output pixel m and tap t is the source pixel given by gat (mode 1), zero outside
  // the image.  Ck % 16 == 0 (a lane's 16 K elements never straddle taps).
  KronGather gat;
  int Ck;
};

// hi/lo split of 8 consecutive fp32 (two float4) into two 16-bit x 8 MFMA fragments
template <typename T>
__device__ __forceinline__ void lr_split8(const f32x4& v0, const f32x4& v1, typename TT<T>::frag& hi,
                                          typename TT<T>::frag& lo) {
  T h[8] __attribute__((aligned(16))), l[8] __attribute__((aligned(16)));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    split_f<T>(v0[e], h[e], l[e]);
    split_f<T>(v1[e], h[4 + e], l[4 + e]);
  }
  hi = *reinterpret_cast<const typename TT<T>::frag*>(h);
  lo = *reinterpret_cast<const typename TT<T>::frag*>(l);
}

template <typename T>
__device__ __forceinline__ void lr_split4(const f32x4& v, typename Mma16<T>::frag& hi, typename Mma16<T>::frag& lo) {
  T h[4] __attribute__((aligned(8))), l[4] __attribute__((aligned(8)));
#pragma unroll
  for (int e = 0; e < 4; ++e) split_f<T>(v[e], h[e], l[e]);
  hi = *reinterpret_cast<const typename Mma16<T>::frag*>(h);
  lo = *reinterpret_cast<const typename Mma16<T>::frag*>(l);
}

// staged stage 1: K elements per wave chunk (128 for the smallest tile, 64 otherwise: static LDS <= 64 KiB), LDS row
// pitch = chunk + 16 elements (8 mod 16 dwords: conflict-free ds_read_b128 fragments)
__host__ __device__ constexpr int bneck_kc(int MI, int RT) { return (16 * MI + 32 * RT <= 48) ? 128 : 64; }
template <int NW, int MI, int RT, bool STG = false>
__host__ __device__ constexpr int bneck_lds_bytes() {
  return NW * MI * RT * 256 * 4 + 16 * MI * (16 * RT + 4) * 4 +
         (STG ? NW * (16 * MI + 32 * RT) * (bneck_kc(MI, RT) + 16) * 2 : 0);
}

// NW waves per workgroup, MI 16-row tiles per workgroup, RT 16-wide rank tiles (R <= 16 RT).
// F1V: F1 rows are K-contiguous and 16-byte aligned (forward: down[r, I]); otherwise element-wise loads with the given
// strides (backward: up^T -- 16 lanes walk the contiguous rank index).  F2V likewise for F2 along k (forward: up[O, r]).
// STG (needs F1V, not with GAT): stage 1 through a per-wave LDS stage with COALESCED global loads -- fragment-direct loads
// give each lane of a quad a different cache line and cost the address unit ~64 cycles per wave instruction instead of 16.
template <typename T, int NW, int MI, int RT, bool F1V, bool F2V, bool GAT = false, bool STG = false>
__device__ __forceinline__ void bneck_body(const BneckArgs& a, const int bx, const int by, const int nby) {
  static_assert(!STG || (F1V && !GAT), "staged stage 1: vector factor layout, no gather");
  constexpr int D = RT == 4 ? 2 : 3;  // stage-1 steps in flight per wave (register budget)
  constexpr int D2 = RT == 1 ? 8 : RT == 2 ? 4 : 2;  // stage-2 column tiles in flight per wave
  constexpr int RP = 16 * RT + 4; // LDS row pitch of mid (floats)
  __shared__ __attribute__((aligned(16))) char smem[bneck_lds_bytes<NW, MI, RT, STG>()];
  float* red = reinterpret_cast<float*>(smem);                         // [NW][MI * RT][256]
  float* mids = red + NW * MI * RT * 256;                              // [16 MI][RP]
  T* stage = reinterpret_cast<T*>(mids + 16 * MI * (16 * RT + 4));     // STG: [NW][16 MI + 32 RT][BN_LDP]
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long m0 = (long)bx * (16 * MI);
  const T* A = static_cast<const T*>(a.A);
  const int K1 = a.K1, R = a.R;
  LYC_TRACE_DECL;
  LYC_STAMP(0);

  // ---------------- stage 1: acc[mi][rt] = sum_k A[m, k] * F1[n, k] over this wave's k-steps ----------------
  const T* arow[MI];
  unsigned long long tmask[MI];  // GAT: taps of this lane's pixel that fall inside the image
  __shared__ int lr_tapoff[GAT ? 64 : 1];  // GAT: source-row offset of tap t relative to tap 0
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    long r = m0 + 16 * mi + li;
    if (r >= a.M) r = a.M - 1;  // rows past the end only feed accumulator rows that are never stored
    if constexpr (!GAT) {
      arow[mi] = A + r * a.lda;
      tmask[mi] = 0ull;
    } else {
      const int hw = a.gat.Hd * a.gat.Wd;
      const int pb = (int)(r / hw);
      const int rem = (int)(r - (long)pb * hw);
      const int ho = rem / a.gat.Wd, wo = rem - ho * a.gat.Wd;
      const int h0 = ho * a.gat.sh - a.gat.ph, w0 = wo * a.gat.sw - a.gat.pw;
      const int kh = a.gat.taps / a.gat.kw;
      unsigned long long wm = 0ull, m = 0ull;
      for (int j = 0; j < a.gat.kw; ++j) {
        const int ws = w0 + j * a.gat.dw;
        if (ws >= 0 && ws < a.gat.Ws) wm |= 1ull << j;
      }
      for (int i = 0; i < kh; ++i) {
        const int hs = h0 + i * a.gat.dh;
        if (hs >= 0 && hs < a.gat.Hs) m |= wm << (i * a.gat.kw);
      }
      tmask[mi] = m;
      arow[mi] = A + (((long)pb * a.gat.Hs + h0) * a.gat.Ws + w0) * a.Ck;  // tap 0's pixel (may lie outside: masked)
    }
  }
  if constexpr (GAT) {
    if (tid < a.gat.taps) {
      const int i = tid / a.gat.kw, j = tid - i * a.gat.kw;
      lr_tapoff[tid] = i * a.gat.dh * a.gat.Ws + j * a.gat.dw;
    }
    __syncthreads();
  }
  const float* frow[RT];
  bool fok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int n = 16 * rt + li;
    fok[rt] = n < R;
    frow[rt] = a.F1 + (fok[rt] ? n : 0) * a.f1n;
  }
  struct Step {
    u32x4 av[MI][2];
    f32x4 fv[RT][4];
    bool k0ok[MI], k1ok[MI];
  };
  // Loads are unconditional from clamped (valid) addresses; the k >= K1 (and, GAT, outside-the-image) mask is applied to
  // A in compute_step, one ring round later (a select next to the load would make the compiler wait for the load right
  // there).  F1 needs no k mask: where k >= K1 the A fragment is zero and the clamped F1 data is ordinary factor data.
  auto load_step = [&](Step& S, int s) {
    const int kk = s * 64 + 16 * g;  // this lane: k = kk .. kk + 15
    const bool k0in = kk < K1, k1in = kk + 8 < K1;  // K1 % 8 == 0: a 16-byte piece is all in or all out
    int tap = 0, ck = kk;
    long toff = 0;
    if constexpr (GAT) {
      tap = k0in ? kk / a.Ck : 0;  // Ck % 16 == 0: both pieces lie in one tap
      ck = kk - tap * a.Ck;
      toff = (long)lr_tapoff[tap] * a.Ck + ck;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      bool v = true;
      if constexpr (GAT) v = (tmask[mi] >> tap) & 1ull;
      const bool o0 = k0in && v, o1 = k1in && v;
      S.k0ok[mi] = o0;
      S.k1ok[mi] = o1;
      const T* p = GAT ? arow[mi] + toff : arow[mi] + kk;
      S.av[mi][0] = *reinterpret_cast<const u32x4*>(o0 ? p : A);
      S.av[mi][1] = *reinterpret_cast<const u32x4*>(o1 ? p + 8 : A);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if constexpr (F1V) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = kk + 4 * q < K1;
          S.fv[rt][q] = *reinterpret_cast<const f32x4*>(frow[rt] + (ok ? kk + 4 * q : 0));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = kk + 4 * q + e;
            S.fv[rt][q][e] = frow[rt][(long)(k < K1 ? k : 0) * a.f1k];
          }
      }
    }
  };
  f32x4 acc[MI][RT];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[mi][rt] = zero4();
  auto compute_step = [&](Step& S) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      LR_USE(S.av[mi][0]);
      LR_USE(S.av[mi][1]);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) LR_USE(S.fv[rt][q]);
      F8 bh0, bl0, bh1, bl1;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      lr_split8<T>(fok[rt] ? S.fv[rt][0] : z, fok[rt] ? S.fv[rt][1] : z, bh0, bl0);
      lr_split8<T>(fok[rt] ? S.fv[rt][2] : z, fok[rt] ? S.fv[rt][3] : z, bh1, bl1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const u32x4 zu = {0u, 0u, 0u, 0u};
        const u32x4 m0v = S.k0ok[mi] ? S.av[mi][0] : zu, m1v = S.k1ok[mi] ? S.av[mi][1] : zu;
        const F8 a0 = *reinterpret_cast<const F8*>(&m0v);
        const F8 a1 = *reinterpret_cast<const F8*>(&m1v);
        acc[mi][rt] = TT<T>::mma(a0, bh0, acc[mi][rt]);
        acc[mi][rt] = TT<T>::mma(a0, bl0, acc[mi][rt]);
        acc[mi][rt] = TT<T>::mma(a1, bh1, acc[mi][rt]);
        acc[mi][rt] = TT<T>::mma(a1, bl1, acc[mi][rt]);
      }
    }
  };
  const int nsteps = (K1 + 63) >> 6;
  if constexpr (STG) {
    // chunk c = K elements [KC c, KC c + KC): this wave takes chunks wave, wave + NW, ...
    //   A  : KC / 8 lanes walk a row (16 bytes each: KC * 2 contiguous bytes), 64 / (KC / 8) rows per instruction
    //   F1 : KC / 4 lanes walk a row (16 bytes each: KC * 4 contiguous bytes), 64 / (KC / 4) rows per instruction
    // registers -> private LDS tiles [row][k] (A as is, F1 split into hi / lo planes) -> ds_read_b128 fragments.
    constexpr int BN_KC = bneck_kc(MI, RT), BN_LDP = BN_KC + 16;
    constexpr int APL = BN_KC / 8, ARW = 64 / APL, NA = 16 * MI / ARW;
    constexpr int FPL = BN_KC / 4, FRW = 64 / FPL, NF = 16 * RT / FRW;
    T* sA = stage + wave * (16 * MI + 32 * RT) * BN_LDP;
    T* sFh = sA + 16 * MI * BN_LDP;
    T* sFl = sFh + 16 * RT * BN_LDP;
    const int ar = lane / APL, ac = 8 * (lane % APL);
    const int fr = lane / FPL, fc = 4 * (lane % FPL);
    const T* aptr[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      long r = m0 + ar + ARW * j;
      if (r >= a.M) r = a.M - 1;
      aptr[j] = A + r * a.lda + ac;
    }
    const float* fptr[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int n = fr + FRW * j;
      fptr[j] = a.F1 + (n < R ? n : 0) * a.f1n + fc;
    }
    struct Chunk {
      u32x4 av[NA];
      f32x4 fv[NF];
    };
    const int nchunks = (K1 + BN_KC - 1) / BN_KC;
    auto load_chunk = [&](Chunk& C, int c) {
      const int k0 = c * BN_KC;
      const bool aok = k0 + ac < K1, fok4 = k0 + fc < K1;  // K1 % 8 == 0: pieces are all in or all out
#pragma unroll
      for (int j = 0; j < NA; ++j) C.av[j] = *reinterpret_cast<const u32x4*>(aok ? aptr[j] + k0 : aptr[j] - ac);
#pragma unroll
      for (int j = 0; j < NF; ++j) C.fv[j] = *reinterpret_cast<const f32x4*>(fok4 ? fptr[j] + k0 : fptr[j] - fc);
    };
    auto write_chunk = [&](Chunk& C, int c) {
      const int k0 = c * BN_KC;
      const bool aok = k0 + ac < K1, fok4 = k0 + fc < K1;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        LR_USE(C.av[j]);
        const u32x4 zu = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(sA + (ar + ARW * j) * BN_LDP + ac) = aok ? C.av[j] : zu;
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        LR_USE(C.fv[j]);
        const bool ok = fok4 && (fr + FRW * j) < R;
        T h4[4] __attribute__((aligned(8))), l4[4] __attribute__((aligned(8)));
#pragma unroll
        for (int e = 0; e < 4; ++e) split_f<T>(ok ? C.fv[j][e] : 0.f, h4[e], l4[e]);
        *reinterpret_cast<u32x2*>(sFh + (fr + FRW * j) * BN_LDP + fc) = *reinterpret_cast<const u32x2*>(h4);
        *reinterpret_cast<u32x2*>(sFl + (fr + FRW * j) * BN_LDP + fc) = *reinterpret_cast<const u32x2*>(l4);
      }
    };
    auto compute_chunk = [&]() {
#pragma unroll
      for (int ks = 0; ks < BN_KC / 32; ++ks) {
        F8 af[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const F8*>(sA + (16 * mi + li) * BN_LDP + 32 * ks + 8 * g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const F8 bh = *reinterpret_cast<const F8*>(sFh + (16 * rt + li) * BN_LDP + 32 * ks + 8 * g);
          const F8 bl = *reinterpret_cast<const F8*>(sFl + (16 * rt + li) * BN_LDP + 32 * ks + 8 * g);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            acc[mi][rt] = TT<T>::mma(af[mi], bh, acc[mi][rt]);
            acc[mi][rt] = TT<T>::mma(af[mi], bl, acc[mi][rt]);
          }
        }
      }
    };
    Chunk c0, c1;
    load_chunk(c0, wave);
    LR_LOAD_FENCE();
    load_chunk(c1, wave + NW);
    LR_LOAD_FENCE();
    for (int c = wave; c < nchunks; c += 2 * NW) {
      write_chunk(c0, c);
      LR_LOAD_FENCE();
      load_chunk(c0, c + 2 * NW);
      LR_LOAD_FENCE();
      compute_chunk();
      LR_LOAD_FENCE();
      if (c + NW < nchunks) {
        write_chunk(c1, c + NW);
        LR_LOAD_FENCE();
        load_chunk(c1, c + 3 * NW);
        LR_LOAD_FENCE();
        compute_chunk();
        LR_LOAD_FENCE();
      }
    }
  } else {
    Step st[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      load_step(st[j], wave + j * NW);
      LR_LOAD_FENCE();
    }
    for (int s = wave; s < nsteps; s += D * NW) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        compute_step(st[j]);  // steps past the end hold zeros
        LR_LOAD_FENCE();
        load_step(st[j], s + (j + D) * NW);
        LR_LOAD_FENCE();
      }
    }
  }

  // this y slice's column tiles: [tbeg, ntiles)
  const int ntiles_all = (a.N2 + 15) >> 4;
  const int tper = (ntiles_all + nby - 1) / nby;
  const int tbeg = by * tper;
  const int ntiles = tbeg + tper < ntiles_all ? tbeg + tper : ntiles_all;
  // F2 needs no masks: columns k >= R of mid are zero (their F1 rows were zeroed), tiles / columns past the end are never
  // stored; the loads only have to come from valid (clamped) addresses.
  auto load_f2 = [&](f32x4 (&fr)[RT], int tile) {
    const int n = 16 * tile + li;
    const float* p = a.F2 + ((tile < ntiles && n < a.N2) ? n : 0) * a.f2n;
#pragma unroll
    for (int kt = 0; kt < RT; ++kt) {
      const int k = 16 * kt + 4 * g;
      if constexpr (F2V) {  // R % 4 == 0: all in or all out
        fr[kt] = *reinterpret_cast<const f32x4*>(p + (k < R ? k : 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) fr[kt][e] = p[(long)(k + e < R ? k + e : 0) * a.f2k];
      }
    }
  };
  // The first D2 factor tiles of the expand stage are requested now, before the cross-wave sum: they do not depend on it
  // and their L2 round trip would otherwise be exposed at the start of stage 2 (results return in issue order: behind
  // the stage-1 loads, which are all consumed by now).
  f32x4 fr[D2][RT];
  if (a.out != nullptr) {
#pragma unroll
    for (int j = 0; j < D2; ++j) {
      load_f2(fr[j], tbeg + wave + j * NW);
      LR_LOAD_FENCE();
    }
  }
  LYC_STAMP(1);
  // ---------------- cross-wave sum -> mid (LDS, and HBM for the backward pass) ----------------
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      *reinterpret_cast<f32x4*>(red + ((wave * MI * RT + mi * RT + rt) * 256 + lane * 4)) = acc[mi][rt];
  __syncthreads();
  for (int e = tid; e < MI * RT * 256; e += NW * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * MI * RT * 256 + e];
    s *= a.alpha1;
    const int t = e >> 8, l = (e >> 2) & 63, q = e & 3;  // accumulator element: column l & 15, row 4 (l >> 4) + q
    const int m = 16 * (t / RT) + 4 * (l >> 4) + q, n = 16 * (t % RT) + (l & 15);
    mids[m * RP + n] = s;
    if (a.mid != nullptr && by == 0 && m0 + m < a.M && n < R) a.mid[(m0 + m) * R + n] = s;
  }
  if (a.out == nullptr) return;  // only mid wanted (no input gradient)
  __syncthreads();
  LYC_STAMP(2);

  // ---------------- stage 2: out[m, n] = alpha2 * sum_k mid[m, k] * F2[n, k], waves split the n tiles ----------------
  F4 bh[MI][RT], bl[MI][RT];  // B operand: lane (m = li, k = 16 kt + 4 g .. + 3)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int kt = 0; kt < RT; ++kt) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(mids + (16 * mi + li) * RP + 16 * kt + 4 * g);
      lr_split4<T>(v, bh[mi][kt], bl[mi][kt]);
    }
  const bool out_vec = a.out_f32 ? ((a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0)
                                 : ((a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7u) == 0);
  // FULL: every tile has 16 valid columns and the rows can take vector stores -> one predicated store per tile; the
  // general version (ragged N2 / unaligned rows) goes element-wise.  Chosen once per launch (uniform), outside the loop:
  // few branches inside the loop keep the counted vmcnt waits of the F2 ring exact.
  auto do_tile = [&](auto full_tag, f32x4 (&fr)[RT], int tile) {
    constexpr bool FULL = decltype(full_tag)::value;
    F4 ah[RT], al[RT];
#pragma unroll
    for (int kt = 0; kt < RT; ++kt) LR_USE(fr[kt]);
#pragma unroll
    for (int kt = 0; kt < RT; ++kt) lr_split4<T>(fr[kt], ah[kt], al[kt]);
    const int n = 16 * tile + 4 * g;  // this lane's 4 output columns
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f32x4 y = zero4();
#pragma unroll
      for (int kt = 0; kt < RT; ++kt) {
        y = Mma16<T>::mma(ah[kt], bh[mi][kt], y);
        y = Mma16<T>::mma(al[kt], bh[mi][kt], y);
        y = Mma16<T>::mma(ah[kt], bl[mi][kt], y);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] *= a.alpha2;
      const long m = m0 + 16 * mi + li;
      const bool ok = tile < ntiles && m < a.M;
      if constexpr (FULL) {
        if (a.out_f32) {
          if (ok) *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + m * a.ldo + n) = y;
        } else {
          T v[4] __attribute__((aligned(8)));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = TT<T>::from_f(y[e]);
          if (ok) *reinterpret_cast<u32x2*>(static_cast<T*>(a.out) + m * a.ldo + n) = *reinterpret_cast<const u32x2*>(v);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ok && n + e < a.N2) {
            if (a.out_f32)
              static_cast<float*>(a.out)[m * a.ldo + n + e] = y[e];
            else
              static_cast<T*>(a.out)[m * a.ldo + n + e] = TT<T>::from_f(y[e]);
          }
      }
    }
  };
  auto stage2 = [&](auto full_tag) {
    LYC_STAMP(3);
    for (int tile = tbeg + wave; tile < ntiles; tile += D2 * NW) {
#pragma unroll
      for (int j = 0; j < D2; ++j) {
        do_tile(full_tag, fr[j], tile + j * NW);
        LR_LOAD_FENCE();
        load_f2(fr[j], tile + (j + D2) * NW);
        LR_LOAD_FENCE();
      }
    }
  };
  if (out_vec && (a.N2 & 15) == 0)
    stage2(std::true_type{});
  else
    stage2(std::false_type{});
  LYC_STAMP(4);
  LYC_TRACE_FLUSH();
}

template <typename T, int NW, int MI, int RT, bool F1V, bool F2V, bool GAT = false, bool STG = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void bneck_kernel(BneckArgs a) {
  bneck_body<T, NW, MI, RT, F1V, F2V, GAT, STG>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// Several problems of ONE shape in one launch (round 5: the to_q / to_k / to_v adapters of an attention block read one tensor --
// reference: one LoConModule.forward per projection, modules/locon.py:309-332): blockIdx.z selects the problem.  A 1024-row LoCon layer
// is 64 row tiles x 4 column slices; three of them planned together are 64 x 1-2 slices each and ONE launch instead of three.
constexpr int BNECK_GROUP_MAX = 4;
struct BneckGroupArgs {
  int n;
  BneckArgs p[BNECK_GROUP_MAX];
};
static_assert(sizeof(BneckGroupArgs) <= 3840, "kernel arguments are limited to 4 KiB");
template <typename T, int NW, int MI, int RT, bool F1V, bool F2V, bool STG = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void bneck_group_kernel(BneckGroupArgs ga) {
  bneck_body<T, NW, MI, RT, F1V, F2V, false, STG>(ga.p[blockIdx.z], (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// ------------------------------------------------------------------------------------------------------------------

struct LowrankTnProb {
  const void* act;   // [M, C] activations (T), row pitch ld
  long ld;
  int C;
  const float* mid;  // [M, R] fp32, row pitch R
  float* out;        // element (c, n) at c * os + n * oj
  long os, oj;
  float alpha;
  int tiles;         // ceil(C / (16 CV))
  int swap;          // 1: the output is contiguous along c (os == 1): compute the transposed tile so that the 16 lanes of
                     //    an atomic instruction walk c instead of n (4 cache lines per instruction instead of 64)
};

struct LowrankTnArgs {
  LowrankTnProb p[2];  // p[1].tiles == 0: single problem
  long M;
  int R;
  int nsplit;          // row slabs
  long rows_per_slab;  // multiple of 4
  int force_atomic;    // != 0: an output may be shared with another problem of the same (grouped) launch
  // implicit Conv2d (GAT kernels), problem p[1] only: its act holds the NHWC input pixel rows [B * Hs * Ws, Ct], its C
  // columns are the flat (tap, channel) index (C = taps * Ct, Ct % (16 CV) == 0) and the Act row of output pixel m and tap
  // t is the source pixel given by gat (mode 1), zero outside the image
  KronGather gat;
  int Ct;
};

template <typename T, int CV>
struct LrVec;
template <typename T>
struct LrVec<T, 8> { typedef u32x4 type; };
template <typename T>
struct LrVec<T, 4> { typedef u32x2 type; };
template <typename T>
struct LrVec<T, 2> { typedef uint32_t type; };
template <typename T>
struct LrVec<T, 1> { typedef uint16_t type; };

// One wave per (column tile of 16 CV, row slab).  CV in {1, 2, 4, 8} columns per lane (C % CV == 0, ld % CV == 0, aligned).
// `w`: index of this wave among the problem's (tiles x slabs) work items
template <typename T, int RT, int CV, bool GAT>
__device__ __forceinline__ void lowrank_tn_body(const LowrankTnArgs& a, const long w) {
  constexpr int D = CV * RT <= 4 ? 16 : 8;  // 4-row steps in flight
  using V = typename LrVec<T, CV>::type;
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const int tiles_total = a.p[0].tiles + a.p[1].tiles;
  if (w >= (long)tiles_total * a.nsplit) return;
  const int slab = (int)(w / tiles_total);
  int tile = (int)(w - (long)slab * tiles_total);
  const bool second = tile >= a.p[0].tiles;
  if (second) tile -= a.p[0].tiles;
  const T* act = static_cast<const T*>(second ? a.p[1].act : a.p[0].act);
  const long ld = second ? a.p[1].ld : a.p[0].ld;
  const int C = second ? a.p[1].C : a.p[0].C;
  const float* mid = second ? a.p[1].mid : a.p[0].mid;
  float* out = second ? a.p[1].out : a.p[0].out;
  const long os = second ? a.p[1].os : a.p[0].os, oj = second ? a.p[1].oj : a.p[0].oj;
  const float alpha = second ? a.p[1].alpha : a.p[0].alpha;
  const bool swap = (second ? a.p[1].swap : a.p[0].swap) != 0;
  const int R = a.R;
  const bool gat = GAT && second;

  const long rbeg = (long)slab * a.rows_per_slab;
  long rend = rbeg + a.rows_per_slab;
  if (rend > a.M) rend = a.M;
  if (rbeg >= rend) return;
  const int c0 = tile * 16 * CV;
  const int cl = c0 + CV * li;        // this lane's first column
  const bool cok = cl < C;            // C % CV == 0: all CV columns in or out
  // GAT: the tile lies inside one tap; the lane walks the output pixels of its slab and keeps (image, h, w) of the pixel
  // of its NEXT load incrementally (load_step is called with r increasing by 4 each time)
  int tap_i = 0, tap_j = 0, gb = 0, gh = 0, gw = 0;
  const T* abase;
  if (gat) {
    const int tap = c0 / a.Ct;
    tap_i = tap / a.gat.kw;
    tap_j = tap - tap_i * a.gat.kw;
    abase = act + (cok ? cl - tap * a.Ct : 0);
    const long m = rbeg + g;
    const int hw = a.gat.Hd * a.gat.Wd;
    gb = (int)(m / hw);
    const int rem = (int)(m - (long)gb * hw);
    gh = rem / a.gat.Wd;
    gw = rem - gh * a.gat.Wd;
  } else {
    abase = act + (cok ? cl : 0);
  }
  bool nok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) nok[rt] = 16 * rt + li < R;

  struct Step {
    V av;
    float mv[RT];
    bool ok;
  };
  // Both loads are unconditional, from clamped (always valid) addresses, and the row / column mask is applied to the
  // converted A value one ring round later: a select next to the load makes hipcc predicate the load and wait for it
  // with vmcnt(0), which serialises the whole ring.  Rank indices >= R read column 0 and only feed accumulator columns
  // that are never stored.
  auto load_step = [&](Step& S, long r) {  // rows r .. r + 3, this lane: row r + g
    long m = r + g;
    bool ok = m < rend && cok;
    if (m >= rend) m = rbeg;
    long arow = m;
    if (GAT && gat) {
      const int hs = gh * a.gat.sh - a.gat.ph + tap_i * a.gat.dh, ws = gw * a.gat.sw - a.gat.pw + tap_j * a.gat.dw;
      ok = ok && hs >= 0 && hs < a.gat.Hs && ws >= 0 && ws < a.gat.Ws;
      arow = ok ? ((long)gb * a.gat.Hs + hs) * a.gat.Ws + ws : 0;
      gw += 4;  // the pixel of the next call
      while (gw >= a.gat.Wd) {
        gw -= a.gat.Wd;
        if (++gh >= a.gat.Hd) {
          gh = 0;
          ++gb;
        }
      }
    }
    S.ok = ok;
    S.av = *reinterpret_cast<const V*>(abase + arow * ld);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) S.mv[rt] = mid[m * R + (nok[rt] ? 16 * rt + li : 0)];
  };
  f32x4 acc[CV][RT];
#pragma unroll
  for (int j = 0; j < CV; ++j)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[j][rt] = zero4();
  auto compute_step = [&](Step& S) {
    LR_USE(S.av);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) LR_USE(S.mv[rt]);
    const T* e = reinterpret_cast<const T*>(&S.av);
#pragma unroll
    for (int j = 0; j < CV; ++j) {
      const float af = S.ok ? TT<T>::to_f(e[j]) : 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float mv = S.mv[rt];
        acc[j][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(swap ? mv : af, swap ? af : mv, acc[j][rt], 0, 0, 0);
      }
    }
  };
  {
    Step st[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      load_step(st[j], rbeg + 4 * j);
      LR_LOAD_FENCE();
    }
    for (long r = rbeg; r < rend; r += 4 * D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        compute_step(st[j]);
        LR_LOAD_FENCE();
        load_step(st[j], r + 4 * (j + D));
        LR_LOAD_FENCE();
      }
    }
  }
  // accumulator element (j, rt)[q]: column c = c0 + CV * (4 g + q) + j, rank index n = 16 rt + li;
  // swapped:                           column c = c0 + CV * li + j,          rank index n = 16 rt + 4 g + q
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        const int n = 16 * rt + (swap ? 4 * g + q : li);
        const int c = c0 + CV * (swap ? li : 4 * g + q) + j;
        if (n >= R || c >= C) continue;
        float* o = out + (long)c * os + (long)n * oj;
        const float v = acc[j][rt][q] * alpha;
        if (a.nsplit == 1 && !a.force_atomic)
          *o += v;
        else
          __hip_atomic_fetch_add(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
}

template <typename T, int RT, int CV, bool GAT = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 4))) void lowrank_tn_kernel(LowrankTnArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lowrank_tn_body<T, RT, CV, GAT>(a, (long)blockIdx.x * NWAVES + wave);
}

// ---- grouped launch: the factor gradients of up to TNG_MAX rank-r layers in ONE grid -------------------------------------
// Like kron_dw2s_group_kernel (kron_dw2s.h): d_down / d_up feed the optimizer only, the host parks (g, x, t, dt) of the
// finished layers and hands batches to this kernel.  A work item here is one WAVE (a 16 CV column tile x a row slab) with no
// cross-wave cooperation, so a batch is simply more independent waves per launch: the single-layer launch is a few hundred
// waves whose time is one wave's serial chain plus the launch; a batch keeps 4 waves per SIMD busy and can afford the widest
// column vector (16-byte loads) and long slabs (few atomics), because its parallelism comes from the other layers.
constexpr int TNG_MAX = 18;
struct LowrankTnItem {
  LowrankTnProb p[2];
  long M;
  long rows_per_slab;
  int R, nsplit, force_atomic;
};
struct LowrankTnGroupArgs {
  int n;
  int wg_end[TNG_MAX];  // exclusive prefix of the workgroup counts (a workgroup = NWAVES work items of ONE problem)
  LowrankTnItem p[TNG_MAX];
};
static_assert(sizeof(LowrankTnGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

template <typename T, int RT, int CV>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 4))) void lowrank_tn_group_kernel(LowrankTnGroupArgs ga) {
  const int b = (int)blockIdx.x;
  int q = 0;
  while (q + 1 < ga.n && b >= ga.wg_end[q]) ++q;
  const int b0 = q ? ga.wg_end[q - 1] : 0;
  const LowrankTnItem& it = ga.p[q];
  LowrankTnArgs a{};
  a.p[0] = it.p[0];
  a.p[1] = it.p[1];
  a.M = it.M; a.R = it.R; a.nsplit = it.nsplit; a.rows_per_slab = it.rows_per_slab; a.force_atomic = it.force_atomic;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lowrank_tn_body<T, RT, CV, false>(a, (long)(b - b0) * NWAVES + wave);
}

// ------------------------------------------------------------------------------------------------------------------
// Input gradient of the implicit Conv2d rank-r layer ("gather, then expand"):
//   dx[p, c] = sum_{tap, n} dt[src(p, tap), n] * down[n, tap, c]          p = input pixel, src = output pixel (gat mode 2)
// A workgroup owns 16 input pixels: it gathers their taps * R intermediate values into LDS (hi / lo planes, zero where
// the tap falls outside the output or between its strides) and then runs the expand stage of bneck_kernel over the C
// output channels with K = taps * R (<= 144) -- no col2im, no atomics.
struct GexpArgs {
  const float* mid;  // dt [B * Hs * Ws, R] (source = OUTPUT pixels)
  const float* F2;   // down as [R][taps][C]: element (n, tap, c) at (n * taps + tap) * C + c
  void* out;         // dx rows [M, C] (destination = INPUT pixels), T
  long M;
  int R, C;
  KronGather gat;    // mode 2
};

constexpr int GEXP_KT = 9;           // k tiles of 16: taps * R <= 144
constexpr int GEXP_KP = 16 * GEXP_KT + 8;  // LDS row pitch (elements): 8-byte fragments, rows 304 bytes apart

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void gexp_kernel(GexpArgs a) {
  constexpr int D2 = 2;
  __shared__ __attribute__((aligned(16))) T midh[16 * GEXP_KP];
  __shared__ __attribute__((aligned(16))) T midl[16 * GEXP_KP];
  using F4 = typename Mma16<T>::frag;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long m0 = (long)blockIdx.x * 16;
  const int R = a.R, C = a.C, taps = a.gat.taps;
  const int K2 = taps * R, KT = (K2 + 15) >> 4;  // R % 4 == 0

  // gather: piece (m, tap, q) = 4 consecutive rank values of one tap of one pixel
  for (int idx = tid; idx < 16 * GEXP_KT * 4; idx += NW * 64) {
    const int m = idx / (GEXP_KT * 4), kq = idx - m * (GEXP_KT * 4);  // k = 4 kq
    const int tap = (4 * kq) / R, nn = 4 * kq - tap * R;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const long pix = m0 + m;
    if (4 * kq < K2 && pix < a.M) {
      const int hw = a.gat.Hd * a.gat.Wd;
      const int pb = (int)(pix / hw);
      const int rem = (int)(pix - (long)pb * hw);
      const int h = rem / a.gat.Wd, w = rem - h * a.gat.Wd;
      const int i = tap / a.gat.kw, j = tap - i * a.gat.kw;
      const int hn = h + a.gat.ph - i * a.gat.dh, wn = w + a.gat.pw - j * a.gat.dw;
      const int hs = hn / a.gat.sh, ws = wn / a.gat.sw;
      if (hn >= 0 && wn >= 0 && hs * a.gat.sh == hn && ws * a.gat.sw == wn && hs < a.gat.Hs && ws < a.gat.Ws)
        v = *reinterpret_cast<const f32x4*>(a.mid + (((long)pb * a.gat.Hs + hs) * a.gat.Ws + ws) * R + nn);
    }
    T h4[4] __attribute__((aligned(8))), l4[4] __attribute__((aligned(8)));
#pragma unroll
    for (int e = 0; e < 4; ++e) split_f<T>(v[e], h4[e], l4[e]);
    *reinterpret_cast<u32x2*>(midh + m * GEXP_KP + 4 * kq) = *reinterpret_cast<const u32x2*>(h4);
    *reinterpret_cast<u32x2*>(midl + m * GEXP_KP + 4 * kq) = *reinterpret_cast<const u32x2*>(l4);
  }
  __syncthreads();

  F4 bh[GEXP_KT], bl[GEXP_KT];  // B operand: lane (pixel li, k = 16 kt + 4 g .. + 3)
  long koff[GEXP_KT];           // F2 offset of this lane's first k of tile kt (its 4 k share one tap: R % 4 == 0)
#pragma unroll
  for (int kt = 0; kt < GEXP_KT; ++kt) {
    bh[kt] = *reinterpret_cast<const F4*>(midh + li * GEXP_KP + 16 * kt + 4 * g);
    bl[kt] = *reinterpret_cast<const F4*>(midl + li * GEXP_KP + 16 * kt + 4 * g);
    const int k = 16 * kt + 4 * g;
    const int tap = k < K2 ? k / R : 0, nn = k < K2 ? k - tap * R : 0;  // k >= K2: mid is zero there, any valid address
    koff[kt] = ((long)nn * taps + tap) * C;
  }
  const long estride = (long)taps * C;  // next rank index
  const int ntiles_all = (C + 15) >> 4;
  const int tper = (ntiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int tbeg = (int)blockIdx.y * tper;
  const int ntiles = tbeg + tper < ntiles_all ? tbeg + tper : ntiles_all;
  auto load_f2 = [&](f32x4 (&fr)[GEXP_KT], int tile) {
    const int n = 16 * tile + li;
    const float* p = a.F2 + ((tile < ntiles && n < C) ? n : 0);
#pragma unroll
    for (int kt = 0; kt < GEXP_KT; ++kt)
      if (kt < KT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fr[kt][e] = p[koff[kt] + (16 * kt + 4 * g + e < K2 ? e : 0) * estride];
      }
  };
  const bool vec = (C & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7u) == 0;
  auto do_tile = [&](f32x4 (&fr)[GEXP_KT], int tile) {
    f32x4 y = zero4();
#pragma unroll
    for (int kt = 0; kt < GEXP_KT; ++kt)
      if (kt < KT) {
        LR_USE(fr[kt]);
        F4 ah, al;
        lr_split4<T>(fr[kt], ah, al);
        y = Mma16<T>::mma(ah, bh[kt], y);
        y = Mma16<T>::mma(al, bh[kt], y);
        y = Mma16<T>::mma(ah, bl[kt], y);
      }
    const long m = m0 + li;
    const int n = 16 * tile + 4 * g;
    const bool ok = tile < ntiles && m < a.M;
    T v[4] __attribute__((aligned(8)));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = TT<T>::from_f(y[e]);
    T* o = static_cast<T*>(a.out) + m * C + n;
    if (vec && n + 4 <= C) {
      if (ok) *reinterpret_cast<u32x2*>(o) = *reinterpret_cast<const u32x2*>(v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ok && n + e < C) o[e] = v[e];
    }
  };
  f32x4 fr[D2][GEXP_KT];
#pragma unroll
  for (int j = 0; j < D2; ++j) {
    load_f2(fr[j], tbeg + wave + j * NW);
    LR_LOAD_FENCE();
  }
  for (int tile = tbeg + wave; tile < ntiles; tile += D2 * NW) {
#pragma unroll
    for (int j = 0; j < D2; ++j) {
      do_tile(fr[j], tile + j * NW);
      LR_LOAD_FENCE();
      load_f2(fr[j], tile + (j + D2) * NW);
      LR_LOAD_FENCE();
    }
  }
}

}  // namespace lyc
