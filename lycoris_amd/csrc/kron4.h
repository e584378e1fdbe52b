// kron4.h -- Kronecker (LoKr) row kernel for 16-bit activations on nn.Linear, gfx950.  Fourth generation (round 4).
//
//   S1[(m,u), n]   = sum_k x3[(m,u), k] * w2[n, k]                      stage 1, v_mfma_f32_16x16x32 (w2 = packed hi + lo planes)
//   y [(m,p), n]   = alpha * sum_u w1[p,u] * S1[(m,u), n] (+ base)       stage 2, v_mfma_f32_16x16x16 in registers
//   dW1[p,u]      += alpha * sum_{m,n} S1[(m,u), n] * xref[(m,p), n]     backward only, in registers (per-workgroup partials)
//
// Why a new kernel (profiles/r03_ktrace_kron3_pl.log): a kron3 workgroup is ONE serial chain of ~9 400 cycles
// at one instruction per 5-7 cycles -- ~2 900 cycles of kernel-argument waits, 64-bit index arithmetic and a scalar loop before the
// first load is issued, x fragments through registers -> ds_write -> ds_read, five code versions of the k loop.  The launches of
// this workload are 5-25 MB, so the chain IS the launch time.  kron4 is built around the instruction count of that chain:
//
//   * every launch constant is precomputed on the host (Kron4Args); rows / columns are addressed with 32-bit offsets through
//     buffer descriptors, so ragged edges (rows >= M*G, the tail of a K that is not a multiple of 32) are zero-filled by the
//     descriptor's bounds check instead of by compares and selects;
//   * BOTH operands go HBM -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction): no staging registers, no
//     ds_write, no conversion.  x3 is a plain row-major [M*G, K] matrix, so a piece is 16 rows x 64 bytes (one k step of one MFMA
//     A tile); the four 16-byte chunks of a row are stored XOR-permuted (on the SOURCE address: the LDS image of a DMA is
//     lane-linear) so that the fragment reads are conflict-free ds_read_b128 for any row pitch;
//   * a ring of D k-step slots with COUNTED vmcnt waits and raw s_barrier: the MFMAs of k step 0 start when k step 0 has
//     landed, the rest of the operands stream in underneath; the same loop serves K = 40 ... 1280 (no chunk variants);
//   * the register epilogue (stage 2 on the matrix cores, fused `base + delta`, dW1 partials) is kron3's, with buffer stores.
//
// Taken when: T in {bf16, fp16}, G = Gin = Gout in {1, 2, 4, 8, 16}, K % 8 == 0, N % 8 == 0, packed planes available, every
// tensor < 2 GiB, rows stored as T.  Everything else stays on kron3.
// Reference math: lycoris/functional/lokr.py:154-247, modules/lokr.py:358-381, 543-566.
#pragma once
#include "kron3.h"

namespace lyc {

struct Kron4Args {
  const void* x;       // [rows_total, K] T, 16-byte aligned
  void* y;             // [rows_total, N] T
  const void* planes;  // packed hi / lo planes of this role (kron_conv.h: units (n tile, k step) of 2 KiB)
  const float* w1;     // element (po, ui) at po * s1o + ui * s1i  (this role's orientation)
  const void* aux;     // EPI 1: base [rows_total, N] (frozen layer output); EPI 2: xref [rows_total, N]
  float* dw1_ws;       // EPI 2: per-workgroup partials [(by * gridDim.x + bx)][G * G]
  int dw1_blocks;      // EPI 2: the reducer sums this many [G * G] blocks (>= the grid size); the workgroups zero the surplus ones
  unsigned x_bytes, y_bytes, plane_bytes;
  int rows_total;      // M * G
  int K, N, KS;        // KS = ceil(K / 32)
  int lg;              // log2 G
  int s1o, s1i;
  float alpha;
};

constexpr int K4_TRASH = 4096;  // dummy DMA target of waves without a plane piece in a k step; dW1 reduction scratch
constexpr unsigned K4_OOB = 0x80000000u;
constexpr int K4_RSRC_FLAGS = 0x00020000;

__host__ __device__ constexpr int kron4_lds_bytes(int MI, int NI, int D) { return D * (NI * 2048 + NWAVES * MI * 1024) + K4_TRASH; }

template <int N>
__device__ __forceinline__ void k4_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `groups` * C vector-memory operations are outstanding (groups <= 7)
template <int C>
__device__ __forceinline__ void k4_wait_groups(int groups) {
  switch (groups) {
    case 0: k4_wait_vm<0>(); break;
    case 1: k4_wait_vm<C>(); break;
    case 2: k4_wait_vm<2 * C>(); break;
    case 3: k4_wait_vm<3 * C>(); break;
    case 4: k4_wait_vm<4 * C>(); break;
    case 5: k4_wait_vm<(5 * C > 63 ? 63 : 5 * C)>(); break;
    case 6: k4_wait_vm<(6 * C > 63 ? 63 : 6 * C)>(); break;
    default: k4_wait_vm<(7 * C > 63 ? 63 : 7 * C)>(); break;
  }
}

typedef __attribute__((address_space(3))) void* k4_lds_ptr;

// ---- column permutation (NP) -------------------------------------------------------------------------------------------------------
// The register epilogue leaves lane (row li, g) with 4 consecutive columns per 16-column MFMA tile: 8-byte stores, 16 rows x 32 bytes
// per wave instruction.  The write path takes that badly: a 21 MB output costs 6.3 us in 32-byte pieces, 3.8 us in 64-byte pieces,
// 3.2 us in full lines (benchmarks/stbench.cpp, profiles/r04_stbench.log; the ablation of profiles/r04_k4_ablation.log shows the
// stores -- not the matrix work, not the operand traffic -- are what the (1024, 1280 -> 10240) forward waits for).  Which w2 rows
// form MFMA tile t is free, so tiles are formed in PAIRS: row i of tile 2q + e holds column 32 q + 8 (i >> 2) + 4 e + (i & 3) of the
// column block.  Lane g then owns 8 CONSECUTIVE columns 32 q + 8 g .. + 7 across the pair: one 16-byte store (and one 16-byte base /
// xref load) per row and pair.  The planes stay as they are (kron_conv.h); the permutation is applied by the per-lane SOURCE address
// of the plane DMA.  An odd last tile of the block keeps the 8-byte form.  Needs N % (16 NI) == 0 (every tile of every block exists).
template <int NI>
__device__ __forceinline__ constexpr bool k4_paired(int ni) {
  return ni < 2 * (NI / 2);
}
// DMA source offset (bytes, relative to the first unit of the tile's pair / of the tile) of lane `lane` for MFMA tile `ni`
template <int NI, bool NP>
__device__ __forceinline__ unsigned k4_plane_voff(int lane, int ni, int KS) {
  if (NP && k4_paired<NI>(ni)) {
    const int i = lane & 15, g = lane >> 4;
    return (unsigned)(i >> 3) * (unsigned)KS * 2048u + (unsigned)(g * 16 + 8 * ((i >> 2) & 1) + 4 * (ni & 1) + (i & 3)) * 16u;
  }
  return (unsigned)lane * 16u;
}
// first n tile (relative to the block) whose units the DMA of MFMA tile `ni` starts from
template <int NI, bool NP>
__device__ __forceinline__ int k4_plane_tile(int ni) {
  return (NP && k4_paired<NI>(ni)) ? (ni & ~1) : ni;
}

// base / xref pieces of one 16 MI-row tile: auxv[mi][ni] = the 4 values (8 bytes) lane (li, g) needs for MFMA tile ni
template <int MI, int NI, bool NP>
__device__ __forceinline__ void k4_load_aux(u32x2 (&auxv)[MI][NI], const __amdgpu_buffer_rsrc_t& rsa, const unsigned (&rofs)[MI], int nt0, int N,
                                            int g) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (NP && k4_paired<NI>(ni)) {
        if (ni & 1) continue;
        const int gn = (nt0 + ni) * 16 + 8 * g;
        const unsigned off = gn < N ? rofs[mi] + (unsigned)gn * 2u : K4_OOB;  // out of bounds reads as zero
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)off, 0, 0));
        auxv[mi][ni] = u32x2{v[0], v[1]};
        auxv[mi][ni + 1] = u32x2{v[2], v[3]};
      } else {
        const int gn = (nt0 + ni) * 16 + 4 * g;
        const unsigned off = gn < N ? rofs[mi] + (unsigned)gn * 2u : K4_OOB;
        auxv[mi][ni] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsa, (int)off, 0, 0));
      }
    }
  }
}

// The stores of one 16 MI x 16 NI tile of T values (`ov[mi][ni]` = the 4 values lane (li, g) holds of MFMA tile ni): 16-byte stores over
// paired column tiles (NP), 8-byte stores otherwise; out-of-range rows / columns are dropped by the descriptor.
template <int MI, int NI, bool NP>
__device__ __forceinline__ void k4_store_row(const u32x2 (&ov)[NI], unsigned rof, const __amdgpu_buffer_rsrc_t& rsy, int nt0, int N, int g) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    if (NP && k4_paired<NI>(ni)) {
      if (ni & 1) continue;
      const int gn = (nt0 + ni) * 16 + 8 * g;  // N % (16 NI) == 0: all in or all out
      const unsigned off = gn < N ? rof + (unsigned)gn * 2u : K4_OOB;  // out-of-bounds stores are dropped
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{ov[ni][0], ov[ni][1], ov[ni + 1][0], ov[ni + 1][1]}, rsy, (int)off, 0, 0);
    } else {
      const int gn = (nt0 + ni) * 16 + 4 * g;
      const unsigned off = gn < N ? rof + (unsigned)gn * 2u : K4_OOB;
      __builtin_amdgcn_raw_buffer_store_b64(ov[ni], rsy, (int)off, 0, 0);
    }
  }
}

// Register epilogue of one 16 MI x 16 NI tile (kron3.h): stage 2 on the matrix cores, fused `base + delta`, dW1 contribution, stores.
// ABL (benchmarks only): bit 1 = no stage-2 matrix work, bit 2 = no stores.
// ACC (round 5, kron4_sum_kernel): nothing is stored; alpha * (stage-2 result) is ADDED to `ysum` in fp32 -- the caller runs several
// problems through one workgroup and stores their sum once.
template <typename T, int MI, int NI, int EPI, bool NP, int ABL, bool ACC = false>
__device__ __forceinline__ void k4_epilogue(f32x4 (&acc)[MI][NI], const typename Mma16<T>::frag& a2h, const typename Mma16<T>::frag& a2l,
                                            const typename Mma16<T>::frag& ident, const u32x2 (&auxv)[MI][NI], const unsigned (&rofs)[MI],
                                            const __amdgpu_buffer_rsrc_t& rsy, int nt0, int N, int g, float alpha, f32x4& cdw,
                                            f32x4 (*ysum)[NI] = nullptr) {
  using F4 = typename Mma16<T>::frag;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    u32x2 ov[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      T h[4], l[4];
      k3_split4<T>(acc[mi][ni], h, l);
      const F4 sh = *reinterpret_cast<F4*>(h), sl = *reinterpret_cast<F4*>(l);
      f32x4 yv = zero4();
      if constexpr ((ABL & 1) != 0) {
        yv = acc[mi][ni];
      } else {
        yv = Mma16<T>::mma(sh, a2h, yv);
        yv = Mma16<T>::mma(sl, a2h, yv);
        yv = Mma16<T>::mma(sh, a2l, yv);
      }
      acc[mi][ni] = zero4();
      float bb[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == 1) {  // fused `base + delta`: fp32 add, one rounding
        T bt[4];
        *reinterpret_cast<u32x2*>(bt) = auxv[mi][ni];
#pragma unroll
        for (int e = 0; e < 4; ++e) bb[e] = TT<T>::to_f(bt[e]);
      }
      if constexpr (ACC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ysum[mi][ni][e] += alpha * yv[e];
      } else {
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(alpha * yv[e] + bb[e]);
        ov[ni] = *reinterpret_cast<u32x2*>(o);
      }
      if constexpr (EPI == 2) {
        // S1 (hi, lo) transposed through the matrix core: lane (li = row, 4g+e = tile column) -- exact, the values are T
        const f32x4 th = Mma16<T>::mma(sh, ident, zero4());
        const f32x4 tl = Mma16<T>::mma(sl, ident, zero4());
        T thv[4], tlv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          thv[e] = TT<T>::from_f(th[e]);
          tlv[e] = TT<T>::from_f(tl[e]);
        }
        const F4 bf = *reinterpret_cast<const F4*>(&auxv[mi][ni]);  // xref fragment B[k = tile column][j = row li]; zero outside
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(thv), bf, cdw);
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(tlv), bf, cdw);
      }
    }
    if constexpr ((ABL & 2) != 0) {
      if (alpha != 123.f) continue;
    }
    if constexpr (!ACC) k4_store_row<MI, NI, NP>(ov, rofs[mi], rsy, nt0, N, g);
  }
}

// w1 operand of stage 2 (raw fp32, converted where it is used) and the identity of the dW1 transposes
template <typename T>
__device__ __forceinline__ void k4_w1_frags(const float (&w1raw)[4], typename Mma16<T>::frag& a2h, typename Mma16<T>::frag& a2l) {
  using F4 = typename Mma16<T>::frag;
  T h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_f<T>(w1raw[j], h[j], l[j]);
  a2h = *reinterpret_cast<F4*>(h);
  a2l = *reinterpret_cast<F4*>(l);
}
template <typename T>
__device__ __forceinline__ typename Mma16<T>::frag k4_identity(int li, int g) {
  using F4 = typename Mma16<T>::frag;
  T idv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) idv[e] = TT<T>::from_f((4 * g + e) == li ? 1.f : 0.f);
  return *reinterpret_cast<F4*>(idv);
}
// cross-wave sum of the dW1 accumulators through `red` (NW KiB) and the per-workgroup partial
template <int NW>
__device__ __forceinline__ void k4_dw1_partial(const Kron4Args& a, float* red, const f32x4& cdw, int tid, int wave, int li, int g, int lg,
                                               int me = (int)(blockIdx.y * gridDim.x + blockIdx.x), int nwg = (int)(gridDim.x * gridDim.y)) {
  const int G = 1 << lg;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * g + r) * 16 + li] = cdw[r];
  __syncthreads();
  if (tid < G * G) {
    // cdw: D[i = (m', u)][j = (m'', po)], lane (col j = li, rows 4g+r); only the diagonal blocks m' == m'' count
    const int u = tid >> lg, po = tid & (G - 1);
    float s = 0.f;
    for (int b = 0; b < (16 >> lg); ++b) {
      const int e = ((b << lg) + u) * 16 + (b << lg) + po;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[256 * w + e];
    }
    const int e = po * a.s1o + u * a.s1i;  // position in dw1 memory order
    a.dw1_ws[(long)me * (G * G) + e] = a.alpha * s;
    // the consumer (lyc_lokr_wgrad_group / the dW2 launch's reducer slice) derives the block count from the layer's dimensions
    // alone (capi.hip: lokr_dx_partial_blocks): blocks beyond this grid are written as zeros
    for (int z = nwg + me; z < a.dw1_blocks; z += nwg) a.dw1_ws[(long)z * (G * G) + e] = 0.f;
  }
}

// EPI: 0 = forward, 1 = forward with the fused `base + delta` epilogue, 2 = backward dx with the dW1 partials
// NP : paired column tiles (16-byte stores, above).  ABL != 0: ablation builds of benchmarks/k4bench.cpp (results are garbage):
//      1 = no stage-2 matrix work, 2 = no stores, 4 = no x DMA, 8 = no plane DMA, 16 = no stage-1 matrix work.
// (bx, by) of (nbx, nby): the workgroup's row / column tile within ITS problem (a launch may carry several problems)
template <typename T, int MI, int NI, int D, int EPI, bool NP = false, int ABL = 0, bool ACC = false>
__device__ __forceinline__ void kron4_body(const Kron4Args& a, const int bx, const int by, const int nbx, const int nby,
                                           f32x4 (*ysum)[NI] = nullptr) {
  extern __shared__ __attribute__((aligned(1024))) char k4_smem[];
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;
  static_assert(D >= 2 && D <= 8, "ring depth");
  constexpr int PPW = (2 * NI + NWAVES - 1) / NWAVES;  // plane pieces per wave and k step (dummies included)
  constexpr int C = PPW + MI;                          // DMA operations per wave and k step
  constexpr int SLOT_P = NI * 2048, SLOT_X = NWAVES * MI * 1024;
  constexpr int OFF_X = D * SLOT_P, OFF_TRASH = OFF_X + D * SLOT_X;
  constexpr bool AUX = EPI != 0;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, N = a.N, KS = a.KS, lg = a.lg, G = 1 << lg;
  const unsigned K2 = (unsigned)K * 2u;
  const int row0 = bx * (64 * MI);
  const int nt0 = by * NI;
  LYC_TRACE_DECL;
  LYC_STAMP(0);

  // ---- loads whose results are first used in the epilogue: issued FIRST (they are then the oldest entries of the vmcnt queue
  //      and never stand between a counted wait and the DMA group it waits for) ------------------------------------------------
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
  // output-row bookkeeping (also the addresses of base / xref): row R = (m, p) of this lane per mi
  unsigned rofs[MI];  // R * N * 2 bytes, or out of bounds for rows >= rows_total
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int R = row0 + (wave * MI + mi) * 16 + li;
    rofs[mi] = R < a.rows_total ? (unsigned)R * (unsigned)N * 2u : K4_OOB;
  }
  u32x2 auxv[MI][NI];
  if constexpr (AUX) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.aux), 0, (int)a.y_bytes, K4_RSRC_FLAGS);
    k4_load_aux<MI, NI, NP>(auxv, rsa, rofs, nt0, N, g);
  }

  // ---- DMA addressing ------------------------------------------------------------------------------------------------------------
  // x: descriptor rebased to the workgroup's first row; rows beyond the matrix are out of bounds (-> zeros in LDS).
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(static_cast<const char*>(a.x)) + (size_t)(unsigned)row0 * K2, 0, (int)(a.x_bytes - (unsigned)row0 * K2),
      K4_RSRC_FLAGS);
  // piece (mi, ks) = 16 rows x 64 bytes; DMA lane l writes LDS slot l = (row r = l >> 2, stored chunk l & 3) which holds the
  // logical chunk c = (l & 3) ^ phi(r >> 2), phi = (0, 3, 2, 1): the 16 rows of a fragment read (row li, chunk g) then fall on 16
  // different 16-byte slots of the 256-byte bank row in every service group of ds_read_b128.
  const int xr = lane >> 2, xc = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int klast = K - 32 * (KS - 1);  // columns of the last k step: 8, 16, 24 or 32
  unsigned vx[MI], vxl[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    vx[mi] = (unsigned)((wave * MI + mi) * 16 + xr) * K2 + (unsigned)xc * 16u;
    vxl[mi] = (8 * xc + 8 <= klast) ? vx[mi] : K4_OOB;  // beyond K: zeros (the neighbouring row's data must not be multiplied in)
  }
  const unsigned rd_x = (unsigned)(OFF_X + (li * 4 + (g ^ ((0 - (li >> 2)) & 3))) * 16) + (unsigned)wave * (MI * 1024);

  // planes: piece p = (ni, half) of a k step; wave w issues p = w, w + 4, ...; p >= 2 NI is a dummy (zeros into the trash slot)
  const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.planes), 0, (int)a.plane_bytes, K4_RSRC_FLAGS);
  const int ntiles = (N + 15) >> 4;
  unsigned pv[PPW], pbase[PPW];
  int pdst[PPW];
  bool preal[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int p = wave + NWAVES * j;
    preal[j] = p < 2 * NI;
    const int ni = p >> 1;
    int nt = nt0 + k4_plane_tile<NI, NP>(ni);
    if (nt > ntiles - 1) nt = ntiles - 1;  // beyond N: a valid duplicate, its columns are never stored
    pbase[j] = ((unsigned)(nt * KS) * 2u + (unsigned)(p & 1)) * 1024u;
    pv[j] = preal[j] ? k4_plane_voff<NI, NP>(lane, ni, KS) : K4_OOB;
    pdst[j] = preal[j] ? p * 1024 : OFF_TRASH + wave * 1024;
  }
  auto issue = [&](int ks, int slot) {
    const bool last = ks == KS - 1;
    const unsigned sx = (unsigned)ks * 64u;
    if constexpr ((ABL & 4) == 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        char* dst = k4_smem + OFF_X + slot * SLOT_X + (wave * MI + mi) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (k4_lds_ptr)dst, 16, (int)(last ? vxl[mi] : vx[mi]), (int)sx, 0, 0);
      }
    }
    if constexpr ((ABL & 8) == 0) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        char* dst = k4_smem + (preal[j] ? slot * SLOT_P : 0) + pdst[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (k4_lds_ptr)dst, 16, (int)pv[j], (int)(pbase[j] + (unsigned)ks * 2048u), 0, 0);
      }
    }
  };

  // ---- stage 1 --------------------------------------------------------------------------------------------------------------------
  const int npro = KS < D ? KS : D;
  for (int s = 0; s < npro; ++s) issue(s, s);
  LYC_STAMP(1);  // prologue DMAs issued
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  int slot = 0, prev = 0;
  for (int ks = 0; ks < KS; ++ks) {
    // groups issued so far: 0 .. min(KS - 1, max(D - 1, ks + D - 2)); all but those newer than `ks` must have landed
    int newest = ks + D - 2;
    if (newest < D - 1) newest = D - 1;
    if (newest > KS - 1) newest = KS - 1;
    k4_wait_groups<C>(newest - ks);
    __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain the DMA queue (vmcnt(0))
    asm volatile("" ::: "memory");
    if (ks == 0) LYC_STAMP(2);  // first k step landed, barrier passed
    if (ks >= 1 && ks - 1 + D < KS) issue(ks - 1 + D, prev);  // every wave is past its reads of `prev` (it arrived at this barrier)
    const char* xs = k4_smem + rd_x + slot * SLOT_X;
    const char* ps = k4_smem + slot * SLOT_P + lane * 16;
    F8 af[MI], bh[NI], bl[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const F8*>(xs + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bh[ni] = *reinterpret_cast<const F8*>(ps + ni * 2048);
      bl[ni] = *reinterpret_cast<const F8*>(ps + ni * 2048 + 1024);
    }
    if constexpr ((ABL & 16) != 0) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][ni] += __builtin_bit_cast(f32x4, af[mi]) + __builtin_bit_cast(f32x4, bh[ni]) + __builtin_bit_cast(f32x4, bl[ni]);
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bh[ni], acc[mi][ni]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bl[ni], acc[mi][ni]);
    }
    prev = slot;
    slot = slot + 1 == D ? 0 : slot + 1;
  }
  LYC_STAMP(4);

  // ---- epilogue, all in registers -------------------------------------------------------------------------------------------------
  F4 a2h, a2l, ident = {};
  k4_w1_frags<T>(w1raw, a2h, a2l);
  if constexpr (EPI == 2) ident = k4_identity<T>(li, g);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, K4_RSRC_FLAGS);
  f32x4 cdw = zero4();
  k4_epilogue<T, MI, NI, EPI, NP, ABL, ACC>(acc, a2h, a2l, ident, auxv, rofs, rsy, nt0, N, g, a.alpha, cdw, ysum);
  LYC_STAMP(5);
  LYC_TRACE_FLUSH();
  // (the trash slot: every DMA of this workgroup has landed -- the last k step waited vmcnt(0) in every wave, of ITS OWN operations;
  //  k4_dw1_partial starts with a barrier)
  if constexpr (EPI == 2) k4_dw1_partial<NWAVES>(a, reinterpret_cast<float*>(k4_smem + OFF_TRASH), cdw, tid, wave, li, g, lg, by * nbx + bx, nbx * nby);
}

template <typename T, int MI, int NI, int D, int EPI, bool NP = false, int ABL = 0>
__global__ __launch_bounds__(NTHREADS, 2) void kron4_kernel(Kron4Args a) {
  kron4_body<T, MI, NI, D, EPI, NP, ABL>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}

// Several problems of ONE tile plan in one launch (round 4: the q / k / v projections of an attention block, the k / v projections
// of a cross-attention -- same shapes, same input or not): blockIdx.z selects the problem, whose own row-tile count may be smaller
// than the grid's.  A 5 us launch that fills a third of the chip becomes one launch that fills it.
constexpr int K4_GROUP_MAX = 4;
struct Kron4GroupArgs {
  int n;
  int nbx[K4_GROUP_MAX];
  Kron4Args p[K4_GROUP_MAX];
};
template <typename T, int MI, int NI, int D, int EPI, bool NP = false>
__global__ __launch_bounds__(NTHREADS, 2) void kron4_group_kernel(Kron4GroupArgs ga) {
  const int z = (int)blockIdx.z;
  if ((int)blockIdx.x >= ga.nbx[z]) return;
  const Kron4Args a = ga.p[z];
  kron4_body<T, MI, NI, D, EPI, NP, 0>(a, (int)blockIdx.x, (int)blockIdx.y, ga.nbx[z], (int)gridDim.y);
}

// The gradient of a tensor that n sibling projections read (round 5): dx = sum_i dx_i.  ONE workgroup per (row tile, column tile)
// runs the n problems one after the other -- each with its own operand ring, stage 1, stage 2 and dW1 partials, exactly as
// kron4_body<EPI 2> does -- adds the stage-2 results in fp32 registers and stores the sum ONCE: no dx_i ever reaches HBM, and the
// separate summation pass (lyc_sum_rows: 4.7 us per set, 140 sets per SDXL step) is gone.  The sum is rounded once (n separate
// nodes: n roundings + n - 1 more in autograd's accumulation).  All problems share rows_total, K, N, G.
template <typename T, int MI, int NI, int D, bool NP = false>
__global__ __launch_bounds__(NTHREADS, 2) void kron4_sum_kernel(Kron4GroupArgs ga, void* dx_sum) {
  const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  f32x4 ysum[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) ysum[mi][ni] = zero4();
  for (int z = 0; z < ga.n; ++z) {
    kron4_body<T, MI, NI, D, 2, NP, 0, true>(ga.p[z], bx, by, (int)gridDim.x, (int)gridDim.y, ysum);
    __syncthreads();  // the dW1 reduction scratch and the ring slots are free before the next problem's first DMA lands in them
  }
  const Kron4Args& a = ga.p[0];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N, nt0 = by * NI, row0 = bx * (64 * MI);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(dx_sum, 0, (int)a.y_bytes, K4_RSRC_FLAGS);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int R = row0 + (wave * MI + mi) * 16 + li;
    const unsigned rof = R < a.rows_total ? (unsigned)R * (unsigned)N * 2u : K4_OOB;
    u32x2 ov[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(ysum[mi][ni][e]);
      ov[ni] = *reinterpret_cast<u32x2*>(o);
    }
    k4_store_row<MI, NI, NP>(ov, rof, rsy, nt0, N, g);
  }
}

}  // namespace lyc
