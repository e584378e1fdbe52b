// kron4r.h -- EXPERIMENT (round 4), not part of the library: the kron4 math with the planes of a column block resident in LDS and
// every wave walking its own row tiles (no barrier after the prologue).  Built and measured by benchmarks/k4bench.cpp only.
// Result (profiles/r04_k4bench3/4/5_*.log): bit-identical to kron3 / kron4, never faster than kron4 on the SDXL / SD1.5 shapes -- the
// big launches wait for the write path (profiles/r04_k4_ablation.log), which this structure does not change.
#pragma once
#include "../../lycoris_amd/csrc/kron4.h"

namespace lyc {

// ----------------------------------------------------------------------------------------------------------------------------------
// kron4r: the same math with the PLANES OF THE COLUMN BLOCK RESIDENT in LDS and every wave walking its own sequence of row tiles.
//
// A workgroup loads the hi / lo planes of its column block once (NI * KS units of 2 KiB, one barrier); then each of its NW waves
// streams its own 16 MI-row tiles through a private ring of k-step slots: no barrier after the prologue, the DMA of tile t + 1, the
// matrix work of tile t and the stores of tile t - 1 overlap inside every wave, operand traffic drops to planes once per workgroup +
// x once per column block.  Counted waits: the vmcnt queue of a wave holds, oldest first, [x groups ...][stores of the finished
// tile, aux loads of the next][x groups ...]; vector-memory operations retire in issue order, so "group s has landed" == "at most
// <operations issued after group s> are outstanding" -- a compile-time constant per (groups in flight, tile boundary inside the
// window) pair.  For K <= ~256 (the planes of a column block must fit the LDS next to the rings).
// Status (profiles/r04_k4bench3/4_*.log): bit-identical to kron3 / kron4, but NOT faster than kron4 on any SDXL shape -- the big
// launches wait for the write path, not for operand traffic or phase overlap (ablation) -- so capi.hip does not dispatch to it yet.
template <int MIc, int E>
__device__ __forceinline__ void k4r_wait(int groups, bool extra) {
  // groups <= 6
  if (!extra) {
    switch (groups) {
      case 0: k4_wait_vm<0>(); break;
      case 1: k4_wait_vm<MIc>(); break;
      case 2: k4_wait_vm<2 * MIc>(); break;
      case 3: k4_wait_vm<3 * MIc>(); break;
      case 4: k4_wait_vm<4 * MIc>(); break;
      case 5: k4_wait_vm<5 * MIc>(); break;
      default: k4_wait_vm<6 * MIc>(); break;
    }
  } else {
    switch (groups) {
      case 0: k4_wait_vm<(E > 63 ? 63 : E)>(); break;
      case 1: k4_wait_vm<(MIc + E > 63 ? 63 : MIc + E)>(); break;
      case 2: k4_wait_vm<(2 * MIc + E > 63 ? 63 : 2 * MIc + E)>(); break;
      case 3: k4_wait_vm<(3 * MIc + E > 63 ? 63 : 3 * MIc + E)>(); break;
      case 4: k4_wait_vm<(4 * MIc + E > 63 ? 63 : 4 * MIc + E)>(); break;
      case 5: k4_wait_vm<(5 * MIc + E > 63 ? 63 : 5 * MIc + E)>(); break;
      default: k4_wait_vm<(6 * MIc + E > 63 ? 63 : 6 * MIc + E)>(); break;
    }
  }
}

__host__ __device__ constexpr int kron4r_lds_bytes(int MI, int NI, int D, int KS, int NW) {
  return NI * KS * 2048 + NW * D * MI * 1024 + NW * 1024;
}

// NW waves per workgroup (4 ... 16) share one resident plane tile: the planes are what limits the workgroups per CU, the waves per
// SIMD (latency hiding: every wave is an in-order instruction stream) come from NW
template <typename T, int MI, int NI, int D, int EPI, int NW, bool NP = false>
__global__ __launch_bounds__(64 * NW) void kron4r_kernel(Kron4Args a) {
  extern __shared__ __attribute__((aligned(1024))) char k4_smem[];
  using F8 = typename TT<T>::frag;
  using F4 = typename Mma16<T>::frag;
  static_assert(D >= 2 && D <= 8, "ring depth");
  constexpr bool AUX = EPI != 0;
  constexpr int NPAIR = NP ? NI / 2 : 0;
  constexpr int NSI = NI - NPAIR;                                   // store (and aux load) instructions per mi
  constexpr int NS = MI * NSI, NA = AUX ? MI * NSI : 0, E = NS + NA;  // vector-memory operations at a tile boundary

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, N = a.N, KS = a.KS, lg = a.lg, G = 1 << lg;
  const unsigned K2 = (unsigned)K * 2u;
  const int nt0 = (int)blockIdx.y * NI;
  const int OFF_X = NI * KS * 2048 + wave * (D * MI * 1024);  // this wave's ring
  const int OFF_RED = NI * KS * 2048 + NW * D * MI * 1024;
  // this wave's row tiles (16 MI rows each): first, stride, count
  const int ntiles = (a.rows_total + 16 * MI - 1) / (16 * MI);
  const int tstride = (int)gridDim.x * NW;
  const int t0 = (int)blockIdx.x * NW + wave;
  const int mytiles = t0 < ntiles ? (ntiles - t0 + tstride - 1) / tstride : 0;
  const int S = mytiles * KS;  // k steps of this wave

  float w1raw[4];
  {
    const int mi_ = li >> lg, po = li & (G - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = 4 * g + j;
      const float v = a.w1[po * a.s1o + (kk & (G - 1)) * a.s1i];
      w1raw[j] = ((kk >> lg) == mi_) ? v : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.aux ? a.aux : a.x), 0, (int)(AUX ? a.y_bytes : 0u), K4_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)a.y_bytes, K4_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, K4_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.planes), 0, (int)a.plane_bytes, K4_RSRC_FLAGS);
  // output-row offsets of a tile: R = tile * 16 MI + 16 mi + li
  auto row_ofs = [&](int tile, unsigned (&rofs)[MI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int R = tile * (16 * MI) + 16 * mi + li;
      rofs[mi] = R < a.rows_total ? (unsigned)R * (unsigned)N * 2u : K4_OOB;
    }
  };
  u32x2 auxv[MI][NI];
  unsigned rofs[MI];
  if (mytiles > 0) {
    row_ofs(t0, rofs);
    if constexpr (AUX) k4_load_aux<MI, NI, NP>(auxv, rsa, rofs, nt0, N, g);
  }

  // ---- planes of this column block: NI * KS * 2 pieces of 1 KiB, round robin over the waves --------------------------------------
  {
    const int ntl = (N + 15) >> 4;
    const int npieces = 2 * NI * KS;
    for (int p = wave; p < npieces; p += NW) {
      const int ni = p / (2 * KS), r = p - ni * (2 * KS);
      int nt = nt0 + k4_plane_tile<NI, NP>(ni);
      if (nt > ntl - 1) nt = ntl - 1;  // beyond N: a valid duplicate, its columns are never stored
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (k4_lds_ptr)(k4_smem + p * 1024), 16, (int)k4_plane_voff<NI, NP>(lane, ni, KS),
                                               (int)(((unsigned)(nt * KS) * 2u + (unsigned)r) * 1024u), 0, 0);
    }
  }
  // ---- x ring ------------------------------------------------------------------------------------------------------------------------
  const int xr = lane >> 2, xc = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int klast = K - 32 * (KS - 1);
  const bool chunk_ok = 8 * xc + 8 <= klast;
  const unsigned vx0 = (unsigned)xr * K2 + (unsigned)xc * 16u;  // + (tile * 16 MI + 16 mi) * K2
  const unsigned rd_x = (unsigned)(OFF_X + (li * 4 + (g ^ ((0 - (li >> 2)) & 3))) * 16);
  int i_tile = t0, i_ks = 0, i_slot = 0, i_left = S;  // issue cursor
  auto issue_next = [&]() {
    const bool last = i_ks == KS - 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const unsigned v = vx0 + (unsigned)(i_tile * (16 * MI) + 16 * mi) * K2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (k4_lds_ptr)(k4_smem + OFF_X + (i_slot * MI + mi) * 1024), 16,
                                               (int)((last && !chunk_ok) ? K4_OOB : v), (int)((unsigned)i_ks * 64u), 0, 0);
    }
    --i_left;
    i_slot = i_slot + 1 == D ? 0 : i_slot + 1;
    if (++i_ks == KS) {
      i_ks = 0;
      i_tile += tstride;
    }
  };
  const int npro = S < D - 1 ? S : D - 1;
  for (int s = 0; s < npro; ++s) issue_next();
  // the planes (issued before the x groups) have landed when at most npro * MI operations are outstanding
  k4r_wait<MI, 0>(npro, false);
  __syncthreads();  // every wave's plane pieces (vmcnt above: this wave's own; the barrier: everybody's)

  F4 a2h, a2l, ident = {};
  k4_w1_frags<T>(w1raw, a2h, a2l);
  if constexpr (EPI == 2) ident = k4_identity<T>(li, g);
  f32x4 cdw = zero4();
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = zero4();

  int tile = t0, ks = 0, slot = 0, tl = 0;  // consume cursor; tl = index of the tile within this wave's sequence
  for (int s = 0; s < S; ++s) {
    int newest = s + D - 2;  // groups issued so far: 0 .. min(S - 1, s + D - 2)
    if (newest > S - 1) newest = S - 1;
    // the stores of the previous tile (and the aux loads of this one) were issued after groups <= s0 + D - 2: they stand
    // between group s and the end of the queue for the first D - 1 steps of a tile
    k4r_wait<MI, E>(newest - s, tl > 0 && ks <= D - 2);
    asm volatile("" ::: "memory");
    if (i_left > 0) issue_next();  // into the slot consumed in the previous step (this wave's reads of it have returned)
    const char* xs = k4_smem + rd_x + slot * (MI * 1024);
    const char* ps = k4_smem + ks * 2048 + lane * 16;
    F8 af[MI], bh[NI], bl[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const F8*>(xs + mi * 1024);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bh[ni] = *reinterpret_cast<const F8*>(ps + ni * KS * 2048);
      bl[ni] = *reinterpret_cast<const F8*>(ps + ni * KS * 2048 + 1024);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bh[ni], acc[mi][ni]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bl[ni], acc[mi][ni]);
    slot = slot + 1 == D ? 0 : slot + 1;
    if (++ks < KS) continue;
    // ---- end of a tile ----------------------------------------------------------------------------------------------------------------
    ks = 0;
    k4_epilogue<T, MI, NI, EPI, NP, 0>(acc, a2h, a2l, ident, auxv, rofs, rsy, nt0, N, g, a.alpha, cdw);
    tile += tstride;
    ++tl;
    if (tl < mytiles) {
      row_ofs(tile, rofs);
      if constexpr (AUX) k4_load_aux<MI, NI, NP>(auxv, rsa, rofs, nt0, N, g);
    }
  }
  if constexpr (EPI == 2) k4_dw1_partial<NW>(a, reinterpret_cast<float*>(k4_smem + OFF_RED), cdw, tid, wave, li, g, lg);
}

}  // namespace lyc
