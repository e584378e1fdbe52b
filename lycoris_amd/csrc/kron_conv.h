// kron_conv.h -- LoKr on nn.Conv2d, activation path (forward and backward-dx / dW1), gfx950.  Round 3.
//
// The round-2 implicit-GEMM kernels (kron3.h, GM = 1 / 2) fetched every MFMA A fragment straight from HBM / L2: each
// source pixel row was read once per tap (9x for a 3x3 window) as 16-byte pieces of 16 different pixel rows per
// instruction, and every workgroup converted its fp32 w2 tile to hi/lo 16-bit planes by itself (49 conv layers = 31 % of the
// SDXL LoKr step at 2.4 % of the HBM roofline).  This kernel changes both ends:
//
//   * SOURCE PATCH IN LDS.  A workgroup owns a TH x TW tile of destination pixels (64 MI stage-1 rows = pixels x groups, MI = 2 /
//     4 / 8 chosen by the host: the operand planes are re-streamed per workgroup, so their traffic goes with 1 / MI) and
//     loads the (TH-1)*s + (kh-1)*dil + 1 rows of source pixels it touches ONCE, with full 16-byte coalesced loads along
//     the NHWC channel dimension, into LDS (zero-filled outside the image: no masks in the main loop).  The A fragment of
//     (pixel, group u, tap, k) is then one ds_read_b128 at  patch[(py + tap_dy) * PW + px + tap_dx][u][k].
//     Layout: [patch pixel][group u][K + pad]: the pad makes the group pitch an ODD number of 16-byte slots, so the 16 rows
//     of a fragment (2 pixels x 8 groups for factor 8) fall into 16 different slots of the 256-byte bank row.
//   * PRE-PACKED OPERAND PLANES.  w2 is a parameter: it changes once per optimizer step, not per workgroup.  A small pack
//     kernel (kron_pack_kernel below) writes it as hi + lo planes in the activation dtype in FRAGMENT-MAJOR order
//     [n tile][k step][hi | lo][lane][8]: the B fragment of a k step is 1 KiB of contiguous memory.  The conv kernel
//     streams those units into a two-stage LDS ring with global_load_lds (LDS-DMA: no registers, no conversion
//     instructions, no ds_write) and reads them back conflict-free as lane-linear ds_read_b128.
//
// Stage 2 (the G x G mix with w1), the fused dW1 partials and the stores are kron3.h's register epilogue.
// Reference math: lycoris/functional/lokr.py:195-247 (conv branch of bypass_forward_diff), modules/lokr.py:358-381.
#pragma once
#include "kron3.h"

namespace lyc {

// ---------------------------------------------------------------------------------------------------------------------
// Packed operand planes
// ---------------------------------------------------------------------------------------------------------------------
// One "role" = the B operand of one stage-1 contraction:  B[n][k],  n < N,  k = tap * Kt + kk  (kk < Kt, Kt % 8 == 0)
//   forward role : n = q (w2 rows, N = c),    kk = v (Kt = d)      B = w2[q, v, tap]
//   backward role: n = v (w2 columns, N = d), kk = q (Kt = c)      B = w2[q, v, tap]
// stored as units of 2 KiB: unit (nt, ks) = {hi[64 lanes][8], lo[64 lanes][8]}, lane (li, g) element e <-> n = 16 nt + li,
// k = 32 ks + 8 g + e; zero outside [N) x [taps * Kt).  Units are ordered ks-major inside an n tile.
__host__ __device__ inline long kron_plane_ksteps(int taps, int Kt) { return ((long)taps * Kt + 31) / 32; }
__host__ __device__ inline long kron_plane_bytes(int N, int taps, int Kt) {
  return (long)((N + 15) / 16) * kron_plane_ksteps(taps, Kt) * 2048;
}

struct KronPackArgs {
  const float* w2;      // full matrix: element (q, v, tap) at q * sq + v * sv + tap * st      (nullptr: low-rank product)
  long sq, sv, st;
  const float* w2a;     // low-rank: w2[q, v, tap] = sum_r w2a[q * a_sq + r * a_sr] * w2b[r * b_sr + v * b_sv + tap * b_st]
  const float* w2b;
  long a_sq, a_sr, b_sr, b_sv, b_st;
  int rank;
  int c, d, taps;
  void* fwd;            // kron_plane_bytes(c, taps, d) bytes, or nullptr
  void* bwd;            // kron_plane_bytes(d, taps, c) bytes, or nullptr
  long units_fwd;       // units of the forward role (the grid covers units_fwd + units_bwd units, 4 per workgroup)
};

template <typename T>
__device__ __forceinline__ void kron_pack_units(const KronPackArgs& a, long unit) {
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const bool fwd = unit < a.units_fwd;
  if (!fwd) unit -= a.units_fwd;
  char* plane = static_cast<char*>(fwd ? a.fwd : a.bwd);
  if (plane == nullptr) return;
  const int N = fwd ? a.c : a.d, Kt = fwd ? a.d : a.c;
  const long KS = kron_plane_ksteps(a.taps, Kt);
  const long nunits = (long)((N + 15) / 16) * KS;
  if (unit >= nunits) return;
  const int nt = (int)(unit / KS), ks = (int)(unit - (long)nt * KS);
  const int n = 16 * nt + li;
  const long k0 = 32L * ks + 8 * g;  // the 8 elements share one tap (Kt % 8 == 0)
  const int tap = (int)(k0 / Kt);
  const int kk0 = (int)(k0 - (long)tap * Kt);
  T h[8], l[8];
  float val[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) val[e] = 0.f;
  if (n < N && tap < a.taps) {
    if (a.w2 != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = fwd ? n : kk0 + e, vv = fwd ? kk0 + e : n;
        val[e] = a.w2[q * a.sq + vv * a.sv + tap * a.st];
      }
    } else if (fwd) {  // low rank, q = n fixed: one w2a value and 8 neighbouring w2b values per rank (16-byte loads for nn.Linear)
      const float* ar = a.w2a + n * a.a_sq;
      const float* bc = a.w2b + kk0 * a.b_sv + tap * a.b_st;
      const bool vec = a.b_sv == 1 && ((reinterpret_cast<uintptr_t>(bc) | (uintptr_t)(a.b_sr * 4)) & 15u) == 0;
      for (int r = 0; r < a.rank; ++r) {
        const float av = ar[r * a.a_sr];
        const float* br = bc + r * a.b_sr;
        float bv[8];
        if (vec) {
          *reinterpret_cast<f32x4*>(bv) = *reinterpret_cast<const f32x4*>(br);
          *reinterpret_cast<f32x4*>(bv + 4) = *reinterpret_cast<const f32x4*>(br + 4);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[e] = br[e * a.b_sv];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = fmaf(av, bv[e], val[e]);
      }
    } else {           // low rank, column n fixed, q = kk0 + e: one w2b value per rank, 8 rows of w2a
      const float* bc = a.w2b + n * a.b_sv + tap * a.b_st;
      const float* a0 = a.w2a + kk0 * a.a_sq;
      const bool vec = a.a_sr == 1 && (a.rank & 3) == 0 && ((reinterpret_cast<uintptr_t>(a0) | (uintptr_t)(a.a_sq * 4)) & 15u) == 0;
      if (vec) {
        for (int r = 0; r < a.rank; r += 4) {
          const float b0 = bc[r * a.b_sr], b1 = bc[(r + 1) * a.b_sr], b2 = bc[(r + 2) * a.b_sr], b3 = bc[(r + 3) * a.b_sr];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + e * a.a_sq + r);
            val[e] = fmaf(av[3], b3, fmaf(av[2], b2, fmaf(av[1], b1, fmaf(av[0], b0, val[e]))));  // rank-ascending, as the scalar loop
          }
        }
      } else {
        for (int r = 0; r < a.rank; ++r) {
          const float bv = bc[r * a.b_sr];
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = fmaf(a0[e * a.a_sq + r * a.a_sr], bv, val[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) split_f<T>(val[e], h[e], l[e]);
  *reinterpret_cast<u32x4*>(plane + unit * 2048 + lane * 16) = *reinterpret_cast<u32x4*>(h);
  *reinterpret_cast<u32x4*>(plane + unit * 2048 + 1024 + lane * 16) = *reinterpret_cast<u32x4*>(l);
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void kron_pack_kernel(KronPackArgs a) {
  kron_pack_units<T>(a, (long)blockIdx.x * NWAVES + (threadIdx.x >> 6));
}

// Many layers per launch (the once-per-optimizer-step refresh of every cached plane set): descriptors by value.
constexpr int KPG_MAX = 28;
struct KronPackGroupArgs {
  int n;
  long unit_end[KPG_MAX];  // exclusive prefix of the (NWAVES-aligned) unit counts
  KronPackArgs p[KPG_MAX];
};
static_assert(sizeof(KronPackGroupArgs) <= 3840, "kernel arguments are limited to 4 KiB");

template <typename T>
__global__ __launch_bounds__(NTHREADS) void kron_pack_group_kernel(KronPackGroupArgs ga) {
  const long u = (long)blockIdx.x * NWAVES + (threadIdx.x >> 6);
  int p = 0;
  while (p + 1 < ga.n && u >= ga.unit_end[p]) ++p;
  const long u0 = p ? ga.unit_end[p - 1] : 0;
  kron_pack_units<T>(ga.p[p], u - u0);
}

// Round 6: EVERY layer in ONE launch.  The refresh above is one launch per 28 layers (4 KiB of kernel arguments): 52 launches of 12 us
// for the 788 layers of the SDXL step (profiles/r06_c14: 0.63 ms for 0.5 GB of traffic -- each launch is one short round of
// workgroups).  Here the descriptors live in a caller-owned device table (written by kron_pack_table_write_kernel from kernel arguments,
// 28 per launch, and only when the set of layers changed -- a training run writes it once) and the pack kernel finds its layer through a
// workgroup -> layer map: one grid over all units of all layers.
//   table layout: KronPackArgs items[n] | int first_wg[n] | int wg_layer[total_wgs]
__host__ __device__ inline long kron_pack_table_items_bytes(int n) { return ((long)n * (long)sizeof(KronPackArgs) + 15) / 16 * 16; }
__host__ __device__ inline long kron_pack_table_bytes(int n, long total_wgs) {
  return kron_pack_table_items_bytes(n) + ((long)n * 4 + 15) / 16 * 16 + (total_wgs * 4 + 15) / 16 * 16;
}
// one workgroup per layer of `ga` (global index base + p; its workgroups start at wg_base + unit_end[p - 1] / NWAVES)
__global__ __launch_bounds__(NTHREADS) void kron_pack_table_write_kernel(KronPackGroupArgs ga, char* table, int n_total, int base, long wg_base) {
  const int p = (int)blockIdx.x;
  if (p >= ga.n) return;
  KronPackArgs* items = reinterpret_cast<KronPackArgs*>(table);
  int* first_wg = reinterpret_cast<int*>(table + kron_pack_table_items_bytes(n_total));
  int* wg_layer = reinterpret_cast<int*>(table + kron_pack_table_items_bytes(n_total) + ((long)n_total * 4 + 15) / 16 * 16);
  const long w0 = wg_base + (p ? ga.unit_end[p - 1] : 0) / NWAVES, w1 = wg_base + ga.unit_end[p] / NWAVES;
  if (threadIdx.x == 0) {
    items[base + p] = ga.p[p];
    first_wg[base + p] = (int)w0;
  }
  for (long w = w0 + threadIdx.x; w < w1; w += NTHREADS) wg_layer[w] = base + p;
}
template <typename T>
__global__ __launch_bounds__(NTHREADS) void kron_pack_table_kernel(const char* table, int n_total) {
  const KronPackArgs* items = reinterpret_cast<const KronPackArgs*>(table);
  const int* first_wg = reinterpret_cast<const int*>(table + kron_pack_table_items_bytes(n_total));
  const int* wg_layer = reinterpret_cast<const int*>(table + kron_pack_table_items_bytes(n_total) + ((long)n_total * 4 + 15) / 16 * 16);
  const int p = __builtin_amdgcn_readfirstlane(wg_layer[blockIdx.x]);
  const long u = ((long)blockIdx.x - first_wg[p]) * NWAVES + (threadIdx.x >> 6);
  const KronPackArgs a = items[p];
  kron_pack_units<T>(a, u);
}

// ---------------------------------------------------------------------------------------------------------------------
// Low-rank w2 = w2a @ w2b (reference modules/lokr.py:131-136, 370; functional/lokr.py:124-151): the chain rule of the product
//     d_w2a[q, r] += sum_v dW2[q, v] * w2b[r, v]        d_w2b[r, v] += sum_q w2a[q, r] * dW2[q, v]
// for MANY layers per launch, from the dense dW2 [c, d] the grouped weight-gradient launch left in a scratch arena (fp32 FMAs:
// 4 c d r flops per layer, a few GFLOP per SDXL step -- the two ATen GEMM launches per layer it replaces cost more in launches
// than in arithmetic).
// ---------------------------------------------------------------------------------------------------------------------
struct KronLrItem {
  const float* dw2;   // [c, taps, d]   (taps > 1: the window-major layout the Conv2d weight-gradient kernels write)
  const float* w2a;   // [c, r]
  const float* w2b;   // [r, d * taps]  (the reference's layout: column v * taps + tap)
  float* d_w2a;       // [c, r] +=          (NULL: not wanted)
  float* d_w2b;       // [r, d * taps] +=   (NULL: not wanted)
  int c, d, r, taps;
};
constexpr int KLR_MAX = 56;
struct KronLrGroupArgs {
  int n;
  int wg_end[KLR_MAX];
  KronLrItem p[KLR_MAX];
};
static_assert(sizeof(KronLrGroupArgs) <= 3840, "kernel arguments are limited to 4 KiB");

// Both products are small fp32 GEMMs (c, d = 40 .. 1280, r = 4 .. 64): they run on the fp32 matrix core (v_mfma_f32_16x16x4_f32:
// exact fp32 products, no hi/lo split), ONE WAVE per 16 x 16 output tile and contraction chunk, partial tiles added atomically:
//   d_w2a tile (16 rows q  x 16 ranks): A = dW2[q, col..]  B = w2b[rank, col..]   16 columns per step = one float4 load per operand
//   d_w2b tile (16 ranks x 16 columns): A = w2a[q.., rank]  B = dW2[q.., col]     16 rows q per step
// History (round 3, SDXL rank-16 step, 739 Linear layers): one thread per output element with a serial contraction loop: 2.0 ms;
// a wave per row with 16 register accumulators + butterfly reductions: 1.25 ms (181 k waves of ~3 loop trips); this version: see
// HISTORY.md 7.4.
constexpr int KLR_CA = 512;   // columns of dW2 per d_w2a work item
constexpr int KLR_CB = 256;   // rows of dW2 per d_w2b work item
__host__ __device__ inline long kron_lr_waves_a(int c, int d, int r, int taps) {
  const long D = (long)d * taps;
  return (long)((c + 15) / 16) * ((r + 15) / 16) * ((D + KLR_CA - 1) / KLR_CA);
}
__host__ __device__ inline long kron_lr_waves(int c, int d, int r, int taps) {
  const long D = (long)d * taps;
  return kron_lr_waves_a(c, d, r, taps) + ((D + 15) / 16) * ((r + 15) / 16) * ((c + KLR_CB - 1) / KLR_CB);
}

__global__ __launch_bounds__(NTHREADS) void kron_lr_chain_kernel(KronLrGroupArgs ga) {
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;
  const KronLrItem& it = ga.p[p];
  const long w = (long)(b - (p ? ga.wg_end[p - 1] : 0)) * NWAVES + (threadIdx.x >> 6);  // work item of this wave
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  const long D = (long)it.d * it.taps;
  const int nrb = (it.r + 15) / 16;
  const long na = kron_lr_waves_a(it.c, it.d, it.r, it.taps);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (w < na) {  // ---- d_w2a[16 q, 16 ranks] over one chunk of columns
    if (!it.d_w2a) return;  // a frozen factor
    const long nch = (D + KLR_CA - 1) / KLR_CA;
    const long ch = w % nch;
    const int rb = (int)((w / nch) % nrb), tq = (int)(w / (nch * nrb));
    const int q = 16 * tq + li, rr = 16 * rb + li;
    const bool qok = q < it.c, rok = rr < it.r;
    const float* g = it.dw2 + (long)q * D;
    const float* wb = it.w2b + (long)rr * D;
    const long col1 = (ch + 1) * KLR_CA < D ? (ch + 1) * KLR_CA : D;
    const bool vec = (((reinterpret_cast<uintptr_t>(it.dw2) | reinterpret_cast<uintptr_t>(it.w2b)) & 15u) == 0) && (D & 3) == 0;
    for (long col = ch * KLR_CA + 4 * kq; col < col1 + 4 * kq; col += 16) {  // every lane runs the same number of trips
      f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
      if (col + 3 < col1 && (it.d & 3) == 0) {  // four columns of one tap (every shape of the fast paths: c, d multiples of 8)
        if (qok) {
          if (vec) av = *reinterpret_cast<const f32x4*>(g + col);
          else { av[0] = g[col]; av[1] = g[col + 1]; av[2] = g[col + 2]; av[3] = g[col + 3]; }
        }
        if (rok) {
          if (it.taps == 1) {
            if (vec) bv = *reinterpret_cast<const f32x4*>(wb + col);
            else { bv[0] = wb[col]; bv[1] = wb[col + 1]; bv[2] = wb[col + 2]; bv[3] = wb[col + 3]; }
          } else {  // dW2 column tap * d + v  <->  w2b column v * taps + tap; the four share the tap (d % 8 == 0)
            const int tap = (int)(col / it.d);
            const long v0 = col - (long)tap * it.d;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = wb[(v0 + e) * it.taps + tap];
          }
        }
      } else {  // ragged tail / odd d: element by element
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long cc = col + e;
          if (cc < col1) {
            const int tap = (int)(cc / it.d);
            if (qok) av[e] = g[cc];
            if (rok) bv[e] = wb[(cc - (long)tap * it.d) * it.taps + tap];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
    }
    const int oc = 16 * rb + li;  // output: rows 16 tq + 4 kq + e, column = rank oc
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oq = 16 * tq + 4 * kq + e;
      if (oq < it.c && oc < it.r) __hip_atomic_fetch_add(it.d_w2a + (long)oq * it.r + oc, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // ---- d_w2b[16 ranks, 16 columns] over one chunk of rows q
  const long wbi = w - na;
  const long nchb = (it.c + KLR_CB - 1) / KLR_CB;
  const long tcols = (D + 15) / 16;
  if (wbi >= tcols * nrb * nchb || !it.d_w2b) return;
  const long ch = wbi % nchb;
  const int rb = (int)((wbi / nchb) % nrb);
  const long tc = wbi / (nchb * nrb);
  const int rr = 16 * rb + li;          // A operand row
  const long col = 16 * tc + li;        // B operand column (dW2 order)
  const bool rok = rr < it.r, cok = col < D;
  const int q1 = (int)((ch + 1) * KLR_CB < it.c ? (ch + 1) * KLR_CB : it.c);
  for (int q0 = (int)(ch * KLR_CB) + 4 * kq; q0 < q1 + 4 * kq; q0 += 16) {
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (q0 + e < q1) {
        if (rok) av[e] = it.w2a[(long)(q0 + e) * it.r + rr];
        if (cok) bv[e] = it.dw2[(long)(q0 + e) * D + col];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
  }
  if (cok) {
    const int tap = (int)(col / it.d);
    const long wcol = (col - (long)tap * it.d) * it.taps + tap;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int orr = 16 * rb + 4 * kq + e;
      if (orr < it.r) __hip_atomic_fetch_add(it.d_w2b + (long)orr * D + wcol, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Conv kernel
// ---------------------------------------------------------------------------------------------------------------------
struct KconvGeom {      // host-computed (capi.hip: plan_kconv)
  int TH, TW;           // destination tile (TH * TW * G == 64 MI rows)
  int PH, PW;           // source patch
  int tiles_h, tiles_w; // tiles per image
  int GP, CP;           // LDS pitch (elements) of a group segment / a patch pixel
  int sy, sx;           // destination -> patch step (stride for the forward, 1 for the backward)
  int oy0, ox0;         // patch origin = tile origin * s + o0 (source coordinates)
  int kss;              // k steps per LDS stage of the B ring
  int patch_bytes;      // PH * PW * CP * 2, rounded up to 1 KiB
};

constexpr int KC_ZERO_BYTES = 16;  // a 16-byte slot of zeros behind the patch (fragments of the K padding read it)

__host__ __device__ inline int kconv_stage_bytes(int NI, int kss) { return NI * kss * 2048; }
__host__ __device__ inline int kconv_lds_bytes(int NI, const KconvGeom& gm) {
  const int ring = 2 * kconv_stage_bytes(NI, gm.kss);
  return gm.patch_bytes + 1024 + (ring > 8192 ? ring : 8192);  // (the dW1 reduction scratch of 8 waves: 8 KiB)
}

// kron3.h's register epilogue for a caller that supplies the row bookkeeping: stage 2 on the matrix cores, optional fused
// `base + delta`, optional dW1 partial (per-workgroup block of a.dw1_ws, or atomics).
//   row_ok[mi], rofs[mi]: validity and element offset (into y / xref / base) of this lane's output row 16 mi + li of the wave
//   (MI 16-row tiles per wave: the workgroup owns 64 MI stage-1 rows)
//   NW: waves of the workgroup (the dW1 partial sums over all of them; red_smem holds NW KiB)
template <typename T, int MI, int NI, bool WITH_DW1, int NW = NWAVES>
__device__ __forceinline__ void k3_epilogue_rows(const KronArgs& a, char* red_smem, f32x4 (&acc)[MI][NI], const float (&w1raw)[4],
                                                 const bool (&row_ok)[MI], const long (&rofs)[MI], long n0, long wg_index) {
  using F4 = typename Mma16<T>::frag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int G = a.Gin, N = a.N;
  const int lg = 31 - __builtin_clz((unsigned)G);
  F4 a2h, a2l;
  {
    T h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_f<T>(w1raw[j], h[j], l[j]);
    a2h = *reinterpret_cast<F4*>(h);
    a2l = *reinterpret_cast<F4*>(l);
  }
  F4 ident;
  {
    T idv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) idv[e] = TT<T>::from_f((4 * g + e) == li ? 1.f : 0.f);
    ident = *reinterpret_cast<F4*>(idv);
  }
  const bool y_vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 7u) == 0);
  const bool xr_vec = WITH_DW1 && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.xref) & 7u) == 0);
  f32x4 cdw = zero4();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long gn = n0 + 16 * ni + 4 * g;
      T h[4], l[4];
      k3_split4<T>(acc[mi][ni], h, l);
      const F4 sh = *reinterpret_cast<F4*>(h), sl = *reinterpret_cast<F4*>(l);
      f32x4 yv = zero4();
      yv = Mma16<T>::mma(sh, a2h, yv);
      yv = Mma16<T>::mma(sl, a2h, yv);
      yv = Mma16<T>::mma(sh, a2l, yv);
      if (row_ok[mi] && gn < N) {
        T* dst = static_cast<T*>(a.y) + rofs[mi] + gn;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(a.alpha * yv[e]);
        if (y_vec && gn + 4 <= N) {
          *reinterpret_cast<u32x2*>(dst) = *reinterpret_cast<u32x2*>(o);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gn + e < N) dst[e] = o[e];
        }
      }
      if constexpr (WITH_DW1) {
        const f32x4 th = Mma16<T>::mma(sh, ident, zero4());
        const f32x4 tl = Mma16<T>::mma(sl, ident, zero4());
        T thv[4], tlv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          thv[e] = TT<T>::from_f(th[e]);
          tlv[e] = TT<T>::from_f(tl[e]);
        }
        const T* xr = static_cast<const T*>(a.xref) + rofs[mi] + gn;
        T bv[4];
        if (xr_vec) {
          const bool ok = row_ok[mi] && gn < N;  // gn + 4 <= N or gn >= N
          const u32x2 z = {0u, 0u};
          const u32x2 v = *reinterpret_cast<const u32x2*>(ok ? xr : static_cast<const T*>(a.xref));
          *reinterpret_cast<u32x2*>(bv) = ok ? v : z;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = (row_ok[mi] && gn + e < N) ? xr[e] : TT<T>::from_f(0.f);
        }
        const F4 bf = *reinterpret_cast<F4*>(bv);
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(thv), bf, cdw);
        cdw = Mma16<T>::mma(*reinterpret_cast<F4*>(tlv), bf, cdw);
      }
    }
  }
  if constexpr (WITH_DW1) {
    float* red = reinterpret_cast<float*>(red_smem);
    __syncthreads();  // other waves may still read the last B stage
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (4 * g + r) * 16 + li] = cdw[r];
    __syncthreads();
    if (tid < G * G) {
      const int u = tid >> lg, po = tid & (G - 1);
      float s = 0.f;
      for (int b = 0; b < (16 >> lg); ++b) {
        const int e = ((b << lg) + u) * 16 + (b << lg) + po;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[256 * w + e];
      }
      const long e = (long)po * a.s1o + (long)u * a.s1i;
      if (a.dw1_ws != nullptr)
        a.dw1_ws[wg_index * (G * G) + e] = a.alpha * s;
      else
        __hip_atomic_fetch_add(a.dw1 + e, a.alpha * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct KconvArgs {
  KronArgs k;          // x = source rows [B * Hs * Ws, G * K], y = destination rows [B * Hd * Wd, G * N]; w2 unused
  const void* planes;  // packed role planes: ceil(N / 16) n tiles x ksteps units
  KconvGeom gm;
  int ksteps;          // k steps of 32 over the flat (tap, k) index
};

// NW = 4: rounds 3 - 5, every wave owns MI row tiles x all NI column tiles.  NW = 8 (round 6): a kconv workgroup is alone on its CU
// (its patch takes 60 - 140 KiB of LDS), so with 4 waves every SIMD runs ONE wave whose fragment reads, MFMAs and operand DMAs follow
// each other (profiles/r06_c37_kconv_waves.log: 5 250 cycles per stage of 4 k steps for 1 024 cycles of MFMAs and 1 250 of LDS reads).
// Two waves per SIMD overlap them: the 64 MI rows x 16 NI columns of the workgroup are split 4 x 2 over the waves (NI even: each wave
// MI x NI / 2 tiles) or 8 x 1 (NI = 3: MI / 2 x NI).  Operand ring, patch and tile plan are unchanged.
//
// PIPE (round 6, second step; needs gm.kss == 4 and K >= 32): the k loop as a software pipeline.  The loop above it (kept: PIPE = false, the
// LYC_KCONV_SERIAL switch, and every plan with fewer than 4 k steps per stage) runs each k step as its own basic block -- tap-table
// ds_read_b32 -> wait -> 6 fragment ds_read_b128 -> wait -> 8 MFMAs (`hipcc -S`: one s_waitcnt lgkmcnt(0) in front of every MFMA pair), a
// dependent chain of ~550 cycles around 128 cycles of matrix-core issue, and every stage ends in a full drain (vmcnt(0) + barrier) before
// the first read of the next one.  Here a stage is ONE straight-line region of four unguarded k steps: the fragments of k step i + 1 are
// read (into a second register set) before the MFMAs of k step i issue, the tap offset advances incrementally (no LDS table, no division), the
// stage barrier sits in FRONT of the last k step's MFMAs (the next stage's first fragments are requested behind it, so the barrier
// latency is covered by 8 MFMAs) and the operand DMA of stage s + 2 is issued there -- a whole stage before anybody waits for it.
// K steps beyond the flat K run on a zero A fragment and a finite duplicate B unit.
template <typename T, int MI, int NI, bool WITH_DW1, int NW = NWAVES, bool PIPE = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && MI * NI <= 24) ? 2 : 1) void kconv_kernel(KconvArgs ca) {
  constexpr int WN = (NW == 8 && NI % 2 == 0) ? 2 : 1;  // waves along the columns
  constexpr int WM = NW / WN;                              // ... along the rows
  constexpr int MW = MI * 4 / WM, NIW = NI / WN;           // row / column tiles of one wave
  extern __shared__ __attribute__((aligned(1024))) char kc_smem[];
  const KronArgs& a = ca.k;
  const KconvGeom& gm = ca.gm;
  using F8 = typename TT<T>::frag;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int G = a.Gin, K = a.K, N = a.N;
  const int lg = 31 - __builtin_clz((unsigned)G);
  // (An XCD-band tile order -- XCD x takes a contiguous eighth of the pixel tiles, so that a halo row is fetched by one L2 -- was measured in
  // round 6 and changed neither the launch time nor the family's fabric traffic, profiles/r06_c45_kcbench_xcd_band_order.log: the 2.1x
  // over the algorithmic bytes is the operand planes and the dW1 partials, not halo duplication.  Identity order.)
  const int bx = (int)blockIdx.x;
  const int by = (int)blockIdx.y;
  const long n0 = (long)by * (16 * NI);

  // ---- tile -> (image, tile row, tile column) -------------------------------------------------------------------------
  const int tpi = gm.tiles_h * gm.tiles_w;
  const int img = bx / tpi;
  const int trem = bx - img * tpi;
  const int th = trem / gm.tiles_w, tw = trem - th * gm.tiles_w;
  const int hd0 = th * gm.TH, wd0 = tw * gm.TW;          // destination tile origin
  const int hs0 = hd0 * gm.sy + gm.oy0, ws0 = wd0 * gm.sx + gm.ox0;  // patch origin in the source image (may be negative)

  char* zero_slot = kc_smem + gm.patch_bytes;
  char* ring = kc_smem + gm.patch_bytes + 1024;
  const int stage_bytes = kconv_stage_bytes(NI, gm.kss);
  LYC_TRACE_DECL;  // benchmarks/ktrace_conv.cpp (-DLYC_TRACE): shader-clock stamps of workgroup 0; nothing in the library build
  LYC_STAMP(0);

  // ---- B ring: stage s = k steps [s * kss, (s + 1) * kss) of the NI n tiles, unit (ni, kk) at ((ni * kss + kk) * 2048) ------
  const char* planes = static_cast<const char*>(ca.planes);
  const int nstage = (ca.ksteps + gm.kss - 1) / gm.kss;
  const int ntiles_n = (N + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs_planes =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(planes), 0, (int)((long)ntiles_n * ca.ksteps * 2048), 0x00020000);
  auto issue_stage = [&](int s, char* buf) {
    if constexpr (PIPE) {  // kss == 4: 8 NI pieces, 8 NI / NW per wave, fully unrolled; k steps past the end re-load the last unit
      if (4 * s >= ca.ksteps) return;  // (wave-uniform)
      constexpr int NPW = 8 * NI / NW;
      static_assert(NPW * NW == 8 * NI, "pieces divide over the waves");
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int p = wave + i * NW;
        const int unit = p >> 1, half = p & 1;
        const int ni = unit >> 2, kk = unit & 3;
        int ks = 4 * s + kk;
        if (ks >= ca.ksteps) ks = ca.ksteps - 1;
        long nt = n0 / 16 + ni;
        if (nt >= ntiles_n) nt = ntiles_n - 1;
        const unsigned off = (unsigned)(((nt * ca.ksteps + ks) * 2 + half) * 1024 + lane * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_planes, (__attribute__((address_space(3))) void*)(buf + p * 1024), 16, (int)off, 0, 0, 0);
      }
      return;
    }
    // NI * kss units of 2 KiB = 2 * NI * kss pieces of 1 KiB; piece p -> wave p % 4 (wave-uniform loop)
    const int npiece = 2 * NI * gm.kss;
    for (int p = wave; p < npiece; p += NW) {
      const int unit = p >> 1, half = p & 1;
      const int ni = unit / gm.kss, kk = unit - ni * gm.kss;
      const int ks = s * gm.kss + kk;
      long nt = n0 / 16 + ni;
      if (nt >= ntiles_n) nt = ntiles_n - 1;  // column tile beyond N: a valid duplicate (its results are never stored; finite
                                              // values keep the dW1 partial, which multiplies them by zeros, finite)
      if (ks < ca.ksteps) {
        // buffer_load ... lds (descriptor form) instead of global_load_lds: -3 .. -5 % per launch (profiles/r04_ktrace_conv.log)
        const unsigned off = (unsigned)(((nt * ca.ksteps + ks) * 2 + half) * 1024 + lane * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_planes, (__attribute__((address_space(3))) void*)(buf + p * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };
  issue_stage(0, ring);
  LYC_STAMP(20);

  // ---- w1 block operand and tap table -------------------------------------------------------------------------------------
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
  // (inside the one dynamic LDS array: a second __shared__ object makes hipcc wait vmcnt(0) before every ds_read that follows
  // an LDS-DMA, i.e. it would serialise the ring -- cdna_hip_programming.md 5, trap 4a)
  int* kc_tapoff = reinterpret_cast<int*>(zero_slot + 64);
  if (tid < a.gat.taps) {
    const int kh = a.gat.taps / a.gat.kw;
    const int i = tid / a.gat.kw, j = tid - i * a.gat.kw;
    // forward: the source of tap (i, j) is patch (py + i dh, px + j dw); backward (stride 1): (py + (kh-1-i) dh, px + (kw-1-j) dw)
    const int oi = a.gat.mode == 1 ? i * a.gat.dh : (kh - 1 - i) * a.gat.dh;
    const int oj = a.gat.mode == 1 ? j * a.gat.dw : (a.gat.kw - 1 - j) * a.gat.dw;
    kc_tapoff[tid] = (oi * gm.PW + oj) * gm.CP * (int)sizeof(T);
  }
  if (tid < 4) reinterpret_cast<uint32_t*>(zero_slot)[tid] = 0u;
  LYC_STAMP(21);

  // ---- source patch: HBM -> LDS by LDS-DMA (buffer_load ... lds), every piece in flight at once ---------------------------------
  // One wave instruction fills 1 KiB of the patch image: lane l supplies the global address of the 16 bytes that belong at
  // (piece base + 16 l) -- a slot of [patch pixel][group u][GP] -- through a buffer descriptor of THIS image, so slots outside
  // the image (zero padding), in the group-pitch padding or beyond the patch get an out-of-range offset and the hardware
  // writes zeros.  No staging registers, no ds_write, and -- what the first version of this kernel lacked (38 us per launch
  // with 4 loads in flight per wave) -- the whole patch is requested before anything waits.
  {
    const T* x = static_cast<const T*>(a.x);
    const int vpp = (G * K) / 8;            // 16-byte vectors of real data per pixel
    const int kv = K / 8;                    // ... per group segment
    const int spg = gm.GP / 8;               // 16-byte slots per group segment (kv, or kv + 1 with the pad)
    const int spp = G * spg;                 // slots per patch pixel
    const int npix = gm.PH * gm.PW;
    const long img_elems = (long)a.gat.Hs * a.gat.Ws * G * K;  // < 2^30 elements (host check)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (long)img * img_elems), 0,
                                                                        (int)(img_elems * (long)sizeof(T)), 0x00020000);
    const float inv_spp = 1.0f / (float)spp, inv_spg = 1.0f / (float)spg, inv_pw = 1.0f / (float)gm.PW;
    const int OOR = 0x7ffffff0;
    const int npiece = gm.patch_bytes >> 10;
    for (int pc = wave; pc < npiece; pc += NW) {
      const int sl = pc * 64 + lane;  // slot index in the patch image
      int pp = (int)(((float)sl + 0.5f) * inv_spp);
      int q = sl - pp * spp;
      if (q < 0) { q += spp; --pp; }
      if (q >= spp) { q -= spp; ++pp; }
      int u = (int)(((float)q + 0.5f) * inv_spg);
      int kq = q - u * spg;
      if (kq < 0) { kq += spg; --u; }
      if (kq >= spg) { kq -= spg; ++u; }
      int py = (int)(((float)pp + 0.5f) * inv_pw);
      int px = pp - py * gm.PW;
      if (px < 0) { px += gm.PW; --py; }
      if (px >= gm.PW) { px -= gm.PW; ++py; }
      const int hs = hs0 + py, ws = ws0 + px;
      const bool ok = pp < npix && kq < kv && hs >= 0 && hs < a.gat.Hs && ws >= 0 && ws < a.gat.Ws;
      const int off = ok ? ((hs * a.gat.Ws + ws) * vpp + u * kv + kq) * 16 : OOR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(kc_smem + pc * 1024), 16, off, 0, 0, 0);
    }
  }

  LYC_STAMP(1);  // stage 0 of the ring and the whole patch requested
  // ---- this lane's two stage-1 rows: local pixel (ly, lx), group u; byte offset of (pixel, u) in the patch ----------------
  int rowbase[MW];
  bool row_ok[MW];
  long rofs[MW];
#pragma unroll
  for (int mi = 0; mi < MW; ++mi) {
    const int r = wm * (16 * MW) + mi * 16 + li;  // local stage-1 row
    const int lp = r >> lg, u = r & (G - 1);
    const int ly = lp / gm.TW, lx = lp - ly * gm.TW;
    rowbase[mi] = ((ly * gm.sy * gm.PW + lx * gm.sx) * gm.CP + u * gm.GP) * (int)sizeof(T);
    const int hd = hd0 + ly, wd = wd0 + lx;
    row_ok[mi] = hd < a.gat.Hd && wd < a.gat.Wd;
    const long dpix = ((long)img * a.gat.Hd + hd) * a.gat.Wd + wd;
    rofs[mi] = dpix * ((long)G * N) + (long)u * N;
  }

  f32x4 acc[MW][NIW];
#pragma unroll
  for (int mi = 0; mi < MW; ++mi)
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) acc[mi][ni] = zero4();

  const int Kflat = a.gat.taps * K;
  const float inv_k = 1.0f / (float)K;
  const int zero_ofs = gm.patch_bytes;  // byte offset of the zero slot from the patch base
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage 0 has landed (LDS-DMA counts on vmcnt)
  __syncthreads();  // patch, tap table, zero slot, stage 0
  LYC_STAMP(2);

  if constexpr (PIPE) {
    const int kh_ = a.gat.taps / a.gat.kw;
    const int sgn = a.gat.mode == 1 ? 1 : -1;  // forward: tap (i, j) reads patch (py + i dh, px + j dw); backward: the flipped window
    const int ci = sgn * a.gat.dh * gm.PW * gm.CP * (int)sizeof(T);
    const int cj = sgn * a.gat.dw * gm.CP * (int)sizeof(T);
    const int c0 = a.gat.mode == 1 ? 0 : ((kh_ - 1) * a.gat.dh * gm.PW + (a.gat.kw - 1) * a.gat.dw) * gm.CP * (int)sizeof(T);
    // LDS byte offsets of this lane's A fragments, k step by k step (K >= 32: at most one tap boundary per step of 32).  The state
    // (flat k, channel v, window column tj, byte offset of (tap, v)) advances with adds, compares and selects only -- the closed form
    // (two float divisions with fix-ups and four v_mul_lo_u32 per k step, as the serial loop has it) is ~45 VALU operations per k step
    // and wave, several of them quarter rate: as many SIMD cycles as the k step's 8 MFMAs
    const int d_j = cj - K * (int)sizeof(T);
    const int d_i = ci - (a.gat.kw - 1) * cj - K * (int)sizeof(T);
    const int kw1 = a.gat.kw - 1;
    int kf = 8 * g, v_ = 8 * g, tj_ = 0, toff = c0 + 8 * g * (int)sizeof(T);
    auto a_offsets = [&](int (&ao)[MW]) {
      const bool kok = kf < Kflat;
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) ao[mi] = kok ? rowbase[mi] + toff : zero_ofs;
    };
    auto advance = [&]() {
      kf += 32;
      v_ += 32;
      const bool wrap = v_ >= K;
      const bool wrapj = wrap && tj_ == kw1;
      toff += 32 * (int)sizeof(T) + (wrap ? (wrapj ? d_i : d_j) : 0);
      v_ -= wrap ? K : 0;
      tj_ = wrap ? (wrapj ? 0 : tj_ + 1) : tj_;
    };
    F8 af[2][MW], bh[2][NIW], bl[2][NIW];
    auto load_frags = [&](F8 (&fa)[MW], F8 (&fh)[NIW], F8 (&fl)[NIW], const char* buf, int kk, const int (&ao)[MW]) {
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) fa[mi] = *reinterpret_cast<const F8*>(kc_smem + ao[mi]);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const char* up = buf + ((wn * NIW + ni) * 4 + kk) * 2048 + lane * 16;
        fh[ni] = *reinterpret_cast<const F8*>(up);
        fl[ni] = *reinterpret_cast<const F8*>(up + 1024);
      }
    };
    auto mfmas = [&](const F8 (&fa)[MW], const F8 (&fh)[NIW], const F8 (&fl)[NIW]) {
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) acc[mi][ni] = TT<T>::mma(fa[mi], fh[ni], acc[mi][ni]);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) acc[mi][ni] = TT<T>::mma(fa[mi], fl[ni], acc[mi][ni]);
    };
    int ao[MW];
    a_offsets(ao);
    load_frags(af[0], bh[0], bl[0], ring, 0, ao);
    for (int s = 0; s < nstage; ++s) {
      char* buf = ring + (s & 1) * stage_bytes;
      const char* nbuf = ring + ((s + 1) & 1) * stage_bytes;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        // stage 1 is requested behind the prologue, not in it: the prologue is fill-rate bound (every CU's workgroup bursts its
        // patch at once, ~12 B / clk / CU), and hipcc drains vmcnt at the loop header whatever was counted before it
        if (kk == 0 && s == 0) issue_stage(1, ring + stage_bytes);
        advance();
        a_offsets(ao);
        if (kk < 3) {
          load_frags(af[nxt], bh[nxt], bl[nxt], buf, kk + 1, ao);
        } else {
          // stage boundary: this wave's reads of `buf` have returned and its share of stage s + 1 has landed; behind the barrier
          // everybody's has, and `buf` is free for stage s + 2
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          issue_stage(s + 2, buf);
          load_frags(af[nxt], bh[nxt], bl[nxt], nbuf, 0, ao);  // (behind the last stage: never consumed)
        }
        // (without the fences the machine scheduler sinks every fragment read to its first use again -- `hipcc -S`: 76 VGPRs, one
        // s_waitcnt lgkmcnt(0..1) in front of every MFMA pair -- i.e. it undoes the pipeline to save the second register set)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(af[cur], bh[cur], bl[cur]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s < 17) LYC_STAMP(3 + s);
    }
  } else
  for (int s = 0; s < nstage; ++s) {
    char* buf = ring + (s & 1) * stage_bytes;
    if (s + 1 < nstage) issue_stage(s + 1, ring + ((s + 1) & 1) * stage_bytes);
    const int ks_lo = s * gm.kss;
    const int nk = (ca.ksteps - ks_lo) < gm.kss ? (ca.ksteps - ks_lo) : gm.kss;
    // one k step: A fragments from the patch, B fragments from the ring, 4 NI MFMAs
    auto kstep = [&](int kk, int kss_) {
      const int k = 32 * (ks_lo + kk) + 8 * g;  // this lane's 8 flat k (one tap: K % 8 == 0)
      int tap = (int)(((float)k + 0.5f) * inv_k);
      int v = k - tap * K;
      if (v < 0) { v += K; --tap; }
      if (v >= K) { v -= K; ++tap; }
      const bool kok = k < Kflat;
      const int toff = kc_tapoff[kok ? tap : 0] + v * (int)sizeof(T);
      F8 af[MW];
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) af[mi] = *reinterpret_cast<const F8*>(kc_smem + (kok ? rowbase[mi] + toff : zero_ofs));
      F8 bh[NIW], bl[NIW];
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const char* up = buf + ((wn * NIW + ni) * kss_ + kk) * 2048 + lane * 16;
        bh[ni] = *reinterpret_cast<const F8*>(up);
        bl[ni] = *reinterpret_cast<const F8*>(up + 1024);
      }
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bh[ni], acc[mi][ni]);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) acc[mi][ni] = TT<T>::mma(af[mi], bl[ni], acc[mi][ni]);
    };
    // ONE code region updates the accumulators: four guarded steps (kss <= 4).  Two alternative regions (a straight-line body for
    // full stages, a loop for the rest) made the register allocator copy every AGPR accumulator at their join -- 318 / 478 / 638
    // v_accvgpr_mov per stage in the MI = 8 instantiations (round 4, `hipcc -S`), more cycles than the stage's MFMAs for NI = 2.
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (kk < nk) kstep(kk, gm.kss);
    if (s < 17) LYC_STAMP(3 + s);  // MFMAs of stage s issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage s + 1 has landed
    __syncthreads();  // ... everybody's has, and nobody still reads stage s's buffer
  }
  LYC_STAMP(28);

  k3_epilogue_rows<T, MW, NIW, WITH_DW1, NW>(a, ring, acc, w1raw, row_ok, rofs, n0 + 16 * (wn * NIW), (long)by * gridDim.x + bx);
  LYC_STAMP(29);
  LYC_TRACE_FLUSH();
}

}  // namespace lyc
