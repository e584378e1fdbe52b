// capi.hip -- C ABI launchers (include/lycoris_amd.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <unordered_map>
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/lycoris_amd.h"
#include "conv_kernels.h"
#include "dense_kernels.h"
#include "ia3_kernels.h"
#include "kron3.h"
#include "kron4.h"
#include "kron_dw2s.h"
#include "kron_dw2f.h"
#include "kron_conv.h"
#ifdef LYC_EXPERIMENT_CONV_DW2_PATCH  // benchmarks/experiments/README.md: measured and dropped (5.9 ms vs 2.2 ms), not in the product build
#include "../../benchmarks/experiments/kron_conv_dw2.h"
#endif
#include "loha_mfma.h"
#include "loha_grad16.h"
#include "gemm16.h"
#include "gemm16d.h"
#include "lokr_kernels.h"
#include "lowrank.h"
#include "lowrank4.h"
#include "skinny_kernels.h"
#include "tucker.h"
#include "wspace.h"

using namespace lyc;

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LYC_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return LYC_OK;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline long round_up(long a, long b) { return cdiv(a, b) * b; }

#define DISPATCH_DTYPE(dtype_, ...)                                  \
  switch ((dtype_) & 0xff) {                                                   \
    case LYC_BF16: { using T = __bf16; __VA_ARGS__; } break;         \
    case LYC_F16: { using T = _Float16; __VA_ARGS__; } break;        \
    case LYC_F32: { using T = float; __VA_ARGS__; } break;           \
    default: return fail(LYC_ERR_ARG, "unknown dtype %d", (int)(dtype_));    \
  }

// ------------------------------------------------------------------------------------------------
template <typename T>
bool kron_fast_ok(const KronArgs& ka) {
  if constexpr (sizeof(T) != 2) {
    return false;
  } else {
    return ka.Gin == ka.Gout && (16 % ka.Gin) == 0 && (ka.K % 8) == 0 &&
           (reinterpret_cast<uintptr_t>(ka.x) & 15u) == 0;
  }
}

template <typename T, int NI, bool DW1, int GATHER, bool BASE = false>
void launch_kron3_inst2(const KronArgs& ka, dim3 grid, hipStream_t st) {
  const long nseg = GATHER == 2 ? cdiv((long)ka.gat.taps * ka.K, kron3_kc(NI))
                                : (ka.gat.mode ? ka.gat.taps : 1) * cdiv(ka.K, kron3_kc(NI));
  const int xs = GATHER == 3 ? kron3_xs_bytes() : 0;  // per-wave x tiles behind the w2 tiles
  const int lds = kron3_lds_bytes(NI, nseg > 1 ? 2 : 1) + xs;
  if constexpr (GATHER == 0 || GATHER == 3) {
    if (ka.w2p != nullptr) {  // pre-packed operand planes: the PL instantiation (same LDS footprint)
      hipLaunchKernelGGL((kron3_kernel<T, NI, DW1, GATHER, BASE, true>), grid, dim3(NTHREADS), lds, st, ka);
      return;
    }
  }
  if (lds > 64 * 1024) {  // more than the default dynamic-LDS window: opt in once per instantiation
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&kron3_kernel<T, NI, DW1, GATHER, BASE>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kron3_lds_bytes(NI, 2) + xs);
    (void)once;
  }
  hipLaunchKernelGGL((kron3_kernel<T, NI, DW1, GATHER, BASE>), grid, dim3(NTHREADS), lds, st, ka);
}

template <typename T, int NI, bool DW1>
void launch_kron3_inst(const KronArgs& ka, dim3 grid, hipStream_t st) {
  constexpr int use_xs = 2;  // 0 = never, 1 = only the launches without dW1, 2 = every plain-row launch (measured best)
  if constexpr (!DW1) {
    if (ka.base != nullptr && !ka.gat.mode) {  // forward with the fused `base + delta` epilogue (plain rows only)
      if (use_xs && (ka.K % 32) == 0) launch_kron3_inst2<T, NI, false, 3, true>(ka, grid, st);
      else launch_kron3_inst2<T, NI, false, 0, true>(ka, grid, st);
      return;
    }
  }
  if (ka.gat.mode && ka.gat.flat) launch_kron3_inst2<T, NI, DW1, 2>(ka, grid, st);
  else if (ka.gat.mode) launch_kron3_inst2<T, NI, DW1, 1>(ka, grid, st);
  // x through the per-wave LDS stage (quad-coalesced loads): -3 % on the forward launches, -2 % on the backward ones
  // (once the w2 tile is requested ahead of x); LYC_K3_XS=0 switches it off, =1 restricts it to the launches without dW1
  else if (use_xs && (use_xs > 1 || !DW1) && (ka.K % 32) == 0) launch_kron3_inst2<T, NI, DW1, 3>(ka, grid, st);
  else launch_kron3_inst2<T, NI, DW1, 0>(ka, grid, st);
}

// 64-column tiles halve the x re-reads and the per-column w2 conversions; 32-column tiles double the workgroup count.
// Small problems (fewer 64-wide tiles than ~1.5 per CU) are latency-bound and take the narrow tile.
inline int kron3_pick_ni(const KronArgs& ka) {
  const long mt = cdiv(ka.M, K3_RT / ka.Gin);
  return (mt * cdiv(ka.N, 64) >= 384 && ka.N > 32) ? 4 : 2;
}

// ---- kron4 (round 4): nn.Linear rows on packed planes -------------------------------------------------------------------------
// Tile plan from the measured shape sweep (benchmarks/k4bench.cpp, profiles/r04_k4bench*.log): the widest column tile whose grid
// still has >= 256 workgroups (N % 80 == 0 -> five 16-column tiles: one or two column blocks for the 640- / 1280-wide layers, x
// re-read 1-2x instead of 3-5x), otherwise the latency plan 64 rows x 32 columns; ring depth 2 when the whole K fits five k steps.
struct Kron4Plan {
  int MI, NI, D;
  bool NP;  // paired column tiles: 16-byte stores (kron4.h); needs every tile of every column block to exist
};
inline Kron4Plan kron4_plan(long rows, int K, int N) {
  const int KS = (K + 31) / 32;
  // wide column tiles (64 columns = one 128-byte line per row and workgroup; 80 for the widths that are not multiples of 64) as long
  // as the grid keeps >= 256 workgroups; otherwise the latency plan of 64 rows x 32 columns
  int ni = (N % 64 == 0) ? 4 : ((N % 80 == 0) ? 5 : 2);
  if (ni > 2 && cdiv(rows, 128) * cdiv(N, 16 * ni) < 256) ni = 2;
  if (ni >= 4) return {2, ni, KS <= 5 ? 2 : 3, true};
  const bool np = (N % 32) == 0;
  if (cdiv(rows, 128) * cdiv(N, 32) >= 1024) return {2, 2, 2, np};  // many short rows (SD1.5 320-wide layers at batch 4)
  return {1, 2, 3, np};
}
inline bool kron4_dims_ok(long M, int G, int K, int N) {
  return G >= 1 && (16 % G) == 0 && (K % 8) == 0 && (N % 8) == 0 && M > 0 && M * G * (long)K * 2 < (1L << 31) &&
         M * G * (long)N * 2 < (1L << 31) && kron_plane_bytes(N, 1, K) < (1L << 31);
}
inline long kron4_blocks(long M, int G, int K, int N) {
  const Kron4Plan p = kron4_plan(M * G, K, N);
  return cdiv(M * G, 64 * p.MI) * cdiv(N, 16 * p.NI);
}
template <typename T>
bool kron4_ok(const KronArgs& ka) {
  if constexpr (sizeof(T) != 2) {
    return false;
  } else {
    return ka.w2p != nullptr && !ka.gat.mode && !ka.out_f32 && ka.Gin == ka.Gout && kron4_dims_ok(ka.M, ka.Gin, ka.K, ka.N) &&
           (reinterpret_cast<uintptr_t>(ka.x) & 15u) == 0 && (reinterpret_cast<uintptr_t>(ka.y) & 15u) == 0 &&
           (!ka.base || (reinterpret_cast<uintptr_t>(ka.base) & 15u) == 0) && (!ka.xref || (reinterpret_cast<uintptr_t>(ka.xref) & 15u) == 0) &&
           (ka.dw1 == nullptr || ka.dw1_ws != nullptr);  // the w1 gradient leaves as per-workgroup partials only
  }
}
template <typename T, int MI, int NI, int D, bool NP>
void launch_kron4_inst(const Kron4Args& a, int epi, dim3 grid, hipStream_t st) {
  constexpr int lds = kron4_lds_bytes(MI, NI, D);
  static_assert(lds <= 160 * 1024, "LDS");
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) {  // more than the default dynamic-LDS window: opt in once per instantiation
      static const hipError_t once =
          hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)once;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, a);
  };
  if (epi == 0) go(kron4_kernel<T, MI, NI, D, 0, NP>);
  else if (epi == 1) go(kron4_kernel<T, MI, NI, D, 1, NP>);
  else go(kron4_kernel<T, MI, NI, D, 2, NP>);
}
inline Kron4Args kron4_args(const KronArgs& ka, int& epi) {
  const int G = ka.Gin;
  const long rows = ka.M * G;
  Kron4Args a{};
  a.x = ka.x; a.y = ka.y; a.planes = ka.w2p; a.w1 = ka.w1;
  epi = ka.dw1 ? 2 : (ka.base ? 1 : 0);
  a.aux = epi == 2 ? ka.xref : ka.base;
  a.dw1_ws = ka.dw1_ws;
  a.x_bytes = (unsigned)(rows * ka.K * 2); a.y_bytes = (unsigned)(rows * ka.N * 2);
  a.plane_bytes = (unsigned)kron_plane_bytes(ka.N, 1, ka.K);
  a.rows_total = (int)rows; a.K = ka.K; a.N = ka.N; a.KS = (ka.K + 31) / 32;
  a.lg = 31 - __builtin_clz((unsigned)G);
  a.s1o = (int)ka.s1o; a.s1i = (int)ka.s1i; a.alpha = ka.alpha;
  return a;
}
// returns the number of dw1 partial blocks the consumer has to sum (0: no w1 gradient requested)
template <typename T>
long launch_kron4(const KronArgs& ka, long dw1_blocks_total, hipStream_t st) {
  const long rows = ka.M * ka.Gin;
  const Kron4Plan p = kron4_plan(rows, ka.K, ka.N);
  int epi = 0;
  Kron4Args a = kron4_args(ka, epi);
  dim3 grid((unsigned)cdiv(rows, 64 * p.MI), (unsigned)cdiv(ka.N, 16 * p.NI));
  const long nwg = (long)grid.x * grid.y;
  a.dw1_blocks = (int)(dw1_blocks_total > nwg ? dw1_blocks_total : nwg);
  if (p.NI == 5) {
    if (p.D == 2) launch_kron4_inst<T, 2, 5, 2, true>(a, epi, grid, st);
    else launch_kron4_inst<T, 2, 5, 3, true>(a, epi, grid, st);
  } else if (p.NI == 4) {
    if (p.D == 2) launch_kron4_inst<T, 2, 4, 2, true>(a, epi, grid, st);
    else launch_kron4_inst<T, 2, 4, 3, true>(a, epi, grid, st);
  } else if (p.MI == 2) {
    if (p.NP) launch_kron4_inst<T, 2, 2, 2, true>(a, epi, grid, st);
    else launch_kron4_inst<T, 2, 2, 2, false>(a, epi, grid, st);
  } else {
    if (p.NP) launch_kron4_inst<T, 1, 2, 3, true>(a, epi, grid, st);
    else launch_kron4_inst<T, 1, 2, 3, false>(a, epi, grid, st);
  }
  return epi == 2 ? (long)a.dw1_blocks : 0;
}

// ---- several problems of one shape class in ONE launch (kron4_group_kernel) ------------------------------------------------------
inline long lokr_dx_partial_blocks(long M, int G, int K, int N);
template <typename T, int MI, int NI, int D, bool NP>
void launch_kron4_group_inst(const Kron4GroupArgs& ga, int epi, dim3 grid, hipStream_t st) {
  constexpr int lds = kron4_lds_bytes(MI, NI, D);
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) {
      static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)once;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, ga);
  };
  if (epi == 0) go(kron4_group_kernel<T, MI, NI, D, 0, NP>);
  else if (epi == 1) go(kron4_group_kernel<T, MI, NI, D, 1, NP>);
  else go(kron4_group_kernel<T, MI, NI, D, 2, NP>);
}
// `kas`: n <= K4_GROUP_MAX problems with equal (G, K, N) and the same epilogue; the tile plan is the one of their rows TAKEN TOGETHER
// (three 1024-row projections are planned like one 3072-row layer: wider column tiles, a full chip)
template <typename T>
int launch_kron4_group(const KronArgs* kas, int n, hipStream_t st) {
  long rows_sum = 0, rows_max = 0;
  for (int i = 0; i < n; ++i) {
    rows_sum += kas[i].M * kas[i].Gin;
    rows_max = std::max<long>(rows_max, kas[i].M * kas[i].Gin);
  }
  const Kron4Plan p = kron4_plan(rows_sum, kas[0].K, kas[0].N);
  Kron4GroupArgs ga{};
  ga.n = n;
  int epi = 0;
  const unsigned gy = (unsigned)cdiv(kas[0].N, 16 * p.NI);
  for (int i = 0; i < n; ++i) {
    ga.p[i] = kron4_args(kas[i], epi);
    ga.nbx[i] = (int)cdiv(kas[i].M * kas[i].Gin, 64 * p.MI);
    const long nwg = (long)ga.nbx[i] * gy;
    const long want = kas[i].dw1 ? lokr_dx_partial_blocks(kas[i].M, kas[i].Gin, kas[i].K, kas[i].N) : 0;
    ga.p[i].dw1_blocks = (int)(want > nwg ? want : nwg);
  }
  const dim3 grid((unsigned)cdiv(rows_max, 64 * p.MI), gy, (unsigned)n);
  if (p.NI == 5) {
    if (p.D == 2) launch_kron4_group_inst<T, 2, 5, 2, true>(ga, epi, grid, st);
    else launch_kron4_group_inst<T, 2, 5, 3, true>(ga, epi, grid, st);
  } else if (p.NI == 4) {
    if (p.D == 2) launch_kron4_group_inst<T, 2, 4, 2, true>(ga, epi, grid, st);
    else launch_kron4_group_inst<T, 2, 4, 3, true>(ga, epi, grid, st);
  } else if (p.MI == 2) {
    if (p.NP) launch_kron4_group_inst<T, 2, 2, 2, true>(ga, epi, grid, st);
    else launch_kron4_group_inst<T, 2, 2, 2, false>(ga, epi, grid, st);
  } else {
    if (p.NP) launch_kron4_group_inst<T, 1, 2, 3, true>(ga, epi, grid, st);
    else launch_kron4_group_inst<T, 1, 2, 3, false>(ga, epi, grid, st);
  }
  return LYC_OK;
}

// n <= K4_GROUP_MAX backward problems with equal (M, G, K, N) whose dx results are wanted as their SUM only (the gradient of the one
// tensor n sibling projections read): kron4_sum_kernel, planned like ONE of the problems -- a workgroup walks all n
template <typename T, int MI, int NI, int D, bool NP>
void launch_kron4_sum_inst(const Kron4GroupArgs& ga, void* dx_sum, dim3 grid, hipStream_t st) {
  constexpr int lds = kron4_lds_bytes(MI, NI, D);
  auto kern = kron4_sum_kernel<T, MI, NI, D, NP>;
  if (lds > 64 * 1024) {
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, ga, dx_sum);
}
template <typename T>
int launch_kron4_sum(const KronArgs* kas, int n, void* dx_sum, hipStream_t st) {
  const long rows = kas[0].M * kas[0].Gin;
  const Kron4Plan p = kron4_plan(rows, kas[0].K, kas[0].N);
  Kron4GroupArgs ga{};
  ga.n = n;
  int epi = 0;
  const dim3 grid((unsigned)cdiv(rows, 64 * p.MI), (unsigned)cdiv(kas[0].N, 16 * p.NI));
  const long nwg = (long)grid.x * grid.y;
  for (int i = 0; i < n; ++i) {
    ga.p[i] = kron4_args(kas[i], epi);
    ga.nbx[i] = (int)grid.x;
    const long want = lokr_dx_partial_blocks(kas[i].M, kas[i].Gin, kas[i].K, kas[i].N);
    ga.p[i].dw1_blocks = (int)(want > nwg ? want : nwg);
  }
  if (p.NI == 5) {
    if (p.D == 2) launch_kron4_sum_inst<T, 2, 5, 2, true>(ga, dx_sum, grid, st);
    else launch_kron4_sum_inst<T, 2, 5, 3, true>(ga, dx_sum, grid, st);
  } else if (p.NI == 4) {
    if (p.D == 2) launch_kron4_sum_inst<T, 2, 4, 2, true>(ga, dx_sum, grid, st);
    else launch_kron4_sum_inst<T, 2, 4, 3, true>(ga, dx_sum, grid, st);
  } else if (p.MI == 2) {
    if (p.NP) launch_kron4_sum_inst<T, 2, 2, 2, true>(ga, dx_sum, grid, st);
    else launch_kron4_sum_inst<T, 2, 2, 2, false>(ga, dx_sum, grid, st);
  } else {
    if (p.NP) launch_kron4_sum_inst<T, 1, 2, 3, true>(ga, dx_sum, grid, st);
    else launch_kron4_sum_inst<T, 1, 2, 3, false>(ga, dx_sum, grid, st);
  }
  return LYC_OK;
}

// returns the number of workgroups (= number of dw1 partials when ka.dw1_ws is set)
template <typename T>
long launch_kron3(const KronArgs& ka, hipStream_t st) {
  const int ni = kron3_pick_ni(ka);
  dim3 grid((unsigned)cdiv(ka.M, K3_RT / ka.Gin), (unsigned)cdiv(ka.N, 16 * ni));
  if (ni == 4) {
    if (ka.dw1) launch_kron3_inst<T, 4, true>(ka, grid, st);
    else launch_kron3_inst<T, 4, false>(ka, grid, st);
  } else {
    if (ka.dw1) launch_kron3_inst<T, 2, true>(ka, grid, st);
    else launch_kron3_inst<T, 2, false>(ka, grid, st);
  }
  return (long)grid.x * grid.y;
}

// Number of [a * b] dw1 partial blocks the dx launch of a 16-bit fast-path layer leaves in its workspace.  The consumer
// (lyc_lokr_wgrad_group, the dW2 launch's reducer slice) knows only the layer's dimensions, not which kernel produced the
// partials, so the count is a function of the dimensions alone: the larger of the two kernels' grids; the producer fills the
// surplus blocks with zeros (kron4: in the kernel; kron3 on kron4-shaped dims: a memset behind the launch).
inline long lokr_dx_partial_blocks(long M, int G, int K, int N) {
  KronArgs ka{};
  ka.M = M; ka.Gin = G; ka.K = K; ka.Gout = G; ka.N = N;
  const long p3 = cdiv(M, K3_RT / G) * cdiv(N, 16 * kron3_pick_ni(ka));
  if (!kron4_dims_ok(M, G, K, N)) return p3;
  const long p4 = kron4_blocks(M, G, K, N);
  return p3 > p4 ? p3 : p4;
}

template <typename T>
long launch_kron(KronArgs ka, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (kron4_ok<T>(ka)) return launch_kron4<T>(ka, ka.dw1 ? lokr_dx_partial_blocks(ka.M, ka.Gin, ka.K, ka.N) : 0, st);
    if (kron_fast_ok<T>(ka)) {
      const long p3 = launch_kron3<T>(ka, st);
      if (ka.dw1 && ka.dw1_ws && !ka.gat.mode) {  // plain rows: the consumer sums the canonical number of blocks
        const long pc = lokr_dx_partial_blocks(ka.M, ka.Gin, ka.K, ka.N);
        if (pc > p3) {
          (void)hipMemsetAsync(ka.dw1_ws + p3 * ka.Gin * ka.Gout, 0, (size_t)(pc - p3) * ka.Gin * ka.Gout * sizeof(float), st);
          return pc;
        }
      }
      return p3;
    }
  }
  ka.dw1_ws = nullptr;  // the generic kernel accumulates dw1 with atomics
  const int TM = KronCfg<T>::RT / ka.Gin;
  const bool wide = (ka.N % 64 == 0) || ka.N >= 256;
  dim3 grid((unsigned)cdiv(ka.M, TM), (unsigned)cdiv(ka.N, wide ? 64 : 32));
  if (wide)
    hipLaunchKernelGGL((kron_kernel<T, 64>), grid, dim3(NTHREADS), 0, st, ka);
  else
    hipLaunchKernelGGL((kron_kernel<T, 32>), grid, dim3(NTHREADS), 0, st, ka);
  return 0;
}

// dW2 streaming kernel: tile and split-K selection (see kron_dw2s.h).  dw1 partial reduction rides in an extra z slice.
template <typename T, int MI, int NJ, int U>
void launch_dw2s_inst(KronDw2sArgs da, long tiles_i, long tiles_j, hipStream_t st) {
  da.tiles_i = (int)tiles_i;
  da.tiles_j = (int)tiles_j;
  dim3 grid((unsigned)(round_up(tiles_i * tiles_j * da.nsplit, 8) + (da.dw1_ws != nullptr ? da.dw1_red : 0)));
  if (da.gat.mode)
    hipLaunchKernelGGL((kron_dw2s_kernel<T, MI, NJ, U, true>), grid, dim3(NTHREADS), 0, st, da);
  else
    hipLaunchKernelGGL((kron_dw2s_kernel<T, MI, NJ, U, false>), grid, dim3(NTHREADS), 0, st, da);
}

// fills rows_per_block / nsplit / dw1_red / tiles of `da`; returns true for the 64 x 64 tile, false for 32 x 32
// `grouped`: the problem is one of many in a launch (kron_dw2s_group_kernel): parallelism comes from the other problems, so
// fewer, longer row slabs (less atomic traffic, the set-up of a workgroup amortised over more rows)
// (the LYC_WG_* macros exist for benchmarks/wgbench.cpp, which compiles this file with other values)
#ifndef LYC_WG_TARGET
#define LYC_WG_TARGET 128
#endif
#ifndef LYC_WG_MAXROWS
#define LYC_WG_MAXROWS 4096
#endif
#ifndef LYC_WG_BIG_ROWS
#define LYC_WG_BIG_ROWS 16384
#endif
#ifndef LYC_WG_U
#define LYC_WG_U 2
#endif
#ifndef LYC_GROUP_MIN
#define LYC_GROUP_MIN 8  // layers in a lyc_*_wgrad_group call from which on the batch plans are used
#endif
#ifndef LYC_WG_WIDE
#define LYC_WG_WIDE 2
#endif
#ifndef LYC_CONV_WIDE
#define LYC_CONV_WIDE 0  // measured on the 49 conv layers of SDXL: 9.98 vs 9.83 ms per step -- not worth it
#endif
#ifndef LYC_CONV_BIG
#define LYC_CONV_BIG 1   // 64 x 64 dW2 tiles for the convs (32 x 32: 10.12 vs 9.82 ms per step over the 49 conv layers)
#endif
// tile configurations of kron_dw2s_kernel: (MI, NJ, U)
enum { DW2_T22 = 0, DW2_T44 = 1, DW2_T52 = 2, DW2_NCFG = 3 };
int plan_dw2s(KronDw2sArgs& da, bool grouped = false) {
  const int target_blocks = grouped ? LYC_WG_TARGET : 512;  // single launch: one resident round, 2 workgroups per CU
  constexpr int atomic_budget = 620000;    // fp32 atomics per launch (~300 / ns)
  const long rows_total = da.M * da.G;
  auto plan = [&](int mi, int nj, long& tiles, long& split) {
    tiles = cdiv(da.I, 16 * mi) * cdiv(da.J, 16 * nj);
    long smax = atomic_budget / ((long)da.I * da.J);
    if (smax < 1) smax = 1;
    const long srows = rows_total / 128 > 0 ? rows_total / 128 : 1;
    split = cdiv(target_blocks, tiles);
    if (split > smax) split = smax;
    if (split > srows) split = srows;
    if (split < 1) split = 1;
  };
  long t44, s44, t22, s22;
  plan(4, 4, t44, s44);
  plan(2, 2, t22, s22);
  // 64 x 64 tiles read 128-byte row segments (the 32 x 32 tile's 64-byte segments halve the address-coalescer rate) and
  // do 4x the matrix work per loaded byte: they win whenever there are enough rows to split (measured: M*G >= 16k) and
  // the padding of I, J to multiples of 64 does not waste more than half of the tile.
  const double eff44 = (double)da.I * da.J / ((double)round_up(da.I, 64) * round_up(da.J, 64));
  bool big = rows_total >= (grouped ? LYC_WG_BIG_ROWS : 16384) && eff44 >= 0.5 && (LYC_CONV_BIG || da.gat.mode == 0);
  // grouped launches are instruction-bound (benchmarks/wgbench.cpp, profiles/r02_wgbench_sweep*.log), and most of a step's
  // instructions belong to the MIXED operand (G x G mix on the matrix cores + hi/lo split of its result), whose cost goes with
  // the tile's J extent only: an 80 x 32 tile does 2.5x the output per mixed column (SDXL mix: 4.65 -> 3.62 ms; it also beats
  // the 64 x 64 tile, which runs at one wave per SIMD).  Every SDXL / SD1.5 w2 has c = O / 8 a multiple of 80 or 40.
  // (also for the single-layer launches of the implicit Conv2d: its 64 x 64 tile needs 340 registers, one wave per SIMD)
  const bool wide = (grouped || (LYC_CONV_WIDE && da.gat.mode != 0)) && LYC_WG_WIDE && (LYC_WG_WIDE == 2 || !big) && da.I >= 80 &&
                    (double)da.I / (double)round_up(da.I, 80) >= 0.8;
  if (wide) big = false;
  long t52 = 0, s52 = 0;
  if (wide) plan(5, 2, t52, s52);
  long split = wide ? s52 : big ? s44 : s22;
  const long tiles = wide ? t52 : big ? t44 : t22;
  while (split > 1 && tiles * split > target_blocks) --split;  // one resident round: 2 workgroups per CU
  if (grouped) {  // ... but no slab longer than 4096 rows (the tail of the launch), atomic budget permitting
    long smax = atomic_budget / ((long)da.I * da.J);
    if (smax < 1) smax = 1;
    while (split < smax && cdiv(rows_total, split) > LYC_WG_MAXROWS) ++split;
  }
  if (split > 8) {
    // the work items are dealt to the 8 XCDs in contiguous eighths of the slab-major order (kernel block mapping): with a
    // multiple of 8 slabs no slab straddles two XCDs (measured: 41 slabs run 1.7x slower than 40).  Round up when the
    // workgroup cap and the atomic budget allow it, else down if that costs little parallelism.
    const long down = split - split % 8, up = down + 8;
    const bool up_ok = tiles * up <= target_blocks && (double)up * da.I * da.J <= (double)atomic_budget;
    if (split % 8 != 0) split = up_ok ? up : (down * 10 >= split * 9 ? down : split);  // give up at most 10 % of the slabs
  }
  da.rows_per_block = round_up(cdiv(rows_total, split), 32);
  da.nsplit = (int)cdiv(rows_total, da.rows_per_block);
  if (da.dw1_ws != nullptr) {
    long r = da.dw1_nblk / 64;
    if (r > 16) r = 16;
    if (r < 1) r = 1;
    da.dw1_red = (int)r;
  }
  da.tiles_i = (int)cdiv(da.I, wide ? 80 : big ? 64 : 32);
  da.tiles_j = (int)cdiv(da.J, big ? 64 : 32);
  return wide ? DW2_T52 : big ? DW2_T44 : DW2_T22;
}

template <typename T>
void launch_dw2s(KronDw2sArgs da, hipStream_t st) {
  const int cfg = plan_dw2s(da);
  if (cfg == DW2_T44) launch_dw2s_inst<T, 4, 4, 1>(da, da.tiles_i, da.tiles_j, st);
  else if (cfg == DW2_T52) launch_dw2s_inst<T, 5, 2, 1>(da, da.tiles_i, da.tiles_j, st);
  else launch_dw2s_inst<T, 2, 2, 4>(da, da.tiles_i, da.tiles_j, st);
}

template <typename T>
void launch_kron_dw2(KronDw2Args da, hipStream_t st) {
  const long rows_total = da.M * da.Gs;
  const int mi = da.I <= 64 ? 4 : da.I <= 128 ? 8 : da.I <= 192 ? 12 : 16;
  const long tiles = cdiv(da.I, 16 * mi) * cdiv(da.J, 64);
  long split = cdiv(512, tiles);
  const long max_split = cdiv(rows_total, 64);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  da.rows_per_block = round_up(cdiv(rows_total, split), 32);
  split = cdiv(rows_total, da.rows_per_block);
  dim3 grid((unsigned)cdiv(da.I, 16 * mi), (unsigned)cdiv(da.J, 64), (unsigned)split);
  switch (mi) {
    case 4: hipLaunchKernelGGL((kron_dw2_kernel<T, 4>), grid, dim3(NTHREADS), 0, st, da); break;
    case 8: hipLaunchKernelGGL((kron_dw2_kernel<T, 8>), grid, dim3(NTHREADS), 0, st, da); break;
    case 12: hipLaunchKernelGGL((kron_dw2_kernel<T, 12>), grid, dim3(NTHREADS), 0, st, da); break;
    default: hipLaunchKernelGGL((kron_dw2_kernel<T, 16>), grid, dim3(NTHREADS), 0, st, da); break;
  }
}

int check_kron_dims(int64_t M, int a, int b, int c, int d) {
  if (M < 0 || a < 1 || b < 1 || c < 1 || d < 1) return fail(LYC_ERR_ARG, "lokr: bad dims M=%ld a=%d b=%d c=%d d=%d", (long)M, a, b, c, d);
  if (a > 128 || b > 128)
    return fail(LYC_ERR_UNSUPPORTED, "lokr: w1 is %dx%d; the small Kronecker factor is limited to 128x128", a, b);
  return LYC_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
void launch_skinny_nt(SkinnyArgs sa, hipStream_t st) {
  constexpr int BK = SkinnyCfg<T>::BK;
  const long mt = cdiv(sa.M, 64);
  long split = cdiv(256, mt);
  const long max_split = cdiv(sa.K, 2 * BK);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  sa.chunk = round_up(cdiv(sa.K, split), BK);
  split = cdiv(sa.K, sa.chunk);
  const int ni = sa.Nn <= 16 ? 1 : sa.Nn <= 32 ? 2 : sa.Nn <= 64 ? 4 : 8;
  dim3 grid((unsigned)mt, (unsigned)split, (unsigned)cdiv(sa.Nn, 16 * ni));
  switch (ni) {
    case 1: hipLaunchKernelGGL((skinny_nt_kernel<T, 1>), grid, dim3(NTHREADS), 0, st, sa); break;
    case 2: hipLaunchKernelGGL((skinny_nt_kernel<T, 2>), grid, dim3(NTHREADS), 0, st, sa); break;
    case 4: hipLaunchKernelGGL((skinny_nt_kernel<T, 4>), grid, dim3(NTHREADS), 0, st, sa); break;
    default: hipLaunchKernelGGL((skinny_nt_kernel<T, 8>), grid, dim3(NTHREADS), 0, st, sa); break;
  }
}

template <typename T>
void launch_expand_nt(const SkinnyArgs& sa, hipStream_t st) {
  dim3 grid((unsigned)cdiv(sa.M, 64), (unsigned)cdiv(sa.Nn, 128));
  hipLaunchKernelGGL((expand_nt_kernel<T>), grid, dim3(NTHREADS), 0, st, sa);
}

template <typename T>
void launch_skinny_tn(SkinnyArgs sa, hipStream_t st) {
  const long it = cdiv(sa.K, 64);  // output row tiles
  long split = cdiv(512, it);
  const long max_split = cdiv(sa.M, 64);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  sa.chunk = round_up(cdiv(sa.M, split), 32);
  split = cdiv(sa.M, sa.chunk);
  const int ni = sa.Nn <= 16 ? 1 : sa.Nn <= 32 ? 2 : sa.Nn <= 64 ? 4 : 8;
  dim3 grid((unsigned)it, (unsigned)split, (unsigned)cdiv(sa.Nn, 16 * ni));
  switch (ni) {
    case 1: hipLaunchKernelGGL((skinny_tn_kernel<T, 1>), grid, dim3(NTHREADS), 0, st, sa); break;
    case 2: hipLaunchKernelGGL((skinny_tn_kernel<T, 2>), grid, dim3(NTHREADS), 0, st, sa); break;
    case 4: hipLaunchKernelGGL((skinny_tn_kernel<T, 4>), grid, dim3(NTHREADS), 0, st, sa); break;
    default: hipLaunchKernelGGL((skinny_tn_kernel<T, 8>), grid, dim3(NTHREADS), 0, st, sa); break;
  }
}
// ------------------------------------------------------------------------------------------------
// rank-r (LoCon) fast path: lowrank.h
bool bneck_ok(const BneckArgs& b, int dtype) {
  const int dt = dtype & 0xff;
  if (dt != LYC_BF16 && dt != LYC_F16) return false;
  return b.R >= 1 && b.R <= 64 && (b.K1 % 8) == 0 && (b.lda % 8) == 0 && (reinterpret_cast<uintptr_t>(b.A) & 15u) == 0;
}

template <typename T, int NW, int MI, int RT>
void launch_bneck_v(BneckArgs b, bool vec, hipStream_t st) {
  // few row tiles (M = 1024 gives 64): also split the output columns over gridDim.y so that every CU gets a share of
  // the expand stage; each slice repeats the reduce stage (its operands come from L2).  Measured best on the SDXL shapes
  // (benchmarks/kt_lowrank.sh): as many slices as keep the grid within one round of 256 workgroups, at most 8.
  const long rows = cdiv(b.M, 16 * MI);
  long ns = b.out != nullptr ? 256 / rows : 1;
  if (ns > cdiv(b.N2, 16 * NW)) ns = cdiv(b.N2, 16 * NW);
  if (ns > 8) ns = 8;
  if (ns < 1) ns = 1;
  b.nsplit = (int)ns;
  const dim3 grid((unsigned)rows, (unsigned)ns);
  if constexpr (RT == 1) {
    if (b.gat.mode != 0) {  // implicit Conv2d forward (vector factor layouts only; checked by the caller)
      hipLaunchKernelGGL((bneck_kernel<T, NW, MI, RT, true, true, true>), grid, dim3(NW * 64), 0, st, b);
      return;
    }
  }
  if constexpr (NW == 4 && RT <= 2 && MI * RT <= 2) {  // coalesced operand loads through the per-wave LDS stage (<= 64 KiB)
    if (vec) {
      hipLaunchKernelGGL((bneck_kernel<T, NW, MI, RT, true, true, false, true>), grid, dim3(NW * 64), 0, st, b);
      return;
    }
  }
  if (vec)
    hipLaunchKernelGGL((bneck_kernel<T, NW, MI, RT, true, true>), grid, dim3(NW * 64), 0, st, b);
  else
    hipLaunchKernelGGL((bneck_kernel<T, NW, MI, RT, false, false>), grid, dim3(NW * 64), 0, st, b);
}

template <typename T, int RT>
void launch_bneck_rt(const BneckArgs& b, hipStream_t st) {
  const bool vec = b.f1k == 1 && (b.f1n % 4) == 0 && (reinterpret_cast<uintptr_t>(b.F1) & 15u) == 0 && b.f2k == 1 &&
                   (b.f2n % 4) == 0 && (b.R % 4) == 0 && (reinterpret_cast<uintptr_t>(b.F2) & 15u) == 0;
  int mi = b.M >= 8192 ? 2 : 1;
  int nw = (mi == 1 && b.K1 >= 8192) ? 8 : 4;  // 8 waves only pay when the reduce stage is very long
  if (mi == 2)
    launch_bneck_v<T, 4, 2, RT>(b, vec, st);
  else if (nw == 8)
    launch_bneck_v<T, 8, 1, RT>(b, vec, st);
  else
    launch_bneck_v<T, 4, 1, RT>(b, vec, st);
}

template <typename T>
void launch_bneck(const BneckArgs& b, hipStream_t st) {
  if (b.R <= 16)
    launch_bneck_rt<T, 1>(b, st);
  else if (b.R <= 32)
    launch_bneck_rt<T, 2>(b, st);
  else
    launch_bneck_rt<T, 4>(b, st);
}

// ---- sibling LoCon projections in one launch (bneck_group_kernel, round 5) ---------------------------------------------------------
// n <= BNECK_GROUP_MAX problems of equal shape and factor layout: the tile plan of launch_bneck_rt / _v with the column slices
// chosen for ALL problems together (rows * n workgroups per slice).  Returns false when the shape class is not instantiated for the
// group kernel (ranks above 32, the 8-wave reduce, gathers): the caller launches layer by layer.
template <typename T, int MI, int RT>
bool launch_bneck_group_v(BneckGroupArgs& ga, bool vec, hipStream_t st) {
  constexpr int NW = 4;
  const BneckArgs& b0 = ga.p[0];
  const long rows = cdiv(b0.M, 16 * MI);
  long ns = b0.out != nullptr ? 256 / (rows * ga.n) : 1;
  if (ns > cdiv(b0.N2, 16 * NW)) ns = cdiv(b0.N2, 16 * NW);
  if (ns > 8) ns = 8;
  if (ns < 1) ns = 1;
  for (int i = 0; i < ga.n; ++i) ga.p[i].nsplit = (int)ns;
  const dim3 grid((unsigned)rows, (unsigned)ns, (unsigned)ga.n);
  if constexpr (MI * RT <= 2) {
    if (vec) {
      hipLaunchKernelGGL((bneck_group_kernel<T, NW, MI, RT, true, true, true>), grid, dim3(NW * 64), 0, st, ga);
      return true;
    }
  }
  if (vec)
    hipLaunchKernelGGL((bneck_group_kernel<T, NW, MI, RT, true, true>), grid, dim3(NW * 64), 0, st, ga);
  else
    hipLaunchKernelGGL((bneck_group_kernel<T, NW, MI, RT, false, false>), grid, dim3(NW * 64), 0, st, ga);
  return true;
}
template <typename T>
bool launch_bneck_group(BneckGroupArgs& ga, hipStream_t st) {
  const BneckArgs& b = ga.p[0];
  if (b.R > 32 || b.gat.mode != 0) return false;
  const bool vec = b.f1k == 1 && (b.f1n % 4) == 0 && b.f2k == 1 && (b.f2n % 4) == 0 && (b.R % 4) == 0;
  for (int i = 0; i < ga.n; ++i) {
    const BneckArgs& q = ga.p[i];
    if (q.M != b.M || q.K1 != b.K1 || q.R != b.R || q.N2 != b.N2 || q.f1n != b.f1n || q.f1k != b.f1k || q.f2n != b.f2n || q.f2k != b.f2k) return false;
    if (vec && ((reinterpret_cast<uintptr_t>(q.F1) | reinterpret_cast<uintptr_t>(q.F2)) & 15u)) return false;
  }
  const int mi = b.M >= 8192 ? 2 : 1;
  if (mi == 1 && b.K1 >= 8192) return false;  // (the 8-wave reduce of very long K: per layer)
  if (b.R <= 16) return mi == 2 ? launch_bneck_group_v<T, 2, 1>(ga, vec, st) : launch_bneck_group_v<T, 1, 1>(ga, vec, st);
  return mi == 2 ? launch_bneck_group_v<T, 2, 2>(ga, vec, st) : launch_bneck_group_v<T, 1, 2>(ga, vec, st);
}

// ---- round 6: the LDS-DMA form of the launch (lowrank4.h) where its conditions hold ------------------------------------------------
// BneckArgs -> Bneck4Args; false when the problem is outside bneck4_kernel (rank above 16, fp32 output rows, gathers, strided or
// unaligned factors, a tensor of 2 GiB or more): the caller keeps bneck_kernel.
bool bneck4_args(const BneckArgs& b, Bneck4Args& a, bool& ft) {
  if (b.gat.mode != 0 || b.out_f32 || b.R < 4 || b.R > 16 || (b.R % 4) != 0 || (b.K1 % 8) != 0 || (b.N2 % 8) != 0 || b.M < 1) return false;
  if (b.f1k == 1 && b.f1n == b.K1 && b.f2k == 1 && b.f2n == b.R) ft = false;         // forward: F1 = down [R, K1], F2 = up [N2, R]
  else if (b.f1n == 1 && b.f1k == b.R && b.f2n == 1 && b.f2k == b.N2) ft = true;     // backward: F1 = up [K1, R], F2 = down [R, N2]
  else return false;
  if ((b.lda % 8) != 0 || (b.out != nullptr && (b.ldo % 8) != 0)) return false;
  if (((reinterpret_cast<uintptr_t>(b.A) | reinterpret_cast<uintptr_t>(b.out) | reinterpret_cast<uintptr_t>(b.F1) | reinterpret_cast<uintptr_t>(b.F2)) & 15u) != 0) return false;
  const long lim = (1L << 31) - 1;
  if (b.M * b.lda * 2 > lim || (b.out != nullptr && b.M * b.ldo * 2 > lim) || (long)b.R * b.K1 * 4 > lim || (long)b.R * b.N2 * 4 > lim) return false;
  a.A = b.A; a.F1 = b.F1; a.F2 = b.F2; a.mid = b.mid; a.out = b.out;
  a.a_bytes = (unsigned)((b.M - 1) * b.lda * 2 + (long)b.K1 * 2); a.out_bytes = b.out ? (unsigned)((b.M - 1) * b.ldo * 2 + (long)b.N2 * 2) : 0u;
  a.f1_bytes = (unsigned)((long)b.R * b.K1 * 4); a.f2_bytes = (unsigned)((long)b.R * b.N2 * 4);
  a.lda = (int)b.lda; a.ldo = (int)b.ldo; a.M = (int)b.M; a.K1 = b.K1; a.KS = (b.K1 + 31) / 32; a.R = b.R; a.N2 = b.N2;
  a.alpha1 = b.alpha1; a.alpha2 = b.alpha2;
  return true;
}
template <typename T, int NW, int MI, bool FT>
void launch_bneck4_inst(const Bneck4Args& a, dim3 grid, int lds, hipStream_t st) {
  static bool attr_set = false;  // dynamic LDS above 64 KiB needs the opt-in (per instantiation)
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck4_kernel<T, NW, MI, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((bneck4_kernel<T, NW, MI, FT>), grid, dim3(NW * 64), lds, st, a);
}
template <typename T, int NW, int MI, bool FT>
void launch_bneck4_group_inst(const Bneck4GroupArgs& ga, dim3 grid, int lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck4_group_kernel<T, NW, MI, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((bneck4_group_kernel<T, NW, MI, FT>), grid, dim3(NW * 64), lds, st, ga);
}
template <typename T>
bool launch_bneck4(const BneckArgs& b, hipStream_t st) {
  Bneck4Args a{};
  bool ft = false;
  Bneck4Plan p{};
  if (!bneck4_args(b, a, ft) || !bneck4_make_plan(b.M, b.K1, b.N2, b.out != nullptr, 1, p)) return false;
  a.D = p.D; a.D2 = p.D2;
  const dim3 grid((unsigned)cdiv(b.M, 16 * p.mi), (unsigned)p.ns);
  if (p.nw == 8) ft ? launch_bneck4_inst<T, 8, 1, true>(a, grid, p.lds, st) : launch_bneck4_inst<T, 8, 1, false>(a, grid, p.lds, st);
  else ft ? launch_bneck4_inst<T, 4, 2, true>(a, grid, p.lds, st) : launch_bneck4_inst<T, 4, 2, false>(a, grid, p.lds, st);
  return true;
}
template <typename T>
bool launch_bneck4_group(const BneckGroupArgs& ga, hipStream_t st) {
  Bneck4GroupArgs g4{};
  if (ga.n > BNECK4_GROUP_MAX) return false;
  g4.n = ga.n;
  bool ft0 = false;
  for (int i = 0; i < ga.n; ++i) {
    bool ft = false;
    if (!bneck4_args(ga.p[i], g4.p[i], ft)) return false;
    if (i == 0) ft0 = ft;
    else if (ft != ft0) return false;
    const BneckArgs &q = ga.p[i], &b0 = ga.p[0];
    if (q.M != b0.M || q.K1 != b0.K1 || q.R != b0.R || q.N2 != b0.N2 || (q.out == nullptr) != (b0.out == nullptr)) return false;
  }
  const BneckArgs& b = ga.p[0];
  Bneck4Plan p{};
  if (!bneck4_make_plan(b.M, b.K1, b.N2, b.out != nullptr, ga.n, p)) return false;
  for (int i = 0; i < ga.n; ++i) { g4.p[i].D = p.D; g4.p[i].D2 = p.D2; }
  const dim3 grid((unsigned)cdiv(b.M, 16 * p.mi), (unsigned)p.ns, (unsigned)ga.n);
  if (p.nw == 8) ft0 ? launch_bneck4_group_inst<T, 8, 1, true>(g4, grid, p.lds, st) : launch_bneck4_group_inst<T, 8, 1, false>(g4, grid, p.lds, st);
  else ft0 ? launch_bneck4_group_inst<T, 4, 2, true>(g4, grid, p.lds, st) : launch_bneck4_group_inst<T, 4, 2, false>(g4, grid, p.lds, st);
  return true;
}

template <typename T, int NW, int MI, bool FT>
void launch_bneck4_sum_inst(const Bneck4GroupArgs& ga, void* out_sum, dim3 grid, int lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck4_sum_kernel<T, NW, MI, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((bneck4_sum_kernel<T, NW, MI, FT>), grid, dim3(NW * 64), lds, st, ga, out_sum);
}
// the n problems of `ga` (equal shapes, `out` unused) in one workgroup per tile, stage-2 results summed into out_sum [M, N2] (row pitch N2)
template <typename T>
bool launch_bneck4_sum(BneckGroupArgs& ga, void* out_sum, hipStream_t st) {
  Bneck4GroupArgs g4{};
  if (ga.n < 2 || ga.n > BNECK4_GROUP_MAX || (reinterpret_cast<uintptr_t>(out_sum) & 15u)) return false;
  g4.n = ga.n;
  bool ft0 = false;
  for (int i = 0; i < ga.n; ++i) {
    bool ft = false;
    ga.p[i].out = out_sum;  // (alignment / extent checks of bneck4_args)
    if (!bneck4_args(ga.p[i], g4.p[i], ft)) return false;
    if (i == 0) ft0 = ft;
    else if (ft != ft0) return false;
    const BneckArgs &q = ga.p[i], &b0 = ga.p[0];
    if (q.M != b0.M || q.K1 != b0.K1 || q.R != b0.R || q.N2 != b0.N2) return false;
  }
  const BneckArgs& b = ga.p[0];
  Bneck4Plan p{};
  if (!bneck4_make_plan(b.M, b.K1, b.N2, true, 1, p, ga.n)) return false;
  for (int i = 0; i < ga.n; ++i) { g4.p[i].D = p.D; g4.p[i].D2 = p.D2; }
  const dim3 grid((unsigned)cdiv(b.M, 16 * p.mi), (unsigned)p.ns);
  if (p.nw == 8) ft0 ? launch_bneck4_sum_inst<T, 8, 1, true>(g4, out_sum, grid, p.lds, st) : launch_bneck4_sum_inst<T, 8, 1, false>(g4, out_sum, grid, p.lds, st);
  else ft0 ? launch_bneck4_sum_inst<T, 4, 2, true>(g4, out_sum, grid, p.lds, st) : launch_bneck4_sum_inst<T, 4, 2, false>(g4, out_sum, grid, p.lds, st);
  return true;
}

void launch_bneck_dt(const BneckArgs& b, int dtype, hipStream_t st) {
  if (!(dtype & LYC_BNECK_REG)) {
    if ((dtype & 0xff) == LYC_BF16 ? launch_bneck4<__bf16>(b, st) : launch_bneck4<_Float16>(b, st)) return;
  }
  if ((dtype & 0xff) == LYC_BF16)
    launch_bneck<__bf16>(b, st);
  else
    launch_bneck<_Float16>(b, st);
}

// largest CV in {8, 4, 2} usable for problem p (0 = none)
int tn_cv_ok(const LowrankTnProb& p, int cv, int ct = 0) {
  if (ct > 0 && (ct % (16 * cv)) != 0) return 0;  // gathered columns: a 16 cv tile must lie inside one tap
  return (p.C % cv) == 0 && (p.ld % cv) == 0 && (reinterpret_cast<uintptr_t>(p.act) % (2 * cv)) == 0;
}

template <typename T, int RT>
void launch_tn_rt(LowrankTnArgs& a, int cv, hipStream_t st) {
  for (int i = 0; i < 2; ++i) a.p[i].tiles = a.p[i].act ? (int)cdiv(a.p[i].C, 16 * cv) : 0;
  const long waves = (long)(a.p[0].tiles + a.p[1].tiles) * a.nsplit;
  const dim3 grid((unsigned)cdiv(waves, NWAVES));
  if constexpr (RT == 1) {
    if (a.gat.mode != 0) {
      switch (cv) {
        case 8: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 8, true>), grid, dim3(NTHREADS), 0, st, a); break;
        case 4: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 4, true>), grid, dim3(NTHREADS), 0, st, a); break;
        case 2: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 2, true>), grid, dim3(NTHREADS), 0, st, a); break;
        default: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 1, true>), grid, dim3(NTHREADS), 0, st, a); break;
      }
      return;
    }
  }
  switch (cv) {
    case 8: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 8>), grid, dim3(NTHREADS), 0, st, a); break;
    case 4: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 4>), grid, dim3(NTHREADS), 0, st, a); break;
    case 1: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 1>), grid, dim3(NTHREADS), 0, st, a); break;
    default: hipLaunchKernelGGL((lowrank_tn_kernel<T, RT, 2>), grid, dim3(NTHREADS), 0, st, a); break;
  }
}

// Launch parameters of lowrank_tn_kernel for the (up to two) problems of `a`: fills rows_per_slab / nsplit / tiles and
// returns the columns per lane (0 = shapes the fast kernel does not take).  `grouped`: the layer is one of many in a launch
// (lowrank_tn_group_kernel): longer slabs (fewer atomics); measured on the SDXL mix (benchmarks/wgbench.cpp,
// profiles/r02_wgbench_sweep3_locon.log): 512-row slabs and 2 columns per lane (2.06 ms for 722 layers; 8 columns: 2.88 ms --
// more, narrower waves keep 4 waves per SIMD in flight; 256 / 1024-row slabs: 3.5 / 3.4 ms).
#ifndef LYC_TNG_ROWS
#define LYC_TNG_ROWS 512
#endif
#ifndef LYC_TNG_CVMAX
#define LYC_TNG_CVMAX 2
#endif
int plan_lowrank_tn(LowrankTnArgs& a, int dtype, bool grouped = false) {
  // measured (profiles/r01_ktrace_lowrank.log); the implicit-Conv2d form (gathered rows: one load per tap and row) has its own target
#ifndef LYC_TN_WAVE_TARGET_GAT
#define LYC_TN_WAVE_TARGET_GAT 1400
#endif
#ifndef LYC_TN_ATOMIC_BUDGET_GAT
#define LYC_TN_ATOMIC_BUDGET_GAT 800000
#endif
  const int atomic_budget = a.gat.mode ? LYC_TN_ATOMIC_BUDGET_GAT : 800000, wave_target = a.gat.mode ? LYC_TN_WAVE_TARGET_GAT : 1400;
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || a.R > 64) return 0;
  long csum = 0;
  for (int i = 0; i < 2; ++i)
    if (a.p[i].act) {
      if (!tn_cv_ok(a.p[i], 1, (i == 1 && a.gat.mode) ? a.Ct : 0)) return 0;
      csum += a.p[i].C;
    }
  if (csum == 0) return 1;
  // Row slabs: ~160-256 rows per wave measured best (benchmarks/kt_lowrank.sh) -- shorter slabs multiply the atomics,
  // longer ones leave the wave a long serial chain; bounded by the atomic budget.
  long split = a.M / (grouped ? LYC_TNG_ROWS : 160);
  const long cap = atomic_budget / ((long)a.R * csum);
  if (split > cap) split = cap;
  if (!grouped && split < 2 && a.M >= 64) split = 2;
  if (split < 1) split = 1;
  a.rows_per_slab = round_up(cdiv(a.M, split), 4);
  a.nsplit = (int)cdiv(a.M, a.rows_per_slab);
  // columns per lane: 1 (most waves) unless the layer is so wide that 2 still gives thousands of waves
  int cv = 1;
  for (int c = grouped ? LYC_TNG_CVMAX : 8; c >= 2; c >>= 1) {
    bool ok = true;
    long tiles = 0;
    for (int i = 0; i < 2; ++i)
      if (a.p[i].act) {
        ok = ok && tn_cv_ok(a.p[i], c, (i == 1 && a.gat.mode) ? a.Ct : 0);
        tiles += cdiv(a.p[i].C, 16 * c);
      }
    if (ok && (grouped || tiles * a.nsplit >= wave_target)) {
      cv = c;
      break;
    }
  }
  for (int i = 0; i < 2; ++i) a.p[i].tiles = a.p[i].act ? (int)cdiv(a.p[i].C, 16 * cv) : 0;
  return cv;
}

// both factor gradients of a rank-r layer in one launch; false = shapes the fast kernel does not take
bool launch_lowrank_tn(LowrankTnArgs a, int dtype, hipStream_t st) {
  const int cv = plan_lowrank_tn(a, dtype);
  if (cv == 0) return false;
  if (!a.p[0].act && !a.p[1].act) return true;
  const int rt = a.R <= 16 ? 1 : a.R <= 32 ? 2 : 4;
  if ((dtype & 0xff) == LYC_BF16) {
    if (rt == 1) launch_tn_rt<__bf16, 1>(a, cv, st);
    else if (rt == 2) launch_tn_rt<__bf16, 2>(a, cv, st);
    else launch_tn_rt<__bf16, 4>(a, cv, st);
  } else {
    if (rt == 1) launch_tn_rt<_Float16, 1>(a, cv, st);
    else if (rt == 2) launch_tn_rt<_Float16, 2>(a, cv, st);
    else launch_tn_rt<_Float16, 4>(a, cv, st);
  }
  return true;
}

template <typename T, int RT>
void launch_tn_group_rt(const LowrankTnGroupArgs& ga, int cv, hipStream_t st) {
  const dim3 grid((unsigned)ga.wg_end[ga.n - 1]);
  switch (cv) {
    case 8: hipLaunchKernelGGL((lowrank_tn_group_kernel<T, RT, 8>), grid, dim3(NTHREADS), 0, st, ga); break;
    case 4: hipLaunchKernelGGL((lowrank_tn_group_kernel<T, RT, 4>), grid, dim3(NTHREADS), 0, st, ga); break;
    case 2: hipLaunchKernelGGL((lowrank_tn_group_kernel<T, RT, 2>), grid, dim3(NTHREADS), 0, st, ga); break;
    default: hipLaunchKernelGGL((lowrank_tn_group_kernel<T, RT, 1>), grid, dim3(NTHREADS), 0, st, ga); break;
  }
}

// the two factor-gradient problems of one rank-r Linear layer (what lyc_locon_linear_bwd launches per layer)
void locon_tn_problems(LowrankTnArgs& ta, const void* g, const void* x, const float* t, const float* dt, float* d_down,
                       float* d_up, int64_t M, int I, int O, int r, float alpha) {
  ta.M = M; ta.R = r;
  int np = 0;
  if (d_up) {
    LowrankTnProb& p = ta.p[np++];
    p.act = g; p.ld = O; p.C = O; p.mid = t; p.out = d_up; p.os = r; p.oj = 1; p.alpha = alpha;
  }
  if (d_down) {
    LowrankTnProb& p = ta.p[np++];
    p.act = x; p.ld = I; p.C = I; p.mid = dt; p.out = d_down; p.os = 1; p.oj = I; p.alpha = 1.0f; p.swap = 1;
  }
}

}  // namespace

extern "C" {

int lyc_abi_version(void) { return LYC_ABI_VERSION; }
const char* lyc_last_error(void) { return g_err; }

extern "C++" {
namespace {
int lokr_linear_fwd_impl(const void* x, const float* w1, const float* w2, const void* planes, const void* base, void* y, int64_t M,
                         int a, int b, int c, int d, float alpha, int dtype, void* stream);
int lokr_linear_bwd_impl(const void* g, const void* x, const float* w1, const float* w2, const void* planes, void* dx, float* dw1,
                         float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype, void* stream);
// the planes serve the 16-bit fast path (kron3, plain rows) only
bool lokr_planes_usable(const void* act, int64_t M, int a, int b, int c, int d, int dtype) {
  const int dt = dtype & 0xff;
  return (dt == LYC_BF16 || dt == LYC_F16) && a == b && a >= 1 && (16 % a) == 0 && (c % 8) == 0 && (d % 8) == 0 && M > 0 &&
         (reinterpret_cast<uintptr_t>(act) & 15u) == 0;
}
}  // namespace
}  // extern "C++"

int lyc_lokr_linear_planes_ok(int64_t M, int a, int b, int c, int d, int dtype) {
  static const int dummy = 0;
  (void)dummy;
  return lokr_planes_usable(nullptr, M, a, b, c, d, dtype) ? 1 : 0;
}

int lyc_lokr_linear_fwd(const void* x, const float* w1, const float* w2, const void* base, void* y, int64_t M, int a, int b,
                        int c, int d, float alpha, int dtype, void* stream) {
  return lokr_linear_fwd_impl(x, w1, w2, nullptr, base, y, M, a, b, c, d, alpha, dtype, stream);
}

int lyc_lokr_linear_fwd_planes(const void* x, const float* w1, const void* planes_fwd, const void* base, void* y, int64_t M, int a,
                               int b, int c, int d, float alpha, int dtype, void* stream) {
  if (!planes_fwd) return fail(LYC_ERR_ARG, "lokr_linear_fwd_planes: null planes");
  if (!lokr_planes_usable(x, M, a, b, c, d, dtype))
    return fail(LYC_ERR_UNSUPPORTED, "lokr_linear_fwd_planes: the planes serve the 16-bit fast path only (lyc_lokr_linear_planes_ok)");
  return lokr_linear_fwd_impl(x, w1, nullptr, planes_fwd, base, y, M, a, b, c, d, alpha, dtype, stream);
}

int lyc_lokr_linear_bwd_planes(const void* g, const void* x, const float* w1, const void* planes_bwd, void* dx, float* dw1,
                               float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype, void* stream) {
  if (!planes_bwd) return fail(LYC_ERR_ARG, "lokr_linear_bwd_planes: null planes");
  if (!lokr_planes_usable(g, M, a, b, c, d, dtype))
    return fail(LYC_ERR_UNSUPPORTED, "lokr_linear_bwd_planes: the planes serve the 16-bit fast path only (lyc_lokr_linear_planes_ok)");
  return lokr_linear_bwd_impl(g, x, w1, nullptr, planes_bwd, dx, dw1, dw2, ws, M, a, b, c, d, alpha, dtype, stream);
}

// ---- sibling projections in one launch (round 4) -----------------------------------------------------------------------
extern "C++" {
namespace {
int lokr_linear_group(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, int dtype, void* stream, bool backward,
                      void* dx_sum = nullptr) {
  const char* who = dx_sum ? "lokr_linear_bwd_group_sum" : (backward ? "lokr_linear_bwd_group" : "lokr_linear_fwd_group");
  if (dx_sum && (n < 1 || n > K4_GROUP_MAX)) return fail(LYC_ERR_UNSUPPORTED, "%s: 1 .. %d problems", who, K4_GROUP_MAX);
  if (dx_sum && (reinterpret_cast<uintptr_t>(dx_sum) & 15u)) return fail(LYC_ERR_ARG, "%s: dx_sum must be 16-byte aligned", who);
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "%s: bad item list", who);
  if (n == 0) return LYC_OK;
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || (dtype & ~0xff)) return fail(LYC_ERR_UNSUPPORTED, "%s: 16-bit activations, no dtype flags", who);
  std::vector<KronArgs> kas((size_t)n);
  for (int k = 0; k < n; ++k) {
    const LycLokrLinearGroupItem& it = items[k];
    if (!it.in || !it.w1 || !it.planes || (!it.out && !dx_sum)) return fail(LYC_ERR_ARG, "%s: item %d: null pointer", who, k);
    if (dx_sum && (it.M != items[0].M || !it.aux || !it.ws))
      return fail(LYC_ERR_ARG, "%s: item %d: the problems of a sum share M and all carry x + their scratch (the w1 partials)", who, k);
    if (int rc = check_kron_dims(it.M, a, b, c, d)) return rc;
    if (it.M < 1 || !lokr_planes_usable(it.in, it.M, a, b, c, d, dtype))
      return fail(LYC_ERR_UNSUPPORTED, "%s: item %d is not on the packed-plane fast path (lyc_lokr_linear_planes_ok)", who, k);
    if ((it.aux != nullptr) != (items[0].aux != nullptr) || (backward && (it.ws != nullptr) != (it.aux != nullptr)))
      return fail(LYC_ERR_ARG, "%s: item %d: every item of a group takes the same operands (base / x + ws: all or none)", who, k);
    KronArgs& ka = kas[(size_t)k];
    ka = KronArgs{};
    ka.x = it.in; ka.y = dx_sum ? dx_sum : it.out; ka.w1 = it.w1; ka.w2p = it.planes; ka.alpha = it.alpha; ka.M = it.M;
    if (!backward) {
      ka.base = it.aux;
      ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c; ka.s1o = b; ka.s1i = 1; ka.s2n = d; ka.s2k = 1;
    } else {
      ka.xref = it.aux; ka.dw1 = it.aux ? reinterpret_cast<float*>(it.ws) : nullptr;  // only a "w1 gradient wanted" flag: the partials go to ws
      ka.dw1_ws = static_cast<float*>(it.ws);
      ka.Gin = a; ka.K = c; ka.Gout = b; ka.N = d; ka.s1o = 1; ka.s1i = b; ka.s2n = 1; ka.s2k = d;
    }
    const bool ok = dt == LYC_BF16 ? kron4_ok<__bf16>(ka) : kron4_ok<_Float16>(ka);
    if (!ok) return fail(LYC_ERR_UNSUPPORTED, "%s: item %d: 16-byte aligned operands and sizes below 2 GiB are required", who, k);
  }
  if (dx_sum) {
    if (dt == LYC_BF16) launch_kron4_sum<__bf16>(kas.data(), n, dx_sum, (hipStream_t)stream);
    else launch_kron4_sum<_Float16>(kas.data(), n, dx_sum, (hipStream_t)stream);
    return check_launch(who);
  }
  for (int lo = 0; lo < n; lo += K4_GROUP_MAX) {
    const int cnt = std::min(K4_GROUP_MAX, n - lo);
    if (dt == LYC_BF16) launch_kron4_group<__bf16>(kas.data() + lo, cnt, (hipStream_t)stream);
    else launch_kron4_group<_Float16>(kas.data() + lo, cnt, (hipStream_t)stream);
    if (int rc = check_launch(who)) return rc;
  }
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

int lyc_lokr_linear_fwd_group(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, int dtype, void* stream) {
  return lokr_linear_group(items, n, a, b, c, d, dtype, stream, false);
}
int lyc_lokr_linear_bwd_group(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, int dtype, void* stream) {
  return lokr_linear_group(items, n, a, b, c, d, dtype, stream, true);
}
int lyc_lokr_linear_bwd_group_sum(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, void* dx_sum, int dtype,
                                  void* stream) {
  if (!dx_sum) return fail(LYC_ERR_ARG, "lokr_linear_bwd_group_sum: null dx_sum");
  return lokr_linear_group(items, n, a, b, c, d, dtype, stream, true, dx_sum);
}

int lyc_sum_rows(const void* const* src, int n, void* dst, int64_t numel, int dtype, void* stream) {
  if (n < 1 || n > 4 || !src || !dst || numel < 0) return fail(LYC_ERR_ARG, "sum_rows: 1 <= n <= 4 sources");
  const int dt = dtype & 0xff;
  if (dt != LYC_BF16 && dt != LYC_F16) return fail(LYC_ERR_UNSUPPORTED, "sum_rows: 16-bit tensors");
  if (numel == 0) return LYC_OK;
  SumArgs sa{};
  for (int i = 0; i < n; ++i) {
    if (!src[i] || (reinterpret_cast<uintptr_t>(src[i]) & 15u)) return fail(LYC_ERR_ARG, "sum_rows: source %d is null or not 16-byte aligned", i);
    sa.src[i] = src[i];
  }
  if ((numel % 8) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15u)) return fail(LYC_ERR_ARG, "sum_rows: numel %% 8 == 0 and an aligned destination");
  sa.dst = dst; sa.total = numel; sa.n = n;
  long blocks = cdiv(numel / 8, (long)NTHREADS);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (dt == LYC_BF16) hipLaunchKernelGGL((sum_rows_kernel<__bf16>), dim3((unsigned)blocks), dim3(NTHREADS), 0, (hipStream_t)stream, sa);
  else hipLaunchKernelGGL((sum_rows_kernel<_Float16>), dim3((unsigned)blocks), dim3(NTHREADS), 0, (hipStream_t)stream, sa);
  return check_launch("sum_rows");
}

// one pack launch for MANY layers (Linear or Conv2d factors given as full matrices): the once-per-optimizer-step refresh
int lyc_lokr_pack_group(const LycLokrPackItem* items, int n, int dtype, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "lokr_pack_group: bad item list");
  const int dt = dtype & 0xff;
  if (n > 0 && dt != LYC_BF16 && dt != LYC_F16) return fail(LYC_ERR_UNSUPPORTED, "lokr_pack_group: 16-bit planes only");
  KronPackGroupArgs ga{};
  auto flush = [&]() -> int {
    if (ga.n == 0) return LYC_OK;
    const dim3 grid((unsigned)cdiv(ga.unit_end[ga.n - 1], NWAVES));
    if (dt == LYC_BF16) hipLaunchKernelGGL((kron_pack_group_kernel<__bf16>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, ga);
    else hipLaunchKernelGGL((kron_pack_group_kernel<_Float16>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, ga);
    ga = KronPackGroupArgs{};
    return check_launch("lokr_pack_group");
  };
  for (int k = 0; k < n; ++k) {
    const LycLokrPackItem& it = items[k];
    if ((!it.w2 && !(it.w2a && it.w2b && it.rank >= 1)) || it.c < 1 || it.d < 1 || it.taps < 1 || (it.c % 8) != 0 || (it.d % 8) != 0)
      return fail(LYC_ERR_ARG, "lokr_pack_group: item %d: bad factor (c, d must be positive multiples of 8)", k);
    if (!it.planes_fwd && !it.planes_bwd) continue;
    if (ga.n == KPG_MAX)
      if (int rc = flush()) return rc;
    KronPackArgs& pa = ga.p[ga.n];
    pa = KronPackArgs{};
    pa.w2 = it.w2; pa.sq = it.sq; pa.sv = it.sv; pa.st = it.st; pa.c = it.c; pa.d = it.d; pa.taps = it.taps;
    pa.fwd = it.planes_fwd; pa.bwd = it.planes_bwd;
    if (!it.w2) {  // low rank: contiguous w2a [c, r], w2b [r, d * taps] (element (r, v, tap) at r * d * taps + v * taps + tap)
      pa.w2a = it.w2a; pa.w2b = it.w2b; pa.rank = it.rank; pa.a_sq = it.rank; pa.a_sr = 1;
      pa.b_sr = (long)it.d * it.taps; pa.b_sv = it.taps; pa.b_st = 1;
    }
    pa.units_fwd = kron_plane_bytes(it.c, it.taps, it.d) / 2048;
    const long units = pa.units_fwd + kron_plane_bytes(it.d, it.taps, it.c) / 2048;
    ga.unit_end[ga.n] = (ga.n ? ga.unit_end[ga.n - 1] : 0) + round_up(units, NWAVES);
    ++ga.n;
  }
  return flush();
}

// ---- every layer in ONE launch through a device table (kron_conv.h: kron_pack_table_kernel) ----------------------------------------
extern "C++" {
namespace {
// KronPackArgs of item `it` and its unit count (rounded to whole workgroups); false: nothing to pack
bool pack_args_of(const LycLokrPackItem& it, KronPackArgs& pa, long& units) {
  if (!it.planes_fwd && !it.planes_bwd) return false;
  pa = KronPackArgs{};
  pa.w2 = it.w2; pa.sq = it.sq; pa.sv = it.sv; pa.st = it.st; pa.c = it.c; pa.d = it.d; pa.taps = it.taps;
  pa.fwd = it.planes_fwd; pa.bwd = it.planes_bwd;
  if (!it.w2) {
    pa.w2a = it.w2a; pa.w2b = it.w2b; pa.rank = it.rank; pa.a_sq = it.rank; pa.a_sr = 1;
    pa.b_sr = (long)it.d * it.taps; pa.b_sv = it.taps; pa.b_st = 1;
  }
  pa.units_fwd = kron_plane_bytes(it.c, it.taps, it.d) / 2048;
  units = round_up(pa.units_fwd + kron_plane_bytes(it.d, it.taps, it.c) / 2048, NWAVES);
  return true;
}
int pack_item_check(const LycLokrPackItem& it, int k) {
  if ((!it.w2 && !(it.w2a && it.w2b && it.rank >= 1)) || it.c < 1 || it.d < 1 || it.taps < 1 || (it.c % 8) != 0 || (it.d % 8) != 0)
    return fail(LYC_ERR_ARG, "lokr_pack_group: item %d: bad factor (c, d must be positive multiples of 8)", k);
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

int64_t lyc_lokr_pack_table_bytes(const LycLokrPackItem* items, int n) {
  if (n < 0 || (n > 0 && !items)) return 0;
  long wgs = 0;
  int m = 0;
  for (int k = 0; k < n; ++k) {
    KronPackArgs pa;
    long units = 0;
    if (items[k].c < 1 || items[k].d < 1 || items[k].taps < 1) continue;
    if (!pack_args_of(items[k], pa, units)) continue;
    wgs += units / NWAVES;
    ++m;
  }
  return kron_pack_table_bytes(m, wgs);
}

int lyc_lokr_pack_group_ws(const LycLokrPackItem* items, int n, int dtype, void* table, int64_t table_bytes, int table_valid, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "lokr_pack_group_ws: bad item list");
  const int dt = dtype & 0xff;
  if (n > 0 && dt != LYC_BF16 && dt != LYC_F16) return fail(LYC_ERR_UNSUPPORTED, "lokr_pack_group_ws: 16-bit planes only");
  if (n == 0) return LYC_OK;
  if (!table || (reinterpret_cast<uintptr_t>(table) & 15u)) return fail(LYC_ERR_ARG, "lokr_pack_group_ws: the table must be 16-byte aligned device memory");
  int m = 0;
  long wgs = 0;
  for (int k = 0; k < n; ++k) {
    if (int rc = pack_item_check(items[k], k)) return rc;
    KronPackArgs pa;
    long units = 0;
    if (!pack_args_of(items[k], pa, units)) continue;
    wgs += units / NWAVES;
    ++m;
  }
  if (m == 0) return LYC_OK;
  if (wgs >= (1L << 31)) return fail(LYC_ERR_UNSUPPORTED, "lokr_pack_group_ws: too many units for one grid");
  if (table_bytes < kron_pack_table_bytes(m, wgs)) return fail(LYC_ERR_ARG, "lokr_pack_group_ws: table of %ld bytes, need %ld", (long)table_bytes, (long)kron_pack_table_bytes(m, wgs));
  hipStream_t st = (hipStream_t)stream;
  if (!table_valid) {  // descriptors: 28 per launch, from kernel arguments (capturable: no host memory is read at replay)
    KronPackGroupArgs ga{};
    int base = 0;
    long wg_base = 0;
    auto flush = [&]() -> int {
      if (ga.n == 0) return LYC_OK;
      hipLaunchKernelGGL(kron_pack_table_write_kernel, dim3((unsigned)ga.n), dim3(NTHREADS), 0, st, ga, static_cast<char*>(table), m, base, wg_base);
      base += ga.n;
      wg_base += ga.unit_end[ga.n - 1] / NWAVES;
      ga = KronPackGroupArgs{};
      return check_launch("lokr_pack_group_ws(table)");
    };
    for (int k = 0; k < n; ++k) {
      KronPackArgs pa;
      long units = 0;
      if (!pack_args_of(items[k], pa, units)) continue;
      if (ga.n == KPG_MAX)
        if (int rc = flush()) return rc;
      ga.p[ga.n] = pa;
      ga.unit_end[ga.n] = (ga.n ? ga.unit_end[ga.n - 1] : 0) + units;
      ++ga.n;
    }
    if (int rc = flush()) return rc;
  }
  if (dt == LYC_BF16) hipLaunchKernelGGL((kron_pack_table_kernel<__bf16>), dim3((unsigned)wgs), dim3(NTHREADS), 0, st, static_cast<const char*>(table), m);
  else hipLaunchKernelGGL((kron_pack_table_kernel<_Float16>), dim3((unsigned)wgs), dim3(NTHREADS), 0, st, static_cast<const char*>(table), m);
  return check_launch("lokr_pack_group_ws");
}

int lyc_lokr_lr_chain_group(const LycLokrLrChainItem* items, int n, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "lokr_lr_chain_group: bad item list");
  KronLrGroupArgs ga{};
  auto flush = [&]() -> int {
    if (ga.n == 0) return LYC_OK;
    hipLaunchKernelGGL(kron_lr_chain_kernel, dim3((unsigned)ga.wg_end[ga.n - 1]), dim3(NTHREADS), 0, (hipStream_t)stream, ga);
    ga = KronLrGroupArgs{};
    return check_launch("lokr_lr_chain_group");
  };
  for (int k = 0; k < n; ++k) {
    const LycLokrLrChainItem& it = items[k];
    if (!it.dw2 || !it.w2a || !it.w2b || (!it.d_w2a && !it.d_w2b) || it.c < 1 || it.d < 1 || it.r < 1 || it.taps < 0)
      return fail(LYC_ERR_ARG, "lokr_lr_chain_group: item %d: bad arguments", k);
    if (ga.n == KLR_MAX)
      if (int rc = flush()) return rc;
    KronLrItem& q = ga.p[ga.n];
    q.dw2 = it.dw2; q.w2a = it.w2a; q.w2b = it.w2b; q.d_w2a = it.d_w2a; q.d_w2b = it.d_w2b; q.c = it.c; q.d = it.d; q.r = it.r;
    q.taps = it.taps > 0 ? it.taps : 1;
    const long wgs = cdiv(kron_lr_waves(it.c, it.d, it.r, q.taps), NWAVES);
    ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
    ++ga.n;
  }
  return flush();
}

extern "C++" {
namespace {
int lokr_linear_fwd_impl(const void* x, const float* w1, const float* w2, const void* planes, const void* base, void* y, int64_t M,
                         int a, int b, int c, int d, float alpha, int dtype, void* stream) {
  if (int rc = check_kron_dims(M, a, b, c, d)) return rc;
  if (!x || !w1 || (!w2 && !planes) || !y) return fail(LYC_ERR_ARG, "lokr_linear_fwd: null pointer");
  if (M == 0) return LYC_OK;
  KronArgs ka{};
  ka.x = x; ka.y = y; ka.w1 = w1; ka.w2 = w2; ka.w2p = planes; ka.dw1 = nullptr; ka.xref = nullptr; ka.base = base;
  if (base) {  // fused `y = base + delta` lives in the epilogue of the 16-bit kron3 kernel only
    ka.M = M; ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c;
    const bool fast = ((dtype & 0xff) == LYC_BF16 && kron_fast_ok<__bf16>(ka)) || ((dtype & 0xff) == LYC_F16 && kron_fast_ok<_Float16>(ka));
    if (!fast)
      return fail(LYC_ERR_UNSUPPORTED, "lokr_linear_fwd: the fused base + delta epilogue needs the 16-bit fast path "
                                       "(a == b dividing 16, d %% 8 == 0, aligned x); add `base` on the caller's side");
  }
  ka.M = M; ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c;
  ka.s1o = b; ka.s1i = 1; ka.s2n = d; ka.s2k = 1; ka.alpha = alpha;
  DISPATCH_DTYPE(dtype, launch_kron<T>(ka, (hipStream_t)stream));
  return check_launch("lokr_linear_fwd");
}
}  // namespace
}  // extern "C++"

int64_t lyc_lokr_bwd_workspace_bytes(int64_t M, int a, int b, int c, int d, int dtype) {
  (void)c; (void)dtype;
  if (M <= 0 || a < 1 || b < 1 || d < 1 || a != b || a > 16) return 0;  // only the 16-bit fast path uses the scratch
  // one G x G fp32 partial per workgroup of the dx launch: kron3's narrowest tiling is 128 rows x 32 columns, kron4's 64 x 32
  const int64_t k3 = (int64_t)cdiv(M, K3_RT / a) * cdiv(d, 32);
  const int64_t k4 = (16 % a) == 0 ? (int64_t)cdiv(M * a, 64) * cdiv(d, 32) : 0;
  return (k3 > k4 ? k3 : k4) * a * b * (int64_t)sizeof(float);
}

int lyc_lokr_linear_bwd(const void* g, const void* x, const float* w1, const float* w2, void* dx, float* dw1,
                        float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype,
                        void* stream) {
  return lokr_linear_bwd_impl(g, x, w1, w2, nullptr, dx, dw1, dw2, ws, M, a, b, c, d, alpha, dtype, stream);
}

extern "C++" {
namespace {
int lokr_linear_bwd_impl(const void* g, const void* x, const float* w1, const float* w2, const void* planes, void* dx, float* dw1,
                         float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype, void* stream) {
  if (int rc = check_kron_dims(M, a, b, c, d)) return rc;
  if (!g || !x || !w1 || (!w2 && !planes)) return fail(LYC_ERR_ARG, "lokr_linear_bwd: null pointer");
  if (M == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool is16 = (dtype & 0xff) != LYC_F32;
  long dw1_partials = 0;  // > 0: the dx launch left that many [a*b] partials in ws
  const bool dw2s_ok = dw2 && is16 && a == b && (16 % a) == 0 && (c % 8) == 0 && (d % 8) == 0 &&
                       (reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  if (dx || dw1) {
    // dx[m, u*d+v] = alpha * sum_p w1[p,u] * sum_q w2[q,v] * g[m, p*c+q]: the same kernel on (w1^T, w2^T),
    // with the w1 gradient taken from its stage-1 result (GZ) against x.
    if (!dx) return fail(LYC_ERR_ARG, "lokr_linear_bwd: dw1 requires dx (they share one pass over g)");
    KronArgs ka{};
    ka.x = g; ka.y = dx; ka.w1 = w1; ka.w2 = w2; ka.w2p = planes; ka.dw1 = dw1; ka.xref = dw1 ? x : nullptr;
    ka.dw1_ws = (dw1 && ws) ? static_cast<float*>(ws) : nullptr;
    ka.M = M; ka.Gin = a; ka.K = c; ka.Gout = b; ka.N = d;
    ka.s1o = 1; ka.s1i = b; ka.s2n = 1; ka.s2k = d; ka.alpha = alpha; ka.out_f32 = (dtype & LYC_F32_ROWS) ? 1 : 0;
    long nblk = 0;
    DISPATCH_DTYPE(dtype, nblk = launch_kron<T>(ka, st));
    if (int rc = check_launch("lokr_linear_bwd(dx)")) return rc;
    if (ka.dw1_ws) dw1_partials = nblk;  // 0 when the generic kernel ran (it used atomics)
  }
  KronDw2sArgs ra{};  // w1-gradient reduction request
  if (dw1_partials > 0) {
    ra.dw1_ws = static_cast<const float*>(ws); ra.dw1 = dw1; ra.dw1_nblk = (int)dw1_partials; ra.dw1_n = a * b;
    ra.dw1_red = 1;
  }
  bool reduced = dw1_partials == 0;
  if (dw2) {
    bool done = false;
    if (dw2s_ok) {
      // rows of the tile run over q (exact operand g), the w1 mix is applied to x on the matrix cores; the output
      // tile is [q][v] with v contiguous in dw2
      KronDw2sArgs da = ra;
      da.Q = g; da.P = x; da.W = w1; da.out = dw2; da.M = M; da.G = a; da.I = c; da.J = d;
      da.ws = b; da.wt = 1; da.os = d; da.alpha = alpha;
      switch (dtype & 0xff) {
        case LYC_BF16: launch_dw2s<__bf16>(da, st); done = true; break;
        case LYC_F16: launch_dw2s<_Float16>(da, st); done = true; break;
        default: break;
      }
      if (done) {
        if (int rc = check_launch("lokr_linear_bwd(dw2)")) return rc;
        reduced = true;
      }
    }
    if (!done) {
      KronDw2Args da{};
      da.M = M; da.alpha = alpha; da.out = dw2;
      if (d <= c) {  // rows of the output tile run over the smaller side (v), mix applied to g
        da.Q = x; da.Gs = b; da.I = d; da.P = g; da.Gt = a; da.J = c;
        da.W = w1; da.ws = 1; da.wt = b;  // W[s=u, t=p] = w1[p, u]
        da.os = 1; da.oj = d;             // out(i=v, j=q) -> dw2[q*d + v]
      } else {
        da.Q = g; da.Gs = a; da.I = c; da.P = x; da.Gt = b; da.J = d;
        da.W = w1; da.ws = b; da.wt = 1;  // W[s=p, t=u] = w1[p, u]
        da.os = d; da.oj = 1;             // out(i=q, j=v) -> dw2[q*d + v]
      }
      DISPATCH_DTYPE(dtype, launch_kron_dw2<T>(da, st));
      if (int rc = check_launch("lokr_linear_bwd(dw2)")) return rc;
    }
  }
  if (!reduced && (dtype & LYC_DEFER_WGRAD)) {
    if (dw2) return fail(LYC_ERR_ARG, "lokr_linear_bwd: LYC_DEFER_WGRAD leaves dw2 to lyc_lokr_wgrad_group (pass dw2 = NULL)");
    return LYC_OK;  // the partials stay in ws; lyc_lokr_wgrad_group reduces them
  }
  if (!reduced) {  // partials were written but no dW2 launch carried the reduction
    long r = dw1_partials / 64;
    ra.dw1_red = (int)(r > 16 ? 16 : r < 1 ? 1 : r);
    hipLaunchKernelGGL(kron_dw1_reduce_kernel, dim3((unsigned)ra.dw1_red), dim3(NTHREADS), 0, st, ra);
    if (int rc = check_launch("lokr_linear_bwd(dw1 reduce)")) return rc;
  }
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

// ---- deferred, grouped weight gradients (kron_dw2s_group_kernel) ------------------------------------------------
extern "C++" {
namespace {
bool lokr_wgrad_fast(const void* g, const void* x, int64_t M, int a, int b, int c, int d, int dtype) {
  const int dt = dtype & 0xff;
  return M > 0 && (dt == LYC_BF16 || dt == LYC_F16) && a == b && a >= 1 && (16 % a) == 0 && (c % 8) == 0 && (d % 8) == 0 &&
         (reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
}
template <typename T, int MI, int NJ, int U>
void launch_dw2s_group(const KronDw2sGroupArgs& ga, hipStream_t st) {
  hipLaunchKernelGGL((kron_dw2s_group_kernel<T, MI, NJ, U>), dim3((unsigned)ga.wg_end[ga.n - 1]), dim3(NTHREADS), 0, st, ga);
}
}  // namespace
}  // extern "C++"

int lyc_lokr_wgrad_deferrable(const void* g, const void* x, int64_t M, int a, int b, int c, int d, int dtype) {
  return lokr_wgrad_fast(g, x, M, a, b, c, d, dtype) ? 1 : 0;
}

extern "C++" {
namespace {
// ---- full-width tiles (kron_dw2f.h) ---------------------------------------------------------------------------------------------
// tile classes by the factor's size: up to 80 (5 MFMA tiles) or up to 160 (10) outputs per side; every wave keeps 5 x 5 MFMA tiles
enum { DW2F_1010 = 0, DW2F_55, DW2F_105, DW2F_510, DW2F_NCFG };
struct Dw2fCfg { int TI, TJ, WK; };
constexpr Dw2fCfg DW2F_CFG[DW2F_NCFG] = {{10, 10, 1}, {5, 5, 4}, {10, 5, 2}, {5, 10, 2}};
inline bool dw2f_ok(const LycLokrWgradItem& it) {
  // small factors (SD1.5's 40 x 40) would leave most of a 80 x 80 tile empty: they stay on the narrow tiles of kron_dw2s.h
  return it.c >= 64 && it.d >= 64 && it.M * it.a * (long)it.c * 2 < (1L << 30) && it.M * it.a * (long)it.d * 2 < (1L << 30);
}
inline int dw2f_cfg_of(const LycLokrWgradItem& it) {
  const bool bi = it.c > 80, bj = it.d > 80;
  return bi ? (bj ? DW2F_1010 : DW2F_105) : (bj ? DW2F_510 : DW2F_55);
}
template <typename T, int TI, int TJ, int WR, int WC, int WK, int D>
void launch_dw2f_inst(const KronDw2fGroupArgs* ga, const KronDw2fItem* items, const int* wg_end, int n, unsigned grid, hipStream_t st) {
  constexpr int lds = kron_dw2f_lds_bytes(TI, TJ, WK, D);
  if (ga) {
    auto kern = kron_dw2f_group_kernel<T, TI, TJ, WR, WC, WK, D>;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, *ga);
  } else {
    auto kern = kron_dw2f_table_kernel<T, TI, TJ, WR, WC, WK, D>;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
#ifdef LYC_TUNE
    if (getenv("LYC_DW2F_OCC")) {
      int nb = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), NTHREADS, lds);
      hipFuncAttributes fa{};
      (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
      fprintf(stderr, "dw2f<%d,%d,%d,%d,%d,%d>: lds %d B, occupancy %d blocks/CU (%s), regs %d, static lds %zu, grid %u\n", TI, TJ, WR, WC, WK, D, lds, nb,
              hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes, grid);
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, items, wg_end, n);
  }
}
template <typename T>
void launch_dw2f(int cfg, const KronDw2fGroupArgs* ga, const KronDw2fItem* items, const int* wg_end, int n, unsigned grid, hipStream_t st) {
  switch (cfg) {
    case DW2F_1010: launch_dw2f_inst<T, 10, 10, 2, 2, 1, 3>(ga, items, wg_end, n, grid, st); break;
    case DW2F_55: launch_dw2f_inst<T, 5, 5, 1, 1, 4, 2>(ga, items, wg_end, n, grid, st); break;
    case DW2F_105: launch_dw2f_inst<T, 10, 5, 2, 1, 2, 3>(ga, items, wg_end, n, grid, st); break;
    default: launch_dw2f_inst<T, 5, 10, 1, 2, 2, 3>(ga, items, wg_end, n, grid, st); break;
  }
}
// Conv2d form: problems in the kernel arguments only (a UNet has a few dozen conv layers, each with thousands of workgroups)
template <typename T>
void launch_dw2f_conv(int cfg, const KronDw2fGroupArgs& ga, unsigned grid, hipStream_t st) {
  auto go = [&](auto kern, int lds) {
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, ga);
  };
  switch (cfg) {
    case DW2F_1010: go(kron_dw2f_group_kernel<T, 10, 10, 2, 2, 1, 3, true>, kron_dw2f_lds_bytes(10, 10, 1, 3)); break;
    case DW2F_55: go(kron_dw2f_group_kernel<T, 5, 5, 1, 1, 4, 2, true>, kron_dw2f_lds_bytes(5, 5, 4, 2)); break;
    case DW2F_105: go(kron_dw2f_group_kernel<T, 10, 5, 2, 1, 2, 3, true>, kron_dw2f_lds_bytes(10, 5, 2, 3)); break;
    default: go(kron_dw2f_group_kernel<T, 5, 10, 1, 2, 2, 3, true>, kron_dw2f_lds_bytes(5, 10, 2, 3)); break;
  }
}
inline long dw2f_item_wgs(const KronDw2fItem& q) {
  return round_up((long)q.tiles_i * q.tiles_j * q.nslab, 8) + round_up(q.dw1_ws ? q.dw1_red : 0, 8);
}
inline size_t dw2f_table_bytes(long n) { return (size_t)round_up(n * (long)sizeof(KronDw2fItem), 16) + (size_t)round_up(n * 4L, 16); }

// the items `take[k]` of the list on full-width tiles.  With a device table: one launch per tile class (+ the table writers);
// without: one launch per DW2F_MAX items.
int lokr_wgrad_group_f(const LycLokrWgradItem* items, const std::vector<char>& take, int n, int dt, hipStream_t st, void* table,
                       int64_t table_bytes) {
  long n_wide = 0;
  for (int k = 0; k < n; ++k) n_wide += take[k] ? 1 : 0;
  const bool use_table = table != nullptr && (reinterpret_cast<uintptr_t>(table) & 15u) == 0 && table_bytes >= (int64_t)dw2f_table_bytes(n_wide);
  KronDw2fItem* t_items = static_cast<KronDw2fItem*>(table);
  int* t_ends = use_table ? reinterpret_cast<int*>(static_cast<char*>(table) + round_up(n_wide * (long)sizeof(KronDw2fItem), 16)) : nullptr;
  long t_used = 0;
  for (int cfg = 0; cfg < DW2F_NCFG; ++cfg) {
    const Dw2fCfg cf = DW2F_CFG[cfg];
    std::vector<KronDw2fItem> qs;
    for (int k = 0; k < n; ++k) {
      if (!take[k] || dw2f_cfg_of(items[k]) != cfg) continue;
      const LycLokrWgradItem& it = items[k];
      KronDw2fItem q{};
      q.Q = it.g; q.P = it.x; q.W = it.w1; q.out = it.dw2; q.rows_total = (int)(it.M * it.a); q.I = it.c; q.J = it.d;
      q.lg = 31 - __builtin_clz((unsigned)it.a);
      q.ws = it.b; q.wt = 1; q.os = it.d; q.alpha = it.alpha;
      q.tiles_i = (int)cdiv(it.c, 16 * cf.TI); q.tiles_j = (int)cdiv(it.d, 16 * cf.TJ);
      // slabs: `slab_rows` rows each, fewer when the output is large (every slab adds the whole tile with fp32 atomics:
      // ~300-400 elements / ns chip-wide)
      const long unit = 32L * cf.WK;
      long slab_rows = 1024, cap = 2000000;
#ifdef LYC_TUNE  // development builds only (benchmarks/dw2_ab.py); the product library is built without it
      if (const char* e = getenv("LYC_DW2F_ROWS")) slab_rows = atol(e);
      if (const char* e = getenv("LYC_DW2F_CAP")) cap = atol(e);
#endif
      long ns = cdiv(q.rows_total, slab_rows);
      const long out_elems = (long)it.c * it.d;
      while (ns > 1 && ns * out_elems > cap) --ns;
      if (ns < 1) ns = 1;
      q.rows_per_slab = (int)(cdiv(cdiv(q.rows_total, ns), unit) * unit);
      q.nslab = (int)cdiv(q.rows_total, q.rows_per_slab);
      q.plain = q.nslab == 1 ? 1 : 0;
#ifdef LYC_TUNE
      if (getenv("LYC_DW2F_NOSTORE")) q.plain = 2;
#endif
      if (it.dw1) {
        q.dw1_ws = static_cast<const float*>(it.ws); q.dw1 = it.dw1; q.dw1_n = it.a * it.b;
        q.dw1_nblk = (int)lokr_dx_partial_blocks(it.M, it.a, it.c, it.d);
        long r = q.dw1_nblk / 256;
        q.dw1_red = (int)(r > 8 ? 8 : r < 1 ? 1 : r);
      }
      qs.push_back(q);
    }
    if (qs.empty()) continue;
    // a parameter that appears twice in one launch (shared module) must be added atomically
    auto unshare = [](KronDw2fItem* p, int cnt) {
      for (int i = 0; i < cnt; ++i)
        for (int j = 0; j < i; ++j)
          if (p[i].out == p[j].out || (p[i].dw1 && p[i].dw1 == p[j].dw1)) {
            if (p[i].plain == 1) p[i].plain = 0;
            if (p[j].plain == 1) p[j].plain = 0;
          }
    };
    if (use_table) {
      // longest workgroups first: the tail of the launch is then made of short ones
      std::stable_sort(qs.begin(), qs.end(), [](const KronDw2fItem& x, const KronDw2fItem& y) { return x.rows_per_slab > y.rows_per_slab; });
      {  // shared outputs: by pointer (the list can hold hundreds of layers)
        std::unordered_map<const void*, int> seen;
        for (const KronDw2fItem& q : qs) { ++seen[q.out]; if (q.dw1) ++seen[q.dw1]; }
        for (KronDw2fItem& q : qs)
          if (q.plain == 1 && (seen[q.out] > 1 || (q.dw1 && seen[q.dw1] > 1))) q.plain = 0;
      }
      long total = 0;
      for (const KronDw2fItem& q : qs) total += dw2f_item_wgs(q);
      if (total > (1L << 30)) return fail(LYC_ERR_ARG, "lokr_wgrad_group: too many workgroups in one launch");
      long wg_base = 0;
      for (size_t lo = 0; lo < qs.size(); lo += DW2F_MAX) {
        KronDw2fGroupArgs ga{};
        ga.n = (int)std::min<size_t>(DW2F_MAX, qs.size() - lo);
        long acc = 0;
        for (int i = 0; i < ga.n; ++i) {
          ga.p[i] = qs[lo + i];
          acc += dw2f_item_wgs(qs[lo + i]);
          ga.wg_end[i] = (int)acc;
        }
        hipLaunchKernelGGL(kron_dw2f_table_write_kernel, dim3(1), dim3(64), 0, st, ga, t_items + t_used, t_ends + t_used, (int)lo, (int)wg_base);
        wg_base += acc;
      }
      if (dt == LYC_BF16) launch_dw2f<__bf16>(cfg, nullptr, t_items + t_used, t_ends + t_used, (int)qs.size(), (unsigned)total, st);
      else launch_dw2f<_Float16>(cfg, nullptr, t_items + t_used, t_ends + t_used, (int)qs.size(), (unsigned)total, st);
      t_used += (long)qs.size();
      if (int rc = check_launch("lokr_wgrad_group(full-width tiles, table)")) return rc;
    } else {
      for (size_t lo = 0; lo < qs.size(); lo += DW2F_MAX) {
        KronDw2fGroupArgs ga{};
        ga.n = (int)std::min<size_t>(DW2F_MAX, qs.size() - lo);
        long acc = 0;
        for (int i = 0; i < ga.n; ++i) {
          ga.p[i] = qs[lo + i];
          acc += dw2f_item_wgs(qs[lo + i]);
          ga.wg_end[i] = (int)acc;
        }
        unshare(ga.p, ga.n);
        if (dt == LYC_BF16) launch_dw2f<__bf16>(cfg, &ga, nullptr, nullptr, 0, (unsigned)acc, st);
        else launch_dw2f<_Float16>(cfg, &ga, nullptr, nullptr, 0, (unsigned)acc, st);
        if (int rc = check_launch("lokr_wgrad_group(full-width tiles)")) return rc;
      }
    }
  }
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

#ifdef LYC_TRACE
extern "C" int lyc_trace_read(unsigned long long* host32) {  // development builds: the shader-clock stamps of the last traced kernel
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(lyc::lyc_trace_buf), 32 * sizeof(unsigned long long));
}
#endif
int64_t lyc_lokr_wgrad_table_bytes(int n) { return n > 0 ? (int64_t)dw2f_table_bytes(n) : 0; }

int lyc_lokr_wgrad_group(const LycLokrWgradItem* items, int n, int dtype, void* stream) {
  return lyc_lokr_wgrad_group_ws(items, n, dtype, nullptr, 0, stream);
}

int lyc_lokr_wgrad_group_ws(const LycLokrWgradItem* items, int n, int dtype, void* table, int64_t table_bytes, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "lokr_wgrad_group: bad item list");
  hipStream_t st = (hipStream_t)stream;
  const int dt = dtype & 0xff;
  // full-width tiles (kron_dw2f.h) for every item they fit, unless the caller pins the round 1-3 plan (LYC_WGRAD_TILE_S)
  std::vector<char> wide((size_t)(n > 0 ? n : 0), 0);
  bool any_wide = false;
  if (!(dtype & LYC_WGRAD_TILE_S) && (dt == LYC_BF16 || dt == LYC_F16))
    for (int k = 0; k < n; ++k) {
      const LycLokrWgradItem& it = items[k];
      if (it.g && it.x && it.w1 && it.dw2 && it.a == it.b && it.a >= 1 && (16 % it.a) == 0 && (it.c % 8) == 0 && (it.d % 8) == 0 && it.M > 0 &&
          lokr_wgrad_fast(it.g, it.x, it.M, it.a, it.b, it.c, it.d, dtype) && (!it.dw1 || it.ws) && dw2f_ok(it))
        wide[k] = 1, any_wide = true;
    }
  // The batch plans (fewer, longer slabs; 80 x 32 tiles) assume that the other layers of the call supply the parallelism: a
  // call with a handful of layers (a backward pass over one or two modules) is planned like single launches.
  const bool batch = n >= LYC_GROUP_MIN;
  // one sequence of launches per tile configuration (plan_dw2s), items keep their order
  for (int cfg = 0; cfg < DW2_NCFG; ++cfg) {
    KronDw2sGroupArgs ga{};
    auto flush = [&]() -> int {
      if (ga.n == 0) return LYC_OK;
      for (int i = 0; i < ga.n; ++i)  // a parameter that appears twice in one grid (shared module) must be added atomically
        for (int j = 0; j < i; ++j)
          if (ga.p[i].out == ga.p[j].out || (ga.p[i].dw1 && ga.p[i].dw1 == ga.p[j].dw1))
            ga.p[i].force_atomic = ga.p[j].force_atomic = 1;
      if (dt == LYC_BF16) {
        if (cfg == DW2_T44) launch_dw2s_group<__bf16, 4, 4, 1>(ga, st);
        else if (cfg == DW2_T52) launch_dw2s_group<__bf16, 5, 2, 1>(ga, st);
        else launch_dw2s_group<__bf16, 2, 2, LYC_WG_U>(ga, st);
      } else {
        if (cfg == DW2_T44) launch_dw2s_group<_Float16, 4, 4, 1>(ga, st);
        else if (cfg == DW2_T52) launch_dw2s_group<_Float16, 5, 2, 1>(ga, st);
        else launch_dw2s_group<_Float16, 2, 2, LYC_WG_U>(ga, st);
      }
      ga = KronDw2sGroupArgs{};
      return check_launch("lokr_wgrad_group");
    };
    for (int k = 0; k < n; ++k) {
      const LycLokrWgradItem& it = items[k];
      if (wide[k]) continue;
      if (cfg == 0) {  // validate once
        if (int rc = check_kron_dims(it.M, it.a, it.b, it.c, it.d)) return rc;
        if (!it.g || !it.x || !it.w1 || !it.dw2) return fail(LYC_ERR_ARG, "lokr_wgrad_group: item %d: null pointer", k);
        if (!lokr_wgrad_fast(it.g, it.x, it.M, it.a, it.b, it.c, it.d, dtype))
          return fail(LYC_ERR_UNSUPPORTED, "lokr_wgrad_group: item %d is not on the 16-bit fast path (see lyc_lokr_wgrad_deferrable)", k);
        if (it.dw1 && !it.ws) return fail(LYC_ERR_ARG, "lokr_wgrad_group: item %d: dw1 needs the ws its dx launch wrote", k);
      }
      KronDw2sArgs da{};
      da.Q = it.g; da.P = it.x; da.W = it.w1; da.out = it.dw2; da.M = it.M; da.G = it.a; da.I = it.c; da.J = it.d;
      da.ws = it.b; da.wt = 1; da.os = it.d; da.alpha = it.alpha;
      if (it.dw1) {  // partials of the dx launch: one [a*b] block per workgroup of launch_kron3's grid
        KronArgs ka{};
        ka.M = it.M; ka.Gin = it.a; ka.K = it.c; ka.Gout = it.b; ka.N = it.d;
        da.dw1_ws = static_cast<const float*>(it.ws); da.dw1 = it.dw1; da.dw1_n = it.a * it.b;
        da.dw1_nblk = (int)lokr_dx_partial_blocks(it.M, it.a, it.c, it.d);
        da.dw1_red = 1;
      }
      if (plan_dw2s(da, batch) != cfg) continue;
      const long wgs = round_up((long)da.tiles_i * da.tiles_j * da.nsplit, 8) + round_up(da.dw1_ws ? da.dw1_red : 0, 8);
      const long before = ga.n ? ga.wg_end[ga.n - 1] : 0;
      if (ga.n == DW2G_MAX || before + wgs > (1L << 30))
        if (int rc = flush()) return rc;
      KronDw2sItem& q = ga.p[ga.n];
      q.Q = da.Q; q.P = da.P; q.W = da.W; q.out = da.out; q.dw1_ws = da.dw1_ws; q.dw1 = da.dw1; q.M = da.M;
      q.rows_per_block = da.rows_per_block; q.G = da.G; q.I = da.I; q.J = da.J; q.nsplit = da.nsplit;
      q.tiles_i = da.tiles_i; q.tiles_j = da.tiles_j; q.dw1_nblk = da.dw1_nblk; q.dw1_n = da.dw1_n; q.dw1_red = da.dw1_red;
      q.ws = (int)da.ws; q.wt = (int)da.wt; q.os = (int)da.os; q.alpha = da.alpha; q.force_atomic = 0;
      ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
      ++ga.n;
    }
    if (int rc = flush()) return rc;
  }
  if (any_wide) return lokr_wgrad_group_f(items, wide, n, dt, st, table, table_bytes);
  return LYC_OK;
}

// ---- LoKr on nn.Conv2d: implicit GEMM on NHWC row matrices (no im2col) -------------------------------------
extern "C++" {
namespace {
struct ConvDims {
  long Ho, Wo;
  int taps;
};
int lokr_conv_check(ConvDims& cd, int64_t B, int64_t H, int64_t W, int a, int b, int c, int d, int kh, int kw, int sh,
                    int sw, int ph, int pw, int dh, int dw, int dtype, const void* p0, const void* p1) {
  if (B < 0 || H < 1 || W < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || dh < 1 || dw < 1)
    return fail(LYC_ERR_ARG, "lokr_conv2d: bad geometry");
  if (int rc = check_kron_dims(B * H * W, a, b, c, d)) return rc;
  cd.Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  cd.Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  cd.taps = kh * kw;
  if (cd.Ho < 1 || cd.Wo < 1) return fail(LYC_ERR_ARG, "lokr_conv2d: empty output");
  const bool ok = (dtype & 0xff) != LYC_F32 && a == b && (a == 4 || a == 8 || a == 16) && (c % 8) == 0 && (d % 8) == 0 &&
                  (reinterpret_cast<uintptr_t>(p0) & 15u) == 0 && (reinterpret_cast<uintptr_t>(p1) & 15u) == 0 &&
                  H * W < (1 << 30) && cd.Ho * cd.Wo < (1 << 30);
  if (!ok)
    return fail(LYC_ERR_UNSUPPORTED,
                "lokr_conv2d: the implicit-GEMM path needs 16-bit activations, a == b in {4, 8, 16}, c, d multiples of 8 "
                "and 16-byte aligned rows; use the im2col lowering (lyc_im2col + lyc_lokr_linear_*) otherwise");
  return LYC_OK;
}
KronGather make_gather(int mode, const ConvDims& cd, int64_t H, int64_t W, int kw, int sh, int sw, int ph, int pw, int dh,
                       int dw, long s2t) {
  KronGather gt{};
  gt.mode = mode; gt.taps = cd.taps; gt.kw = kw;
  if (mode == 1) { gt.Hs = (int)H; gt.Ws = (int)W; gt.Hd = (int)cd.Ho; gt.Wd = (int)cd.Wo; }
  else { gt.Hs = (int)cd.Ho; gt.Ws = (int)cd.Wo; gt.Hd = (int)H; gt.Wd = (int)W; }
  gt.sh = sh; gt.sw = sw; gt.ph = ph; gt.pw = pw; gt.dh = dh; gt.dw = dw; gt.s2t = s2t;
  return gt;
}
}  // namespace
}  // extern "C++"

int lyc_lokr_conv2d_fwd(const void* x_rows, const float* w1, const float* w2p, void* y_rows, int64_t B, int64_t H,
                        int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                        int dw, float alpha, int dtype, void* stream) {
  if (!x_rows || !w1 || !w2p || !y_rows) return fail(LYC_ERR_ARG, "lokr_conv2d_fwd: null pointer");
  ConvDims cd{};
  if (int rc = lokr_conv_check(cd, B, H, W, a, b, c, d, kh, kw, sh, sw, ph, pw, dh, dw, dtype, x_rows, y_rows)) return rc;
  if (B == 0) return LYC_OK;
  KronArgs ka{};
  ka.x = x_rows; ka.y = y_rows; ka.w1 = w1; ka.w2 = w2p;
  ka.M = B * cd.Ho * cd.Wo; ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c;
  ka.s1o = b; ka.s1i = 1; ka.s2n = (long)cd.taps * d; ka.s2k = 1; ka.alpha = alpha;
  ka.gat = make_gather(1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
  ka.gat.flat = cd.taps <= 64;  // w2p[q][t][v] is contiguous in the flat (tap, v) index: s2n = taps * d, s2k = 1
  switch (dtype & 0xff) {
    case LYC_BF16: launch_kron3<__bf16>(ka, (hipStream_t)stream); break;
    default: launch_kron3<_Float16>(ka, (hipStream_t)stream); break;
  }
  return check_launch("lokr_conv2d_fwd");
}

int64_t lyc_lokr_conv2d_bwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int a, int b, int d) {
  if (B <= 0 || H < 1 || W < 1 || a < 1 || a != b || a > 16 || d < 1) return 0;
  // one [a*b] partial per workgroup of the dx launch: the row kernel tiles 128 / a consecutive pixels, the patch kernel
  // (kron_conv.h) TH x TW pixel tiles per image; both take column tiles of >= 32
  const int tmp = K3_RT / a;  // the patch kernel's smallest tile (MI = 2: 128 rows) has the most workgroups
  int lt = 0;
  while ((2 << lt) <= tmp) ++lt;
  const int th = 1 << (lt / 2), tw = tmp / th;
  const long flat = cdiv(B * H * W, tmp), tiled = B * cdiv(H, th) * cdiv(W, tw), strip = B * H * cdiv(W, tmp);  // strip: 1 x 1 windows
  long most = flat > tiled ? flat : tiled;
  if (strip > most) most = strip;
  return (int64_t)most * cdiv(d, 32) * a * b * (int64_t)sizeof(float);
}

extern "C++" {
namespace {
// ---- the patch kernel (kron_conv.h): plan, launch ---------------------------------------------------------------------
// `ka` carries the problem as for kron3 (x = source rows, M = destination pixels, K per tap, N, gat with mode 1 / 2).
// Tile of 64 mi / G destination pixels for one MI; false when the patch + operand ring do not fit the LDS.
bool plan_kconv_mi(const KronArgs& ka, int mi, KconvGeom& gm, int ni, int ksteps) {
  const KronGather& gt = ka.gat;
  const int G = ka.Gin, kh = gt.taps / gt.kw;
  const int tmp = 64 * mi / G;
  if (tmp < 2) return false;
  int lt = 0;
  while ((2 << lt) <= tmp) ++lt;       // log2(tmp)
  gm.TH = 1 << (lt / 2);
  gm.TW = tmp / gm.TH;
  if (gt.taps == 1) { gm.TH = 1; gm.TW = tmp; }  // 1 x 1 window: no halo, a strip of consecutive pixels
  if (gt.mode == 1) {
    gm.sy = gt.sh; gm.sx = gt.sw; gm.oy0 = -gt.ph; gm.ox0 = -gt.pw;
    gm.PH = (gm.TH - 1) * gt.sh + (kh - 1) * gt.dh + 1;
    gm.PW = (gm.TW - 1) * gt.sw + (gt.kw - 1) * gt.dw + 1;
  } else {
    gm.sy = gm.sx = 1; gm.oy0 = gt.ph - (kh - 1) * gt.dh; gm.ox0 = gt.pw - (gt.kw - 1) * gt.dw;
    gm.PH = gm.TH + (kh - 1) * gt.dh;
    gm.PW = gm.TW + (gt.kw - 1) * gt.dw;
  }
  gm.tiles_h = (int)cdiv(gt.Hd, gm.TH);
  gm.tiles_w = (int)cdiv(gt.Wd, gm.TW);
  gm.GP = ka.K + (((ka.K / 8) % 2 == 0) ? 8 : 0);  // an odd number of 16-byte slots per group segment
  gm.CP = G * gm.GP;
  const long pbytes = round_up((long)gm.PH * gm.PW * gm.CP * 2, 1024);
  if (pbytes > 140 * 1024) return false;
  gm.patch_bytes = (int)pbytes;
  gm.kss = ksteps < 4 ? ksteps : 4;
  while (gm.kss > 1 && kconv_lds_bytes(ni, gm) > 160 * 1024) gm.kss >>= 1;
  return kconv_lds_bytes(ni, gm) <= 160 * 1024;
}

inline int kconv_pin(int dtype) { return (dtype >> 12) & 0xf; }  // LYC_KCONV_ROW_TILE
bool plan_kconv(const KronArgs& ka, int64_t B, KconvGeom& gm, int& mi, int& ni, int& ksteps, int pin_mi = 0) {
  const KronGather& gt = ka.gat;
  const int G = ka.Gin;
  if (G != ka.Gout || (G != 4 && G != 8 && G != 16) || (ka.K % 8) != 0 || (ka.N % 8) != 0 || gt.taps < 1 || gt.taps > 64) return false;
  if (gt.mode == 2 && (gt.sh != 1 || gt.sw != 1)) return false;  // the transposed convolution of a strided layer: row kernel
  if ((long)gt.Hs * gt.Ws * G * ka.K >= (1L << 30)) return false;  // buffer descriptor offsets are 32-bit
  ksteps = (int)kron_plane_ksteps(gt.taps, ka.K);
  // column tile: the widest of 64 / 48 / 32 that pads N by <= 25 %, else the one that pads least
  ni = 0;
  long best = 1L << 40;
  for (int cand = 4; cand >= 2; --cand) {
    const long padded = round_up(ka.N, 16 * cand);
    if (padded * 4 <= (long)ka.N * 5) { ni = cand; break; }
    if (padded < best) { best = padded; ni = cand; }
  }
#ifdef LYC_KCONV_NI_FILL  // experiment builds (benchmarks/kcbench_nifill): prefer the column tile that fills one round of 256 CUs
  if (ni == 4 && B * cdiv(gt.Hd, 4) * cdiv(gt.Wd, 4) * cdiv(ka.N, 64) < 256 && B * cdiv(gt.Hd, 4) * cdiv(gt.Wd, 4) * cdiv(ka.N, 48) <= 256) ni = 3;
#endif
  // row tile: every workgroup streams the operand planes of its column tile once, so their traffic goes with 1 / MI -- take the
  // largest pixel tile that fits the LDS and still leaves about one workgroup per CU; else the smallest (most workgroups)
  const long ctiles = cdiv(ka.N, 16 * ni);
  KconvGeom g2{};
  bool have = false;
  const int only = pin_mi;  // LYC_KCONV_ROW_TILE(mi) in the call's dtype argument: tests pin the row tile (2 / 4 / 8) so that small problems reach every instantiation
  for (int cand = 8; cand >= 2; cand >>= 1) {
    if (only && cand != only) continue;
    KconvGeom t{};
    if (!plan_kconv_mi(ka, cand, t, ni, ksteps)) continue;
    g2 = t; mi = cand; have = true;
    if (B * t.tiles_h * t.tiles_w * ctiles >= 200) break;
  }
  if (!have) return false;
  gm = g2;
  return true;
}

template <typename T, int MI, int NI, bool DW1, int NW, bool PIPE = false>
void launch_kconv_nw(const KconvArgs& ca, dim3 grid, int lds, hipStream_t st) {
  static bool attr_set = false;  // dynamic LDS above 64 KiB needs the opt-in (per instantiation)
  if (lds > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kconv_kernel<T, MI, NI, DW1, NW, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((kconv_kernel<T, MI, NI, DW1, NW, PIPE>), grid, dim3(NW * 64), lds, st, ca);
}
// w4: the 4-wave workgroups of rounds 3 - 5 (LYC_KCONV_W4 in the call's dtype argument: A/B and regression tests); default: 8 waves,
// with the software-pipelined k loop where the plan has 4 k steps per ring stage (LYC_KCONV_SERIAL: the serial loop, A/B and tests)
template <typename T, int MI, int NI, bool DW1>
void launch_kconv_inst(const KconvArgs& ca, dim3 grid, int lds, hipStream_t st, bool w4, bool serial) {
  if (w4) launch_kconv_nw<T, MI, NI, DW1, 4>(ca, grid, lds, st);
  else if (!serial && ca.gm.kss == 4 && ca.k.K >= 32) launch_kconv_nw<T, MI, NI, DW1, 8, true>(ca, grid, lds, st);
  else launch_kconv_nw<T, MI, NI, DW1, 8>(ca, grid, lds, st);
}
template <typename T, int MI, bool DW1>
void launch_kconv_ni(int ni, const KconvArgs& ca, dim3 grid, int lds, hipStream_t st, bool w4, bool serial) {
  switch (ni) {
    case 2: launch_kconv_inst<T, MI, 2, DW1>(ca, grid, lds, st, w4, serial); break;
    case 3: launch_kconv_inst<T, MI, 3, DW1>(ca, grid, lds, st, w4, serial); break;
    default: launch_kconv_inst<T, MI, 4, DW1>(ca, grid, lds, st, w4, serial); break;
  }
}

// returns the number of workgroups, or -1 when the problem is not plannable
template <typename T>
long launch_kconv(const KronArgs& ka, const void* planes, int64_t B, hipStream_t st, int pin_mi, int flags = 0) {
  bool w4 = (flags & LYC_KCONV_W4) != 0;
  const bool serial = (flags & LYC_KCONV_SERIAL) != 0;
  KconvArgs ca{};
  int mi = 0, ni = 0, ksteps = 0;
  if (!plan_kconv(ka, B, ca.gm, mi, ni, ksteps, pin_mi)) return -1;
  ca.k = ka;
  ca.planes = planes;
  ca.ksteps = ksteps;
  dim3 grid((unsigned)(B * ca.gm.tiles_h * ca.gm.tiles_w), (unsigned)cdiv(ka.N, 16 * ni));
  const int lds = kconv_lds_bytes(ni, ca.gm);
  const bool dw1 = ka.dw1 != nullptr || ka.dw1_ws != nullptr;
  // 8 waves (two per SIMD) by default: -10 % on the 1280-channel layers, -11 % on the 320-channel backward; the 256-row x 48-column
  // tile of the 640-channel layers (8 x 1 split, two row tiles per wave) measured 3 % slower and keeps 4 (profiles/r06_c37_kconv_waves.log)
  if (mi == 4 && ni == 3 && (serial || ca.gm.kss != 4 || ka.K < 32)) w4 = true;
  switch (mi) {
    case 8: dw1 ? launch_kconv_ni<T, 8, true>(ni, ca, grid, lds, st, w4, serial) : launch_kconv_ni<T, 8, false>(ni, ca, grid, lds, st, w4, serial); break;
    case 4: dw1 ? launch_kconv_ni<T, 4, true>(ni, ca, grid, lds, st, w4, serial) : launch_kconv_ni<T, 4, false>(ni, ca, grid, lds, st, w4, serial); break;
    default: dw1 ? launch_kconv_ni<T, 2, true>(ni, ca, grid, lds, st, w4, serial) : launch_kconv_ni<T, 2, false>(ni, ca, grid, lds, st, w4, serial); break;
  }
  return (long)grid.x * grid.y;
}

int lokr_conv2d_bwd_impl(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const float* w2t,
                         const void* planes_bwd, void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W,
                         int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha,
                         int dtype, void* stream);
}  // namespace
}  // extern "C++"

int64_t lyc_lokr_planes_bytes(int c, int d, int taps, int backward) {
  if (c < 1 || d < 1 || taps < 1) return 0;
  return backward ? kron_plane_bytes(d, taps, c) : kron_plane_bytes(c, taps, d);
}

int lyc_lokr_pack_w2(const float* w2, int64_t sq, int64_t sv, int64_t st, const float* w2a, int64_t a_sq, int64_t a_sr,
                     const float* w2b, int64_t b_sr, int64_t b_sv, int64_t b_st, int rank, int c, int d, int taps,
                     void* planes_fwd, void* planes_bwd, int dtype, void* stream) {
  if (c < 1 || d < 1 || taps < 1 || (c % 8) != 0 || (d % 8) != 0) return fail(LYC_ERR_ARG, "lokr_pack_w2: c, d must be positive multiples of 8");
  if (!w2 && !(w2a && w2b && rank >= 1)) return fail(LYC_ERR_ARG, "lokr_pack_w2: pass w2 or the low-rank pair (w2a, w2b, rank)");
  if (!planes_fwd && !planes_bwd) return LYC_OK;
  KronPackArgs pa{};
  pa.w2 = w2; pa.sq = sq; pa.sv = sv; pa.st = st;
  pa.w2a = w2a; pa.w2b = w2b; pa.a_sq = a_sq; pa.a_sr = a_sr; pa.b_sr = b_sr; pa.b_sv = b_sv; pa.b_st = b_st; pa.rank = rank;
  pa.c = c; pa.d = d; pa.taps = taps; pa.fwd = planes_fwd; pa.bwd = planes_bwd;
  pa.units_fwd = kron_plane_bytes(c, taps, d) / 2048;
  const long units = pa.units_fwd + kron_plane_bytes(d, taps, c) / 2048;
  const dim3 grid((unsigned)cdiv(units, NWAVES));
  switch (dtype & 0xff) {
    case LYC_BF16: hipLaunchKernelGGL((kron_pack_kernel<__bf16>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, pa); break;
    case LYC_F16: hipLaunchKernelGGL((kron_pack_kernel<_Float16>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, pa); break;
    default: return fail(LYC_ERR_UNSUPPORTED, "lokr_pack_w2: operand planes exist for 16-bit activations only");
  }
  return check_launch("lokr_pack_w2");
}

int lyc_lokr_conv2d_planes_ok(int64_t B, int64_t H, int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph,
                              int pw, int dh, int dw, int dtype, int backward) {
  ConvDims cd{};
  if (B < 1 || (dtype & 0xff) == LYC_F32 || a != b) return 0;
  if (H < 1 || W < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || dh < 1 || dw < 1) return 0;
  cd.Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  cd.Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  cd.taps = kh * kw;
  if (cd.Ho < 1 || cd.Wo < 1) return 0;
  KronArgs ka{};
  if (!backward) { ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c; }
  else { ka.Gin = a; ka.K = c; ka.Gout = b; ka.N = d; }
  ka.gat = make_gather(backward ? 2 : 1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
  KconvGeom gm{};
  int mi, ni, ks;
  return plan_kconv(ka, B, gm, mi, ni, ks, kconv_pin(dtype)) ? mi : 0;  // != 0: covered; the value is the row tile (64 * value stage-1 rows per workgroup)
}

int lyc_lokr_conv2d_fwd_planes(const void* x_rows, const float* w1, const void* planes_fwd, void* y_rows, int64_t B, int64_t H,
                               int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                               int dw, float alpha, int dtype, void* stream) {
  if (!x_rows || !w1 || !planes_fwd || !y_rows) return fail(LYC_ERR_ARG, "lokr_conv2d_fwd_planes: null pointer");
  ConvDims cd{};
  if (int rc = lokr_conv_check(cd, B, H, W, a, b, c, d, kh, kw, sh, sw, ph, pw, dh, dw, dtype, x_rows, y_rows)) return rc;
  if (B == 0) return LYC_OK;
  KronArgs ka{};
  ka.x = x_rows; ka.y = y_rows; ka.w1 = w1;
  ka.M = B * cd.Ho * cd.Wo; ka.Gin = b; ka.K = d; ka.Gout = a; ka.N = c;
  ka.s1o = b; ka.s1i = 1; ka.alpha = alpha;
  ka.gat = make_gather(1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
  const long n = (dtype & 0xff) == LYC_BF16 ? launch_kconv<__bf16>(ka, planes_fwd, B, (hipStream_t)stream, kconv_pin(dtype), dtype & (LYC_KCONV_W4 | LYC_KCONV_SERIAL))
                                            : launch_kconv<_Float16>(ka, planes_fwd, B, (hipStream_t)stream, kconv_pin(dtype), dtype & (LYC_KCONV_W4 | LYC_KCONV_SERIAL));
  if (n < 0) return fail(LYC_ERR_UNSUPPORTED, "lokr_conv2d_fwd_planes: geometry outside the patch kernel (see lyc_lokr_conv2d_planes_ok)");
  return check_launch("lokr_conv2d_fwd_planes");
}

int lyc_lokr_conv2d_bwd_planes(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const void* planes_bwd,
                               void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W, int a, int b,
                               int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha, int dtype,
                               void* stream) {
  if (!planes_bwd) return fail(LYC_ERR_ARG, "lokr_conv2d_bwd_planes: null planes");
  return lokr_conv2d_bwd_impl(g_rows, x_rows, w1, w2p, nullptr, planes_bwd, dx_rows, dw1, dw2p, ws, B, H, W, a, b, c, d, kh, kw, sh, sw,
                              ph, pw, dh, dw, alpha, dtype, stream);
}

int lyc_lokr_conv2d_bwd(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const float* w2t,
                        void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W, int a, int b, int c, int d,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha, int dtype,
                        void* stream) {
  return lokr_conv2d_bwd_impl(g_rows, x_rows, w1, w2p, w2t, nullptr, dx_rows, dw1, dw2p, ws, B, H, W, a, b, c, d, kh, kw, sh, sw, ph,
                              pw, dh, dw, alpha, dtype, stream);
}

extern "C++" {
namespace {
int lokr_conv2d_bwd_impl(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const float* w2t,
                         const void* planes_bwd, void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W,
                         int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha,
                         int dtype, void* stream) {
  if (!g_rows || !x_rows || !w1 || (!w2p && !planes_bwd)) return fail(LYC_ERR_ARG, "lokr_conv2d_bwd: null pointer");
  ConvDims cd{};
  if (int rc = lokr_conv_check(cd, B, H, W, a, b, c, d, kh, kw, sh, sw, ph, pw, dh, dw, dtype, x_rows, g_rows)) return rc;
  if (dw1 && !dx_rows) return fail(LYC_ERR_ARG, "lokr_conv2d_bwd: dw1 requires dx (they share one pass over g)");
  if (B == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = (dtype & 0xff) == LYC_BF16;
  long dw1_partials = 0;
  if (dx_rows) {
    // transposed convolution: destination rows are INPUT pixels, the operand rows come from g (output pixels)
    KronArgs ka{};
    ka.x = g_rows; ka.y = dx_rows; ka.w1 = w1; ka.w2 = w2p; ka.dw1 = dw1; ka.xref = dw1 ? x_rows : nullptr;
    ka.dw1_ws = (dw1 && ws) ? static_cast<float*>(ws) : nullptr;
    ka.M = B * H * W; ka.Gin = a; ka.K = c; ka.Gout = b; ka.N = d;
    ka.s1o = 1; ka.s1i = b; ka.s2n = 1; ka.s2k = (long)cd.taps * d; ka.alpha = alpha;
    ka.gat = make_gather(2, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
    long nblk = -1;
    if (planes_bwd) {  // LDS source patch + pre-packed operand planes (kron_conv.h)
      nblk = bf ? launch_kconv<__bf16>(ka, planes_bwd, B, st, kconv_pin(dtype), dtype & (LYC_KCONV_W4 | LYC_KCONV_SERIAL)) : launch_kconv<_Float16>(ka, planes_bwd, B, st, kconv_pin(dtype), dtype & (LYC_KCONV_W4 | LYC_KCONV_SERIAL));
      if (nblk < 0 && !w2p)
        return fail(LYC_ERR_UNSUPPORTED, "lokr_conv2d_bwd_planes: geometry outside the patch kernel (see lyc_lokr_conv2d_planes_ok)");
    }
    if (nblk < 0) {
      if (w2t && sh == 1 && sw == 1 && cd.taps <= 64) {
        // stride 1: the source pixel of tap t is base - offset[t], so K can run over the flat (tap, q) index with full
        // segments; that needs the factor as [taps, c, d]: element (n = v, k = t*c + q) at w2t[k * d + v]
        ka.w2 = w2t; ka.s2n = 1; ka.s2k = d; ka.gat.flat = 1;
      }
      nblk = bf ? launch_kron3<__bf16>(ka, st) : launch_kron3<_Float16>(ka, st);
    }
    if (int rc = check_launch("lokr_conv2d_bwd(dx)")) return rc;
    if (ka.dw1_ws) dw1_partials = nblk;
  }
  KronDw2sArgs ra{};
  if (dw1_partials > 0) {
    ra.dw1_ws = static_cast<const float*>(ws); ra.dw1 = dw1; ra.dw1_nblk = (int)dw1_partials; ra.dw1_n = a * b;
    ra.dw1_red = 1;
  }
  if (dw2p) {
    KronDw2sArgs da = ra;
    da.Q = g_rows; da.P = x_rows; da.W = w1; da.out = dw2p; da.M = B * cd.Ho * cd.Wo; da.G = a; da.I = c;
    da.J = cd.taps * d; da.Jt = d; da.ws = b; da.wt = 1; da.os = (long)cd.taps * d; da.alpha = alpha;
    da.gat = make_gather(1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
    if (bf) launch_dw2s<__bf16>(da, st);
    else launch_dw2s<_Float16>(da, st);
    if (int rc = check_launch("lokr_conv2d_bwd(dw2)")) return rc;
  } else if (dw1_partials > 0 && !(dtype & LYC_DEFER_WGRAD)) {  // deferred: lyc_lokr_conv_wgrad_group reduces the partials
    long r = dw1_partials / 64;
    ra.dw1_red = (int)(r > 16 ? 16 : r < 1 ? 1 : r);
    hipLaunchKernelGGL(kron_dw1_reduce_kernel, dim3((unsigned)ra.dw1_red), dim3(NTHREADS), 0, st, ra);
    if (int rc = check_launch("lokr_conv2d_bwd(dw1 reduce)")) return rc;
  }
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

// ---- deferred, grouped weight gradients of the implicit Conv2d form (kron_dw2s_conv_group_kernel) ---------------------------
extern "C++" {
namespace {
template <typename T, int MI, int NJ, int U>
void launch_dw2s_conv_group(const KronDw2sConvGroupArgs& ga, hipStream_t st) {
  hipLaunchKernelGGL((kron_dw2s_conv_group_kernel<T, MI, NJ, U>), dim3((unsigned)ga.wg_end[ga.n - 1]), dim3(NTHREADS), 0, st, ga);
}
}  // namespace
}  // extern "C++"

int64_t lyc_lokr_conv2d_dx_blocks(int64_t B, int64_t H, int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph,
                                  int pw, int dh, int dw, int dtype, int with_planes) {
  // number of [a*b] dw1 partials the dx launch of lyc_lokr_conv2d_bwd / _bwd_planes leaves in `ws` (its workgroup count)
  ConvDims cd{};
  if (B < 1 || a != b || a < 1) return 0;
  cd.Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  cd.Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  cd.taps = kh * kw;
  KronArgs ka{};
  ka.M = B * H * W; ka.Gin = a; ka.K = c; ka.Gout = b; ka.N = d;
  ka.gat = make_gather(2, cd, H, W, kw, sh, sw, ph, pw, dh, dw, d);
  if (with_planes) {
    KconvGeom gm{};
    int mi, ni, ks;
    if (plan_kconv(ka, B, gm, mi, ni, ks, kconv_pin(dtype))) return (int64_t)B * gm.tiles_h * gm.tiles_w * cdiv(ka.N, 16 * ni);
  }
  return (int64_t)cdiv(ka.M, K3_RT / ka.Gin) * cdiv(ka.N, 16 * kron3_pick_ni(ka));
}

extern "C++" {
namespace {
#ifdef LYC_EXPERIMENT_CONV_DW2_PATCH
// plan of the patch kernel (kron_conv_dw2.h) for one layer; false: the row-gather kernel takes it
bool plan_kd(const LycLokrConvWgradItem& it, const ConvDims& cd, KdItem& k) {
  const int G = it.a;
  if (it.a != it.b || (G != 4 && G != 8 && G != 16) || (it.c % 8) != 0 || (it.d % 8) != 0) return false;
  // Measured (profiles/r03_c5_*): correct, but 5.9 ms per SDXL step against 2.2 ms for the grouped row-gather kernel -- its
  // per-block address arithmetic (runtime divisions in a 168-way unrolled loop) issues ~8000 instructions per tile and wave.
  // Experiment builds only (-DLYC_EXPERIMENT_CONV_DW2_PATCH), selected per call with LYC_CONV_WGRAD_PATCH in `dtype`.
  k = KdItem{};
  k.g = it.g_rows; k.x = it.x_rows; k.w1 = it.w1; k.out = it.dw2p;
  k.B = (int)it.B; k.G = G; k.I = it.c; k.J = it.d;
  k.Hs = (int)it.H; k.Ws = (int)it.W; k.Hd = (int)cd.Ho; k.Wd = (int)cd.Wo;
  k.taps = cd.taps; k.kw = it.kw; k.sh = it.sh; k.sw = it.sw; k.ph = it.ph; k.pw = it.pw; k.dh = it.dh; k.dw = it.dw;
  const int tmp = K3_RT / G;
  int lt = 0;
  while ((2 << lt) <= tmp) ++lt;
  k.TH = 1 << (lt / 2);
  k.TW = tmp / k.TH;
  const int pv = 16 / G;
  if (k.TH % pv != 0) return false;
  k.PH = (k.TH - 1) * it.sh + (it.kh - 1) * it.dh + 1;
  k.PW = (k.TW - 1) * it.sw + (it.kw - 1) * it.dw + 1;
  k.tiles_h = (int)cdiv(cd.Ho, k.TH);
  k.tiles_w = (int)cdiv(cd.Wo, k.TW);
  k.nq = (int)cdiv(it.c, KD_WIN); k.KQ = (int)round_up(cdiv(it.c, k.nq), 8);
  k.nv = (int)cdiv(it.d, KD_WIN); k.KV = (int)round_up(cdiv(it.d, k.nv), 8);
  k.nq = (int)cdiv(it.c, k.KQ);
  k.nv = (int)cdiv(it.d, k.KV);
  if (cd.taps * cdiv(k.KQ, 16) * cdiv(k.KV, 16) > 4 * KD_MAXC) return false;
  if ((long)it.H * it.W * G * it.d >= (1L << 30) || cd.Ho * cd.Wo * G * it.c >= (1L << 30)) return false;
  if (kd_lds(k).total() > 160 * 1024) return false;
  const long ntile = it.B * k.tiles_h * k.tiles_w;
  // split over pixel-tile slabs: ~1 M fp32 atomics per layer at most (each slab adds its whole [c, taps, d] block once)
  long slabs = 1000000 / ((long)it.c * cd.taps * it.d);
  if (slabs < 1) slabs = 1;
  if (slabs > ntile) slabs = ntile;
  k.tiles_per_slab = (int)cdiv(ntile, slabs);
  k.slabs = (int)cdiv(ntile, k.tiles_per_slab);
  k.ws = it.b; k.wt = 1; k.os = cd.taps * it.d; k.alpha = it.alpha;
  return true;
}
#endif
}  // namespace
}  // extern "C++"

int lyc_lokr_conv_wgrad_group(const LycLokrConvWgradItem* items_in, int n, int dtype, void* stream) {
  if (n < 0 || (n > 0 && !items_in)) return fail(LYC_ERR_ARG, "lokr_conv_wgrad_group: bad item list");
  hipStream_t st = (hipStream_t)stream;
  const int dt = dtype & 0xff;
  if (n > 0 && dt != LYC_BF16 && dt != LYC_F16) return fail(LYC_ERR_UNSUPPORTED, "lokr_conv_wgrad_group: 16-bit activations only");
  // ---- full-width tiles (kron_dw2f.h, Conv2d form: the taps are column blocks of one virtual [rows, taps * d] operand whose shifted
  //      source rows arrive by per-lane DMA offsets) for every layer they fit, unless the caller pins the round 1-3 plan ---------------
  std::vector<char> wide((size_t)n, 0);
  if (!(dtype & LYC_WGRAD_TILE_S)) {
    for (int cfg = 0; cfg < DW2F_NCFG; ++cfg) {
      const Dw2fCfg cf = DW2F_CFG[cfg];
      KronDw2fGroupArgs ga{};
      auto flush = [&]() -> int {
        if (ga.n == 0) return LYC_OK;
        for (int i = 0; i < ga.n; ++i)
          for (int j = 0; j < i; ++j)
            if (ga.p[i].out == ga.p[j].out || (ga.p[i].dw1 && ga.p[i].dw1 == ga.p[j].dw1)) ga.p[i].plain = ga.p[j].plain = 0;
        const unsigned grid = (unsigned)ga.wg_end[ga.n - 1];
        if (dt == LYC_BF16) launch_dw2f_conv<__bf16>(cfg, ga, grid, st);
        else launch_dw2f_conv<_Float16>(cfg, ga, grid, st);
        ga = KronDw2fGroupArgs{};
        return check_launch("lokr_conv_wgrad_group(full-width tiles)");
      };
      for (int k = 0; k < n; ++k) {
        const LycLokrConvWgradItem& it = items_in[k];
        if (!it.g_rows || !it.x_rows || !it.w1 || !it.dw2p || it.B < 1 || (it.dw1 && (!it.ws || it.dw1_blocks < 1))) continue;  // reported below
        ConvDims cd{};
        if (lokr_conv_check(cd, it.B, it.H, it.W, it.a, it.b, it.c, it.d, it.kh, it.kw, it.sh, it.sw, it.ph, it.pw, it.dh, it.dw, dtype, it.x_rows,
                            it.g_rows))
          continue;  // reported by the loop below
        const long J = (long)cd.taps * it.d, rows = it.B * cd.Ho * cd.Wo * it.a;
        const int my = (it.c > 80 ? (J > 80 ? DW2F_1010 : DW2F_105) : (J > 80 ? DW2F_510 : DW2F_55));
        const bool fits = it.c >= 40 && J >= 64 && cd.taps <= 255 && it.kw <= 255 && it.sh <= 255 && it.sw <= 255 && it.ph <= 255 && it.pw <= 255 &&
                          it.dh <= 255 && it.dw <= 255 && rows * it.c * 2 < (1L << 30) && it.B * it.H * it.W * (long)it.a * it.d * 2 < (1L << 30) &&
                          rows < (1L << 30) && (long)it.c * J < (1L << 24);
        if (!fits || my != cfg) continue;
        KronDw2fItem q{};
        q.Q = it.g_rows; q.P = it.x_rows; q.W = it.w1; q.out = it.dw2p; q.rows_total = (int)rows; q.I = it.c; q.J = (int)J;
        q.lg = 31 - __builtin_clz((unsigned)it.a);
        q.ws = it.b; q.wt = 1; q.os = (int)J; q.alpha = it.alpha;
        q.tiles_i = (int)cdiv(it.c, 16 * cf.TI); q.tiles_j = (int)cdiv(J, 16 * cf.TJ);
        q.p_bytes = (unsigned)(it.B * it.H * it.W * (long)it.a * it.d * 2);
        q.Hs = (int)it.H; q.Ws = (int)it.W; q.Hd = (int)cd.Ho; q.Wd = (int)cd.Wo; q.dtap = it.d;
        q.taps = (unsigned char)cd.taps; q.kw = (unsigned char)it.kw; q.sh = (unsigned char)it.sh; q.sw = (unsigned char)it.sw;
        q.ph = (unsigned char)it.ph; q.pw = (unsigned char)it.pw; q.dh = (unsigned char)it.dh; q.dw = (unsigned char)it.dw;
        const long unit = 32L * cf.WK;
        long ns = cdiv(q.rows_total, 2048);  // rows are cheap here (narrow factors, 9 taps of columns): longer slabs, fewer atomics
        const long out_elems = (long)it.c * J;
        while (ns > 1 && ns * out_elems > 4000000) --ns;
        q.rows_per_slab = (int)(cdiv(cdiv(q.rows_total, ns), unit) * unit);
        q.nslab = (int)cdiv(q.rows_total, q.rows_per_slab);
        q.plain = q.nslab == 1 ? 1 : 0;
        if (it.dw1) {
          q.dw1_ws = static_cast<const float*>(it.ws); q.dw1 = it.dw1; q.dw1_n = it.a * it.b; q.dw1_nblk = (int)it.dw1_blocks;
          long r = it.dw1_blocks / 256;
          q.dw1_red = (int)(r > 8 ? 8 : r < 1 ? 1 : r);
        }
        const long wgs = dw2f_item_wgs(q);
        const long before = ga.n ? ga.wg_end[ga.n - 1] : 0;
        if (ga.n == DW2F_MAX || before + wgs > (1L << 30))
          if (int rc = flush()) return rc;
        ga.p[ga.n] = q;
        ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
        ++ga.n;
        wide[(size_t)k] = 1;
      }
      if (int rc = flush()) return rc;
    }
  }
  std::vector<LycLokrConvWgradItem> narrow_items;
  for (int k = 0; k < n; ++k)
    if (!wide[(size_t)k]) narrow_items.push_back(items_in[k]);
  items_in = narrow_items.data();
  n = (int)narrow_items.size();
#ifdef LYC_EXPERIMENT_CONV_DW2_PATCH
  // ---- layers the patch kernel covers (stride-1 / small-patch geometries): kconv_dw2_group_kernel, 12 layers per launch ------
  std::vector<LycLokrConvWgradItem> rest;
  {
    KdGroupArgs ga{};
    int lds = 0;
    auto flush = [&]() -> int {
      if (ga.n == 0) return LYC_OK;
      for (int i = 0; i < ga.n; ++i)
        for (int j = 0; j < i; ++j)
          if (ga.p[i].out == ga.p[j].out || (ga.p[i].dw1 && ga.p[i].dw1 == ga.p[j].dw1)) ga.p[i].force_atomic = ga.p[j].force_atomic = 1;
      const dim3 grid((unsigned)ga.wg_end[ga.n - 1]);
      if (dt == LYC_BF16) {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kconv_dw2_group_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL((kconv_dw2_group_kernel<__bf16>), grid, dim3(NTHREADS), lds, st, ga);
      } else {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kconv_dw2_group_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL((kconv_dw2_group_kernel<_Float16>), grid, dim3(NTHREADS), lds, st, ga);
      }
      ga = KdGroupArgs{};
      lds = 0;
      return check_launch("lokr_conv_wgrad_group(patch)");
    };
    for (int k = 0; k < n; ++k) {
      const LycLokrConvWgradItem& it = items_in[k];
      if (!it.g_rows || !it.x_rows || !it.w1 || !it.dw2p) return fail(LYC_ERR_ARG, "lokr_conv_wgrad_group: item %d: null pointer", k);
      if (it.dw1 && (!it.ws || it.dw1_blocks < 1)) return fail(LYC_ERR_ARG, "lokr_conv_wgrad_group: item %d: dw1 needs ws and dw1_blocks", k);
      ConvDims cd{};
      if (int rc = lokr_conv_check(cd, it.B, it.H, it.W, it.a, it.b, it.c, it.d, it.kh, it.kw, it.sh, it.sw, it.ph, it.pw, it.dh, it.dw, dtype,
                                   it.x_rows, it.g_rows))
        return rc;
      KdItem kd{};
      if (it.B < 1 || !(dtype & LYC_CONV_WGRAD_PATCH) || !plan_kd(it, cd, kd)) {
        rest.push_back(it);
        continue;
      }
      if (it.dw1) {
        kd.dw1_ws = static_cast<const float*>(it.ws); kd.dw1 = it.dw1; kd.dw1_n = it.a * it.b; kd.dw1_nblk = (int)it.dw1_blocks;
        long r = it.dw1_blocks / 64;
        kd.dw1_red = (int)(r > 16 ? 16 : r < 1 ? 1 : r);
      }
      if (ga.n == KD_MAX_ITEMS)
        if (int rc = flush()) return rc;
      const int wgs = kd.slabs * kd.nq * kd.nv + (kd.dw1_ws ? kd.dw1_red : 0);
      ga.p[ga.n] = kd;
      ga.wg_end[ga.n] = (ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs;
      const int need = kd_lds(kd).total();
      if (need > lds) lds = need;
      ++ga.n;
    }
    if (int rc = flush()) return rc;
  }
#else
  std::vector<LycLokrConvWgradItem> rest(items_in, items_in + n);
#endif
  const LycLokrConvWgradItem* items = rest.data();
  n = (int)rest.size();
  const bool batch = n >= 4;  // a few conv layers already fill the chip (their row counts are 8 x those of the Linear layers)
  for (int cfg = 0; cfg < DW2_NCFG; ++cfg) {
    KronDw2sConvGroupArgs ga{};
    auto flush = [&]() -> int {
      if (ga.n == 0) return LYC_OK;
      for (int i = 0; i < ga.n; ++i)
        for (int j = 0; j < i; ++j)
          if (ga.p[i].it.out == ga.p[j].it.out || (ga.p[i].it.dw1 && ga.p[i].it.dw1 == ga.p[j].it.dw1))
            ga.p[i].it.force_atomic = ga.p[j].it.force_atomic = 1;
      if (dt == LYC_BF16) {
        if (cfg == DW2_T44) launch_dw2s_conv_group<__bf16, 4, 4, 1>(ga, st);
        else if (cfg == DW2_T52) launch_dw2s_conv_group<__bf16, 5, 2, 1>(ga, st);
        else launch_dw2s_conv_group<__bf16, 2, 2, LYC_WG_U>(ga, st);
      } else {
        if (cfg == DW2_T44) launch_dw2s_conv_group<_Float16, 4, 4, 1>(ga, st);
        else if (cfg == DW2_T52) launch_dw2s_conv_group<_Float16, 5, 2, 1>(ga, st);
        else launch_dw2s_conv_group<_Float16, 2, 2, LYC_WG_U>(ga, st);
      }
      ga = KronDw2sConvGroupArgs{};
      return check_launch("lokr_conv_wgrad_group");
    };
    for (int k = 0; k < n; ++k) {
      const LycLokrConvWgradItem& it = items[k];
      ConvDims cd{};
      if (cfg == 0) {
        if (!it.g_rows || !it.x_rows || !it.w1 || !it.dw2p) return fail(LYC_ERR_ARG, "lokr_conv_wgrad_group: item %d: null pointer", k);
        if (it.dw1 && (!it.ws || it.dw1_blocks < 1)) return fail(LYC_ERR_ARG, "lokr_conv_wgrad_group: item %d: dw1 needs ws and dw1_blocks", k);
      }
      if (int rc = lokr_conv_check(cd, it.B, it.H, it.W, it.a, it.b, it.c, it.d, it.kh, it.kw, it.sh, it.sw, it.ph, it.pw, it.dh, it.dw, dtype,
                                   it.x_rows, it.g_rows))
        return rc;
      KronDw2sArgs da{};
      da.Q = it.g_rows; da.P = it.x_rows; da.W = it.w1; da.out = it.dw2p; da.M = it.B * cd.Ho * cd.Wo; da.G = it.a; da.I = it.c;
      da.J = cd.taps * it.d; da.Jt = it.d; da.ws = it.b; da.wt = 1; da.os = (long)cd.taps * it.d; da.alpha = it.alpha;
      da.gat = make_gather(1, cd, it.H, it.W, it.kw, it.sh, it.sw, it.ph, it.pw, it.dh, it.dw, it.d);
      if (it.dw1) {
        da.dw1_ws = static_cast<const float*>(it.ws); da.dw1 = it.dw1; da.dw1_n = it.a * it.b; da.dw1_nblk = (int)it.dw1_blocks;
        da.dw1_red = 1;
      }
      if (plan_dw2s(da, batch) != cfg) continue;
      const long wgs = round_up((long)da.tiles_i * da.tiles_j * da.nsplit, 8) + round_up(da.dw1_ws ? da.dw1_red : 0, 8);
      const long before = ga.n ? ga.wg_end[ga.n - 1] : 0;
      if (ga.n == DW2GC_MAX || before + wgs > (1L << 30))
        if (int rc = flush()) return rc;
      KronDw2sConvItem& ci = ga.p[ga.n];
      KronDw2sItem& q = ci.it;
      q.Q = da.Q; q.P = da.P; q.W = da.W; q.out = da.out; q.dw1_ws = da.dw1_ws; q.dw1 = da.dw1; q.M = da.M;
      q.rows_per_block = da.rows_per_block; q.G = da.G; q.I = da.I; q.J = da.J; q.nsplit = da.nsplit;
      q.tiles_i = da.tiles_i; q.tiles_j = da.tiles_j; q.dw1_nblk = da.dw1_nblk; q.dw1_n = da.dw1_n; q.dw1_red = da.dw1_red;
      q.ws = (int)da.ws; q.wt = (int)da.wt; q.os = (int)da.os; q.alpha = da.alpha; q.force_atomic = 0;
      ci.gat = da.gat;
      ci.Jt = da.Jt;
      ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
      ++ga.n;
    }
    if (int rc = flush()) return rc;
  }
  return LYC_OK;
}

int lyc_locon_linear_fwd(const void* x, const float* down, const float* up, float* t, void* y, int64_t M, int I,
                         int O, int r, float alpha, int dtype, void* stream) {
  if (M < 0 || I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "locon_linear_fwd: bad dims");
  if (!x || !down || !up || !t || !y) return fail(LYC_ERR_ARG, "locon_linear_fwd: null pointer");
  if (M == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  {  // one launch: t = x down^T (kept for the backward pass), y = alpha * t up^T
    BneckArgs b{};
    b.A = x; b.lda = I; b.M = M; b.K1 = I; b.F1 = down; b.f1n = I; b.f1k = 1; b.R = r; b.mid = t;
    b.F2 = up; b.f2n = r; b.f2k = 1; b.N2 = O; b.out = y; b.ldo = O; b.out_f32 = 0; b.alpha1 = 1.0f; b.alpha2 = alpha;
    if (bneck_ok(b, dtype)) {
      launch_bneck_dt(b, dtype, st);
      return check_launch("locon_linear_fwd");
    }
  }
  // general shapes / fp32 activations: split-K kernels that accumulate into t
  if (hipMemsetAsync(t, 0, (size_t)M * r * sizeof(float), st) != hipSuccess) return check_launch("locon_linear_fwd(memset)");
  SkinnyArgs s1{};
  s1.A = x; s1.B = down; s1.out = t; s1.M = M; s1.K = I; s1.Nn = r; s1.lda = I;
  s1.bn = I; s1.bk = 1; s1.os = r; s1.oj = 1; s1.alpha = 1.0f;
  DISPATCH_DTYPE(dtype, launch_skinny_nt<T>(s1, st));
  SkinnyArgs s2{};
  s2.A = t; s2.B = up; s2.out = y; s2.M = M; s2.K = r; s2.Nn = O; s2.lda = r;
  s2.bn = r; s2.bk = 1; s2.os = O; s2.oj = 1; s2.alpha = alpha;
  DISPATCH_DTYPE(dtype, launch_expand_nt<T>(s2, st));
  return check_launch("locon_linear_fwd");
}

extern "C++" {
namespace {
int locon_linear_group(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, int dtype, void* stream, bool backward) {
  const char* who = backward ? "locon_linear_bwd_group" : "locon_linear_fwd_group";
  if (n < 1 || n > BNECK_GROUP_MAX || !items) return fail(LYC_ERR_ARG, "%s: 1 .. %d items", who, BNECK_GROUP_MAX);
  if (I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "%s: bad dims", who);
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || (dtype & ~(0xff | LYC_BNECK_REG))) return fail(LYC_ERR_UNSUPPORTED, "%s: 16-bit activations, no dtype flags but LYC_BNECK_REG", who);
  BneckGroupArgs ga{};
  ga.n = n;
  for (int k = 0; k < n; ++k) {
    const LycLoconLinearGroupItem& it = items[k];
    if (!it.in || !it.down || !it.up || !it.mid || !it.out || it.M < 1) return fail(LYC_ERR_ARG, "%s: item %d: null pointer or empty", who, k);
    BneckArgs& b = ga.p[k];
    if (!backward) {  // t = x down^T (kept for the backward pass), y = alpha * t up^T      (lyc_locon_linear_fwd)
      b.A = it.in; b.lda = I; b.M = it.M; b.K1 = I; b.F1 = it.down; b.f1n = I; b.f1k = 1; b.R = r; b.mid = it.mid;
      b.F2 = it.up; b.f2n = r; b.f2k = 1; b.N2 = O; b.out = it.out; b.ldo = O; b.out_f32 = 0; b.alpha1 = 1.0f; b.alpha2 = it.alpha;
    } else {          // dt = alpha * g up (kept for d_down), dx = dt down                  (lyc_locon_linear_bwd, gradient pointers NULL)
      b.A = it.in; b.lda = O; b.M = it.M; b.K1 = O; b.F1 = it.up; b.f1n = 1; b.f1k = r; b.R = r; b.mid = it.mid;
      b.F2 = it.down; b.f2n = 1; b.f2k = I; b.N2 = I; b.out = it.out; b.ldo = I; b.out_f32 = 0; b.alpha1 = it.alpha; b.alpha2 = 1.0f;
    }
    if (!bneck_ok(b, dtype)) return fail(LYC_ERR_UNSUPPORTED, "%s: item %d is not on the fused rank-r path (16-bit, K %% 8 == 0, r <= 64, aligned rows)", who, k);
  }
  if (!(dtype & LYC_BNECK_REG)) {  // round 6: the LDS-DMA kernel where it covers the set
    if (dt == LYC_BF16 ? launch_bneck4_group<__bf16>(ga, (hipStream_t)stream) : launch_bneck4_group<_Float16>(ga, (hipStream_t)stream)) return check_launch(who);
  }
  const bool ok = dt == LYC_BF16 ? launch_bneck_group<__bf16>(ga, (hipStream_t)stream) : launch_bneck_group<_Float16>(ga, (hipStream_t)stream);
  if (!ok) return fail(LYC_ERR_UNSUPPORTED, "%s: this shape class has no grouped instantiation (r > 32, K >= 8192 at M < 8192, unequal shapes)", who);
  return check_launch(who);
}
}  // namespace
}  // extern "C++"

int lyc_locon_linear_bwd_group_sum(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, void* dx_sum, int dtype, void* stream) {
  const char* who = "locon_linear_bwd_group_sum";
  if (n < 2 || n > BNECK4_GROUP_MAX || !items || !dx_sum) return fail(LYC_ERR_ARG, "%s: 2 .. %d items and dx_sum", who, BNECK4_GROUP_MAX);
  if (I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "%s: bad dims", who);
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || (dtype & ~0xff)) return fail(LYC_ERR_UNSUPPORTED, "%s: 16-bit activations, no dtype flags", who);
  BneckGroupArgs ga{};
  ga.n = n;
  for (int k = 0; k < n; ++k) {
    const LycLoconLinearGroupItem& it = items[k];
    if (!it.in || !it.down || !it.up || !it.mid || it.M < 1) return fail(LYC_ERR_ARG, "%s: item %d: null pointer or empty", who, k);
    BneckArgs& b = ga.p[k];  // dt = alpha * g up (kept for d_down), dx += dt down
    b.A = it.in; b.lda = O; b.M = it.M; b.K1 = O; b.F1 = it.up; b.f1n = 1; b.f1k = r; b.R = r; b.mid = it.mid;
    b.F2 = it.down; b.f2n = 1; b.f2k = I; b.N2 = I; b.out = dx_sum; b.ldo = I; b.out_f32 = 0; b.alpha1 = it.alpha; b.alpha2 = 1.0f;
  }
  const bool ok = dt == LYC_BF16 ? launch_bneck4_sum<__bf16>(ga, dx_sum, (hipStream_t)stream) : launch_bneck4_sum<_Float16>(ga, dx_sum, (hipStream_t)stream);
  if (!ok) return fail(LYC_ERR_UNSUPPORTED, "%s: outside the LDS-DMA rank-r kernel (r <= 16, r %% 4 == 0, I %% 8 == 0, O %% 8 == 0, aligned, n tiles in the LDS)", who);
  return check_launch(who);
}

int lyc_locon_linear_fwd_group(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, int dtype, void* stream) {
  return locon_linear_group(items, n, I, O, r, dtype, stream, false);
}
int lyc_locon_linear_bwd_group(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, int dtype, void* stream) {
  return locon_linear_group(items, n, I, O, r, dtype, stream, true);
}

int lyc_locon_linear_bwd(const void* g, const void* x, const float* down, const float* up, const float* t,
                         float* dt, void* dx, float* d_down, float* d_up, int64_t M, int I, int O, int r,
                         float alpha, int dtype, void* stream) {
  if (M < 0 || I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "locon_linear_bwd: bad dims");
  if (!g || !x || !down || !up || !dt) return fail(LYC_ERR_ARG, "locon_linear_bwd: null pointer");
  if (d_up && !t) return fail(LYC_ERR_ARG, "locon_linear_bwd: d_up needs t from the forward call");
  if (M == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  bool fast_rows = false;
  {  // one launch: dt = alpha * g up (kept for d_down), dx = dt down
    BneckArgs b{};
    b.A = g; b.lda = O; b.M = M; b.K1 = O; b.F1 = up; b.f1n = 1; b.f1k = r; b.R = r; b.mid = dt;
    b.F2 = down; b.f2n = 1; b.f2k = I; b.N2 = I; b.out = dx; b.ldo = I; b.out_f32 = (dtype & LYC_F32_ROWS) ? 1 : 0;
    b.alpha1 = alpha; b.alpha2 = 1.0f;
    if (bneck_ok(b, dtype)) {
      launch_bneck_dt(b, dtype, st);
      if (int rc = check_launch("locon_linear_bwd(dx)")) return rc;
      fast_rows = true;
    }
  }
  if (fast_rows) {  // both factor gradients in one launch
    LowrankTnArgs ta{};
    if (!d_up && !d_down) return LYC_OK;  // dx only: the caller defers the factor gradients (lyc_locon_wgrad_group)
    locon_tn_problems(ta, g, x, t, dt, d_down, d_up, M, I, O, r, alpha);
    if (launch_lowrank_tn(ta, dtype, st)) return check_launch("locon_linear_bwd(factor gradients)");
  } else if (hipMemsetAsync(dt, 0, (size_t)M * r * sizeof(float), st) != hipSuccess) {
    return check_launch("locon_linear_bwd(memset)");
  }
  if (!fast_rows) {  // dt[m, n] = alpha * sum_o g[m, o] * up[o, n]
    SkinnyArgs s{};
    s.A = g; s.B = up; s.out = dt; s.M = M; s.K = O; s.Nn = r; s.lda = O;
    s.bn = 1; s.bk = r; s.os = r; s.oj = 1; s.alpha = alpha;
    DISPATCH_DTYPE(dtype, launch_skinny_nt<T>(s, st));
  }
  if (d_up) {  // d_up[o, n] += alpha * sum_m g[m, o] * t[m, n]
    SkinnyArgs s{};
    s.A = g; s.B = t; s.out = d_up; s.M = M; s.K = O; s.Nn = r; s.lda = O;
    s.bn = 1; s.bk = r; s.os = r; s.oj = 1; s.alpha = alpha;
    DISPATCH_DTYPE(dtype, launch_skinny_tn<T>(s, st));
  }
  if (dx && !fast_rows) {  // dx[m, i] = sum_n dt[m, n] * down[n, i]
    SkinnyArgs s{};
    s.A = dt; s.B = down; s.out = dx; s.M = M; s.K = r; s.Nn = I; s.lda = r;
    s.bn = 1; s.bk = I; s.os = I; s.oj = 1; s.alpha = 1.0f; s.out_f32 = (dtype & LYC_F32_ROWS) ? 1 : 0;
    DISPATCH_DTYPE(dtype, launch_expand_nt<T>(s, st));
  }
  if (d_down) {  // d_down[n, i] += sum_m dt[m, n] * x[m, i]
    SkinnyArgs s{};
    s.A = x; s.B = dt; s.out = d_down; s.M = M; s.K = I; s.Nn = r; s.lda = I;
    s.bn = 1; s.bk = r; s.os = 1; s.oj = I; s.alpha = 1.0f;
    DISPATCH_DTYPE(dtype, launch_skinny_tn<T>(s, st));
  }
  return check_launch("locon_linear_bwd");
}

// ---- deferred, grouped LoCon factor gradients (lowrank_tn_group_kernel) -------------------------------------------------
int lyc_locon_wgrad_deferrable(const void* g, const void* x, int64_t M, int I, int O, int r, int dtype) {
  const int dt = dtype & 0xff;
  if (M < 1 || (dt != LYC_BF16 && dt != LYC_F16) || r < 1 || r > 64) return 0;
  // the same conditions under which lyc_locon_linear_bwd runs its fused dx launch (bneck_ok) and the wave kernel
  return (O % 8) == 0 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (reinterpret_cast<uintptr_t>(x) & 1u) == 0 ? 1 : 0;
}

int lyc_locon_wgrad_group(const LycLoconWgradItem* items, int n, int dtype, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "locon_wgrad_group: bad item list");
  hipStream_t st = (hipStream_t)stream;
  const int dt = dtype & 0xff;
  // one sequence of launches per kernel instantiation (rank tiles 1 / 2 / 4 x columns per lane 1 / 2 / 4 / 8)
  for (int rt = 1; rt <= 4; rt <<= 1)
    for (int cvw = 1; cvw <= 8; cvw <<= 1) {
      LowrankTnGroupArgs ga{};
      auto flush = [&]() -> int {
        if (ga.n == 0) return LYC_OK;
        for (int i = 0; i < ga.n; ++i)  // a parameter that appears twice in one grid must be added atomically
          for (int j = 0; j < i; ++j)
            for (int u = 0; u < 2; ++u)
              for (int v = 0; v < 2; ++v)
                if (ga.p[i].p[u].out && ga.p[i].p[u].out == ga.p[j].p[v].out) ga.p[i].force_atomic = ga.p[j].force_atomic = 1;
        if (dt == LYC_BF16) {
          if (rt == 1) launch_tn_group_rt<__bf16, 1>(ga, cvw, st);
          else if (rt == 2) launch_tn_group_rt<__bf16, 2>(ga, cvw, st);
          else launch_tn_group_rt<__bf16, 4>(ga, cvw, st);
        } else {
          if (rt == 1) launch_tn_group_rt<_Float16, 1>(ga, cvw, st);
          else if (rt == 2) launch_tn_group_rt<_Float16, 2>(ga, cvw, st);
          else launch_tn_group_rt<_Float16, 4>(ga, cvw, st);
        }
        ga = LowrankTnGroupArgs{};
        return check_launch("locon_wgrad_group");
      };
      for (int k = 0; k < n; ++k) {
        const LycLoconWgradItem& it = items[k];
        if (rt == 1 && cvw == 1) {  // validate once
          if (it.M < 1 || it.I < 1 || it.O < 1 || it.r < 1) return fail(LYC_ERR_ARG, "locon_wgrad_group: item %d: bad dims", k);
          if (!it.g || !it.x || !it.dt || (it.d_up && !it.t))
            return fail(LYC_ERR_ARG, "locon_wgrad_group: item %d: null pointer (d_up needs t from the forward call)", k);
          if (!lyc_locon_wgrad_deferrable(it.g, it.x, it.M, it.I, it.O, it.r, dtype))
            return fail(LYC_ERR_UNSUPPORTED, "locon_wgrad_group: item %d is not on the 16-bit fast path (see lyc_locon_wgrad_deferrable)", k);
        }
        if (!it.d_up && !it.d_down) continue;
        if ((it.r <= 16 ? 1 : it.r <= 32 ? 2 : 4) != rt) continue;
        LowrankTnArgs ta{};
        locon_tn_problems(ta, it.g, it.x, it.t, it.dt, it.d_down, it.d_up, it.M, it.I, it.O, it.r, it.alpha);
        const int cv = plan_lowrank_tn(ta, dtype, n >= LYC_GROUP_MIN);
        if (cv == 0) return fail(LYC_ERR_UNSUPPORTED, "locon_wgrad_group: item %d: unaligned activations", k);
        if (cv != cvw) continue;
        const long wgs = cdiv((long)(ta.p[0].tiles + ta.p[1].tiles) * ta.nsplit, NWAVES);
        const long before = ga.n ? ga.wg_end[ga.n - 1] : 0;
        if (ga.n == TNG_MAX || before + wgs > (1L << 30))
          if (int rc = flush()) return rc;
        LowrankTnItem& q = ga.p[ga.n];
        q.p[0] = ta.p[0]; q.p[1] = ta.p[1]; q.M = ta.M; q.rows_per_slab = ta.rows_per_slab; q.R = ta.R; q.nsplit = ta.nsplit;
        q.force_atomic = 0;
        ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
        ++ga.n;
      }
      if (int rc = flush()) return rc;
    }
  return LYC_OK;
}

// ---- LoCon on Conv2d without im2col (reference: modules/locon.py:286-332 with F.conv2d; lora_down is the kh x kw conv,
// lora_up the 1x1 conv).  Rows are NHWC pixel rows; down_p is lora_down as [r, kh, kw, C] (a channels_last parameter
// viewed with permute(0, 2, 3, 1)), up is [O, r].
extern "C++" {
namespace {
int locon_conv_check(ConvDims& cd, int64_t B, int64_t H, int64_t W, int C, int O, int r, int kh, int kw, int sh, int sw,
                     int ph, int pw, int dh, int dw, int dtype, const void* p0, const void* p1) {
  if (B < 0 || H < 1 || W < 1 || C < 1 || O < 1 || r < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 ||
      dh < 1 || dw < 1)
    return fail(LYC_ERR_ARG, "locon_conv2d: bad geometry");
  cd.Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  cd.Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  cd.taps = kh * kw;
  if (cd.Ho < 1 || cd.Wo < 1) return fail(LYC_ERR_ARG, "locon_conv2d: empty output");
  const int dt = dtype & 0xff;
  const bool ok = (dt == LYC_BF16 || dt == LYC_F16) && (C % 16) == 0 && (O % 8) == 0 && (r % 4) == 0 && r <= 16 &&
                  cd.taps <= 64 && cd.taps * r <= 16 * GEXP_KT && (reinterpret_cast<uintptr_t>(p0) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(p1) & 15u) == 0 && H * W < (1 << 30) && cd.Ho * cd.Wo < (1 << 30);
  if (!ok)
    return fail(LYC_ERR_UNSUPPORTED,
                "locon_conv2d: the implicit-GEMM path needs 16-bit activations, C %% 16 == 0, O %% 8 == 0, rank in {4, 8, 12, "
                "16} with kh * kw * rank <= 144 and 16-byte aligned rows; use the im2col lowering (lyc_im2col + "
                "lyc_locon_linear_*) otherwise");
  return LYC_OK;
}
}  // namespace
}  // extern "C++"

int lyc_locon_conv2d_fwd(const void* x_rows, const float* down_p, const float* up, float* t, void* y_rows, int64_t B,
                         int64_t H, int64_t W, int C, int O, int r, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                         int dw, float alpha, int dtype, void* stream) {
  if (!x_rows || !down_p || !up || !t || !y_rows) return fail(LYC_ERR_ARG, "locon_conv2d_fwd: null pointer");
  ConvDims cd{};
  if (int rc = locon_conv_check(cd, B, H, W, C, O, r, kh, kw, sh, sw, ph, pw, dh, dw, dtype, x_rows, y_rows)) return rc;
  if (B == 0) return LYC_OK;
  if ((reinterpret_cast<uintptr_t>(down_p) & 15u) || (reinterpret_cast<uintptr_t>(up) & 15u))
    return fail(LYC_ERR_ARG, "locon_conv2d_fwd: factors must be 16-byte aligned");
  BneckArgs b{};
  b.A = x_rows; b.lda = C; b.M = B * cd.Ho * cd.Wo; b.K1 = cd.taps * C; b.F1 = down_p; b.f1n = (long)cd.taps * C; b.f1k = 1;
  b.R = r; b.mid = t; b.F2 = up; b.f2n = r; b.f2k = 1; b.N2 = O; b.out = y_rows; b.ldo = O; b.alpha1 = 1.0f; b.alpha2 = alpha;
  b.gat = make_gather(1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, 0); b.Ck = C;
  launch_bneck_dt(b, dtype, (hipStream_t)stream);
  return check_launch("locon_conv2d_fwd");
}

int lyc_locon_conv2d_bwd(const void* g_rows, const void* x_rows, const float* down_p, const float* up, const float* t,
                         float* dt, void* dx_rows, float* d_down_p, float* d_up, int64_t B, int64_t H, int64_t W, int C,
                         int O, int r, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha,
                         int dtype, void* stream) {
  if (!g_rows || !x_rows || !down_p || !up || !dt) return fail(LYC_ERR_ARG, "locon_conv2d_bwd: null pointer");
  if (d_up && !t) return fail(LYC_ERR_ARG, "locon_conv2d_bwd: d_up needs t from the forward call");
  ConvDims cd{};
  if (int rc = locon_conv_check(cd, B, H, W, C, O, r, kh, kw, sh, sw, ph, pw, dh, dw, dtype, x_rows, g_rows)) return rc;
  if (B == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  const long Mo = B * cd.Ho * cd.Wo, Mi = B * H * W;
  {  // dt[p_out, n] = alpha * sum_o g[p_out, o] * up[o, n]   (reduce stage only)
    BneckArgs b{};
    b.A = g_rows; b.lda = O; b.M = Mo; b.K1 = O; b.F1 = up; b.f1n = 1; b.f1k = r; b.R = r; b.mid = dt;
    b.F2 = up; b.f2n = 1; b.f2k = 1; b.N2 = 0; b.out = nullptr; b.alpha1 = alpha; b.alpha2 = 1.0f;
    launch_bneck_dt(b, dtype, st);
    if (int rc = check_launch("locon_conv2d_bwd(dt)")) return rc;
  }
  if (dx_rows) {  // transposed convolution of dt with lora_down
    GexpArgs ga{};
    ga.mid = dt; ga.F2 = down_p; ga.out = dx_rows; ga.M = Mi; ga.R = r; ga.C = C;
    ga.gat = make_gather(2, cd, H, W, kw, sh, sw, ph, pw, dh, dw, 0);
    const long rows = cdiv(Mi, 16);
    long ns = 256 / rows;
    if (ns > cdiv(C, 64)) ns = cdiv(C, 64);
    if (ns > 8) ns = 8;
    if (ns < 1) ns = 1;
    const dim3 grid((unsigned)rows, (unsigned)ns);
    if ((dtype & 0xff) == LYC_BF16)
      hipLaunchKernelGGL((gexp_kernel<__bf16, 4>), grid, dim3(256), 0, st, ga);
    else
      hipLaunchKernelGGL((gexp_kernel<_Float16, 4>), grid, dim3(256), 0, st, ga);
    if (int rc = check_launch("locon_conv2d_bwd(dx)")) return rc;
  }
  if (d_up || d_down_p) {
    LowrankTnArgs ta{};
    ta.M = Mo; ta.R = r;
    if (d_up) {
      LowrankTnProb& p = ta.p[0];
      p.act = g_rows; p.ld = O; p.C = O; p.mid = t; p.out = d_up; p.os = r; p.oj = 1; p.alpha = alpha;
    }
    if (d_down_p) {  // d_down_p[n, (tap, c)] += sum_p dt[p, n] * x[src(p, tap), c]
      LowrankTnProb& p = ta.p[1];
      p.act = x_rows; p.ld = C; p.C = cd.taps * C; p.mid = dt; p.out = d_down_p; p.os = 1; p.oj = (long)cd.taps * C;
      p.alpha = 1.0f; p.swap = 1;
      ta.gat = make_gather(1, cd, H, W, kw, sh, sw, ph, pw, dh, dw, 0); ta.Ct = C;
    }
    if (!launch_lowrank_tn(ta, dtype, st)) return fail(LYC_ERR_UNSUPPORTED, "locon_conv2d_bwd: factor-gradient shapes");
    return check_launch("locon_conv2d_bwd(factor gradients)");
  }
  return LYC_OK;
}

int lyc_chan_scale(const void* in, const float* w, const float* bias, void* out, int64_t outer, int64_t C,
                   int64_t inner, float s0, float mult, int dtype, void* stream) {
  if (outer < 0 || C < 1 || inner < 1) return fail(LYC_ERR_ARG, "chan_scale: bad dims");
  if (!in || !w || !out) return fail(LYC_ERR_ARG, "chan_scale: null pointer");
  const long total = outer * C * inner;
  if (total == 0) return LYC_OK;
  ChanArgs ca{};
  ca.a_in = in; ca.out = out; ca.w = w; ca.bias = bias; ca.outer = outer; ca.C = C; ca.inner = inner;
  ca.s0 = s0; ca.mult = mult;
  {  // rows of 16-byte vectors (nn.Linear activations, channels_last tensors): the slab layout of chan_bwd_kernel, factors in registers
    const int vec = (dtype & 0xff) == LYC_F32 ? 4 : 8;
    if (inner == 1 && (C % vec) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
      const long ct = cdiv(C, 64 * vec);
      long slabs = cdiv(1024, ct);
      const long max_slabs = cdiv(outer, 16);
      if (slabs > max_slabs) slabs = max_slabs;
      if (slabs < 1) slabs = 1;
      DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_bwd_kernel<T, true>), dim3((unsigned)ct, (unsigned)slabs), dim3(NTHREADS), 0,
                                               (hipStream_t)stream, ca));
      return check_launch("chan_scale(rows)");
    }
  }
  long blocks = cdiv(total, (long)NTHREADS * 8);
  if (blocks > 256 * 8) blocks = 256 * 8;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_scale_kernel<T>), dim3((unsigned)blocks), dim3(NTHREADS), 0,
                                           (hipStream_t)stream, ca));
  return check_launch("chan_scale");
}

// the backward of lyc_chan_scale in one pass: da = g * (s0 + w*mult), dw += mult * sum g * (a - bias).  inner == 1 with 16-byte rows:
// chan_bwd_kernel; every other layout: the two kernels above, back to back (same results)
int lyc_chan_bwd(const void* g, const void* a, const float* w, const float* bias, void* da, float* dw, int64_t outer, int64_t C,
                 int64_t inner, float s0, float mult, int dtype, void* stream) {
  if (outer < 0 || C < 1 || inner < 1) return fail(LYC_ERR_ARG, "chan_bwd: bad dims");
  if (!g || !a || !w || (!da && !dw)) return fail(LYC_ERR_ARG, "chan_bwd: null pointer");
  if (outer == 0) return LYC_OK;
  const int vec = (dtype & 0xff) == LYC_F32 ? 4 : 8;
  const bool vec_ok = inner == 1 && (C % vec) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(da)) & 15u) == 0;
  if (!vec_ok) {
    if (da) if (int rc = lyc_chan_scale(g, w, nullptr, da, outer, C, inner, s0, mult, dtype, stream)) return rc;
    if (dw) if (int rc = lyc_chan_reduce(g, a, bias, dw, outer, C, inner, mult, dtype, stream)) return rc;
    return LYC_OK;
  }
  ChanArgs ca{};
  ca.a_in = g; ca.b_in = a; ca.out = da; ca.w = w; ca.bias = bias; ca.dw = dw; ca.outer = outer; ca.C = C; ca.inner = 1;
  ca.s0 = s0; ca.mult = mult; ca.vec = 1;
  const long ct = cdiv(C, 64 * vec);
  long slabs = cdiv(1024, ct);
  const long max_slabs = cdiv(outer, 16);  // 16 rows = one round of 4 rows per wave
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs < 1) slabs = 1;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_bwd_kernel<T>), dim3((unsigned)ct, (unsigned)slabs), dim3(NTHREADS), 0, (hipStream_t)stream, ca));
  return check_launch("chan_bwd");
}

int lyc_chan_reduce(const void* a, const void* b, const float* bias, float* dw, int64_t outer, int64_t C,
                    int64_t inner, float mult, int dtype, void* stream) {
  if (outer < 0 || C < 1 || inner < 1) return fail(LYC_ERR_ARG, "chan_reduce: bad dims");
  if (!a || !b || !dw) return fail(LYC_ERR_ARG, "chan_reduce: null pointer");
  if (outer == 0) return LYC_OK;
  ChanArgs ca{};
  ca.a_in = a; ca.b_in = b; ca.dw = dw; ca.bias = bias; ca.outer = outer; ca.C = C; ca.inner = inner;
  ca.mult = mult;
  dim3 grid;
  if (inner == 1) {
    const int vec = (dtype & 0xff) == LYC_F32 ? 4 : 8;  // elements per 16-byte load
    const bool vec_ok = (C % vec) == 0 && (reinterpret_cast<uintptr_t>(a) & 15u) == 0 && (reinterpret_cast<uintptr_t>(b) & 15u) == 0;
    const long ct = cdiv(C, vec_ok ? 64 * vec : 64);
    long slabs = cdiv(vec_ok ? 768 : 1024, ct);
    const long max_slabs = cdiv(outer, 16);
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    grid = dim3((unsigned)ct, (unsigned)slabs);
    ca.vec = vec_ok ? 1 : 0;
  } else {
    if (outer > 65535) return fail(LYC_ERR_UNSUPPORTED, "chan_reduce: outer=%ld too large for the conv layout", (long)outer);
    grid = dim3((unsigned)C, (unsigned)outer);
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_reduce_kernel<T>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, ca));
  return check_launch("chan_reduce");
}


// ---- LoHa ------------------------------------------------------------------------------------------
extern "C++" {
namespace {
struct LohaPlanes {
  char *nh, *nl, *th, *tl;
  long ldn, ldt;
};
// 16-bit activations with O % 8 == I % 8 == 0 (every real layer): one plane; the three contractions run on gemm16.h.
// Other 16-bit dims and fp32 activations: the plane and its transpose for the generic NT / TN kernels of dense_kernels.h.
bool loha_dims_fast(long O, long I) { return (O % 8) == 0 && (I % 8) == 0; }
LohaPlanes loha_planes(void* base, int O, int I, size_t esz) {
  LohaPlanes p;
  p.ldn = round_up(I, 16 / (long)esz);
  p.ldt = round_up(O, 16 / (long)esz);
  char* b = static_cast<char*>(base);
  const size_t n = (size_t)O * p.ldn * esz;
  const bool transposed = esz == 4 || !loha_dims_fast(O, I);
  p.nh = b; p.nl = nullptr; p.th = transposed ? b + n : nullptr; p.tl = nullptr;
  return p;
}
size_t esize(int dtype) { return (dtype & 0xff) == LYC_F32 ? 4 : 2; }

// LoHa's Hadamard product of two rank-r matrices is full rank: once the dW operand plane exists, the three contractions
// (y = x dW^T, dx = g dW, G = g^T x) are plain dense GEMMs -- gemm16.h (rounds 1-2 called rocBLAS here).
// 64 x 128 output tiles: the 128 x 128 instantiation needs > 256 registers for the two-deep prefetch (one wave per SIMD, or
// a compiler that serialises the pipeline to fit the cap) -- measured slower on this workload's M = 1024 ... 4096 problems.
template <typename T, bool A_KS, bool B_KS>
void launch_gemm16_group(const Gemm16Group& ga, hipStream_t st) {
  const dim3 grid((unsigned)ga.wg_end[ga.n - 1]);
  hipLaunchKernelGGL((gemm16_kernel<T, G16_TM, A_KS, B_KS>), grid, dim3(NTHREADS), gemm16_lds_bytes<G16_TM>(), st, ga);
}
// ---- gemm16d.h (round 6): both operands by LDS-DMA, transposed LDS reads, register epilogue -------------------------------------
// Tile classes (profiles/r06_c4_g16bench_two_phase_pipeline.log, every SDXL / SD1.5 shape x the three contractions): what bounds these
// kernels is the operand delivery of a CU (16-25 B / clk through its L1: r06_c6_g16_pmc_tcc_sq_tcp.log), so a problem wants as many
// CUs as it can get before it wants a bigger tile:
//   >= 300 tiles of 128 x 128: 128 x 128, two-slot ring, two workgroups per CU        (FFN up, the grouped G = g^T x launches)
//   >= 128                   : 128 x 128, three-slot ring                             (4096-row attention, FFN down of the 640 blocks)
//   otherwise                : 128 x 64 if that gives >= 128 tiles, else 64 x 64      (1024-row attention, FFN down, text-context layers)
enum { G16D_128_D2 = 0, G16D_128_D3 = 1, G16D_12864 = 2, G16D_64 = 3, G16D_CLASSES = 4 };
inline int gemm16d_class(long M, long N) {
  const long t128 = cdiv(M, 128) * cdiv(N, 128);
  if (t128 >= 300) return G16D_128_D2;
  if (t128 >= 128) return G16D_128_D3;
  return cdiv(M, 128) * cdiv(N, 64) >= 128 ? G16D_12864 : G16D_64;
}
inline void gemm16d_tile(int cls, int& bm, int& bn) {
  bm = cls == G16D_64 ? 64 : 128;
  bn = (cls == G16D_64 || cls == G16D_12864) ? 64 : 128;
}
inline int gemm16d_problem_wgs(const Gemm16Prob& p, int cls) {
  int bm, bn;
  gemm16d_tile(cls, bm, bn);
  return gemm16d_wgs(p.M, p.N, bm, bn);
}
template <typename T, int BM, int BN, bool A_KS, bool B_KS, int D>
void launch_gemm16d_inst(const Gemm16Group& ga, hipStream_t st) {
  constexpr int lds = gemm16d_lds_bytes(BM, BN, D);
  auto kern = gemm16d_kernel<T, BM, BN, A_KS, B_KS, D>;
  if (lds > 64 * 1024) {  // more than the default dynamic-LDS window: opt in once per instantiation
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)once;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)ga.wg_end[ga.n - 1]), dim3(NTHREADS), lds, st, ga);
}
// every problem of `ga` is planned for tile class `cls` (wg_end from gemm16d_problem_wgs)
template <typename T, bool A_KS, bool B_KS>
void launch_gemm16d_group(const Gemm16Group& ga, int cls, hipStream_t st) {
  switch (cls) {
    case G16D_128_D2: launch_gemm16d_inst<T, 128, 128, A_KS, B_KS, 2>(ga, st); break;
    case G16D_128_D3: launch_gemm16d_inst<T, 128, 128, A_KS, B_KS, 3>(ga, st); break;
    case G16D_12864: launch_gemm16d_inst<T, 128, 64, A_KS, B_KS, 3>(ga, st); break;
    default: launch_gemm16d_inst<T, 64, 64, A_KS, B_KS, 3>(ga, st); break;
  }
}

// one problem; mode: 0 = NT, 1 = NN (B K-strided), 2 = TN (both K-strided).  gemm16d where the operands allow it, else gemm16.
template <typename T>
void launch_gemm16(const Gemm16Prob& p, int mode, int out_f32, hipStream_t st) {
  Gemm16Group ga{};
  ga.n = 1; ga.out_f32 = out_f32; ga.p[0] = p;
  if (gemm16d_ok(p, mode == 2, mode >= 1, out_f32 != 0)) {
    const int cls = gemm16d_class(p.M, p.N);
    ga.wg_end[0] = gemm16d_problem_wgs(p, cls);
    if (mode == 0) launch_gemm16d_group<T, false, false>(ga, cls, st);
    else if (mode == 1) launch_gemm16d_group<T, false, true>(ga, cls, st);
    else launch_gemm16d_group<T, true, true>(ga, cls, st);
    return;
  }
  ga.wg_end[0] = (int)(cdiv(p.M, G16_TM) * cdiv(p.N, 128));
  if (mode == 0) launch_gemm16_group<T, false, false>(ga, st);
  else if (mode == 1) launch_gemm16_group<T, false, true>(ga, st);
  else launch_gemm16_group<T, true, true>(ga, st);
}

// HadaWeight.backward on a dense fp32 gradient G [O, I] (functional/loha.py:18-30): shared by the activation path
// (G = g^T x) and the weight-space path (DoRA's norm gradient).
// NO x nt tiles per workgroup (fewer atomics: the a-side gradients stay in registers over nt column tiles, the b-side over
// NO row tiles).  One layer per launch: as long as ~512 workgroups remain.  `grouped` (loha_factor_grad_group_kernel): the
// other layers of the batch supply the parallelism, ~32 workgroups per layer are enough.  Ranks > 32 go tile by tile.
#ifndef LYC_LHG_TARGET
#define LYC_LHG_TARGET 32
#endif
#ifndef LYC_LHG_NTMAX
#define LYC_LHG_NTMAX 8
#endif
// one rank chunk and float4 staging (the conditions of lh_vec_ok): the lean instantiation of the factor-gradient kernels
bool loha_grad_fast(const float* w1a, const float* w1b, const float* w2a, const float* w2b, long I, int r) {
  return r <= LOHA_RC && (r % 4) == 0 && (I % 4) == 0 &&
         (((reinterpret_cast<uintptr_t>(w1a) | reinterpret_cast<uintptr_t>(w2a) | reinterpret_cast<uintptr_t>(w1b) |
            reinterpret_cast<uintptr_t>(w2b)) & 15u) == 0);
}

void plan_loha_grad(long O, long I, int r, bool grouped, int& no, int& nt) {
  const long tiles_o = cdiv(O, LOHA_T), tiles_j = cdiv(I, LOHA_T);
  const long per = (tiles_o * tiles_j) / (grouped ? LYC_LHG_TARGET : 512);
  no = 1;
  nt = 1;
  if (r <= LOHA_RC && per >= 2) {
    no = 2;  // (4 row tiles per workgroup need > 256 registers: one wave per SIMD, measured slower than 2 x nt)
    if (no > tiles_o) no = 1;
    nt = (int)(per / no);
    if (nt < 1) nt = 1;
    if (grouped && nt > LYC_LHG_NTMAX) nt = LYC_LHG_NTMAX;
    if (nt > tiles_j) nt = (int)tiles_j;
  }
}

// loha_grad16.h's kernels use 74 KiB of dynamic LDS: opt in once per instantiation
template <typename K>
void loha_grad16_optin(K kern) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, loha_grad16_lds_bytes());
}
// split16: the caller's activations are 16-bit (the factor gradients carry the 1e-4 bound of DESIGN.md 4): bf16 hi / lo operands on the
// 16-bit matrix cores (loha_grad16.h) where the layer allows it.  fp32 callers (the weight-space path, fp32 activations) stay exact.
void launch_loha_factor_grad(const float* gw, const float* w1a, const float* w1b, const float* w2a, const float* w2b,
                             float* d_w1a, float* d_w1b, float* d_w2a, float* d_w2b, long O, long I, int r, float alpha,
                             hipStream_t st, bool split16 = false) {
  LohaArgs la{};
  la.w1a = w1a; la.w1b = w1b; la.w2a = w2a; la.w2b = w2b; la.O = O; la.I = I; la.R = r; la.scale = alpha;
  la.G = gw; la.d_w1a = d_w1a; la.d_w1b = d_w1b; la.d_w2a = d_w2a; la.d_w2b = d_w2b;
  int no = 1;
  LohaGradGeom gm{1};
  plan_loha_grad(O, I, r, false, no, gm.nt);
  const long tiles_o = cdiv(O, LOHA_T), tiles_j = cdiv(I, LOHA_T);
  dim3 fg((unsigned)cdiv(tiles_o, no), (unsigned)cdiv(tiles_j, gm.nt));
  if (split16 && loha_grad16_ok(w1a, w1b, w2a, w2b, gw, I, r)) {
    static const int once = (loha_grad16_optin(loha_factor_grad16_kernel<1>), loha_grad16_optin(loha_factor_grad16_kernel<2>), 0);
    (void)once;
    if (no == 2) hipLaunchKernelGGL((loha_factor_grad16_kernel<2>), fg, dim3(NTHREADS), loha_grad16_lds_bytes(), st, la, gm);
    else hipLaunchKernelGGL((loha_factor_grad16_kernel<1>), fg, dim3(NTHREADS), loha_grad16_lds_bytes(), st, la, gm);
    return;
  }
  if (loha_grad_fast(w1a, w1b, w2a, w2b, I, r)) {
    switch (no) {
      case 2: hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<2, true>), fg, dim3(NTHREADS), 0, st, la, gm); break;
      default: hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<1, true>), fg, dim3(NTHREADS), 0, st, la, gm); break;
    }
    return;
  }
  switch (no) {
    case 2: hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<2>), fg, dim3(NTHREADS), 0, st, la, gm); break;
    default: hipLaunchKernelGGL((loha_factor_grad_mfma_kernel<1>), fg, dim3(NTHREADS), 0, st, la, gm); break;
  }
}
}  // namespace
}  // extern "C++"

int64_t lyc_loha_workspace_bytes(int O, int I, int dtype) {
  const long esz = (long)esize(dtype);
  const int64_t n = (int64_t)O * round_up(I, 16 / esz) * esz;
  return (esz == 4 || !loha_dims_fast(O, I)) ? n + (int64_t)I * round_up(O, 16 / esz) * esz : n;
}

int lyc_loha_linear_fwd(const void* x, const float* w1a, const float* w1b, const float* w2a, const float* w2b,
                        void* wplanes, void* y, int64_t M, int I, int O, int r, float alpha, int dtype,
                        void* stream) {
  if (M < 0 || I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "loha_linear_fwd: bad dims");
  if (!x || !w1a || !w1b || !w2a || !w2b || !wplanes || !y) return fail(LYC_ERR_ARG, "loha_linear_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  LohaPlanes pl = loha_planes(wplanes, O, I, esize(dtype));
  LohaArgs la{};
  la.w1a = w1a; la.w1b = w1b; la.w2a = w2a; la.w2b = w2b; la.O = O; la.I = I; la.R = r; la.scale = alpha;
  la.Wn_h = pl.nh; la.Wn_l = pl.nl; la.Wt_h = pl.th; la.Wt_l = pl.tl; la.ldn = pl.ldn; la.ldt = pl.ldt;
  dim3 rg((unsigned)cdiv(O, LOHA_T), (unsigned)cdiv(I, LOHA_T));
  const bool wt16 = pl.th != nullptr;  // the transposed plane is read by the generic dx kernels only
  const bool fast16 = !wt16 && (dtype & 0xff) != LYC_F32 && loha_rebuild16_ok(la);  // hi / lo operands on the 16-bit matrix cores
  if (dtype & LYC_PLANE_READY) {  // the caller's plane cache holds the operand plane of these factors (lyc_loha_rebuild_group)
    if (!fast16) return fail(LYC_ERR_ARG, "loha_linear_fwd: LYC_PLANE_READY needs a layer lyc_loha_plane_cacheable accepts");
  } else
  switch (fast16 ? -1 : (dtype & 0xff)) {
    case -1:
      if ((dtype & 0xff) == LYC_BF16) hipLaunchKernelGGL((loha_rebuild16_kernel<__bf16>), rg, dim3(NTHREADS), 0, st, la);
      else hipLaunchKernelGGL((loha_rebuild16_kernel<_Float16>), rg, dim3(NTHREADS), 0, st, la);
      break;
    case LYC_BF16:
      if (wt16) hipLaunchKernelGGL((loha_rebuild_mfma_kernel<__bf16, true>), rg, dim3(NTHREADS), 0, st, la);
      else hipLaunchKernelGGL((loha_rebuild_mfma_kernel<__bf16, false>), rg, dim3(NTHREADS), 0, st, la);
      break;
    case LYC_F16:
      if (wt16) hipLaunchKernelGGL((loha_rebuild_mfma_kernel<_Float16, true>), rg, dim3(NTHREADS), 0, st, la);
      else hipLaunchKernelGGL((loha_rebuild_mfma_kernel<_Float16, false>), rg, dim3(NTHREADS), 0, st, la);
      break;
    case LYC_F32: hipLaunchKernelGGL((loha_rebuild_mfma_kernel<float, true>), rg, dim3(NTHREADS), 0, st, la); break;
    default: return fail(LYC_ERR_ARG, "unknown dtype %d", dtype);
  }
  if (int rc = check_launch("loha_linear_fwd(rebuild)")) return rc;
  if (M > 0 && (dtype & 0xff) != LYC_F32) {
    // y = x dW^T with dW in the activation type (one plane, one pass: the reference's own semantics)
    Gemm16Prob gp{};
    gp.A = x; gp.B = pl.nh; gp.C = y; gp.M = (int)M; gp.N = O; gp.K = I; gp.lda = I; gp.ldb = (int)pl.ldn; gp.ldc = O; gp.alpha = 1.0f;
    if (!wt16 && gemm16_ok(gp, false, false)) {
      if ((dtype & 0xff) == LYC_BF16) launch_gemm16<__bf16>(gp, 0, 0, st);
      else launch_gemm16<_Float16>(gp, 0, 0, st);
    } else {  // odd dims / unaligned rows: the generic NT kernel on the single plane
      GemmArgs ga{};
      ga.A = x; ga.Bh = pl.nh; ga.Bl = pl.nh; ga.out = y; ga.M = M; ga.N = O; ga.K = I;
      ga.lda = I; ga.ldb = pl.ldn; ga.ldo = O; ga.alpha = 1.0f;
      dim3 gg((unsigned)cdiv(M, 128), (unsigned)cdiv(O, 128));
      if ((dtype & 0xff) == LYC_BF16) hipLaunchKernelGGL((gemm_nt_kernel<__bf16, false>), gg, dim3(NTHREADS), 0, st, ga);
      else hipLaunchKernelGGL((gemm_nt_kernel<_Float16, false>), gg, dim3(NTHREADS), 0, st, ga);
    }
    return check_launch("loha_linear_fwd");
  }
  if (M > 0) {  // fp32 activations: exact fp32 MFMA kernel
    GemmArgs ga{};
    ga.A = x; ga.Bh = pl.nh; ga.Bl = pl.nl; ga.out = y; ga.M = M; ga.N = O; ga.K = I;
    ga.lda = I; ga.ldb = pl.ldn; ga.ldo = O; ga.alpha = 1.0f;
    dim3 gg((unsigned)cdiv(M, 128), (unsigned)cdiv(O, 128));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gemm_nt_kernel<T>), gg, dim3(NTHREADS), 0, st, ga));
  }
  return check_launch("loha_linear_fwd");
}

// ---- operand planes of many layers in one launch (the plane cache's refresh) -----------------------------------------------------
int lyc_loha_plane_cacheable(const float* w1a, const float* w1b, const float* w2a, const float* w2b, int I, int O, int r, int dtype) {
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || I < 1 || O < 1 || r < 1 || !loha_dims_fast(O, I)) return 0;
  LohaArgs la{};
  la.w1a = w1a; la.w1b = w1b; la.w2a = w2a; la.w2b = w2b; la.O = O; la.I = I; la.R = r; la.ldn = round_up(I, 8);
  la.Wn_h = nullptr;
  return loha_rebuild16_ok(la) ? 1 : 0;
}
int lyc_loha_rebuild_group(const LycLohaPlaneItem* items, int n, int dtype, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "loha_rebuild_group: bad item list");
  const int dt = dtype & 0xff;
  if (n > 0 && dt != LYC_BF16 && dt != LYC_F16) return fail(LYC_ERR_UNSUPPORTED, "loha_rebuild_group: 16-bit planes only");
  hipStream_t st = (hipStream_t)stream;
  LohaRebuildGroupArgs ga{};
  auto flush = [&]() -> int {
    if (ga.n == 0) return LYC_OK;
    const dim3 grid((unsigned)ga.wg_end[ga.n - 1]);
    if (dt == LYC_BF16) hipLaunchKernelGGL((loha_rebuild16_group_kernel<__bf16>), grid, dim3(NTHREADS), 0, st, ga);
    else hipLaunchKernelGGL((loha_rebuild16_group_kernel<_Float16>), grid, dim3(NTHREADS), 0, st, ga);
    ga = LohaRebuildGroupArgs{};
    return check_launch("loha_rebuild_group");
  };
  for (int k = 0; k < n; ++k) {
    const LycLohaPlaneItem& it = items[k];
    if (!it.w1a || !it.w1b || !it.w2a || !it.w2b || !it.plane || (reinterpret_cast<uintptr_t>(it.plane) & 15u))
      return fail(LYC_ERR_ARG, "loha_rebuild_group: item %d: null / unaligned pointer", k);
    if (!lyc_loha_plane_cacheable(it.w1a, it.w1b, it.w2a, it.w2b, it.I, it.O, it.r, dtype))
      return fail(LYC_ERR_UNSUPPORTED, "loha_rebuild_group: item %d: rank <= 32, rank %% 4 == 0, I %% 8 == O %% 8 == 0, 16-byte aligned factors", k);
    const long gx = cdiv(it.O, 128), wgs = gx * cdiv(cdiv(it.I, LOHA_T), LRG_NCT);
    if (ga.n == LRG_MAX || (ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs > (1L << 30))
      if (int rc = flush()) return rc;
    LohaRebuildItem& q = ga.p[ga.n];
    q.w1a = it.w1a; q.w1b = it.w1b; q.w2a = it.w2a; q.w2b = it.w2b; q.plane = it.plane;
    q.O = it.O; q.I = it.I; q.R = it.r; q.ldn = (int)round_up(it.I, 8); q.gx = (int)gx; q.scale = it.alpha;
    ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
    ++ga.n;
  }
  return flush();
}

int lyc_loha_linear_bwd(const void* g, const void* x, const float* w1a, const float* w1b, const float* w2a,
                        const float* w2b, const void* wplanes, float* gw, void* dx, float* d_w1a, float* d_w1b,
                        float* d_w2a, float* d_w2b, int64_t M, int I, int O, int r, float alpha, int dtype,
                        void* stream) {
  if (M < 0 || I < 1 || O < 1 || r < 1) return fail(LYC_ERR_ARG, "loha_linear_bwd: bad dims");
  if (!g || !x || !w1a || !w1b || !w2a || !w2b || !wplanes) return fail(LYC_ERR_ARG, "loha_linear_bwd: null pointer");
  const bool want_factors = d_w1a || d_w1b || d_w2a || d_w2b;
  if (want_factors && !(d_w1a && d_w1b && d_w2a && d_w2b && gw))
    return fail(LYC_ERR_ARG, "loha_linear_bwd: factor gradients come as a set of four and need the gw scratch");
  if (M == 0) return LYC_OK;
  hipStream_t st = (hipStream_t)stream;
  LohaPlanes pl = loha_planes(const_cast<void*>(wplanes), O, I, esize(dtype));
  const bool lib = (dtype & 0xff) != LYC_F32;
  const bool bf = (dtype & 0xff) == LYC_BF16;
  if (dx && lib) {  // dx = g dW: [M,O] x [O,I] on the same plane (B operand K-strided)
    Gemm16Prob gp{};
    gp.A = g; gp.B = pl.nh; gp.C = dx; gp.M = (int)M; gp.N = I; gp.K = O; gp.lda = O; gp.ldb = (int)pl.ldn; gp.ldc = I; gp.alpha = 1.0f;
    const int of32 = (dtype & LYC_F32_ROWS) ? 1 : 0;
    if (pl.th == nullptr && gemm16_ok(gp, false, true)) {
      if (bf) launch_gemm16<__bf16>(gp, 1, of32, st);
      else launch_gemm16<_Float16>(gp, 1, of32, st);
    } else {
      if (pl.th == nullptr) return fail(LYC_ERR_UNSUPPORTED, "loha_linear_bwd: unaligned activations need the transposed plane (odd dims)");
      GemmArgs ga{};
      ga.A = g; ga.Bh = pl.th; ga.Bl = pl.th; ga.out = dx; ga.M = M; ga.N = I; ga.K = O;
      ga.lda = O; ga.ldb = pl.ldt; ga.ldo = I; ga.alpha = 1.0f; ga.out_f32 = of32;
      dim3 gg((unsigned)cdiv(M, 128), (unsigned)cdiv(I, 128));
      if (bf) hipLaunchKernelGGL((gemm_nt_kernel<__bf16, false>), gg, dim3(NTHREADS), 0, st, ga);
      else hipLaunchKernelGGL((gemm_nt_kernel<_Float16, false>), gg, dim3(NTHREADS), 0, st, ga);
    }
  } else if (dx) {  // dx = g @ dW : B operand rows = i, K = o  -> the transposed planes
    GemmArgs ga{};
    ga.A = g; ga.Bh = pl.th; ga.Bl = pl.tl; ga.out = dx; ga.M = M; ga.N = I; ga.K = O;
    ga.lda = O; ga.ldb = pl.ldt; ga.ldo = I; ga.alpha = 1.0f; ga.out_f32 = (dtype & LYC_F32_ROWS) ? 1 : 0;
    dim3 gg((unsigned)cdiv(M, 128), (unsigned)cdiv(I, 128));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gemm_nt_kernel<T>), gg, dim3(NTHREADS), 0, st, ga));
  }
  if (want_factors) {  // G = g^T x (fp32, [O, I]), then the Hadamard chain rule on the factors
    Gemm16Prob gp{};
    gp.A = g; gp.B = x; gp.C = gw; gp.M = O; gp.N = I; gp.K = (int)M; gp.lda = O; gp.ldb = I; gp.ldc = I; gp.alpha = 1.0f;
    if (lib && gemm16_ok(gp, true, true)) {
      if (bf) launch_gemm16<__bf16>(gp, 2, 1, st);
      else launch_gemm16<_Float16>(gp, 2, 1, st);
    } else {
      GemmArgs ga{};
      ga.A = g; ga.Bh = x; ga.out = gw; ga.M = O; ga.N = I; ga.K = M;
      ga.lda = O; ga.ldb = I; ga.ldo = I; ga.alpha = 1.0f; ga.atomic = 0; ga.chunk = M;
      dim3 gg((unsigned)cdiv(O, 128), (unsigned)cdiv(I, 128), 1);
      DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gemm_tn_kernel<T>), gg, dim3(NTHREADS), 0, st, ga));
    }
    launch_loha_factor_grad(gw, w1a, w1b, w2a, w2b, d_w1a, d_w1b, d_w2a, d_w2b, O, I, r, alpha, st, lib);
  }
  return check_launch("loha_linear_bwd");
}


// ---- deferred, grouped LoHa factor gradients (loha_factor_grad_group_kernel) ---------------------------------------------
int lyc_loha_wgrad_deferrable(const void* g, const void* x, int64_t M, int I, int O, int r, int dtype) {
  const int dt = dtype & 0xff;
  (void)g; (void)x;
  return (M >= 1 && (dt == LYC_BF16 || dt == LYC_F16) && r >= 1 && O < (1 << 30) && I < (1 << 30)) ? 1 : 0;
}

int lyc_loha_wgrad_group(const LycLohaWgradItem* items, int n, int dtype, void* stream) {
  if (n < 0 || (n > 0 && !items)) return fail(LYC_ERR_ARG, "loha_wgrad_group: bad item list");
  hipStream_t st = (hipStream_t)stream;
  for (int k = 0; k < n; ++k) {  // G_k = g_k^T x_k (fp32, [O, I]): one library GEMM per layer, back to back
    const LycLohaWgradItem& it = items[k];
    if (it.M < 1 || it.I < 1 || it.O < 1 || it.r < 1) return fail(LYC_ERR_ARG, "loha_wgrad_group: item %d: bad dims", k);
    if (!it.g || !it.x || !it.w1a || !it.w1b || !it.w2a || !it.w2b || !it.gw || !it.d_w1a || !it.d_w1b || !it.d_w2a || !it.d_w2b)
      return fail(LYC_ERR_ARG, "loha_wgrad_group: item %d: null pointer (the four gradients come as a set and need the gw scratch)", k);
    if (!lyc_loha_wgrad_deferrable(it.g, it.x, it.M, it.I, it.O, it.r, dtype))
      return fail(LYC_ERR_UNSUPPORTED, "loha_wgrad_group: item %d needs 16-bit activations", k);
  }
  {  // G_k = g_k^T x_k (fp32, [O, I]) of ALL layers: gemm16d TN (gemm16 for operands it does not take), up to 24 layers per launch
    const bool bf = (dtype & 0xff) == LYC_BF16;
    // the batch supplies the parallelism: one tile class for the whole call, chosen for the sum of its problems
    long t128 = 0;
    for (int k = 0; k < n; ++k) t128 += cdiv(items[k].O, 128) * cdiv(items[k].I, 128);
    const int cls = t128 >= 300 ? G16D_128_D2 : (t128 >= 128 ? G16D_128_D3 : G16D_64);
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: gemm16d problems, pass 1: the rest on gemm16
      Gemm16Group ga{};
      ga.out_f32 = 1;
      auto flush = [&]() -> int {
        if (ga.n == 0) return LYC_OK;
        if (pass == 0) {
          if (bf) launch_gemm16d_group<__bf16, true, true>(ga, cls, st);
          else launch_gemm16d_group<_Float16, true, true>(ga, cls, st);
        } else {
          if (bf) launch_gemm16_group<__bf16, true, true>(ga, st);
          else launch_gemm16_group<_Float16, true, true>(ga, st);
        }
        ga = Gemm16Group{};
        ga.out_f32 = 1;
        return check_launch("loha_wgrad_group(G)");
      };
      for (int k = 0; k < n; ++k) {
        const LycLohaWgradItem& it = items[k];
        Gemm16Prob gp{};
        gp.A = it.g; gp.B = it.x; gp.C = it.gw; gp.M = it.O; gp.N = it.I; gp.K = (int)it.M; gp.lda = it.O; gp.ldb = it.I; gp.ldc = it.I;
        gp.alpha = 1.0f;
        const bool dma = gemm16d_ok(gp, true, true, true);
        if (dma != (pass == 0)) continue;
        if (!dma && !gemm16_ok(gp, true, true)) {  // odd dims: the generic TN kernel, one launch
          GemmArgs ta{};
          ta.A = it.g; ta.Bh = it.x; ta.out = it.gw; ta.M = it.O; ta.N = it.I; ta.K = it.M;
          ta.lda = it.O; ta.ldb = it.I; ta.ldo = it.I; ta.alpha = 1.0f; ta.atomic = 0; ta.chunk = it.M;
          dim3 gg((unsigned)cdiv(it.O, 128), (unsigned)cdiv(it.I, 128), 1);
          if (bf) hipLaunchKernelGGL((gemm_tn_kernel<__bf16>), gg, dim3(NTHREADS), 0, st, ta);
          else hipLaunchKernelGGL((gemm_tn_kernel<_Float16>), gg, dim3(NTHREADS), 0, st, ta);
          continue;
        }
        const long wgs = dma ? gemm16d_problem_wgs(gp, cls) : cdiv(it.O, G16_TM) * cdiv(it.I, 128);
        if (ga.n == G16_MAX || (ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs > (1L << 30))
          if (int rc = flush()) return rc;
        ga.p[ga.n] = gp;
        ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + wgs);
        ++ga.n;
      }
      if (int rc = flush()) return rc;
    }
  }
  // fast: 0 = general form, 1 = one rank chunk + float4 staging on the fp32 matrix core, 2 = bf16 hi / lo split (loha_grad16.h)
  for (int fast = 0; fast < 3; ++fast)
  for (int no = 1; no <= 2; no <<= 1) {  // one sequence of launches per kernel instantiation
    LohaGradGroupArgs ga{};
    auto flush = [&]() -> int {
      if (ga.n == 0) return LYC_OK;
      const dim3 grid((unsigned)ga.wg_end[ga.n - 1]);
      if (fast == 2) {
        static const int once = (loha_grad16_optin(loha_factor_grad16_group_kernel<1>), loha_grad16_optin(loha_factor_grad16_group_kernel<2>), 0);
        (void)once;
        if (no == 2) hipLaunchKernelGGL((loha_factor_grad16_group_kernel<2>), grid, dim3(NTHREADS), loha_grad16_lds_bytes(), st, ga);
        else hipLaunchKernelGGL((loha_factor_grad16_group_kernel<1>), grid, dim3(NTHREADS), loha_grad16_lds_bytes(), st, ga);
      } else if (fast) {
        switch (no) {
          case 2: hipLaunchKernelGGL((loha_factor_grad_group_kernel<2, true>), grid, dim3(NTHREADS), 0, st, ga); break;
          default: hipLaunchKernelGGL((loha_factor_grad_group_kernel<1, true>), grid, dim3(NTHREADS), 0, st, ga); break;
        }
      } else {
        switch (no) {
          case 2: hipLaunchKernelGGL((loha_factor_grad_group_kernel<2>), grid, dim3(NTHREADS), 0, st, ga); break;
          default: hipLaunchKernelGGL((loha_factor_grad_group_kernel<1>), grid, dim3(NTHREADS), 0, st, ga); break;
        }
      }
      ga = LohaGradGroupArgs{};
      return check_launch("loha_wgrad_group");
    };
    for (int k = 0; k < n; ++k) {
      const LycLohaWgradItem& it = items[k];
      int pno = 1, nt = 1;
      plan_loha_grad(it.O, it.I, it.r, n >= LYC_GROUP_MIN, pno, nt);
      const int lvl = loha_grad16_ok(it.w1a, it.w1b, it.w2a, it.w2b, it.gw, it.I, it.r) ? 2 : (loha_grad_fast(it.w1a, it.w1b, it.w2a, it.w2b, it.I, it.r) ? 1 : 0);
      if (pno != no || lvl != fast) continue;
      const long gx = cdiv(cdiv(it.O, LOHA_T), no), gy = cdiv(cdiv(it.I, LOHA_T), nt);
      const long before = ga.n ? ga.wg_end[ga.n - 1] : 0;
      if (ga.n == LHG_MAX || before + gx * gy > (1L << 30))
        if (int rc = flush()) return rc;
      LohaGradItem& q = ga.p[ga.n];
      q.w1a = it.w1a; q.w1b = it.w1b; q.w2a = it.w2a; q.w2b = it.w2b; q.G = it.gw;
      q.d_w1a = it.d_w1a; q.d_w1b = it.d_w1b; q.d_w2a = it.d_w2a; q.d_w2b = it.d_w2b;
      q.O = it.O; q.I = it.I; q.R = it.r; q.nt = nt; q.gx = (int)gx; q.scale = it.alpha;
      ga.wg_end[ga.n] = (int)((ga.n ? ga.wg_end[ga.n - 1] : 0) + gx * gy);
      ++ga.n;
    }
    if (int rc = flush()) return rc;
  }
  return LYC_OK;
}

// ---- weight space: merge / diff weight / max-norm / DoRA (wspace.h) ------------------------------------------------
int lyc_wspace(int algo, const float* f0, const float* f1, const float* f2, const float* f3, int64_t O, int64_t J, int r,
               int a, int b, int c, int kk, const void* W, int w_dtype, float w_scale, const float* coef, int chan_mode,
               void* out, int out_dtype, float beta, float* sums, float alpha, void* stream) {
  if (O < 1 || J < 1 || kk < 1) return fail(LYC_ERR_ARG, "wspace: bad dims O=%ld J=%ld kk=%d", (long)O, (long)J, kk);
  if (!f0 || !f1) return fail(LYC_ERR_ARG, "wspace: null factor");
  if (!out && !sums) return fail(LYC_ERR_ARG, "wspace: nothing to do (out and sums are both NULL)");
  if (chan_mode < WS_CH_ONE || chan_mode > WS_CH_COL) return fail(LYC_ERR_ARG, "wspace: bad chan_mode %d", chan_mode);
  if ((W && (w_dtype < 0 || w_dtype > 2)) || (out && (out_dtype < 0 || out_dtype > 2))) return fail(LYC_ERR_ARG, "wspace: bad dtype");
  WspaceArgs wa{};
  wa.f0 = f0; wa.f1 = f1; wa.f2 = f2; wa.f3 = f3; wa.O = O; wa.J = J; wa.R = r; wa.a = a; wa.b = b; wa.c = c; wa.kk = kk;
  wa.W = W; wa.w_dtype = w_dtype; wa.w_scale = w_scale; wa.coef = coef; wa.chan_mode = chan_mode; wa.out = out;
  wa.out_dtype = out_dtype; wa.beta = beta; wa.sums = sums; wa.alpha = alpha;
  if (cdiv(J, WS_T) > 65535) return fail(LYC_ERR_UNSUPPORTED, "wspace: J=%ld too wide", (long)J);
  const dim3 grid((unsigned)cdiv(O, WS_T), (unsigned)cdiv(J, WS_T));
  hipStream_t st = (hipStream_t)stream;
  switch (algo) {
    case WS_LOCON:
      if (r < 1) return fail(LYC_ERR_ARG, "wspace(locon): rank %d", r);
      hipLaunchKernelGGL((wspace_kernel<WS_LOCON>), grid, dim3(NTHREADS), 0, st, wa);
      break;
    case WS_LOHA:
      if (r < 1 || !f2 || !f3) return fail(LYC_ERR_ARG, "wspace(loha): needs four factors and a rank");
      hipLaunchKernelGGL((wspace_kernel<WS_LOHA>), grid, dim3(NTHREADS), 0, st, wa);
      break;
    case WS_LOKR:
      if (a < 1 || b < 1 || c < 1 || (long)a * c != O || J % b != 0)
        return fail(LYC_ERR_ARG, "wspace(lokr): w1 %dx%d, w2 rows %d do not tile [%ld, %ld]", a, b, c, (long)O, (long)J);
      wa.dk = J / b;
      hipLaunchKernelGGL((wspace_kernel<WS_LOKR>), grid, dim3(NTHREADS), 0, st, wa);
      break;
    default: return fail(LYC_ERR_ARG, "wspace: unknown algo %d", algo);
  }
  return check_launch("wspace");
}

int lyc_locon_wgrad(const float* gw, const float* down, const float* up, float* d_down, float* d_up, int64_t O, int64_t J,
                    int r, float alpha, void* stream) {
  if (O < 1 || J < 1 || r < 1 || !gw || !down || !up) return fail(LYC_ERR_ARG, "locon_wgrad: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (d_up) {  // d_up[o, n] += alpha sum_j Gw[o, j] down[n, j]
    SkinnyArgs s{};
    s.A = gw; s.B = down; s.out = d_up; s.M = O; s.K = J; s.Nn = r; s.lda = J; s.bn = J; s.bk = 1; s.os = r; s.oj = 1;
    s.alpha = alpha;
    launch_skinny_nt<float>(s, st);
  }
  if (d_down) {  // d_down[n, j] += alpha sum_o up[o, n] Gw[o, j]
    SkinnyArgs s{};
    s.A = gw; s.B = up; s.out = d_down; s.M = O; s.K = J; s.Nn = r; s.lda = J; s.bn = 1; s.bk = r; s.os = 1; s.oj = J;
    s.alpha = alpha;
    launch_skinny_tn<float>(s, st);
  }
  return check_launch("locon_wgrad");
}

int lyc_loha_wgrad(const float* gw, const float* w1a, const float* w1b, const float* w2a, const float* w2b, float* d_w1a,
                   float* d_w1b, float* d_w2a, float* d_w2b, int64_t O, int64_t J, int r, float alpha, void* stream) {
  if (O < 1 || J < 1 || r < 1 || !gw || !w1a || !w1b || !w2a || !w2b || !d_w1a || !d_w1b || !d_w2a || !d_w2b)
    return fail(LYC_ERR_ARG, "loha_wgrad: bad arguments (the four gradients come as a set)");
  launch_loha_factor_grad(gw, w1a, w1b, w2a, w2b, d_w1a, d_w1b, d_w2a, d_w2b, O, J, r, alpha, (hipStream_t)stream);
  return check_launch("loha_wgrad");
}

int lyc_lokr_wgrad(const float* gw, const float* w1, const float* w2, float* d_w1, float* d_w2, int a, int b, int c,
                   int64_t dk, float alpha, void* stream) {
  if (a < 1 || b < 1 || c < 1 || dk < 1 || !gw || !w1 || !w2) return fail(LYC_ERR_ARG, "lokr_wgrad: bad arguments");
  if (cdiv(dk, NTHREADS) > 65535) return fail(LYC_ERR_UNSUPPORTED, "lokr_wgrad: dk=%ld too wide", (long)dk);
  KronWgradArgs k{};
  k.gw = gw; k.w1 = w1; k.w2 = w2; k.d_w1 = d_w1; k.d_w2 = d_w2; k.a = a; k.b = b; k.c = c; k.dk = dk; k.alpha = alpha;
  hipLaunchKernelGGL(kron_wgrad_kernel, dim3((unsigned)c, (unsigned)cdiv(dk, NTHREADS)), dim3(NTHREADS), 0,
                     (hipStream_t)stream, k);
  return check_launch("lokr_wgrad");
}


// ---- Tucker / conv-CP core fold (tucker.h) ---------------------------------------------------------------------------
int lyc_tucker_core_fwd(const float* t, const float* wb, float* out, int r1, int r2, int64_t Q, int kk, void* stream) {
  if (r1 < 1 || r2 < 1 || Q < 1 || kk < 1 || !t || !wb || !out) return fail(LYC_ERR_ARG, "tucker_core_fwd: bad arguments");
  TuckerArgs a{};
  a.t = t; a.wb = wb; a.B = out; a.r1 = r1; a.r2 = r2; a.Q = Q; a.kk = kk;
  long blocks = cdiv((long)r1 * Q * kk, NTHREADS);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(tucker_core_fwd_kernel, dim3((unsigned)blocks), dim3(NTHREADS), 0, (hipStream_t)stream, a);
  return check_launch("tucker_core_fwd");
}

int lyc_tucker_core_bwd(const float* dout, const float* t, const float* wb, float* d_t, float* d_wb, int r1, int r2, int64_t Q,
                        int kk, void* stream) {
  if (r1 < 1 || r2 < 1 || Q < 1 || kk < 1 || !dout || !t || !wb) return fail(LYC_ERR_ARG, "tucker_core_bwd: bad arguments");
  TuckerArgs a{};
  a.t = t; a.wb = wb; a.dB = dout; a.d_t = d_t; a.d_wb = d_wb; a.r1 = r1; a.r2 = r2; a.Q = Q; a.kk = kk;
  hipStream_t st = (hipStream_t)stream;
  if (d_t) hipLaunchKernelGGL(tucker_core_dt_kernel, dim3((unsigned)(r1 * r2)), dim3(NTHREADS), 0, st, a);
  if (d_wb) {
    long blocks = cdiv((long)r2 * Q, NTHREADS);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tucker_core_dwb_kernel, dim3((unsigned)blocks), dim3(NTHREADS), 0, st, a);
  }
  return check_launch("tucker_core_bwd");
}

// ---- Conv2d lowering -------------------------------------------------------------------------------
extern "C++" {
namespace {
int conv_geom(ConvGeom& cg, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int ph,
              int pw, int dh, int dw) {
  if (B < 0 || C < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1 || ph < 0 || pw < 0 || dh < 1 || dw < 1)
    return fail(LYC_ERR_ARG, "conv lowering: bad geometry");
  cg.B = B; cg.C = C; cg.H = H; cg.W = W; cg.kh = kh; cg.kw = kw; cg.sh = sh; cg.sw = sw; cg.ph = ph; cg.pw = pw;
  cg.dh = dh; cg.dw = dw;
  cg.Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
  cg.Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  if (cg.Ho < 1 || cg.Wo < 1) return fail(LYC_ERR_ARG, "conv lowering: empty output");
  return LYC_OK;
}
unsigned stream_blocks(long total) {
  long b = cdiv(total, (long)NTHREADS * 4);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}
}  // namespace
}  // extern "C++"

int lyc_im2col(const void* x, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh,
               int sw, int ph, int pw, int dh, int dw, int dtype, void* stream) {
  ConvGeom cg{};
  if (int rc = conv_geom(cg, B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw)) return rc;
  if (!x || !cols) return fail(LYC_ERR_ARG, "im2col: null pointer");
  if (B == 0) return LYC_OK;
  const long total = cg.B * cg.Ho * cg.Wo * cg.C * kh * kw;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((im2col_kernel<T>), dim3(stream_blocks(total)), dim3(NTHREADS), 0,
                                           (hipStream_t)stream, static_cast<const T*>(x), static_cast<T*>(cols), cg));
  return check_launch("im2col");
}

int lyc_col2im(const void* dcols, void* dx, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh,
               int sw, int ph, int pw, int dh, int dw, int dtype, void* stream) {
  ConvGeom cg{};
  if (int rc = conv_geom(cg, B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw)) return rc;
  if (!dcols || !dx) return fail(LYC_ERR_ARG, "col2im: null pointer");
  if (B == 0) return LYC_OK;
  const long total = cg.B * cg.C * cg.H * cg.W;
  if (dtype & LYC_F32_ROWS) {
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((col2im_kernel<T, float>), dim3(stream_blocks(total)), dim3(NTHREADS), 0,
                                             (hipStream_t)stream, static_cast<const float*>(dcols), static_cast<T*>(dx), cg));
  } else {
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((col2im_kernel<T, T>), dim3(stream_blocks(total)), dim3(NTHREADS), 0,
                                             (hipStream_t)stream, static_cast<const T*>(dcols), static_cast<T*>(dx), cg));
  }
  return check_launch("col2im");
}

// NHWC row matrices, window-major columns (conv_kernels.h); 16-bit tensors, C % 8 == 0, 16-byte aligned
int lyc_im2col_rows(const void* x_rows, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int ph,
                    int pw, int dh, int dw, int dtype, void* stream) {
  ConvGeom cg{};
  if (int rc = conv_geom(cg, B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw)) return rc;
  if (!x_rows || !cols) return fail(LYC_ERR_ARG, "im2col_rows: null pointer");
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || (C % 8) != 0 || ((reinterpret_cast<uintptr_t>(x_rows) | reinterpret_cast<uintptr_t>(cols)) & 15u))
    return fail(LYC_ERR_UNSUPPORTED, "im2col_rows: 16-bit tensors, C %% 8 == 0, 16-byte aligned");
  if (B == 0) return LYC_OK;
  const long total = cg.B * cg.Ho * cg.Wo * kh * kw * (cg.C / 8);
  if (dt == LYC_BF16) hipLaunchKernelGGL((im2col_rows_kernel<__bf16>), dim3(stream_blocks(total)), dim3(NTHREADS), 0, (hipStream_t)stream,
                                         static_cast<const __bf16*>(x_rows), static_cast<__bf16*>(cols), cg);
  else hipLaunchKernelGGL((im2col_rows_kernel<_Float16>), dim3(stream_blocks(total)), dim3(NTHREADS), 0, (hipStream_t)stream,
                          static_cast<const _Float16*>(x_rows), static_cast<_Float16*>(cols), cg);
  return check_launch("im2col_rows");
}

int lyc_col2im_rows(const void* dcols, void* dx_rows, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int ph,
                    int pw, int dh, int dw, int dtype, void* stream) {
  ConvGeom cg{};
  if (int rc = conv_geom(cg, B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw)) return rc;
  if (!dcols || !dx_rows) return fail(LYC_ERR_ARG, "col2im_rows: null pointer");
  const int dt = dtype & 0xff;
  if ((dt != LYC_BF16 && dt != LYC_F16) || (C % 8) != 0 || ((reinterpret_cast<uintptr_t>(dcols) | reinterpret_cast<uintptr_t>(dx_rows)) & 15u))
    return fail(LYC_ERR_UNSUPPORTED, "col2im_rows: 16-bit tensors, C %% 8 == 0, 16-byte aligned");
  if (B == 0) return LYC_OK;
  const long total = cg.B * cg.H * cg.W * (cg.C / 8);
  const dim3 grid(stream_blocks(total));
  hipStream_t st = (hipStream_t)stream;
  if (dtype & LYC_F32_ROWS) {
    if (dt == LYC_BF16) hipLaunchKernelGGL((col2im_rows_kernel<__bf16, float>), grid, dim3(NTHREADS), 0, st, static_cast<const float*>(dcols), static_cast<__bf16*>(dx_rows), cg);
    else hipLaunchKernelGGL((col2im_rows_kernel<_Float16, float>), grid, dim3(NTHREADS), 0, st, static_cast<const float*>(dcols), static_cast<_Float16*>(dx_rows), cg);
  } else {
    if (dt == LYC_BF16) hipLaunchKernelGGL((col2im_rows_kernel<__bf16, __bf16>), grid, dim3(NTHREADS), 0, st, static_cast<const __bf16*>(dcols), static_cast<__bf16*>(dx_rows), cg);
    else hipLaunchKernelGGL((col2im_rows_kernel<_Float16, _Float16>), grid, dim3(NTHREADS), 0, st, static_cast<const _Float16*>(dcols), static_cast<_Float16*>(dx_rows), cg);
  }
  return check_launch("col2im_rows");
}

extern "C++" {
namespace {
template <bool TO_ROWS>
int launch_nchw_rows(const void* in, void* out, int64_t B, int64_t C, int64_t P, int dtype, void* stream) {
  if (B < 0 || C < 1 || P < 1) return fail(LYC_ERR_ARG, "nchw<->rows: bad dims");
  if (!in || !out) return fail(LYC_ERR_ARG, "nchw<->rows: null pointer");
  if (B == 0) return LYC_OK;
  if (B > 65535 || cdiv(C, 32) > 65535) return fail(LYC_ERR_UNSUPPORTED, "nchw<->rows: grid too large");
  dim3 grid((unsigned)cdiv(P, 32), (unsigned)cdiv(C, 32), (unsigned)B);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nchw_rows_kernel<T, TO_ROWS>), grid, dim3(NTHREADS), 0, (hipStream_t)stream,
                                           static_cast<const T*>(in), static_cast<T*>(out), (long)B, (long)C, (long)P));
  return check_launch("nchw<->rows");
}
}  // namespace
}  // extern "C++"

int lyc_nchw_to_rows(const void* t, void* rows, int64_t B, int64_t C, int64_t P, int dtype, void* stream) {
  return launch_nchw_rows<true>(t, rows, B, C, P, dtype, stream);
}
int lyc_rows_to_nchw(const void* rows, void* t, int64_t B, int64_t C, int64_t P, int dtype, void* stream) {
  return launch_nchw_rows<false>(rows, t, B, C, P, dtype, stream);
}

}  // extern "C"
