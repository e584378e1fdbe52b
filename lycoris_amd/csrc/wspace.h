// wspace.h -- "weight-space" kernels: everything the reference does on the [O, I(,kh,kw)]-shaped dW OFF the activation
// path -- merge_to / get_diff_weight (modules/base.py:326-342, locon.py:221-237), apply_max_norm (locon.py:273-284,
// loha.py:281-292, lokr.py:442-466), DoRA's norm of W + dW and its rescaled weight (apply_weight_decompose,
// locon.py:239-260, loha.py:244-265, lokr.py:399-420) -- with the dW tile REBUILT ON CHIP from the factors:
//
//     v[o, j] = coef[ch(o, j)] * (w_scale * W[o, j] + alpha * dW[o, j])          j = flattened (i, kh, kw)
//
//   * out  != NULL : out[o, j] = v (+ beta * out[o, j])      merge (out = W, beta = 1), diff weight, DoRA weight, Gw
//   * sums != NULL : sums[ch(o, j)] += v^2                    Frobenius norm^2 (one channel) / DoRA row or column norms^2
//
// Nothing [O, J]-sized is read or written except what the mode asks for (W when given, out when given); the reference
// writes dW, W + dW and the normalised weight to HBM and reads them back (3-5 passes).
//
// dW per algorithm (tile 64 x 64, 256 threads, 4 x 4 fp32 values per thread, plain fp32 FMAs -- exact fp32, these
// launches are O * J * r flops, three orders of magnitude below the activation path):
//   LOCON  dW = up[O, r] down[r, J]
//   LOHA   dW = (w1a w1b) * (w2a w2b)
//   LOKR   dW[(p, q), (u, rem)] = w1[p, u] * w2[q, rem]     O = a c, J = b dk, w2 viewed [c, dk = d kh kw]
//   TUCKER variants: see wtucker below (the core tensor is contracted per tap).
#pragma once
#include "tile.h"

namespace lyc {

enum { WS_LOCON = 0, WS_LOHA = 1, WS_LOKR = 2 };
enum { WS_CH_ONE = 0, WS_CH_ROW = 1, WS_CH_COL = 2 };

struct WspaceArgs {
  const float *f0, *f1, *f2, *f3;  // LOCON: down, up | LOHA: w1a, w1b, w2a, w2b | LOKR: w1, w2
  long O, J;
  int R;          // rank (LOCON / LOHA)
  int a, b, c;    // LOKR: w1 [a, b], w2 [c, dk]
  long dk;        // LOKR: columns of the w2 view
  int kk;         // kh * kw: a column channel is j / kk
  const void* W;  // optional base weight [O, J]
  int w_dtype;    // LYC_F32 / LYC_F16 / LYC_BF16
  float w_scale;
  const float* coef;  // optional per-channel coefficient
  int chan_mode;
  void* out;      // optional [O, J]
  int out_dtype;
  float beta;
  float* sums;    // optional
  float alpha;
};

constexpr int WS_T = 64;    // tile edge
constexpr int WS_RC = 32;   // rank chunk staged in LDS

__device__ __forceinline__ float ws_load(const void* p, int dtype, long idx) {
  if (dtype == 0) return static_cast<const float*>(p)[idx];
  if (dtype == 1) return (float)static_cast<const _Float16*>(p)[idx];
  return (float)static_cast<const __bf16*>(p)[idx];
}
__device__ __forceinline__ void ws_store(void* p, int dtype, long idx, float v) {
  if (dtype == 0) static_cast<float*>(p)[idx] = v;
  else if (dtype == 1) static_cast<_Float16*>(p)[idx] = (_Float16)v;
  else static_cast<__bf16*>(p)[idx] = (__bf16)v;
}

template <int ALGO>
__global__ __launch_bounds__(NTHREADS) void wspace_kernel(WspaceArgs a) {
  __shared__ float sA[2][WS_T][WS_RC + 1];  // a-side factor rows  [o][k]   (LOCON: up; LOHA: w1a, w2a)
  __shared__ float sB[2][WS_RC][WS_T + 4];  // b-side factor rows  [k][j]   (LOCON: down; LOHA: w1b, w2b)
  __shared__ float sCol[WS_T];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const long o0 = (long)blockIdx.x * WS_T, j0 = (long)blockIdx.y * WS_T;
  float dw[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[i][j] = 0.f;

  if constexpr (ALGO == WS_LOKR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long o = o0 + ty * 4 + i;
      if (o >= a.O) continue;
      const int p = (int)(o / a.c), q = (int)(o % a.c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long jj = j0 + tx * 4 + j;
        if (jj >= a.J) continue;
        const int u = (int)(jj / a.dk);
        const long rem = jj % a.dk;
        dw[i][j] = a.f0[p * a.b + u] * a.f1[(long)q * a.dk + rem];
      }
    }
  } else {
    constexpr int NP = ALGO == WS_LOHA ? 2 : 1;
    float acc[NP][4][4];
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[n][i][j] = 0.f;
    const float* fa[2] = {ALGO == WS_LOCON ? a.f1 : a.f0, a.f2};  // [O, R]
    const float* fb[2] = {ALGO == WS_LOCON ? a.f0 : a.f1, a.f3};  // [R, J]
    for (int r0 = 0; r0 < a.R; r0 += WS_RC) {
      if (r0) __syncthreads();
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        for (int e = tid; e < WS_T * WS_RC; e += NTHREADS) {
          const int o = e / WS_RC, k = e % WS_RC;
          const bool ok = (o0 + o < a.O) && (r0 + k < a.R);
          sA[n][o][k] = ok ? fa[n][(o0 + o) * a.R + r0 + k] : 0.f;
        }
        for (int e = tid; e < WS_RC * WS_T; e += NTHREADS) {
          const int k = e / WS_T, j = e % WS_T;
          const bool ok = (r0 + k < a.R) && (j0 + j < a.J);
          sB[n][k][j] = ok ? fb[n][(long)(r0 + k) * a.J + j0 + j] : 0.f;
        }
      }
      __syncthreads();
      const int kmax = (a.R - r0) < WS_RC ? (a.R - r0) : WS_RC;
      for (int k = 0; k < kmax; ++k) {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
          float av[4], bv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) av[i] = sA[n][ty * 4 + i][k];
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = sB[n][k][tx * 4 + j];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[n][i][j] = fmaf(av[i], bv[j], acc[n][i][j]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ALGO == WS_LOHA) dw[i][j] = acc[0][i][j] * acc[1][i][j];
        else dw[i][j] = acc[0][i][j];
      }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  float rowsum[4] = {0.f, 0.f, 0.f, 0.f}, colsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long o = o0 + ty * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long jj = j0 + tx * 4 + j;
      if (o >= a.O || jj >= a.J) continue;
      const long idx = o * a.J + jj;
      float v = a.alpha * dw[i][j];
      if (a.W) v += a.w_scale * ws_load(a.W, a.w_dtype, idx);
      if (a.coef) v *= a.coef[a.chan_mode == WS_CH_ROW ? o : (a.chan_mode == WS_CH_COL ? jj / a.kk : 0)];
      if (a.out) {
        float res = v;
        if (a.beta != 0.f) res += a.beta * ws_load(a.out, a.out_dtype, idx);
        ws_store(a.out, a.out_dtype, idx, res);
      }
      rowsum[i] += v * v;
      colsum[j] += v * v;
    }
  }
  if (!a.sums) return;
  if (a.chan_mode == WS_CH_ROW) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = rowsum[i];
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);  // the 16 tx lanes
      const long o = o0 + ty * 4 + i;
      if (tx == 0 && o < a.O) __hip_atomic_fetch_add(a.sums + o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (tid < WS_T) sCol[tid] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(&sCol[tx * 4 + j], colsum[j]);
  __syncthreads();
  if (a.chan_mode == WS_CH_COL) {
    const long jj = j0 + tid;
    if (tid < WS_T && jj < a.J)
      __hip_atomic_fetch_add(a.sums + jj / a.kk, sCol[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (tid == 0) {
    float s = 0.f;
    for (int j = 0; j < WS_T; ++j) s += sCol[j];
    __hip_atomic_fetch_add(a.sums, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- gradient of a weight-space quantity w.r.t. the LoKr factors ----------------------------------------------------
// Gw: [O, J] fp32 (O = a c, J = b dk).  d_w1[p, u] += alpha sum_{q, rem} Gw[(p,q), (u,rem)] w2[q, rem]
//                                       d_w2[q, rem] += alpha sum_{p, u} w1[p, u] Gw[(p,q), (u,rem)]
// (the reference: torch.kron backward on the dense gradient).  One workgroup per (q, 256-wide rem slab): Gw is read once.
struct KronWgradArgs {
  const float *gw, *w1, *w2;
  float *d_w1, *d_w2;
  int a, b, c;
  long dk;
  float alpha;
};

__global__ __launch_bounds__(NTHREADS) void kron_wgrad_kernel(KronWgradArgs k) {
  __shared__ float red[NWAVES];
  const int q = blockIdx.x;
  const long rem = (long)blockIdx.y * NTHREADS + threadIdx.x;
  const bool ok = rem < k.dk;
  const long J = (long)k.b * k.dk;
  const float w2v = ok ? k.w2[(long)q * k.dk + rem] : 0.f;
  float acc2 = 0.f;
  for (int p = 0; p < k.a; ++p)
    for (int u = 0; u < k.b; ++u) {
      const float gv = ok ? k.gw[((long)p * k.c + q) * J + (long)u * k.dk + rem] : 0.f;
      acc2 = fmaf(k.w1[p * k.b + u], gv, acc2);
      float s = gv * w2v;  // -> d_w1[p, u]: reduce over the workgroup
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0 && k.d_w1)
        __hip_atomic_fetch_add(k.d_w1 + p * k.b + u, k.alpha * (red[0] + red[1] + red[2] + red[3]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
  if (ok && k.d_w2) k.d_w2[(long)q * k.dk + rem] += k.alpha * acc2;  // each (q, rem) belongs to exactly one thread
}

}  // namespace lyc
