// tucker.h -- the core contraction of the Tucker / conv-CP forms (reference: rebuild_tucker, functional/general.py:9-11;
// HadaWeightTucker, functional/loha.py:33-75; lora_mid, modules/locon.py:85-90; lokr_t2, modules/lokr.py:121-128).
//
//   W[p, q, k] = sum_ij t[i, j, k] wa[i, p] wb[j, q]  =  sum_i wa[i, p] * B[i, (q, k)],   B = core(t, wb)
//
// i.e. every Tucker form is the NON-Tucker form of the same algorithm with the k x k core folded into the input-side
// factor: B[i, q, k] = sum_j t[i, j, k] wb[j, q]  ([r1, Q * kk], the layout of a flattened conv factor [r, I, kh, kw]).
// The adapter kernels then run unchanged on (wa^T, B).  These three kernels are that fold and its gradients; they are
// O(r^2 Q kk) flops on tensors of a few hundred KB (one thread per output element, no atomics).
#pragma once
#include "tile.h"

namespace lyc {

struct TuckerArgs {
  const float *t, *wb, *dB;  // t [r1, r2, kk], wb [r2, Q], dB [r1, Q, kk]
  float *B, *d_t, *d_wb;
  int r1, r2, kk;
  long Q;
};

__global__ __launch_bounds__(NTHREADS) void tucker_core_fwd_kernel(TuckerArgs a) {
  const long total = (long)a.r1 * a.Q * a.kk;
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += (long)gridDim.x * NTHREADS) {
    const int k = (int)(e % a.kk);
    const long q = (e / a.kk) % a.Q;
    const int i = (int)(e / ((long)a.kk * a.Q));
    float s = 0.f;
    for (int j = 0; j < a.r2; ++j) s = fmaf(a.t[((long)i * a.r2 + j) * a.kk + k], a.wb[(long)j * a.Q + q], s);
    a.B[e] = s;
  }
}

// d_t[i, j, k] = sum_q dB[i, q, k] wb[j, q]      one workgroup per (i, j), the kk taps x Q split over its threads
__global__ __launch_bounds__(NTHREADS) void tucker_core_dt_kernel(TuckerArgs a) {
  __shared__ float red[NTHREADS];
  const int i = blockIdx.x / a.r2, j = blockIdx.x % a.r2;
  for (int k = 0; k < a.kk; ++k) {
    float s = 0.f;
    for (long q = threadIdx.x; q < a.Q; q += NTHREADS) s = fmaf(a.dB[((long)i * a.Q + q) * a.kk + k], a.wb[(long)j * a.Q + q], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = NTHREADS / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.d_t[((long)i * a.r2 + j) * a.kk + k] = red[0];
    __syncthreads();
  }
}

// d_wb[j, q] = sum_{i, k} t[i, j, k] dB[i, q, k]
__global__ __launch_bounds__(NTHREADS) void tucker_core_dwb_kernel(TuckerArgs a) {
  const long total = (long)a.r2 * a.Q;
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += (long)gridDim.x * NTHREADS) {
    const long q = e % a.Q;
    const int j = (int)(e / a.Q);
    float s = 0.f;
    for (int i = 0; i < a.r1; ++i)
      for (int k = 0; k < a.kk; ++k) s = fmaf(a.t[((long)i * a.r2 + j) * a.kk + k], a.dB[((long)i * a.Q + q) * a.kk + k], s);
    a.d_wb[e] = s;
  }
}

}  // namespace lyc
