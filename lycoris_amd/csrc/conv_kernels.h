// conv_kernels.h -- Conv2d lowering helpers (NCHW), gfx950.  HBM-bound gather / transpose kernels.
//
// v1 of the Conv2d path evaluates every adapter as its Linear kernel on the im2col view:
//   cols[(b, oh, ow), c*kh*kw + i*kw + j] = x[b, c, oh*sh - ph + i*dh, ow*sw - pw + j*dw]      (0 outside)
// which is exactly how the reference lays out its non-Tucker conv factors ([r, I*kh*kw], modules/loha.py:76,
// modules/lokr.py:131-136, and locon.py:198-219 via .view(r, -1)); for LoKr the channel index c = u*d + v makes
// the plain im2col column order the grouped (u, v, kh, kw) order the Kronecker kernel expects.
//   im2col_kernel        x[B,C,H,W]          -> cols[B*Ho*Wo, C*kh*kw]
//   col2im_kernel        dcols               -> dx[B,C,H,W]      (gather form: no atomics, fp32 accumulate)
//   nchw_to_rows_kernel  t[B,C,P]            -> rows[B*P, C]
//   rows_to_nchw_kernel  rows[B*P, C]        -> t[B,C,P]
#pragma once
#include "tile.h"

namespace lyc {

struct ConvGeom {
  long B, C, H, W, Ho, Wo;
  int kh, kw, sh, sw, ph, pw, dh, dw;
};

template <typename T>
__global__ __launch_bounds__(NTHREADS) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, ConvGeom g) {
  const long KK = (long)g.C * g.kh * g.kw;
  const long total = g.B * g.Ho * g.Wo * KK;
  const long stride = (long)gridDim.x * NTHREADS;
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += stride) {
    const long m = e / KK, col = e % KK;
    const int j = (int)(col % g.kw), i = (int)((col / g.kw) % g.kh);
    const long c = col / (g.kw * g.kh);
    const long ow = m % g.Wo, oh = (m / g.Wo) % g.Ho, b = m / (g.Wo * g.Ho);
    const long ih = oh * g.sh - g.ph + (long)i * g.dh, iw = ow * g.sw - g.pw + (long)j * g.dw;
    T v = TT<T>::from_f(0.f);
    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = x[((b * g.C + c) * g.H + ih) * g.W + iw];
    cols[e] = v;
  }
}

template <typename T, typename RT>
__global__ __launch_bounds__(NTHREADS) void col2im_kernel(const RT* __restrict__ dcols, T* __restrict__ dx, ConvGeom g) {
  const long KK = (long)g.C * g.kh * g.kw;
  const long total = g.B * g.C * g.H * g.W;
  const long stride = (long)gridDim.x * NTHREADS;
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += stride) {
    const long iw = e % g.W, ih = (e / g.W) % g.H, c = (e / (g.W * g.H)) % g.C, b = e / (g.W * g.H * g.C);
    float s = 0.f;
    for (int i = 0; i < g.kh; ++i) {
      const long th = ih + g.ph - (long)i * g.dh;
      if (th < 0 || th % g.sh) continue;
      const long oh = th / g.sh;
      if (oh >= g.Ho) continue;
      for (int j = 0; j < g.kw; ++j) {
        const long tw = iw + g.pw - (long)j * g.dw;
        if (tw < 0 || tw % g.sw) continue;
        const long ow = tw / g.sw;
        if (ow >= g.Wo) continue;
        s += (float)dcols[((b * g.Ho + oh) * g.Wo + ow) * KK + (c * g.kh + i) * g.kw + j];
      }
    }
    dx[e] = TT<T>::from_f(s);
  }
}

// 32x32 LDS tile transpose between [B, C, P] and [B*P, C]
template <typename T, bool TO_ROWS>
__global__ __launch_bounds__(NTHREADS) void nchw_rows_kernel(const T* __restrict__ in, T* __restrict__ out, long B,
                                                             long C, long P) {
  __shared__ T tile[32][33];
  const long b = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32, c0 = (long)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_ROWS) {
    for (int r = ty; r < 32; r += 8) {  // read in[b, c0 + r, p0 + tx]
      const long c = c0 + r, p = p0 + tx;
      tile[r][tx] = (c < C && p < P) ? in[(b * C + c) * P + p] : TT<T>::from_f(0.f);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {  // write out[(b, p0 + r), c0 + tx]
      const long p = p0 + r, c = c0 + tx;
      if (p < P && c < C) out[(b * P + p) * C + c] = tile[tx][r];
    }
  } else {
    for (int r = ty; r < 32; r += 8) {  // read in[(b, p0 + r), c0 + tx]
      const long p = p0 + r, c = c0 + tx;
      tile[r][tx] = (p < P && c < C) ? in[(b * P + p) * C + c] : TT<T>::from_f(0.f);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {  // write out[b, c0 + r, p0 + tx]
      const long c = c0 + r, p = p0 + tx;
      if (c < C && p < P) out[(b * C + c) * P + p] = tile[tx][r];
    }
  }
}

}  // namespace lyc
