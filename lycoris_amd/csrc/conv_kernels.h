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

// ---- round 6: the same lowering on NHWC ROW matrices, window-major columns ----------------------------------------------------------
//   cols[(b, oh, ow), (i*kw + j)*C + c] = x_rows[(b, ih, iw), c]      (0 outside the image)
// A `channels_last` tensor IS its row matrix, a column block of one tap is a contiguous run of channels: every access is a 16-byte vector
// of 8 channels, fully coalesced, one index computation per vector (the NCHW kernels above compute three divisions per ELEMENT and read
// with a stride of H*W: 183 us per SDXL conv layer for 26 MB, profiles/r06_c9_loha_kernel_stats.csv).  The factors that meet these
// columns are the reference's [r, C*kh*kw] ones with their columns permuted to window-major order (csrc/torch_ops.cpp).  C % 8 == 0.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void im2col_rows_kernel(const T* __restrict__ x, T* __restrict__ cols, ConvGeom g) {
  const long CV = g.C / 8, taps = (long)g.kh * g.kw;
  const long total = g.B * g.Ho * g.Wo * taps * CV;
  const long stride = (long)gridDim.x * NTHREADS;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += stride) {
    const long cv = e % CV, t = (e / CV) % taps, m = e / (CV * taps);
    const int j = (int)(t % g.kw), i = (int)(t / g.kw);
    const long ow = m % g.Wo, oh = (m / g.Wo) % g.Ho, b = m / (g.Wo * g.Ho);
    const long ih = oh * g.sh - g.ph + (long)i * g.dh, iw = ow * g.sw - g.pw + (long)j * g.dw;
    u32x4 v = zero;
    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = *reinterpret_cast<const u32x4*>(x + ((b * g.H + ih) * g.W + iw) * g.C + cv * 8);
    *reinterpret_cast<u32x4*>(cols + e * 8) = v;
  }
}

// dx_rows[(b, ih, iw), c] = sum over the taps that reach the pixel of dcols[(b, oh, ow), tap*C + c]: gather form, fp32 accumulation, one
// rounding; RT = float (the GEMM emits un-rounded rows) or T
template <typename T, typename RT>
__global__ __launch_bounds__(NTHREADS) void col2im_rows_kernel(const RT* __restrict__ dcols, T* __restrict__ dx, ConvGeom g) {
  const long CV = g.C / 8, KK = (long)g.C * g.kh * g.kw;
  const long total = g.B * g.H * g.W * CV;
  const long stride = (long)gridDim.x * NTHREADS;
  for (long e = (long)blockIdx.x * NTHREADS + threadIdx.x; e < total; e += stride) {
    const long cv = e % CV, pix = e / CV;
    const long iw = pix % g.W, ih = (pix / g.W) % g.H, b = pix / (g.W * g.H);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        const RT* src = dcols + ((b * g.Ho + oh) * g.Wo + ow) * KK + (long)(i * g.kw + j) * g.C + cv * 8;
        if constexpr (sizeof(RT) == 4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s[q] += a[q];
            s[4 + q] += c[q];
          }
        } else {
          RT v[8];
          *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
          for (int q = 0; q < 8; ++q) s[q] += (float)v[q];
        }
      }
    }
    T o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = TT<T>::from_f(s[q]);
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(dx + e * 8) = *reinterpret_cast<const u32x4*>(o);
    } else {
      *reinterpret_cast<f32x4*>(dx + e * 8) = *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(dx + e * 8 + 4) = *reinterpret_cast<const f32x4*>(o + 4);
    }
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
