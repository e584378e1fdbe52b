// kron_conv_dw2.h -- LoKr on nn.Conv2d, the w2 gradient from LDS patches (gfx950).  Round 3.
//
//   dW2[q, tap, v] += alpha * sum_{pix, u} Qm[(pix, u), q] * x[(src(pix, tap), u), v],   Qm[(pix, u), q] = sum_p w1[p, u] g[(pix, p), q]
//
// The row-gather kernel (kron_dw2s.h, GATHER) streams 32-row steps through per-wave transposes: every tap is a different
// gather of x (9 L2 reads of the tensor for a 3x3 window), g is re-read once per 64-column tile of the (tap, v) axis (6 times
// for a 320-channel layer) and the w1 mix is recomputed per tap: 15 tensor-sized L2 reads per layer, instruction-bound.
// Here a workgroup walks destination pixel tiles (the 128 rows of kron_conv.h's MI = 2 tile) and per tile
//   1. pulls the source patch of x (its v window) and the g tile (its q window) into LDS by LDS-DMA: each pixel once;
//   2. turns them into matrix-core operands whose K dimension is the ROW index (pixels x groups) -- the transposition a
//      "TN" product needs -- with the matrix core itself: a 16 x 16 tile read K-contiguous along the channels, multiplied by
//      the identity, comes back in the accumulator layout, which IS the operand layout with rows as K (exact: the values
//      are 16-bit).  x tiles are "vertical runs" of 16 / G pixels, one per patch pixel as its top, so that every tap shift
//      maps a destination run to another run of the same set; g runs are mixed with (I (x) w1^T) on the way (hi/lo, once per
//      tile instead of once per tap);
//   3. runs 2 MFMAs (Qm hi, lo) per (destination run, tap, 16 x 16 output block); the output blocks of all taps are spread
//      over the four waves and stay in registers across the workgroup's slab of pixel tiles; one atomic pass at the end.
// The DMA of the next tile is issued as soon as step 2 has consumed the raw images, i.e. it runs under step 3.
// Output block [v][q] (dW2 transposed): lane (q = li, v = 4g + r).
#pragma once
#include "kron_conv.h"
#include "kron_dw2s.h"

namespace lyc {

constexpr int KD_MAXC = 21;   // output blocks per wave: taps * NVB * NQB <= 4 * KD_MAXC
constexpr int KD_WIN = 48;    // widest q / v window of a workgroup (3 blocks of 16)

struct KdItem {
  const void* g;        // [B * Ho * Wo, G * I] destination rows (upstream gradient)
  const void* x;        // [B * H * W,  G * J]  source rows (layer input)
  const float* w1;      // element (p, u) at p * ws + u * wt
  float* out;           // dW2p: element (q, tap, v) at q * os + tap * Jd + v
  const float* dw1_ws;  // dw1 partials of the dx launch (nullptr: none)
  float* dw1;
  int B, G, I, J;       // I = c (q extent), J = d (v extent per tap)
  int Hs, Ws, Hd, Wd;   // source (input) / destination (output) image
  int taps, kw, sh, sw, ph, pw, dh, dw;
  int TH, TW, PH, PW;   // destination pixel tile, source patch
  int tiles_h, tiles_w; // tiles per image
  int KQ, KV;           // q / v window widths (multiples of 8, <= KD_WIN)
  int nq, nv;           // windows
  int slabs, tiles_per_slab;
  int ws, wt, os;
  int dw1_nblk, dw1_n, dw1_red;
  int force_atomic;
  float alpha;
};
constexpr int KD_MAX_ITEMS = 12;
struct KdGroupArgs {
  int n;
  int wg_end[KD_MAX_ITEMS];
  KdItem p[KD_MAX_ITEMS];
};
static_assert(sizeof(KdGroupArgs) <= 3584, "kernel arguments are limited to 4 KiB");

__host__ __device__ inline int kd_pitch(int win) { return win + (((win / 8) % 2 == 0) ? 8 : 0); }  // odd number of 16-byte slots
struct KdLds {
  int x_bytes, g_bytes, xt_bytes, qm_bytes;
  int total() const { return x_bytes + g_bytes + xt_bytes + qm_bytes + 1024; }
};
__host__ __device__ inline KdLds kd_lds(const KdItem& it) {
  KdLds l;
  const int pv = 16 / it.G;
  const int nvb = (it.KV + 15) / 16, nqb = (it.KQ + 15) / 16;
  const int ntop = (it.PH - (pv - 1) * it.sh) * it.PW;
  l.x_bytes = ((it.PH * it.PW * it.G * kd_pitch(it.KV) * 2 + 64 + 1023) / 1024) * 1024;  // + slack: block reads run 16 B past a segment
  l.g_bytes = ((it.TH * it.TW * it.G * kd_pitch(it.KQ) * 2 + 64 + 1023) / 1024) * 1024;
  l.xt_bytes = ntop * nvb * 512;
  l.qm_bytes = 8 * nqb * 1024;
  return l;
}

// LDS-DMA of a pixel window of an NHWC image: pixels [h0, h0 + nh) x [w0, w0 + nw) (zeros outside the image), per pixel G groups
// of Kt channels of which the window [c0, c0 + cw) is taken, into dst[(pixel)][group][pitch] (16-bit elements).
template <typename T>
__device__ __forceinline__ void kd_stage_window(char* lds_base, int dst_bytes, const T* img_base, long img_elems, int Hi, int Wi, int G,
                                                int Kt, int c0, int cw, int pitch, int h0, int w0, int nh, int nw, int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(img_base), 0, (int)(img_elems * (long)sizeof(T)), 0x00020000);
  const int kv = cw / 8, spg = pitch / 8, spp = G * spg, npix = nh * nw;
  const int vpp = (G * Kt) / 8, kvt = Kt / 8, c0v = c0 / 8;
  const float inv_spp = 1.0f / (float)spp, inv_spg = 1.0f / (float)spg, inv_nw = 1.0f / (float)nw;
  const int OOR = 0x7ffffff0;
  const int npiece = dst_bytes >> 10;
  for (int pc = wave; pc < npiece; pc += NWAVES) {
    const int sl = pc * 64 + lane;
    int pp = (int)(((float)sl + 0.5f) * inv_spp);
    int q = sl - pp * spp;
    if (q < 0) { q += spp; --pp; }
    if (q >= spp) { q -= spp; ++pp; }
    int u = (int)(((float)q + 0.5f) * inv_spg);
    int kq = q - u * spg;
    if (kq < 0) { kq += spg; --u; }
    if (kq >= spg) { kq -= spg; ++u; }
    int py = (int)(((float)pp + 0.5f) * inv_nw);
    int px = pp - py * nw;
    if (px < 0) { px += nw; --py; }
    if (px >= nw) { px -= nw; ++py; }
    const int hs = h0 + py, ws = w0 + px;
    const bool ok = pp < npix && kq < kv && (c0v + kq) < kvt && hs >= 0 && hs < Hi && ws >= 0 && ws < Wi;
    const int off = ok ? ((hs * Wi + ws) * vpp + u * kvt + c0v + kq) * 16 : OOR;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_base + pc * 1024), 16, off, 0, 0, 0);
  }
}

template <typename T>
__device__ __forceinline__ void kconv_dw2_body(const KdItem& it, char* smem, int b_) {
  using F4 = typename Mma16<T>::frag;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int G = it.G;
  const int lg = 31 - __builtin_clz((unsigned)G);
  const int pv = 16 >> lg;  // pixels of a 16-row run
  const int nwork = it.slabs * it.nq * it.nv;
  if (b_ >= nwork) {  // dw1 reducer workgroups ride at the end of the item's range
    const int r = b_ - nwork;
    if (it.dw1_ws != nullptr && r < it.dw1_red) {
      KronDw2sArgs a{};
      a.dw1_ws = it.dw1_ws; a.dw1 = it.dw1; a.dw1_nblk = it.dw1_nblk; a.dw1_n = it.dw1_n; a.dw1_red = it.dw1_red;
      a.force_atomic = it.force_atomic;
      dw1_reduce_role(a, r, reinterpret_cast<float*>(smem));
    }
    return;
  }
  // work item -> (slab, q window, v window); neighbours share the slab (the same g / x pixels in L2)
  const int slab = b_ / (it.nq * it.nv);
  const int rem = b_ - slab * (it.nq * it.nv);
  const int jq = rem / it.nv, jv = rem - jq * it.nv;
  const int q0 = jq * it.KQ, v0 = jv * it.KV;
  const int kq_w = (it.I - q0) < it.KQ ? (it.I - q0) : it.KQ;  // valid widths of this window
  const int kv_w = (it.J - v0) < it.KV ? (it.J - v0) : it.KV;
  const int nqb = (it.KQ + 15) / 16, nvb = (it.KV + 15) / 16;
  const int ncombo = it.taps * nvb * nqb;
  const KdLds L = kd_lds(it);
  char* RX = smem;
  char* RG = RX + L.x_bytes;
  char* RXT = RG + L.g_bytes;
  char* RQM = RXT + L.xt_bytes;
  const int gpx = kd_pitch(it.KV), gpq = kd_pitch(it.KQ);
  const int ntile_total = it.B * it.tiles_h * it.tiles_w;
  int t_lo = slab * it.tiles_per_slab, t_hi = t_lo + it.tiles_per_slab;
  if (t_hi > ntile_total) t_hi = ntile_total;

  // constant operands: the identity (B operand) and (I (x) w1^T) hi / lo (A operand: i = (t, u), k = (t', p))
  F4 ident, wbh, wbl;
  {
    T idv[4], h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      idv[e] = TT<T>::from_f((4 * g + e) == li ? 1.f : 0.f);
      const int k = 4 * g + e;
      const float w = it.w1[(k & (G - 1)) * it.ws + (li & (G - 1)) * it.wt];  // w1[p = k % G][u = li % G]
      split_f<T>(((k >> lg) == (li >> lg)) ? w : 0.f, h[e], l[e]);
    }
    ident = *reinterpret_cast<F4*>(idv);
    wbh = *reinterpret_cast<F4*>(h);
    wbl = *reinterpret_cast<F4*>(l);
  }

  f32x4 acc[KD_MAXC];
#pragma unroll
  for (int c = 0; c < KD_MAXC; ++c) acc[c] = zero4();

  const T* gx = static_cast<const T*>(it.x);
  const T* gg = static_cast<const T*>(it.g);
  const long x_img = (long)it.Hs * it.Ws * G * it.J, g_img = (long)it.Hd * it.Wd * G * it.I;
  const int tpi = it.tiles_h * it.tiles_w;
  auto issue_tile = [&](int t) {
    const int img = t / tpi, trem = t - img * tpi;
    const int th = trem / it.tiles_w, tw = trem - th * it.tiles_w;
    const int hd0 = th * it.TH, wd0 = tw * it.TW;
    kd_stage_window<T>(RX, L.x_bytes, gx + (long)img * x_img, x_img, it.Hs, it.Ws, G, it.J, v0, it.KV, gpx, hd0 * it.sh - it.ph,
                       wd0 * it.sw - it.pw, it.PH, it.PW, wave, lane);
    kd_stage_window<T>(RG, L.g_bytes, gg + (long)img * g_img, g_img, it.Hd, it.Wd, G, it.I, q0, it.KQ, gpq, hd0, wd0, it.TH, it.TW, wave,
                       lane);
  };
  if (t_lo < t_hi) issue_tile(t_lo);

  const int ntop = (it.PH - (pv - 1) * it.sh) * it.PW;
  const int kh = it.taps / it.kw;
  (void)kh;
  for (int t = t_lo; t < t_hi; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // raw images of tile t are in LDS; everybody is done with the operand images of tile t - 1
    // ---- step 2: operand images --------------------------------------------------------------------------------------------
    // x runs: job = (top pixel, v block); rows of the run: r = (tt, u), pixel top + tt * sh * PW
    for (int job = wave; job < ntop * nvb; job += NWAVES) {
      const int top = job / nvb, vb = job - top * nvb;
      const int tt = li >> lg, u = li & (G - 1);
      const F4 af = *reinterpret_cast<const F4*>(RX + (((top + tt * it.sh * it.PW) * G + u) * gpx + 16 * vb + 4 * g) * 2);
      const f32x4 d = Mma16<T>::mma(af, ident, zero4());
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(d[e]);
      *reinterpret_cast<u32x2*>(RXT + job * 512 + lane * 8) = *reinterpret_cast<u32x2*>(o);
    }
    // g runs: job = (destination run, q block); run td: pixels (ly0 + tt, lx), ly0 = (td / TW) * pv, lx = td % TW
    for (int job = wave; job < 8 * nqb; job += NWAVES) {
      const int td = job / nqb, qb = job - td * nqb;
      const int ly0 = (td / it.TW) * pv, lx = td - (td / it.TW) * it.TW;
      const int tt = li >> lg, p = li & (G - 1);
      const F4 af = *reinterpret_cast<const F4*>(RG + ((((ly0 + tt) * it.TW + lx) * G + p) * gpq + 16 * qb + 4 * g) * 2);
      const f32x4 d = Mma16<T>::mma(af, ident, zero4());  // lane (q = li, rows (tt, p) = 4g + r): the B operand layout
      T o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = TT<T>::from_f(d[e]);
      const F4 gt = *reinterpret_cast<F4*>(o);
      f32x4 qm = Mma16<T>::mma(wbh, gt, zero4());
      qm = Mma16<T>::mma(wbl, gt, qm);  // Qm[(tt, u) = 4g + r][q = li]
      T h[4], l[4];
      k3_split4<T>(qm, h, l);
      *reinterpret_cast<u32x2*>(RQM + (job * 2 + 0) * 512 + lane * 8) = *reinterpret_cast<u32x2*>(h);
      *reinterpret_cast<u32x2*>(RQM + (job * 2 + 1) * 512 + lane * 8) = *reinterpret_cast<u32x2*>(l);
    }
    __syncthreads();  // operand images complete; the raw images are free again
    if (t + 1 < t_hi) issue_tile(t + 1);  // runs under step 3
    // ---- step 3: output blocks of this wave -----------------------------------------------------------------------------------
#pragma unroll
    for (int ci = 0; ci < KD_MAXC; ++ci) {
      const int c = wave + NWAVES * ci;
      if (c < ncombo) {
        const int tap = c / (nvb * nqb);
        const int r2 = c - tap * (nvb * nqb);
        const int vb = r2 / nqb, qb = r2 - vb * nqb;
        const int ti = tap / it.kw, tj = tap - ti * it.kw;
        const int tapoff = ti * it.dh * it.PW + tj * it.dw;
        f32x4 a = acc[ci];
#pragma unroll
        for (int td = 0; td < 8; ++td) {
          const int ly0 = (td / it.TW) * pv, lx = td - (td / it.TW) * it.TW;
          const int top = (ly0 * it.sh) * it.PW + lx * it.sw + tapoff;
          const F4 xa = *reinterpret_cast<const F4*>(RXT + (top * nvb + vb) * 512 + lane * 8);
          const F4 qh = *reinterpret_cast<const F4*>(RQM + ((td * nqb + qb) * 2 + 0) * 512 + lane * 8);
          const F4 ql = *reinterpret_cast<const F4*>(RQM + ((td * nqb + qb) * 2 + 1) * 512 + lane * 8);
          a = Mma16<T>::mma(xa, qh, a);
          a = Mma16<T>::mma(xa, ql, a);
        }
        acc[ci] = a;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- output: block [v = 16 vb + 4g + r][q = 16 qb + li] of tap -> out[q * os + tap * J + v] -----------------------------------
  const bool plain = it.slabs == 1 && !it.force_atomic;
#pragma unroll
  for (int ci = 0; ci < KD_MAXC; ++ci) {
    const int c = wave + NWAVES * ci;
    if (c < ncombo) {
      const int tap = c / (nvb * nqb);
      const int r2 = c - tap * (nvb * nqb);
      const int vb = r2 / nqb, qb = r2 - vb * nqb;
      const int ql = 16 * qb + li;
      if (ql < kq_w) {
        float* dst = it.out + (long)(q0 + ql) * it.os + (long)tap * it.J + v0 + 16 * vb + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (16 * vb + 4 * g + r < kv_w) {
            if (plain) dst[r] += it.alpha * acc[ci][r];
            else __hip_atomic_fetch_add(dst + r, it.alpha * acc[ci][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void kconv_dw2_group_kernel(KdGroupArgs ga) {
  extern __shared__ __attribute__((aligned(1024))) char kd_smem[];
  const int b = (int)blockIdx.x;
  int p = 0;
  while (p + 1 < ga.n && b >= ga.wg_end[p]) ++p;
  const int b0 = p ? ga.wg_end[p - 1] : 0;
  kconv_dw2_body<T>(ga.p[p], kd_smem, b - b0);
}

}  // namespace lyc
