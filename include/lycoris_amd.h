/* lycoris_amd.h -- C ABI of the MI355X (gfx950) adapter hot path.
 *
 * Drop-in boundary for the LyCORIS adapter forward/backward.  The
 * reference is pure Python over ATen; each entry point below replaces the ATen call sequence of one
 * reference function, named in the comment above it (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM); activations are row-major and contiguous;
 *   - `dtype` selects the activation element type (x, g, y, dx, base ...): LYC_F32 / LYC_F16 / LYC_BF16;
 *   - adapter factors and their gradients are always fp32 (callers with 16-bit factors convert the
 *     small tensors); arithmetic accumulates in fp32; fp32 factors are fed to the matrix cores as
 *     hi + lo pairs so the only rounding is the final store of a `dtype` output;
 *   - gradient outputs marked "+=" are accumulated with fp32 atomics: the caller zero-fills them
 *     (or passes a slice of its gradient arena); buffers marked "scratch, zeroed" likewise;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work: no allocation, no
 *     synchronisation, no host reads, so they are legal inside hipGraph capture;
 *   - return value 0 = success; otherwise an LYC_ERR_* code and lyc_last_error() (thread-local) says why.
 *     Nothing is enqueued when an argument check fails.
 */
#ifndef LYCORIS_AMD_H
#define LYCORIS_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LYC_F32 = 0, LYC_F16 = 1, LYC_BF16 = 2 };
/* OR-able into `dtype`: the row-matrix side of the call (dx of the *_linear_bwd entry points, dcols of lyc_col2im)
 * is fp32 instead of `dtype`.  Used by the Conv2d lowering so that col2im sums un-rounded rows and rounds once. */
enum { LYC_F32_ROWS = 0x100, LYC_DEFER_WGRAD = 0x200,
       /* lyc_lokr_wgrad_group only: use the round 1-3 tile plan (80 x 32 outputs per wave, rows split over the waves: kron_dw2s.h) for
        * every item instead of the full-width tiles of kron_dw2f.h -- the A/B and regression-test switch (an argument, not getenv) */
       LYC_WGRAD_TILE_S = 0x400,
       /* lyc_lokr_conv_wgrad_group, experiment builds (-DLYC_EXPERIMENT_CONV_DW2_PATCH) only: the LDS-patch weight-gradient kernel of
        * benchmarks/experiments/kron_conv_dw2.h where its plan covers the layer; ignored by the product library */
       LYC_CONV_WGRAD_PATCH = 0x800,
       /* lyc_loha_linear_fwd only (round 6): `wplanes` already holds the operand plane of these factors (lyc_loha_rebuild_group: the
        * caller's once-per-optimizer-step plane cache) -- the entry point skips its rebuild launch */
       LYC_PLANE_READY = 0x10000,
       /* the rank-r (LoCon) entry points, round 6: keep the reduce / expand launch on the register-staged kernel of rounds 1-5
        * (bneck_kernel, lowrank.h) instead of the LDS-DMA kernel (bneck4_kernel, lowrank4.h) where both cover the problem -- the
        * A/B and regression-test switch (an argument, not getenv) */
       LYC_BNECK_REG = 0x20000,
       /* the *_planes Conv2d entry points of LoKr, round 6: run the patch kernel with the 4-wave workgroups of rounds 3 - 5 instead of
        * 8 waves (two per SIMD) -- the A/B and regression-test switch */
       LYC_KCONV_W4 = 0x40000,
       /* same entry points, round 6: keep the k loop of the 8-wave patch kernel in its serial form (one basic block per k step) instead of
        * the software-pipelined loop (next k step's fragments read before this one's MFMAs) -- the A/B and regression-test switch */
       LYC_KCONV_SERIAL = 0x80000 };
/* The *_planes Conv2d entry points and lyc_lokr_conv2d_planes_ok / _dx_blocks: pin the patch kernel's row tile (mi = 2, 4 or 8: 64 * mi
 * stage-1 rows per workgroup) instead of letting the host plan it -- for tests, which otherwise reach only the smallest tile with
 * their small problems (rounds 2-3 read an environment variable for this). */
#define LYC_KCONV_ROW_TILE(mi) (((mi) & 0xf) << 12)
enum { LYC_OK = 0, LYC_ERR_ARG = 1, LYC_ERR_UNSUPPORTED = 2, LYC_ERR_LAUNCH = 3 };

#define LYC_ABI_VERSION 13
int lyc_abi_version(void);
const char* lyc_last_error(void);

/* ---- LoKr on nn.Linear -------------------------------------------------------------------------
 * replaces lycoris/modules/lokr.py:543-566 (forward: make_kron + dense F.linear) and the factored
 * bypass lycoris/functional/lokr.py:154-247 / modules/lokr.py:468-538.
 *   w1:[a,b]  w2:[c,d]  x:[M, b*d]  y:[M, a*c]
 *   y[m, p*c+q] = alpha * sum_u w1[p,u] * sum_v w2[q,v] * x[m, u*d+v]          (== x @ (kron(w1,w2)*alpha)^T)
 * bwd: dx = g @ (kron(w1,w2)*alpha);  dw1 += , dw2 +=  (gradients of sum(g*y) w.r.t. w1, w2)
 * `base`: optional [M, a*c] output of the frozen layer (same dtype): with it y = base + delta is formed in the kernel's
 *       epilogue (fp32 add, one rounding) -- the reference's `return base + delta` (modules/lokr.py:566) without the extra
 *       elementwise launch and its 3 * M * O elements of traffic.  16-bit fast path only (LYC_ERR_UNSUPPORTED otherwise).
 * `ws`: optional device scratch of lyc_lokr_bwd_workspace_bytes(...) bytes (no need to clear).  With it the w1
 *       gradient is reduced from per-workgroup partials in a fixed order (deterministic, and ~10 us faster per call
 *       than the same-address fp32 atomics used when ws == NULL).                                          */
int lyc_lokr_linear_fwd(const void* x, const float* w1, const float* w2, const void* base, void* y, int64_t M, int a, int b,
                        int c, int d, float alpha, int dtype, void* stream);
int64_t lyc_lokr_bwd_workspace_bytes(int64_t M, int a, int b, int c, int d, int dtype);
int lyc_lokr_linear_bwd(const void* g, const void* x, const float* w1, const float* w2, void* dx, float* dw1,
                        float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype,
                        void* stream);

/* ---- deferred, grouped LoKr weight gradients ---------------------------------------------------------
 * The factor gradients are consumed by the optimizer only, so the backward pass of a network does not have to wait for
 * them layer by layer: call lyc_lokr_linear_bwd with `dtype | LYC_DEFER_WGRAD`, dw1 != NULL, dw2 == NULL and a `ws` --
 * it runs the dx launch only and leaves the w1-gradient partials in `ws` -- keep (g, x, ws) of the layer alive, and hand
 * batches of layers to lyc_lokr_wgrad_group, which finishes dw1 and accumulates dw2 for ALL of them in
 * ceil(n / 24) launches per tile configuration (reference: the same autograd products of modules/lokr.py:543-566, just
 * scheduled together).  Items must satisfy lyc_lokr_wgrad_deferrable (16-bit activations, a == b dividing 16, c and d
 * multiples of 8, 16-byte aligned g and x); `dw1` may be NULL (only dw2 wanted).  A parameter may appear in several items. */
typedef struct LycLokrWgradItem {
  const void* g;    /* [M, a*c] upstream gradient            */
  const void* x;    /* [M, b*d] layer input                  */
  const float* w1;  /* [a, b]                                */
  float* dw1;       /* [a, b]  += (NULL: skip)               */
  float* dw2;       /* [c, d]  +=                            */
  void* ws;         /* the scratch the layer's dx launch used */
  int64_t M;
  int a, b, c, d;
  float alpha;
} LycLokrWgradItem;
int lyc_lokr_wgrad_deferrable(const void* g, const void* x, int64_t M, int a, int b, int c, int d, int dtype);
int lyc_lokr_wgrad_group(const LycLokrWgradItem* items, int n, int dtype, void* stream);
/* The same with a caller-owned device scratch of lyc_lokr_wgrad_table_bytes(n) bytes (16-byte aligned, alive until the launches have
 * run): the problems are then written to a table in that scratch and ALL layers of a tile class run in ONE launch (round 4: no
 * launch tails between groups of 24 layers, two workgroups per CU throughout) instead of ceil(n / 24).  table == NULL or
 * table_bytes too small: exactly lyc_lokr_wgrad_group.  Capture-safe: the table is written by kernels whose arguments carry the
 * problems, not by a host copy. */
int64_t lyc_lokr_wgrad_table_bytes(int n);
int lyc_lokr_wgrad_group_ws(const LycLokrWgradItem* items, int n, int dtype, void* table, int64_t table_bytes, void* stream);

/* The same for LoCon on nn.Linear: call lyc_locon_linear_bwd with d_down == d_up == NULL (dx launch only; `dt` is still
 * written), keep (g, x, t, dt) alive, and hand batches to lyc_locon_wgrad_group: d_up += alpha * g^T t, d_down += dt^T x for
 * ALL layers in ceil(n / 18) launches per kernel configuration (reference: the autograd products of modules/locon.py:309-332). */
typedef struct LycLoconWgradItem {
  const void* g;    /* [M, O] upstream gradient                          */
  const void* x;    /* [M, I] layer input                                */
  const float* t;   /* [M, r] forward intermediate (needed for d_up)     */
  const float* dt;  /* [M, r] written by the layer's lyc_locon_linear_bwd */
  float* d_down;    /* [r, I] += (NULL: skip)                            */
  float* d_up;      /* [O, r] += (NULL: skip)                            */
  int64_t M;
  int I, O, r;
  float alpha;
} LycLoconWgradItem;
int lyc_locon_wgrad_deferrable(const void* g, const void* x, int64_t M, int I, int O, int r, int dtype);
int lyc_locon_wgrad_group(const LycLoconWgradItem* items, int n, int dtype, void* stream);

/* And for LoHa on nn.Linear: lyc_loha_linear_bwd with the four gradient pointers NULL runs only dx = g dW; lyc_loha_wgrad_group
 * then computes G = g^T x per layer (library GEMM into the item's `gw` scratch, [O, I] fp32) and HadaWeight.backward
 * (functional/loha.py:18-30) for up to 24 layers per launch with blocks of tiles per workgroup: 1/nt + 1/NO fp32 atomics per
 * element of G instead of 2.  The gradients always accumulate with atomics here, so a parameter may appear in several items. */
typedef struct LycLohaWgradItem {
  const void* g;    /* [M, O] */
  const void* x;    /* [M, I] */
  const float *w1a, *w1b, *w2a, *w2b;     /* w*a:[O, r]  w*b:[r, I] */
  float *d_w1a, *d_w1b, *d_w2a, *d_w2b;   /* +=, all four           */
  float* gw;        /* [O, I] fp32 scratch */
  int64_t M;
  int I, O, r;
  float alpha;
} LycLohaWgradItem;
int lyc_loha_wgrad_deferrable(const void* g, const void* x, int64_t M, int I, int O, int r, int dtype);
int lyc_loha_wgrad_group(const LycLohaWgradItem* items, int n, int dtype, void* stream);

/* ---- LoKr on nn.Conv2d (groups = 1): implicit GEMM, no im2col -------------------------------------
 * replaces lycoris/modules/lokr.py:543-566 with F.conv2d (functional/general.py:6) and the grouped-conv bypass
 * lycoris/functional/lokr.py:195-247.  Activations are NHWC ROW matrices (channels contiguous):
 *   x_rows:[B*H*W, b*d]   y_rows / g_rows:[B*Ho*Wo, a*c]   dx_rows:[B*H*W, b*d]
 * (lyc_nchw_to_rows / lyc_rows_to_nchw convert; a channels_last tensor already is such a matrix).
 *   w1:[a,b]   w2p:[c, kh*kw, d] = the reference's lokr_w2 [c, d, kh, kw] with the kernel window moved in front of
 *   the input-channel index (w2.permute(0,2,3,1): free when the parameter is kept in channels_last memory format);
 *   dw2p has the same layout.
 *   y[(b,ho,wo), p*c+q] = alpha * sum_u w1[p,u] * sum_{i,j,v} w2[q,v,i,j] * x[(b, ho*sh-ph+i*dh, wo*sw-pw+j*dw), u*d+v]
 * The kernel window is walked inside the kernels (pixel-row gather, zero outside the image); dx is the transposed
 * convolution with all taps summed in fp32 registers and rounded once.  Requires 16-bit activations, a == b in
 * {4, 8, 16}, c % 8 == d % 8 == 0, 16-byte aligned rows; returns LYC_ERR_UNSUPPORTED otherwise (use the im2col
 * lowering below).  `ws`: optional scratch of lyc_lokr_conv2d_bwd_workspace_bytes(...) bytes, as for the Linear form.
 * `w2t`: optional second copy of the factor as [kh*kw, c, d] (w2.permute(2,3,0,1)); with it (and stride 1) the
 * transposed convolution runs over the flat (tap, q) index in full K segments instead of one short segment per tap. */
int lyc_lokr_conv2d_fwd(const void* x_rows, const float* w1, const float* w2p, void* y_rows, int64_t B, int64_t H,
                        int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                        int dw, float alpha, int dtype, void* stream);
int64_t lyc_lokr_conv2d_bwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int a, int b, int d);
int lyc_lokr_conv2d_bwd(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const float* w2t,
                        void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W, int a, int b, int c, int d,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha, int dtype,
                        void* stream);

/* ---- LoKr: pre-packed operand planes + the LDS-patch Conv2d kernels (ABI v8, csrc/kron_conv.h) ------------------------
 * w2 (lokr_w2, or lokr_w2_a @ lokr_w2_b: reference lycoris/modules/lokr.py:358-381 get_weight / functional/lokr.py:124-151)
 * is a PARAMETER: it changes once per optimizer step.  lyc_lokr_pack_w2 writes it once as hi + lo planes in the activation
 * dtype, in MFMA-fragment order, for the two contractions that use it: `planes_fwd` (n = q, k = (tap, v): y = ... w2 x) and
 * `planes_bwd` (n = v, k = (tap, q): dx = ... w2^T g); either may be NULL.  lyc_lokr_planes_bytes gives their sizes.
 *   full matrix : w2 element (q, v, tap) at w2[q*sq + v*sv + tap*st]            (w2a == w2b == NULL)
 *                 [c, d, kh, kw] contiguous: sq = d*kh*kw, sv = kh*kw, st = 1; channels_last / [c, kh*kw, d]: sq = kh*kw*d, sv = 1, st = d
 *   low rank    : w2 == NULL; w2[q, v, tap] = sum_r w2a[q*a_sq + r*a_sr] * w2b[r*b_sr + v*b_sv + tap*b_st]   (the product is
 *                 formed inside the pack kernel: no [c, d] tensor, no library GEMM)
 * The *_planes Conv2d entry points are lyc_lokr_conv2d_fwd / _bwd with the stage-1 operand taken from the planes and the
 * kh x kw window applied from a source patch held in LDS (each source pixel is loaded once per workgroup instead of once
 * per tap).  lyc_lokr_conv2d_planes_ok(..., backward) tells whether a geometry is covered (16-bit activations, a == b in
 * {4, 8, 16}, c % 8 == d % 8 == 0, the patch + operand ring within 160 KiB of LDS; backward: stride 1); a refused call
 * returns LYC_ERR_UNSUPPORTED and does nothing.  In lyc_lokr_conv2d_bwd_planes `w2p` may be NULL when the geometry is
 * covered (it is only the fallback operand); dw2p is computed as by lyc_lokr_conv2d_bwd. */
/* The same planes serve nn.Linear (taps = 1: planes_fwd n = q, k = v for y; planes_bwd n = v, k = q for dx): the *_planes forms of
 * lyc_lokr_linear_fwd / _bwd take them instead of the fp32 w2, so that the workgroups of the row kernel (csrc/kron3.h) no longer
 * convert their w2 tile themselves (a fifth of a workgroup's instruction chain) but stream the planes into LDS by LDS-DMA.
 * 16-bit fast path only (lyc_lokr_linear_planes_ok: a == b dividing 16, c % 8 == d % 8 == 0, 16-bit dtype; the activation
 * pointers 16-byte aligned); dw2 is computed from g and x alone, as before (also by lyc_lokr_wgrad_group).
 * lyc_lokr_pack_group refreshes the planes of MANY layers in one launch per 28 factors (full matrices, any strides): the
 * once-per-optimizer-step form. */
typedef struct LycLokrPackItem {
  const float* w2;     /* element (q, v, tap) at q*sq + v*sv + tap*st; NULL: the low-rank pair below */
  int64_t sq, sv, st;
  int c, d, taps;
  void* planes_fwd;    /* lyc_lokr_planes_bytes(c, d, taps, 0) bytes, or NULL */
  void* planes_bwd;    /* lyc_lokr_planes_bytes(c, d, taps, 1) bytes, or NULL */
  const float* w2a;    /* low rank (w2 == NULL): contiguous lokr_w2_a [c, rank] ...                       */
  const float* w2b;    /* ... and lokr_w2_b [rank, d * taps] (the reference's layout, modules/lokr.py:131-136) */
  int rank;
} LycLokrPackItem;
int lyc_lokr_pack_group(const LycLokrPackItem* items, int n, int dtype, void* stream);
/* Round 6: the same refresh as ONE launch over all layers, through a caller-owned device table of lyc_lokr_pack_table_bytes(items, n)
 * bytes (16-byte aligned, alive until the launch has run).  table_valid != 0: the table already holds exactly these items (written by
 * an earlier call with the same list) -- only the pack launch is enqueued; otherwise the descriptors are written first (28 per
 * launch, from kernel arguments: capturable).  Same results as lyc_lokr_pack_group. */
int64_t lyc_lokr_pack_table_bytes(const LycLokrPackItem* items, int n);
int lyc_lokr_pack_group_ws(const LycLokrPackItem* items, int n, int dtype, void* table, int64_t table_bytes, int table_valid, void* stream);
/* Low-rank w2 = w2a @ w2b: the activation path needs only the planes (packed straight from the two factors); the weight
 * gradient dW2 [c, d] is taken in a caller-owned fp32 scratch (lyc_lokr_linear_bwd / lyc_lokr_wgrad_group with dw2 = scratch,
 * zero-filled) and pushed through the product by ONE launch per 56 layers:
 *     d_w2a += dW2 w2b^T,   d_w2b += w2a^T dW2          (fp32, accumulated atomically)
 * replaces the reference's `w2a @ w2b` + its two autograd GEMMs per layer and step (functional/lokr.py:124-151). */
typedef struct LycLokrLrChainItem {
  const float* dw2;   /* [c, d]; Conv2d: [c, taps, d], the window-major layout lyc_lokr_conv_wgrad_group writes */
  const float* w2a;   /* [c, r]  */
  const float* w2b;   /* [r, d]; Conv2d: [r, d * taps] with column v * taps + tap (reference layout, modules/lokr.py:131-136) */
  float* d_w2a;       /* [c, r] +=   (NULL: factor frozen; not both) */
  float* d_w2b;       /* same layout as w2b, +=   (NULL: factor frozen) */
  int c, d, r;
  int taps;           /* kh * kw; 0 or 1: nn.Linear */
} LycLokrLrChainItem;
int lyc_lokr_lr_chain_group(const LycLokrLrChainItem* items, int n, void* stream);
int lyc_lokr_linear_planes_ok(int64_t M, int a, int b, int c, int d, int dtype);
int lyc_lokr_linear_fwd_planes(const void* x, const float* w1, const void* planes_fwd, const void* base, void* y, int64_t M, int a,
                               int b, int c, int d, float alpha, int dtype, void* stream);
int lyc_lokr_linear_bwd_planes(const void* g, const void* x, const float* w1, const void* planes_bwd, void* dx, float* dw1,
                               float* dw2, void* ws, int64_t M, int a, int b, int c, int d, float alpha, int dtype, void* stream);
int64_t lyc_lokr_planes_bytes(int c, int d, int taps, int backward);
/* Sibling projections in ONE launch (round 4): up to 4 problems per launch (longer lists run in chunks) that share (a, b, c, d) --
 * the to_q / to_k / to_v projections of an attention block, to_k / to_v of a cross-attention (reference call sites: one
 * LokrModule.forward per projection, modules/lokr.py:436-486, each on the same hidden state).  A single 1024-token projection fills
 * a third of the chip for ~5 us; the group is planned like one layer with the rows of all of them.  Results are bit-identical to
 * the per-layer *_planes entry points.
 *   forward : in = x [M, b*d], planes = planes_fwd, aux = base [M, a*c] or NULL, out = y [M, a*c], ws unused
 *   backward: in = g [M, a*c], planes = planes_bwd, out = dx [M, b*d]; aux = x [M, b*d] and ws = the layer's scratch
 *             (lyc_lokr_bwd_workspace_bytes) when the w1 gradient is wanted -- its partials are left in ws exactly as by
 *             lyc_lokr_linear_bwd_planes(dtype | LYC_DEFER_WGRAD), to be finished (with dW2) by lyc_lokr_wgrad_group[_ws]
 * All items of a call agree on which optional operands they pass.  LYC_ERR_UNSUPPORTED (nothing launched) when an item is not on
 * the packed-plane fast path: call the per-layer entry points instead. */
typedef struct LycLokrLinearGroupItem {
  const void* in;
  const float* w1;     /* [a, b] */
  const void* planes;
  const void* aux;
  void* out;
  void* ws;
  int64_t M;
  float alpha;
} LycLokrLinearGroupItem;
int lyc_lokr_linear_fwd_group(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, int dtype, void* stream);
int lyc_lokr_linear_bwd_group(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, int dtype, void* stream);
/* The same backward for n <= 4 problems that read ONE tensor (equal M): dx_sum [M, b*d] = sum_i dx_i, formed in registers -- a
 * workgroup walks the n problems of its tile and stores the fp32 sum once (one rounding; no dx_i is written, `out` of the items is
 * ignored); aux = x and ws (the w1 partials, as above) are required.  Replaces n - 1 passes of autograd's gradient accumulation for
 * lokr.py:543-566 called n times on one tensor. */
int lyc_lokr_linear_bwd_group_sum(const LycLokrLinearGroupItem* items, int n, int a, int b, int c, int d, void* dx_sum, int dtype,
                                  void* stream);
/* dst = src[0] + ... + src[n - 1], n <= 4, `numel` elements of the 16-bit `dtype` each (numel % 8 == 0, 16-byte aligned pointers):
 * fp32 accumulation, one rounding.  The gradient of a tensor that n sibling projections read -- the n dx results of
 * lyc_lokr_linear_bwd_group -- in ONE pass (autograd's accumulation, which this replaces for a grouped set, makes n - 1 passes with
 * a rounding each; reference: the implicit gradient accumulation of torch.autograd for lokr.py:543-566 called n times on one
 * tensor).  dst may be src[0]. */
int lyc_sum_rows(const void* const* src, int n, void* dst, int64_t numel, int dtype, void* stream);
int lyc_lokr_pack_w2(const float* w2, int64_t sq, int64_t sv, int64_t st, const float* w2a, int64_t a_sq, int64_t a_sr,
                     const float* w2b, int64_t b_sr, int64_t b_sv, int64_t b_st, int rank, int c, int d, int taps,
                     void* planes_fwd, void* planes_bwd, int dtype, void* stream);
int lyc_lokr_conv2d_planes_ok(int64_t B, int64_t H, int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph,
                              int pw, int dh, int dw, int dtype, int backward);
int lyc_lokr_conv2d_fwd_planes(const void* x_rows, const float* w1, const void* planes_fwd, void* y_rows, int64_t B, int64_t H,
                               int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                               int dw, float alpha, int dtype, void* stream);
/* Deferred, grouped weight gradients of the Conv2d form (as lyc_lokr_wgrad_group for nn.Linear): call lyc_lokr_conv2d_bwd[_planes]
 * with dw2p == NULL and dtype | LYC_DEFER_WGRAD (dx launch only; the dw1 partials stay in `ws`), keep (g_rows, x_rows, ws) alive,
 * and hand batches of layers to lyc_lokr_conv_wgrad_group: dw2p += for up to 14 layers per launch, dw1 += reduced from the
 * partials.  `dw1_blocks` = lyc_lokr_conv2d_dx_blocks(..., with_planes) for the dx call that wrote `ws` (with_planes: it was
 * lyc_lokr_conv2d_bwd_planes on a geometry lyc_lokr_conv2d_planes_ok(backward = 1) covers). */
typedef struct LycLokrConvWgradItem {
  const void* g_rows;  /* [B*Ho*Wo, a*c] */
  const void* x_rows;  /* [B*H*W, b*d]   */
  const float* w1;     /* [a, b]         */
  float* dw1;          /* [a, b] += (NULL: skip) */
  float* dw2p;         /* [c, kh*kw, d] +=       */
  void* ws;            /* the scratch the layer's dx launch used */
  int64_t B, H, W;
  int64_t dw1_blocks;
  int a, b, c, d, kh, kw, sh, sw, ph, pw, dh, dw;
  float alpha;
} LycLokrConvWgradItem;
int64_t lyc_lokr_conv2d_dx_blocks(int64_t B, int64_t H, int64_t W, int a, int b, int c, int d, int kh, int kw, int sh, int sw, int ph,
                                  int pw, int dh, int dw, int dtype, int with_planes);
int lyc_lokr_conv_wgrad_group(const LycLokrConvWgradItem* items, int n, int dtype, void* stream);
int lyc_lokr_conv2d_bwd_planes(const void* g_rows, const void* x_rows, const float* w1, const float* w2p, const void* planes_bwd,
                               void* dx_rows, float* dw1, float* dw2p, void* ws, int64_t B, int64_t H, int64_t W, int a, int b,
                               int c, int d, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha, int dtype,
                               void* stream);

/* ---- LoCon on nn.Linear ------------------------------------------------------------------------
 * replaces lycoris/modules/locon.py:309-332 (forward: make_weight + dense F.linear) and the bypass
 * lycoris/functional/locon.py:64-85 / modules/locon.py:286-304.
 *   down:[r,I]  up:[O,r]  x:[M,I]  y:[M,O];   y = alpha * (x @ down^T) @ up^T
 *   t :[M,r] fp32, written by fwd (t = x @ down^T; keep it for the backward call), no need to clear it
 *   dt:[M,r] fp32 scratch written by bwd; d_down += , d_up += (caller-zeroed or .grad)
 * 16-bit activations with I % 8 == 0 (fwd) / O % 8 == 0 (bwd) and r <= 64 run the fused kernels of csrc/lowrank.h:
 * one launch forward (reduce + expand), two backward (dt/dx, both factor gradients); other shapes and fp32
 * activations take the general split-K kernels.                                                       */
int lyc_locon_linear_fwd(const void* x, const float* down, const float* up, float* t, void* y, int64_t M, int I,
                         int O, int r, float alpha, int dtype, void* stream);
int lyc_locon_linear_bwd(const void* g, const void* x, const float* down, const float* up, const float* t,
                         float* dt, void* dx, float* d_down, float* d_up, int64_t M, int I, int O, int r,
                         float alpha, int dtype, void* stream);
/* Sibling projections in ONE launch (round 5), as lyc_lokr_linear_fwd_group for LoCon: n <= 4 layers of equal (I, O, r) and M --
 * the to_q / to_k / to_v adapters of an attention block (reference: one LoConModule.forward per projection, modules/locon.py:309-332).
 *   forward : in = x [M, I], mid = t [M, r] fp32 (written; keep it for the backward call), out = y [M, O] = alpha * (x down^T) up^T
 *   backward: in = g [M, O], mid = dt [M, r] fp32 (written), out = dx [M, I] = (alpha * g up) down -- the dx launch of
 *             lyc_locon_linear_bwd with NULL gradient pointers; the factor gradients follow through lyc_locon_wgrad_group
 * Bit-identical to the per-layer entry points.  LYC_ERR_UNSUPPORTED (nothing launched) off the fused 16-bit rank-r path or for shape
 * classes without a grouped instantiation (r > 32): call the per-layer entry points instead. */
typedef struct LycLoconLinearGroupItem {
  const void* in;
  const float* down;   /* [r, I] */
  const float* up;     /* [O, r] */
  float* mid;
  void* out;
  int64_t M;
  float alpha;
} LycLoconLinearGroupItem;
int lyc_locon_linear_fwd_group(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, int dtype, void* stream);
int lyc_locon_linear_bwd_group(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, int dtype, void* stream);
/* The same backward for 2 <= n <= 4 problems that read ONE tensor (equal M; round 6): dx_sum [M, I] = sum_i dt_i down_i, formed as ONE
 * expand stage over the n `mid` tiles of a workgroup (lowrank4.h: bneck4_sum_kernel) -- fp32 sum, one rounding, no dx_i written (`out`
 * of the items is ignored), every dt_i still goes to its `mid`.  Replaces lyc_locon_linear_bwd_group + lyc_sum_rows for
 * locon.py:309-332 called n times on one tensor.  LYC_ERR_UNSUPPORTED (nothing launched) where bneck4_kernel does not cover the
 * shape (r > 16, r % 4, I % 8, O % 8, unaligned pointers) or the n tiles do not fit the LDS: use the two calls above. */
int lyc_locon_linear_bwd_group_sum(const LycLoconLinearGroupItem* items, int n, int I, int O, int r, void* dx_sum, int dtype, void* stream);

/* LoCon on nn.Conv2d without im2col (reference: lycoris/modules/locon.py:286-332 with F.conv2d -- lora_down is the
 * kh x kw convolution [r, C, kh, kw], lora_up the 1x1 convolution [O, r, 1, 1]; functional/locon.py:64-85).
 * x_rows / g_rows / y_rows / dx_rows are NHWC pixel rows ([B*H*W, C] resp. [B*Ho*Wo, O]; a channels_last tensor as is,
 * lyc_nchw_to_rows otherwise).  down_p is lora_down as [r, kh, kw, C] (a channels_last parameter viewed with
 * permute(0, 2, 3, 1)), up is [O, r]; d_down_p / d_up have the same layouts and are accumulated ("+=").
 * t ([B*Ho*Wo, r] fp32) is written by fwd and read by bwd; dt is scratch of the same size.
 * Taken for 16-bit activations, C % 16 == 0, O % 8 == 0, r in {4, 8, 12, 16} with kh*kw*r <= 144; LYC_ERR_UNSUPPORTED
 * otherwise (use lyc_im2col + lyc_locon_linear_*).  Three launches in bwd: dt = alpha g up; dx = transposed convolution
 * of dt (gathered in LDS, no col2im); both factor gradients. */
int lyc_locon_conv2d_fwd(const void* x_rows, const float* down_p, const float* up, float* t, void* y_rows, int64_t B,
                         int64_t H, int64_t W, int C, int O, int r, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                         int dw, float alpha, int dtype, void* stream);
int lyc_locon_conv2d_bwd(const void* g_rows, const void* x_rows, const float* down_p, const float* up, const float* t,
                         float* dt, void* dx_rows, float* d_down_p, float* d_up, int64_t B, int64_t H, int64_t W, int C,
                         int O, int r, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float alpha,
                         int dtype, void* stream);

/* ---- (IA)^3 per-channel scale ------------------------------------------------------------------
 * replaces lycoris/modules/ia3.py:91-102,129-144 (rebuild path).  Tensors are [outer, C, inner]
 * (Linear: inner = 1; NCHW Conv2d: inner = H*W).
 *   lyc_chan_scale : out = in * (s0 + w[c]*mult) - bias[c] * w[c]*mult        (bias may be NULL)
 *   lyc_chan_reduce: dw[c] += mult * sum a * (b - bias[c])                                             */
int lyc_chan_scale(const void* in, const float* w, const float* bias, void* out, int64_t outer, int64_t C,
                   int64_t inner, float s0, float mult, int dtype, void* stream);
int lyc_chan_reduce(const void* a, const void* b, const float* bias, float* dw, int64_t outer, int64_t C,
                    int64_t inner, float mult, int dtype, void* stream);
/* Round 6: the backward of lyc_chan_scale in ONE pass over g (ia3.py's autograd of `x * w` / `base + (base - bias) * w`):
 *   da = g * (s0 + w[c]*mult)   and   dw[c] += mult * sum g * (a - bias[c])         (da or dw may be NULL: that half is skipped)  */
int lyc_chan_bwd(const void* g, const void* a, const float* w, const float* bias, void* da, float* dw, int64_t outer, int64_t C,
                 int64_t inner, float s0, float mult, int dtype, void* stream);

/* ---- LoHa on nn.Linear -------------------------------------------------------------------------
 * replaces lycoris/modules/loha.py:301-322 (forward) and lycoris/functional/loha.py:10-30
 * (HadaWeight.forward / .backward).   w1a,w2a:[O,r]  w1b,w2b:[r,I]
 *   dW = ((w1a @ w1b) * (w2a @ w2b)) * alpha ;  y = x @ dW^T
 * `wplanes` is caller-owned scratch of lyc_loha_workspace_bytes(O, I, dtype) bytes: fwd fills it with the
 * matrix-core operand image of dW -- ONE plane in the activation type for 16-bit activations (dW rounded once, the
 * reference's `diff_weight.to(base_weight.dtype)`, modules/loha.py:310), the fp32 plane and its transpose for fp32
 * activations -- and bwd reads it (pass the same buffer).
 * `gw` is [O,I] fp32 scratch (no need to clear).  d_w* +=                                              */
int64_t lyc_loha_workspace_bytes(int O, int I, int dtype);
/* Round 6: the operand planes of MANY layers in one launch per 48 layers -- the refresh of a plane cache keyed on the parameters
 * (csrc/torch_ops.cpp: the factors change once per optimizer step, not per layer call; reference lycoris/functional/loha.py:10-16
 * HadaWeight.forward rebuilds dW in every forward).  `plane`: lyc_loha_workspace_bytes(O, I, dtype) bytes, 16-byte aligned; the layer must
 * satisfy lyc_loha_plane_cacheable (16-bit activations, rank <= 32, rank % 4 == 0, O % 8 == I % 8 == 0, 16-byte aligned factors).
 * The plane written is bit-identical to the one lyc_loha_linear_fwd writes for the same factors. */
typedef struct LycLohaPlaneItem {
  const float *w1a, *w1b, *w2a, *w2b;   /* w*a:[O, r]  w*b:[r, I] */
  void* plane;
  int O, I, r;
  float alpha;
} LycLohaPlaneItem;
int lyc_loha_plane_cacheable(const float* w1a, const float* w1b, const float* w2a, const float* w2b, int I, int O, int r, int dtype);
int lyc_loha_rebuild_group(const LycLohaPlaneItem* items, int n, int dtype, void* stream);
int lyc_loha_linear_fwd(const void* x, const float* w1a, const float* w1b, const float* w2a, const float* w2b,
                        void* wplanes, void* y, int64_t M, int I, int O, int r, float alpha, int dtype,
                        void* stream);
int lyc_loha_linear_bwd(const void* g, const void* x, const float* w1a, const float* w1b, const float* w2a,
                        const float* w2b, const void* wplanes, float* gw, void* dx, float* d_w1a, float* d_w1b,
                        float* d_w2a, float* d_w2b, int64_t M, int I, int O, int r, float alpha, int dtype,
                        void* stream);

/* ---- weight space: merge / diff weight / max-norm / DoRA -----------------------------------------------
 * Everything the reference does on the [O, I(,kh,kw)]-shaped dW OFF the activation path, with the dW tile rebuilt on
 * chip from the factors (csrc/wspace.h).  J = I*kh*kw (a conv weight [O, I, kh, kw] is the row-major matrix [O, J]).
 *
 *   v[o, j] = coef[ch(o, j)] * (w_scale * W[o, j] + alpha * dW[o, j])
 *   out  != NULL : out[o, j] = v + beta * out[o, j]        (out_dtype: LYC_F32 / LYC_F16 / LYC_BF16)
 *   sums != NULL : sums[ch(o, j)] += v * v                  (caller-zeroed)
 *   ch: chan_mode 0 = one channel (index 0), 1 = output row o, 2 = input channel j / kk  (kk = kh*kw)
 *   W, coef may be NULL (then w_scale*W = 0, coef = 1).
 *
 *   algo 0 LoCon: f0 = lora_down [r, J], f1 = lora_up [O, r]                        dW = up down
 *        1 LoHa : f0..f3 = w1a [O, r], w1b [r, J], w2a [O, r], w2b [r, J]            dW = (w1a w1b) * (w2a w2b)
 *        2 LoKr : f0 = w1 [a, b], f1 = w2 viewed [c, J / b]  (O == a*c)              dW = kron(w1, w2)
 *
 * replaces: get_diff_weight / merge_to (lycoris/modules/base.py:326-342, locon.py:221-237, loha.py:228-242,
 * lokr.py:383-397): out = W, beta = 0, w_scale = 1, or out = W in place with W = NULL, beta = 1;
 * apply_max_norm's dW.norm() (locon.py:273-284, loha.py:281-292, lokr.py:442-466): sums only, chan_mode 0;
 * apply_weight_decompose (locon.py:239-260, loha.py:244-265, lokr.py:399-420): the norms of W + dW per output row
 * (wd_on_out) or per input channel are `sums` with W given; the rescaled weight is `out` with coef = scale. */
int lyc_wspace(int algo, const float* f0, const float* f1, const float* f2, const float* f3, int64_t O, int64_t J, int r,
               int a, int b, int c, int kk, const void* W, int w_dtype, float w_scale, const float* coef, int chan_mode,
               void* out, int out_dtype, float beta, float* sums, float alpha, void* stream);

/* Gradient of a weight-space quantity w.r.t. the factors: gw is the dense fp32 gradient [O, J] w.r.t. dW (for DoRA:
 * 2 * dL/dnorm2[ch] * (W + dW), written by lyc_wspace with coef); the factor gradients are accumulated ("+=").
 * replaces autograd through `up @ down` (locon.py:198-219), HadaWeight.backward (functional/loha.py:18-30) and
 * torch.kron's backward (functional/lokr.py:11-20) on a weight-shaped gradient. */
int lyc_locon_wgrad(const float* gw, const float* down, const float* up, float* d_down, float* d_up, int64_t O, int64_t J,
                    int r, float alpha, void* stream);
int lyc_loha_wgrad(const float* gw, const float* w1a, const float* w1b, const float* w2a, const float* w2b, float* d_w1a,
                   float* d_w1b, float* d_w2a, float* d_w2b, int64_t O, int64_t J, int r, float alpha, void* stream);
int lyc_lokr_wgrad(const float* gw, const float* w1, const float* w2, float* d_w1, float* d_w2, int a, int b, int c,
                   int64_t dk, float alpha, void* stream);

/* ---- Tucker / conv-CP forms ------------------------------------------------------------------------------------
 * rebuild_tucker (lycoris/functional/general.py:9-11): W[p, q, k] = sum_ij t[i, j, k] wa[i, p] wb[j, q].  Every Tucker
 * form of the reference (lora_mid, modules/locon.py:85-90; hada_t1 / hada_t2 with HadaWeightTucker, functional/loha.py:
 * 33-75; lokr_t2, modules/lokr.py:121-128) is the non-Tucker form of the same algorithm with the k x k core folded into
 * the input-side factor:  B[i, (q, k)] = sum_j t[i, j, k] wb[j, q]   (t [r1, r2, kk], wb [r2, Q], B [r1, Q * kk] = the
 * layout of a flattened conv factor [r, I, kh, kw]); the adapter kernels then run unchanged on (wa^T, B).
 * fwd writes B; bwd writes d_t [r1, r2, kk] and d_wb [r2, Q] (either may be NULL) from dB.  Not accumulated. */
int lyc_tucker_core_fwd(const float* t, const float* wb, float* out, int r1, int r2, int64_t Q, int kk, void* stream);
int lyc_tucker_core_bwd(const float* dout, const float* t, const float* wb, float* d_t, float* d_wb, int r1, int r2, int64_t Q,
                        int kk, void* stream);

/* ---- Conv2d lowering (NCHW, groups = 1) --------------------------------------------------------
 * The Conv2d form of every adapter (F.conv2d in lycoris/functional/general.py:6, kw_dict of
 * lycoris/modules/base.py:101-121) is evaluated as its Linear entry point on the im2col view
 *   cols[(b,oh,ow), c*kh*kw + i*kw + j] = x[b, c, oh*sh-ph+i*dh, ow*sw-pw+j*dw]
 * which matches the reference's own [r, I*kh*kw] / [c, d*kh*kw] conv factor layouts:
 *   rows = lyc_im2col(x); y_rows = lyc_<algo>_linear_fwd(rows, ...); y = lyc_rows_to_nchw(y_rows)
 *   backward: g_rows = lyc_nchw_to_rows(g); ... ; dx = lyc_col2im(d_rows)                              */
int lyc_im2col(const void* x, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh,
               int sw, int ph, int pw, int dh, int dw, int dtype, void* stream);
int lyc_col2im(const void* dcols, void* dx, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh,
               int sw, int ph, int pw, int dh, int dw, int dtype, void* stream);
/* Round 6 (ABI 12): the same lowering on NHWC ROW matrices with WINDOW-MAJOR columns
 *   cols[(b,oh,ow), (i*kw + j)*C + c] = x_rows[(b, oh*sh-ph+i*dh, ow*sw-pw+j*dw), c]
 * -- a channels_last tensor is its own row matrix and a tap's column block is a contiguous run of channels, so every access is a
 * 16-byte vector (the NCHW forms above are element-wise gathers: 183 us per SDXL conv layer).  The factor that meets these columns is the
 * reference's [r, C*kh*kw] one with its columns permuted to window-major order ([r, C, kh*kw] -> [r, kh*kw, C]; csrc/torch_ops.cpp does
 * that for LoHa, whose Conv2d form is reference lycoris/modules/loha.py:301-322 with F.conv2d of functional/general.py:6).
 * 16-bit tensors, C % 8 == 0, 16-byte aligned; LYC_ERR_UNSUPPORTED otherwise.  lyc_col2im_rows: gather form, fp32 accumulation, one
 * rounding; `dtype | LYC_F32_ROWS`: dcols is fp32. */
int lyc_im2col_rows(const void* x_rows, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int ph,
                    int pw, int dh, int dw, int dtype, void* stream);
int lyc_col2im_rows(const void* dcols, void* dx_rows, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int ph,
                    int pw, int dh, int dw, int dtype, void* stream);
int lyc_nchw_to_rows(const void* t, void* rows, int64_t B, int64_t C, int64_t P, int dtype, void* stream);
int lyc_rows_to_nchw(const void* rows, void* t, int64_t B, int64_t C, int64_t P, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LYCORIS_AMD_H */
