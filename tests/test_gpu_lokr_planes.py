"""GPU: pre-packed LoKr operand planes (lyc_lokr_pack_w2) and the LDS-patch Conv2d kernels that consume them
(csrc/kron_conv.h, lyc_lokr_conv2d_{fwd,bwd}_planes), through the C ABI, against the float64 oracle.

The pack kernel is checked BIT-EXACTLY: a plane element is hi = round_T(w), lo = round_T(w - hi) of the fp32 value at a
known (n, k) position, so the expected byte image can be built with torch on the host.  The conv kernels are checked against
oracle.lokr (reference semantics: lycoris/functional/lokr.py:195-247, modules/lokr.py:358-381) on geometries chosen to hit
every plan branch: ragged images (tiles partly outside), batch > 1, stride 2 (forward only), dilation, 5x5 windows, factor 4 /
8 / 16, group widths with and without LDS padding (K / 8 even / odd), column counts that need 2 / 3 / 4 n tiles and padding.
"""
import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, check, dev, err, rnd
from lycoris_amd import _native as N

pytestmark = pytest.mark.gpu


def _expected_planes(w2, c, d, taps, backward, dtype):
    """byte image of one role: units [nt][ks] of {hi[64][8], lo[64][8]}; w2: fp32 [c, d, taps] (cpu)"""
    Nn, Kt = (d, c) if backward else (c, d)
    ks_n = (taps * Kt + 31) // 32
    nt_n = (Nn + 15) // 16
    B = torch.zeros(nt_n * 16, ks_n * 32, dtype=torch.float32)
    src = w2.permute(1, 2, 0).reshape(d, taps * c) if backward else w2.permute(0, 2, 1).reshape(c, taps * d)  # [n, (tap, kk)]
    B[:Nn, :taps * Kt] = src
    hi = B.to(dtype)
    lo = (B - hi.float()).to(dtype)
    # unit (nt, ks): lane = li + 16 g holds n = 16 nt + li, k = 32 ks + 8 g + e
    def units(t):
        t = t.view(nt_n, 16, ks_n, 4, 8)            # nt, li, ks, g, e
        return t.permute(0, 2, 3, 1, 4).contiguous()  # nt, ks, g, li, e  -> lane = g * 16 + li
    img = torch.stack([units(hi), units(lo)], dim=2)  # nt, ks, {hi, lo}, g, li, e
    return img.contiguous().view(torch.uint8).flatten()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("c,d,kh,layout", [(40, 40, 3, "nchw"), (160, 80, 3, "cl"), (24, 56, 1, "nchw"), (8, 8, 5, "cl"), (48, 120, 3, "nchw")])
def test_pack_full_matrix_bit_exact(c, d, kh, layout, dtype):
    taps = kh * kh
    gen = torch.Generator().manual_seed(c * d + kh)
    w2 = torch.randn(c, d, kh, kh, generator=gen) * 0.1
    wd = w2.to(dev())
    if layout == "cl":
        wd = wd.contiguous(memory_format=torch.channels_last)
    lib = N.load()
    code = N.dtype_code(dtype)
    pf = torch.full((int(lib.lyc_lokr_planes_bytes(c, d, taps, 0)),), 0xCD, dtype=torch.uint8, device=dev())
    pb = torch.full((int(lib.lyc_lokr_planes_bytes(c, d, taps, 1)),), 0xCD, dtype=torch.uint8, device=dev())
    N.call("lyc_lokr_pack_w2", N.ptr(wd), wd.stride(0), wd.stride(1), wd.stride(3), None, 0, 0, None, 0, 0, 0, 0, c, d, taps,
           N.ptr(pf), N.ptr(pb), code, N.stream_ptr(dev()))
    torch.cuda.synchronize()
    w3 = w2.reshape(c, d, taps)
    assert torch.equal(pf.cpu(), _expected_planes(w3, c, d, taps, False, dtype))
    assert torch.equal(pb.cpu(), _expected_planes(w3, c, d, taps, True, dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("kh,r", [(3, 16), (1, 16), (1, 6)], ids=["conv3x3_r16", "linear_r16", "linear_r6"])  # Linear: the 16-byte-load paths
def test_pack_low_rank_product(dtype, kh, r):
    """w2 = w2a @ w2b formed inside the pack kernel (reference modules/lokr.py:131-136, 370: lokr_w2_a [c, r], lokr_w2_b [r, d*k*k])"""
    c, d = 80, 40
    taps = kh * kh
    gen = torch.Generator().manual_seed(5)
    w2a, w2b = torch.randn(c, r, generator=gen) * 0.3, torch.randn(r, d * taps, generator=gen) * 0.3
    lib = N.load()
    code = N.dtype_code(dtype)
    pf = torch.zeros(int(lib.lyc_lokr_planes_bytes(c, d, taps, 0)), dtype=torch.uint8, device=dev())
    pb = torch.zeros(int(lib.lyc_lokr_planes_bytes(c, d, taps, 1)), dtype=torch.uint8, device=dev())
    ad, bd = w2a.to(dev()), w2b.to(dev())
    N.call("lyc_lokr_pack_w2", None, 0, 0, 0, N.ptr(ad), r, 1, N.ptr(bd), d * taps, taps, 1, r, c, d, taps, N.ptr(pf), N.ptr(pb), code,
           N.stream_ptr(dev()))
    torch.cuda.synchronize()
    w2 = (w2a.double() @ w2b.double()).reshape(c, d, taps)

    def unpack(img, Nn, Kt):  # -> hi + lo as float64 [Nn, taps * Kt]
        ks_n, nt_n = (taps * Kt + 31) // 32, (Nn + 15) // 16
        t = img.cpu().view(dtype).view(nt_n, ks_n, 2, 4, 16, 8).double()
        t = t[:, :, 0] + t[:, :, 1]                        # nt, ks, g, li, e
        return t.permute(0, 3, 1, 2, 4).reshape(nt_n * 16, ks_n * 32)[:Nn, :taps * Kt]

    got_f = unpack(pf, c, d)
    got_b = unpack(pb, d, c)
    want_f = w2.permute(0, 2, 1).reshape(c, taps * d)
    want_b = w2.permute(1, 2, 0).reshape(d, taps * c)
    # hi + lo of a T pair resolves 2 * mantissa bits: 2^-16 (bf16) / 2^-22 (fp16) relative; the product itself is an fp32 fma chain
    tol = 1e-4 if dtype == torch.bfloat16 else 1e-5
    assert float((got_f - want_f).norm() / want_f.norm()) < tol
    assert float((got_b - want_b).norm() / want_b.norm()) < tol


# (B, H, W, a, c, d, k, stride, pad, dil)
GEOMS = [
    (1, 16, 16, 8, 40, 40, 3, 1, 1, 1),     # SDXL 320-channel conv in small: K / 8 odd (no LDS pad), 3 n tiles of 16 (N = 40 -> 48)
    (2, 9, 13, 8, 80, 80, 3, 1, 1, 1),      # ragged image, batch 2, K / 8 even (padded group pitch)
    (1, 12, 12, 8, 160, 16, 3, 1, 1, 1),    # wide N (4 n tiles x 3 column tiles), tiny K
    (1, 11, 7, 8, 24, 120, 3, 1, 1, 1),     # K = 120 (the C = 960 convs), N = 24
    (1, 17, 17, 8, 40, 40, 3, 2, 1, 1),     # stride 2: planes forward, row kernel backward
    (1, 10, 10, 8, 16, 24, 3, 1, 2, 2),     # dilation 2
    (1, 9, 9, 8, 16, 16, 5, 1, 2, 1),       # 5x5 window: 25 taps
    (1, 8, 12, 4, 32, 48, 3, 1, 1, 1),      # factor 4: 4 x 8 pixel tiles
    (1, 8, 8, 16, 8, 8, 3, 1, 1, 1),        # factor 16: 2 x 4 pixel tiles
    (1, 6, 6, 8, 40, 40, 3, 1, 0, 1),       # no padding: output smaller than the input
    (3, 5, 5, 8, 8, 8, 1, 1, 0, 1),         # 1x1 window through the conv entry points
    # K >= 32 in both roles: the software-pipelined k loop of round 6 (tap offsets advanced incrementally) on the odd windows too
    (1, 10, 10, 8, 40, 32, 3, 1, 2, 2),     # dilation 2
    (1, 9, 9, 8, 32, 40, 5, 1, 2, 1),       # 5x5 window: 25 taps
    (2, 7, 9, 8, 48, 32, 3, 1, 1, 1),       # K = 32: a tap boundary on every k step
]


@pytest.mark.parametrize("waves", [8, 80, 4], ids=["w8", "w8serial", "w4"])
@pytest.mark.parametrize("row_tile", [0, 4, 8], ids=["mi_auto", "mi4", "mi8"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("geom", GEOMS, ids=[f"B{g[0]}_{g[1]}x{g[2]}_a{g[3]}_c{g[4]}_d{g[5]}_k{g[6]}s{g[7]}p{g[8]}d{g[9]}" for g in GEOMS])
def test_conv_planes_vs_oracle(geom, dtype, row_tile, waves):
    B, H, W, a, c, d, k, s, p, dl = geom
    lib = N.load()
    # LYC_KCONV_ROW_TILE(mi) in the dtype argument pins the patch kernel's row tile (64 * mi stage-1 rows per workgroup): the host
    # otherwise picks the smallest one for these small problems (include/lycoris_amd.h; an environment variable until round 3)
    # LYC_KCONV_W4: the 4-wave workgroups of rounds 3 - 5 (default since round 6: 8 waves, two per SIMD, where the plan pays)
    # LYC_KCONV_SERIAL: the 8-wave kernel with the serial k loop (default: the software-pipelined loop where the plan has 4 k steps per stage)
    code = N.dtype_code(dtype) | (row_tile << 12) | {8: 0, 80: 0x80000, 4: 0x40000}[waves]
    taps = k * k
    Ho, Wo = (H + 2 * p - dl * (k - 1) - 1) // s + 1, (W + 2 * p - dl * (k - 1) - 1) // s + 1
    gen = torch.Generator().manual_seed(sum(geom))
    x, x64 = rnd((B, a * d, H, W), dtype, gen)
    g, g64 = rnd((B, a * c, Ho, Wo), dtype, gen, 1.0 / np.sqrt(a * c))
    w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d, k, k), torch.float32, gen, 0.1)
    geo = (B, H, W, a, a, c, d, k, k, s, s, p, p, dl, dl)
    fwd_ok = lib.lyc_lokr_conv2d_planes_ok(*geo, code, 0) != 0
    bwd_ok = lib.lyc_lokr_conv2d_planes_ok(*geo, code, 1) != 0
    if row_tile:  # a pinned tile may not fit the LDS for this geometry (the host then refuses it: nothing to test)
        if not fwd_ok:
            pytest.skip(f"row tile {row_tile} does not fit the LDS for this geometry")
        assert lib.lyc_lokr_conv2d_planes_ok(*geo, code, 0) == row_tile
        assert not bwd_ok or s == 1
    else:
        assert fwd_ok, "every geometry of this list fits the forward patch kernel"
        assert bwd_ok == (s == 1)
    x_rows = x.permute(0, 2, 3, 1).reshape(B * H * W, a * d).contiguous()
    g_rows = g.permute(0, 2, 3, 1).reshape(B * Ho * Wo, a * c).contiguous()
    pf = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, taps, 0)), dtype=torch.uint8, device=dev())
    pb = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, taps, 1)), dtype=torch.uint8, device=dev())
    N.call("lyc_lokr_pack_w2", N.ptr(w2), w2.stride(0), w2.stride(1), w2.stride(3), None, 0, 0, None, 0, 0, 0, 0, c, d, taps, N.ptr(pf),
           N.ptr(pb), code, N.stream_ptr(dev()))
    y_rows = torch.full((B * Ho * Wo, a * c), float("nan"), dtype=dtype, device=dev())
    N.call("lyc_lokr_conv2d_fwd_planes", N.ptr(x_rows), N.ptr(w1), N.ptr(pf), N.ptr(y_rows), *geo, 0.5, code, N.stream_ptr(dev()))
    ca = {"stride": s, "padding": p, "dilation": dl}
    y_ref = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.5, kshape=(k, k), conv_args=ca)
    y = y_rows.view(B, Ho, Wo, a * c).permute(0, 3, 1, 2)
    errs = {"y": err(y, y_ref, dtype)}
    bounds = {"y": TOL["store_out"][dtype]}
    if bwd_ok:
        dx_rows = torch.full((B * H * W, a * d), float("nan"), dtype=dtype, device=dev())
        dw1 = torch.zeros(a, a, device=dev())
        dw2p = torch.zeros(c, taps, d, device=dev())
        ws = torch.empty(max(int(lib.lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, a, a, d)), 16), dtype=torch.uint8, device=dev())
        N.call("lyc_lokr_conv2d_bwd_planes", N.ptr(g_rows), N.ptr(x_rows), N.ptr(w1), None, N.ptr(pb), N.ptr(dx_rows), N.ptr(dw1),
               N.ptr(dw2p), N.ptr(ws), *geo, 0.5, code, N.stream_ptr(dev()))
        gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=0.5, kshape=(k, k), conv_args=ca)
        dx = dx_rows.view(B, H, W, a * d).permute(0, 3, 1, 2)
        errs.update({"dx": err(dx, gr["dx"], dtype), "dw1": err(dw1, gr["w1"]),
                     "dw2": err(dw2p.view(c, k, k, d).permute(0, 3, 1, 2), gr["w2"])})
        bounds.update({"dx": TOL["store_out"][dtype], "dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]})
    torch.cuda.synchronize()
    check(f"conv_planes[{geom},{dtype}]", errs, bounds)


def test_every_sdxl_conv_takes_the_patch_kernel_where_lds_allows():
    """which SDXL / SD1.5 Conv2d layers run on the patch kernel: all 3x3 forward passes with C_in <= 1280 and every stride-1
    backward pass (the transposed convolution's operand rows are the OUTPUT channels, <= 1280)"""
    from benchmarks.sdxl_shapes import sdxl_unet_layers
    lib = N.load()
    code = N.dtype_code(torch.bfloat16)
    n_f = n_b = n = 0
    for l in sdxl_unet_layers(1):
        if l["kind"] != "conv" or l["k"] == 1:
            continue
        geo = (l["B"], l["H"], l["W"], 8, 8, l["O"] // 8, l["C"] // 8, l["k"], l["k"], l["stride"], l["stride"], l["pad"], l["pad"], 1, 1)
        f, b = lib.lyc_lokr_conv2d_planes_ok(*geo, code, 0), lib.lyc_lokr_conv2d_planes_ok(*geo, code, 1)
        assert (f != 0) == (l["C"] <= 1920), l
        assert (b != 0) == (l["stride"] == 1), l
        n += l["count"]
        n_f += (f != 0) * l["count"]
        n_b += (b != 0) * l["count"]
    assert n_f >= n - 6 and n_b >= n - 2, (n, n_f, n_b)


@pytest.mark.parametrize("tiles", ["full_width", "narrow"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_conv_wgrad_group_vs_oracle(dtype, tiles):
    """lyc_lokr_conv_wgrad_group over ALL geometries of the list in one call (the deferred form: dx launches with LYC_DEFER_WGRAD
    leave the dw1 partials in `ws`, the grouped launch computes dw2p and reduces dw1).  `full_width`: the Conv2d form of
    kron_dw2f.h (round 4: the taps as column blocks of one virtual operand, shifted source rows by per-lane DMA offsets) for the
    layers it fits, the row-gather kernel of kron_dw2s.h for the rest; `narrow`: LYC_WGRAD_TILE_S pins the row-gather kernel for
    every layer.  (The LDS-patch kernel of round 3 lives in benchmarks/experiments/, outside the product build.)"""
    import ctypes
    kernel = tiles
    lib = N.load()
    code = N.dtype_code(dtype)
    # + the tile classes of the full-width kernel the list above does not reach: 160 x 80 (c > 80, one narrow column block: a 1x1
    # window with stride 2 and padding), 80 x 80 (1x1), dilation and factor 4 / 16 on wide factors, three slabs of rows (B = 3)
    wide = [(1, 9, 9, 8, 168, 72, 1, 2, 1, 1), (2, 7, 6, 8, 40, 64, 1, 1, 0, 1), (1, 10, 10, 8, 48, 24, 3, 1, 2, 2), (1, 8, 12, 4, 64, 48, 3, 1, 1, 1),
            (1, 8, 8, 16, 48, 16, 3, 1, 1, 1), (3, 20, 20, 8, 88, 40, 3, 1, 1, 1)]
    geoms = GEOMS + [GEOMS[0]] + wide  # (one layer twice: two items on different tensors)
    items = (N.LokrConvWgradItem * len(geoms))()
    keep, want = [], []
    for k, geom in enumerate(geoms):
        B, H, W, a, c, d, kk, s, p, dl = geom
        taps = kk * kk
        Ho, Wo = (H + 2 * p - dl * (kk - 1) - 1) // s + 1, (W + 2 * p - dl * (kk - 1) - 1) // s + 1
        gen = torch.Generator().manual_seed(sum(geom) + 31 * k)
        x, x64 = rnd((B, a * d, H, W), dtype, gen)
        g, g64 = rnd((B, a * c, Ho, Wo), dtype, gen, 1.0 / np.sqrt(a * c))
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d, kk, kk), torch.float32, gen, 0.1)
        geo = (B, H, W, a, a, c, d, kk, kk, s, s, p, p, dl, dl)
        x_rows = x.permute(0, 2, 3, 1).reshape(B * H * W, a * d).contiguous()
        g_rows = g.permute(0, 2, 3, 1).reshape(B * Ho * Wo, a * c).contiguous()
        dx_rows = torch.empty_like(x_rows)
        dw1 = torch.zeros(a, a, device=dev())
        dw2p = torch.zeros(c, taps, d, device=dev())
        ws = torch.empty(max(int(lib.lyc_lokr_conv2d_bwd_workspace_bytes(B, H, W, a, a, d)), 16), dtype=torch.uint8, device=dev())
        with_planes = int(lib.lyc_lokr_conv2d_planes_ok(*geo, code, 1) != 0)
        if with_planes:
            pb = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, taps, 1)), dtype=torch.uint8, device=dev())
            N.call("lyc_lokr_pack_w2", N.ptr(w2), w2.stride(0), w2.stride(1), w2.stride(3), None, 0, 0, None, 0, 0, 0, 0, c, d, taps, None,
                   N.ptr(pb), code, N.stream_ptr(dev()))
            N.call("lyc_lokr_conv2d_bwd_planes", N.ptr(g_rows), N.ptr(x_rows), N.ptr(w1), None, N.ptr(pb), N.ptr(dx_rows), N.ptr(dw1), None,
                   N.ptr(ws), *geo, 0.5, code | 0x200, N.stream_ptr(dev()))
            keep.append(pb)
        else:
            w2p = w2.permute(0, 2, 3, 1).contiguous()
            N.call("lyc_lokr_conv2d_bwd", N.ptr(g_rows), N.ptr(x_rows), N.ptr(w1), N.ptr(w2p), None, N.ptr(dx_rows), N.ptr(dw1), None,
                   N.ptr(ws), *geo, 0.5, code | 0x200, N.stream_ptr(dev()))
            keep.append(w2p)
        blocks = int(lib.lyc_lokr_conv2d_dx_blocks(*geo, code, with_planes))
        items[k] = N.LokrConvWgradItem(N.ptr(g_rows), N.ptr(x_rows), N.ptr(w1), N.ptr(dw1), N.ptr(dw2p), N.ptr(ws), B, H, W, blocks,
                                       a, a, c, d, kk, kk, s, s, p, p, dl, dl, 0.5)
        keep += [x_rows, g_rows, dx_rows, ws, w1]
        gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=0.5, kshape=(kk, kk), conv_args={"stride": s, "padding": p, "dilation": dl})
        want.append((geom, dw1, dw2p, gr, (c, kk, d)))
    N.call("lyc_lokr_conv_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(geoms), code | (0x400 if tiles == "narrow" else 0),
           N.stream_ptr(dev()))
    torch.cuda.synchronize()
    for k, (geom, dw1, dw2p, gr, (c, kk, d)) in enumerate(want):
        errs = {"dw1": err(dw1, gr["w1"]), "dw2": err(dw2p.view(c, kk, kk, d).permute(0, 3, 1, 2), gr["w2"])}
        check(f"conv_wgrad_group[{k},{geom},{dtype},{kernel}]", errs, {"dw1": TOL["f32_out"][dtype], "dw2": TOL["f32_out"][dtype]})


def test_conv_module_grads_complete_when_backward_returns():
    """Conv2d LoKr layers in the training configuration (.grad buffers exist, fused accumulation): the dx launch runs inside the
    backward pass, the weight gradients of all parked conv layers in the grouped launch at its end -- .grad must be complete and
    equal to the per-layer path when backward() returns; every parameter is reported once."""
    import torch.nn as nn
    from lycoris_amd import ops
    torch.manual_seed(11)
    w1s = [nn.Parameter(torch.randn(8, 8, device=dev()) * 0.3) for _ in range(3)]
    w2s = [nn.Parameter((torch.randn(16, 16, 3, 3, device=dev()) * 0.1).contiguous(memory_format=torch.channels_last)) for _ in range(3)]
    params = w1s + w2s
    x = (torch.randn(2, 128, 12, 12, device=dev()) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = (torch.randn(2, 128, 12, 12, device=dev()) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(defer):
        seen = []
        for p in params:
            p.grad = torch.zeros_like(p)
        x.grad = None
        ops.fused_grad_accumulation(True, callback=lambda p: seen.append(id(p)))
        ops.deferred_weight_gradients(defer, 48)
        try:
            h = x
            for w1, w2 in zip(w1s, w2s):
                h = h + ops.lokr_conv2d(h, w1, w2, 0.5, (1, 1), (1, 1), (1, 1))
            h.backward(gy)
            assert ops._DISPATCH["ext"].deferred_pending() == 0
            torch.cuda.synchronize()
            return x.grad.clone(), [p.grad.clone() for p in params], seen
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.deferred_weight_gradients(True, 48)

    dx0, g0, seen0 = run(False)
    dx1, g1, seen1 = run(True)
    assert torch.equal(dx0, dx1)
    for u, v in zip(g0, g1):  # two kernels, two summation orders: norm-wise
        assert float(u.abs().max()) > 0 and float((u - v).norm() / u.norm()) < 1e-4, float((u - v).norm() / u.norm())
    assert sorted(seen0) == sorted(seen1) == sorted(id(p) for p in params)


# ---- nn.Linear on cached planes (csrc/torch_ops.cpp planes_for, csrc/kron3.h PL) ----------------------------------------------
def _lin(x, w1, w2):
    from lycoris_amd import ops
    return ops.lokr_linear(x, w1, w2, 0.5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_linear_plane_cache_follows_the_parameter(dtype):
    """the operand planes are cached per leaf tensor with its version counter: an in-place update (what an optimizer does), a
    storage swap (`.data = ...`) and a second activation dtype must all be seen by the next call; cache off == cache on"""
    from lycoris_amd import ops
    gen = torch.Generator().manual_seed(77)
    M, a, c, d = 200, 8, 80, 160
    x, x64 = rnd((M, a * d), dtype, gen)
    w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
    w2 = torch.nn.Parameter(w2)
    ext = ops._DISPATCH["ext"] or (ops._cpp() and ops._DISPATCH["ext"])
    y0 = _lin(x, w1, w2)
    assert ext.planes_cache_size() >= 1
    assert err(y0, oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.5), dtype) < TOL["store_out"][dtype]
    # 1. in-place update
    delta, d64 = rnd((c, d), torch.float32, gen, 0.05)
    with torch.no_grad():
        w2.add_(delta)
    y1 = _lin(x, w1, w2)
    assert err(y1, oracle.lokr.forward(x64, w1=w1_64, w2=w2_64 + d64, scale=0.5), dtype) < TOL["store_out"][dtype]
    # 2. backward on the refreshed planes (dx uses the transposed role)
    xg = x.clone().requires_grad_(True)
    g, g64 = rnd((M, a * c), dtype, gen, 0.05)
    dx, = torch.autograd.grad(_lin(xg, w1, w2), [xg], g)
    gr = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64 + d64, scale=0.5)
    assert err(dx, gr["dx"], dtype) < TOL["store_out"][dtype]
    # 3. storage swap
    new, n64 = rnd((c, d), torch.float32, gen, 0.1)
    w2.data = new
    y2 = _lin(x, w1, w2)
    assert err(y2, oracle.lokr.forward(x64, w1=w1_64, w2=n64, scale=0.5), dtype) < TOL["store_out"][dtype]
    # 4. explicit refresh (what a graph-replaying caller captures) after a raw write that does not bump the version counter
    w2.data.copy_(torch.from_numpy(w2_64).float().to(dev()))  # a write through .data does NOT bump w2._version: only force sees it
    ops.refresh_lokr_planes(force=True)
    y3 = _lin(x, w1, w2)
    assert err(y3, oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.5), dtype) < TOL["store_out"][dtype]
    # 5. the cache is an optimisation, not a semantic: off == on, bit for bit (same operand values, same summation order)
    ops.lokr_planes_cache(False)
    try:
        y4 = _lin(x, w1, w2)
    finally:
        ops.lokr_planes_cache(True)
    assert torch.equal(y3, y4)


def test_linear_plane_cache_one_grouped_refresh_per_step():
    """many layers, one optimizer step: the first layer call of the next step repacks ALL stale planes; results equal the oracle"""
    from lycoris_amd import ops
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(78)
    layers = []
    for k in range(40):
        c, d = [(40, 40), (160, 160), (80, 640), (160, 40)][k % 4]
        w1, w1_64 = rnd((8, 8), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
        x, x64 = rnd((64, 8 * d), dtype, gen)
        layers.append((x, x64, w1, w1_64, torch.nn.Parameter(w2), w2_64))
    for x, x64, w1, w1_64, w2, w2_64 in layers:
        _lin(x, w1, w2)
    opt = torch.optim.SGD([l[4] for l in layers], lr=0.5)
    for l in layers:
        l[4].grad = torch.full_like(l[4], 0.01)
    opt.step()  # every w2 -= 0.005, in place
    for x, x64, w1, w1_64, w2, w2_64 in layers:
        y = _lin(x, w1, w2)
        assert err(y, oracle.lokr.forward(x64, w1=w1_64, w2=w2_64 - 0.005, scale=0.5), dtype) < TOL["store_out"][dtype]


# ---- step boundaries the version counter does not see -------------------------------------------------------------------
class _DataSGD(torch.optim.Optimizer):
    """An optimizer that updates through `p.data` (Prodigy, DAdaptation, 8-bit optimizers with raw kernels, master-weight copies
    do): `p._version` does NOT move."""

    def __init__(self, params, lr):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    p.data.add_(p.grad.to(p.dtype), alpha=-g["lr"])


@pytest.mark.parametrize("how", ["backward_then_data_write", "optimizer_step_hook"])
def test_linear_plane_cache_sees_data_style_updates(how):
    """`p.data.add_()` leaves the autograd version counter where it was (asserted below): the cache must take the END OF A BACKWARD
    PASS and every torch.optim step as step boundaries of their own, otherwise forward / dx keep using the planes of step 0 while
    dW2 and the checkpoint follow the real parameter (silently wrong training)."""
    from lycoris_amd import ops
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(91)
    M, a, c, d = 160, 8, 40, 80
    x, x64 = rnd((M, a * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 0.05)
    w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
    w1 = torch.nn.Parameter(w1)
    w2 = torch.nn.Parameter(w2)
    xg = x.clone().requires_grad_(True)
    y0 = _lin(xg, w1, w2)
    assert err(y0, oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.5), dtype) < TOL["store_out"][dtype]
    v0 = w2._version
    if how == "backward_then_data_write":
        y0.backward(g)  # plain autograd: w2.grad is produced, the pass ends -> the cache is marked dirty
        with torch.no_grad():
            w2.data.add_(w2.grad, alpha=-3.0)
    else:
        w2.grad = (torch.randn(c, d, generator=gen) * 0.05).to(dev())
        _DataSGD([w2], lr=3.0).step()  # no backward of ours in between: the global optimizer-step post hook marks the cache
    assert w2._version == v0, "this torch bumps the version counter on .data writes: the test no longer tests anything"
    new64 = w2.detach().double().cpu().numpy()
    assert oracle.general.rel_err(new64, w2_64) > 1e-2  # a real step
    y1 = _lin(x, w1, w2)
    assert err(y1, oracle.lokr.forward(x64, w1=w1_64, w2=new64, scale=0.5), dtype) < TOL["store_out"][dtype]
    xg2 = x.clone().requires_grad_(True)
    dx, = torch.autograd.grad(_lin(xg2, w1, w2), [xg2], g)
    assert err(dx, oracle.lokr.backward(x64, g64, w1=w1_64, w2=new64, scale=0.5)["dx"], dtype) < TOL["store_out"][dtype]


def test_flat_parameter_arena_updates_reach_the_planes():
    """AdapterGradSync.flat_parameters(): the factors become views of one arena and the optimizer updates the ARENA (a different
    tensor: the factors' version counters never move, and re-homing swaps their storage after the planes were packed once).  The
    forward and dx after the step must use the updated w2."""
    from lycoris_amd.grad_sync import AdapterGradSync
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(93)
    M, a, c, d = 160, 8, 40, 80
    x, x64 = rnd((M, a * d), dtype, gen)
    g, g64 = rnd((M, a * c), dtype, gen, 0.05)
    w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
    w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
    w1, w2 = torch.nn.Parameter(w1), torch.nn.Parameter(w2)
    y0 = _lin(x, w1, w2)  # planes packed from the ORIGINAL storage
    sync = AdapterGradSync([w1, w2])
    flats = sync.flat_parameters()
    assert w2.data.untyped_storage().data_ptr() == flats[0].data.untyped_storage().data_ptr()
    assert err(_lin(x, w1, w2), oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=0.5), dtype) < TOL["store_out"][dtype]  # re-homed: repacked
    opt = torch.optim.SGD(flats, lr=3.0)
    sync.zero_grad()
    xg = x.clone().requires_grad_(True)
    _lin(xg, w1, w2).backward(g)
    sync.finish()
    v0 = w2._version
    opt.step()
    assert w2._version == v0  # updated through the flat leaf
    new1, new2 = w1.detach().double().cpu().numpy(), w2.detach().double().cpu().numpy()
    assert oracle.general.rel_err(new2, w2_64) > 1e-3  # a real step
    assert err(_lin(x, w1, w2), oracle.lokr.forward(x64, w1=new1, w2=new2, scale=0.5), dtype) < TOL["store_out"][dtype]
    xg2 = x.clone().requires_grad_(True)
    dx, = torch.autograd.grad(_lin(xg2, w1, w2), [xg2], g)
    assert err(dx, oracle.lokr.backward(x64, g64, w1=new1, w2=new2, scale=0.5)["dx"], dtype) < TOL["store_out"][dtype]
    sync.remove()


def test_plane_refresh_drops_entries_whose_storage_moved():
    """refresh_lokr_planes(force=True) packs from pointers cached per entry: a parameter whose storage was replaced since its last
    own layer call (`.data =` swap, `.to()`, offload) must be dropped, not read -- and its next call repacks."""
    from lycoris_amd import ops
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(92)
    M, a, c, d = 96, 8, 40, 40
    x, x64 = rnd((M, a * d), dtype, gen)
    w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
    ws = []
    for _ in range(3):
        w, w64 = rnd((c, d), torch.float32, gen, 0.1)
        ws.append((torch.nn.Parameter(w), w64))
    for w, _ in ws:
        _lin(x, w1, w)
    ext = ops._DISPATCH["ext"]
    ops.refresh_lokr_planes(force=True)  # (also purges the entries of parameters earlier tests have dropped)
    n0 = ext.planes_cache_size()
    # swap the storage of the middle one (new allocation), free the old one
    new, new64 = rnd((c, d), torch.float32, gen, 0.1)
    ws[1][0].data = new.clone()
    del new
    torch.cuda.empty_cache()
    ops.refresh_lokr_planes(force=True)  # must not touch the dead pointer
    torch.cuda.synchronize()
    assert ext.planes_cache_size() == n0 - 1
    y = _lin(x, w1, ws[1][0])
    assert err(y, oracle.lokr.forward(x64, w1=w1_64, w2=new64, scale=0.5), dtype) < TOL["store_out"][dtype]
    assert ext.planes_cache_size() == n0


def test_planes_skipped_for_inference_tensors():
    """a parameter created under torch.inference_mode() has no version counter (`_version` raises): no planes, no crash"""
    from lycoris_amd import ops
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(93)
    M, a, c, d = 64, 8, 40, 40
    x, x64 = rnd((M, a * d), dtype, gen)
    with torch.inference_mode():
        w1 = (torch.randn(a, a, generator=gen) * 0.3).to(dev())
        w2 = torch.nn.Parameter((torch.randn(c, d, generator=gen) * 0.1).to(dev()), requires_grad=False)
        y = _lin(x, w1, w2)
    ref = oracle.lokr.forward(x64, w1=w1.double().cpu().numpy(), w2=w2.double().cpu().numpy(), scale=0.5)
    assert err(y, ref, dtype) < TOL["store_out"][dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_one_launch_refresh_through_a_table_equals_the_grouped_launches(dtype):
    """lyc_lokr_pack_group_ws (round 6): every layer in ONE pack launch through a device table of the descriptors -- the same plane bytes
    as lyc_lokr_pack_group (one launch per 28 layers); a second call on the written table (table_valid = 1) after the factors changed in
    place packs the NEW values.  70 layers: nn.Linear factors, a channels_last Conv2d factor, low-rank pairs, one item without planes."""
    import ctypes
    lib = N.load()
    code = N.dtype_code(dtype)
    gen = torch.Generator().manual_seed(123)
    st = N.stream_ptr(dev())
    keep, ref_items, new_items, planes = [], [], [], []
    n = 70
    for k in range(n):
        c, d = [(40, 40), (160, 160), (80, 640), (160, 40), (1280, 160)][k % 5]
        taps = 9 if k % 11 == 3 else 1
        lowrank = k % 7 == 5 and taps == 1
        nf, nb = int(lib.lyc_lokr_planes_bytes(c, d, taps, 0)), int(lib.lyc_lokr_planes_bytes(c, d, taps, 1))
        bufs = [torch.full((nf,), 0xCD, dtype=torch.uint8, device=dev()) for _ in range(2)] + [torch.full((nb,), 0xCD, dtype=torch.uint8, device=dev()) for _ in range(2)]
        if k == 17:  # nothing to pack for this one (its buffers keep their fill pattern in both runs)
            w = (torch.randn(c, d, generator=gen) * 0.1).to(dev())
            it = lambda pf, pb, w=w: N.LokrPackItem(N.ptr(w), d, 1, 0, c, d, 1, None, None, None, None, 0)
            taps = 1
            srcs = [w]
        elif lowrank:
            wa = (torch.randn(c, 16, generator=gen) * 0.1).to(dev())
            wb = (torch.randn(16, d, generator=gen) * 0.1).to(dev())
            it = lambda pf, pb, wa=wa, wb=wb: N.LokrPackItem(None, 0, 0, 0, c, d, 1, N.ptr(pf), N.ptr(pb), N.ptr(wa), N.ptr(wb), 16)
            srcs = [wa, wb]
        elif taps == 9:
            w = (torch.randn(c, d, 3, 3, generator=gen) * 0.1).to(dev()).contiguous(memory_format=torch.channels_last)
            it = lambda pf, pb, w=w: N.LokrPackItem(N.ptr(w), w.stride(0), w.stride(1), w.stride(3), c, d, 9, N.ptr(pf), N.ptr(pb), None, None, 0)
            srcs = [w]
        else:
            w = (torch.randn(c, d, generator=gen) * 0.1).to(dev())
            it = lambda pf, pb, w=w: N.LokrPackItem(N.ptr(w), d, 1, 0, c, d, 1, N.ptr(pf), N.ptr(pb), None, None, 0)
            srcs = [w]
        ref_items.append(it(bufs[0], bufs[2]))
        new_items.append(it(bufs[1], bufs[3]))
        planes.append(bufs)
        keep += srcs
    ra, na = (N.LokrPackItem * n)(*ref_items), (N.LokrPackItem * n)(*new_items)
    tb = int(lib.lyc_lokr_pack_table_bytes(ctypes.cast(na, ctypes.c_void_p), n))
    assert tb > 0
    table = torch.empty(tb, dtype=torch.uint8, device=dev())
    for valid in (0, 1):
        N.call("lyc_lokr_pack_group", ctypes.cast(ra, ctypes.c_void_p), n, code, st)
        N.call("lyc_lokr_pack_group_ws", ctypes.cast(na, ctypes.c_void_p), n, code, N.ptr(table), tb, valid, st)
        torch.cuda.synchronize()
        for k, b in enumerate(planes):
            assert torch.equal(b[0], b[1]) and torch.equal(b[2], b[3]), (k, valid)
        for t in keep:  # the factors move (an optimizer step): the second round packs the new values through the SAME table
            t.mul_(1.25).add_(0.01)
        for b in planes:
            for t in b:
                t.fill_(0xCD)
