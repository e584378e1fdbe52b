"""CPU: the HOST side of the launch planning, through the C ABI (pure host functions: no GPU call).

`lyc_lokr_conv2d_planes_ok` (capi.hip plan_kconv: does the LDS-patch kernel take this Conv2d geometry, and with which row tile),
`lyc_lokr_linear_planes_ok`, `lyc_lokr_planes_bytes`, the workspace-size functions and `lyc_*_wgrad_deferrable`'s shape part, over
every Conv2d / Linear shape of the SDXL and SD1.5 workloads (benchmarks/sdxl_shapes.py, sd15_shapes.py) -- what bench.py's layers
actually run through is pinned here without a GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.sd15_shapes import sd15_unet_layers  # noqa: E402
from benchmarks.sdxl_shapes import sdxl_unet_layers  # noqa: E402
from lycoris_amd import _native as N  # noqa: E402

BF16 = 2  # LYC_BF16 (include/lycoris_amd.h)


def _convs(layers):
    return [l for l in layers if l["kind"] == "conv"]


def test_dtype_codes_match_the_header():
    assert (N.LYC_F32, N.LYC_F16, N.LYC_BF16) == (0, 1, 2) and BF16 == N.LYC_BF16


@pytest.mark.parametrize("model", ["sdxl", "sd15"])
def test_every_3x3_stride1_conv_of_the_workloads_takes_the_patch_kernel(model):
    lib = N.load()
    layers = sdxl_unet_layers(1) if model == "sdxl" else sd15_unet_layers(4)
    seen = 0
    for l in _convs(layers):
        a = 8
        if l["C"] % a or l["O"] % a or (l["C"] // a) % 8 or (l["O"] // a) % 8:
            continue  # conv_in / conv_out (4 channels): the im2col lowering, not this kernel
        c, d, k, s, p = l["O"] // a, l["C"] // a, l["k"], l["stride"], l["pad"]
        geo = (l["B"], l["H"], l["W"], a, a, c, d, k, k, s, s, p, p, 1, 1)
        fwd = lib.lyc_lokr_conv2d_planes_ok(*geo, BF16, 0)
        bwd = lib.lyc_lokr_conv2d_planes_ok(*geo, BF16, 1)
        if k == 1:
            continue  # 1x1: the row kernels (ops.py routes them before this check)
        # the patch of the smallest tile (4 x 4 pixels + halo, ALL channels resident) must fit 140 KiB of LDS: the two C = 2560
        # up-block convs of SDXL (d = 320: 6 * 6 * 8 * 328 * 2 B = 189 KB) do not -- they stay on the row-gather kernels (kron3 GM = 2)
        gp = d + (8 if (d // 8) % 2 == 0 else 0)
        ph = (4 - 1) * s + (k - 1) + 1  # source rows under a 4 x 4 destination tile
        fits = ph * ph * a * gp * 2 <= 140 * 1024
        fits_b = (4 - 1 + k) * (4 - 1 + k) * a * (c + (8 if (c // 8) % 2 == 0 else 0)) * 2 <= 140 * 1024
        assert (fwd in (2, 4, 8)) == fits, (l, fwd)             # the forward patch kernel takes every other 3x3 conv, stride 1 and 2
        assert (bwd in (2, 4, 8)) == (s == 1 and fits_b), (l, bwd)  # transposed convolution of a strided layer: row-gather kernel
        # planes: both roles, multiples of the 2 KiB unit, the same number of units for square factors
        bf, bb = lib.lyc_lokr_planes_bytes(c, d, k * k, 0), lib.lyc_lokr_planes_bytes(c, d, k * k, 1)
        assert bf % 2048 == 0 and bb % 2048 == 0 and bf >= c * d * k * k * 4 and bb >= c * d * k * k * 4
        seen += 1
    assert seen >= (10 if model == "sdxl" else 3), seen  # distinct 3x3 shapes (the tables carry a count per shape)


def test_row_tile_choice_keeps_about_one_workgroup_per_cu():
    """plan_kconv takes the largest pixel tile that still leaves >= 200 workgroups: the big images get MI = 8, the 32 x 32 ones less"""
    lib = N.load()
    big = lib.lyc_lokr_conv2d_planes_ok(1, 128, 128, 8, 8, 40, 40, 3, 3, 1, 1, 1, 1, 1, 1, BF16, 0)
    small = lib.lyc_lokr_conv2d_planes_ok(1, 32, 32, 8, 8, 160, 160, 3, 3, 1, 1, 1, 1, 1, 1, BF16, 0)
    assert big == 8 and small in (2, 4) and small < big
    assert lib.lyc_lokr_conv2d_planes_ok(1, 32, 32, 8, 8, 160, 160, 3, 3, 1, 1, 1, 1, 1, 1, N.LYC_F32, 0) == 0      # 16-bit only
    assert lib.lyc_lokr_conv2d_planes_ok(1, 32, 32, 3, 3, 160, 160, 3, 3, 1, 1, 1, 1, 1, 1, BF16, 0) == 0          # factor 3
    assert lib.lyc_lokr_conv2d_planes_ok(1, 32, 32, 8, 8, 156, 160, 3, 3, 1, 1, 1, 1, 1, 1, BF16, 0) == 0          # c % 8 != 0


@pytest.mark.parametrize("model", ["sdxl", "sd15"])
def test_every_linear_layer_of_the_workloads_is_on_the_plane_path_and_has_a_workspace(model):
    lib = N.load()
    layers = sdxl_unet_layers(1) if model == "sdxl" else sd15_unet_layers(4)
    n = 0
    for l in layers:
        if l["kind"] != "linear":
            continue
        a, c, d = 8, l["O"] // 8, l["I"] // 8
        assert lib.lyc_lokr_linear_planes_ok(l["M"], a, a, c, d, BF16) == 1, l
        assert lib.lyc_lokr_linear_planes_ok(l["M"], a, a, c, d, N.LYC_F32) == 0
        assert lib.lyc_lokr_bwd_workspace_bytes(l["M"], a, a, c, d, BF16) > 0, l
        assert lib.lyc_lokr_planes_bytes(c, d, 1, 0) % 2048 == 0
        n += 1
    assert n >= 5
