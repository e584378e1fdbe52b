"""GPU: the LDS-DMA rank-r launch of round 6 (csrc/lowrank4.h: bneck4_kernel / bneck4_group_kernel / bneck4_sum_kernel) behind the LoCon
entry points, against the float64 oracle AND next to the register-staged kernel of rounds 1-5 on the same inputs (ops.locon_reg_staged,
`LYC_BNECK_REG` in the C ABI's dtype argument).

Reference math: lycoris/functional/locon.py:64-85, modules/locon.py:286-332 (one LoConModule.forward per projection)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, check, err, rnd
from lycoris_amd import _native as N
from lycoris_amd import ops

pytestmark = pytest.mark.gpu

# (M, I, O, r): what bneck4_kernel covers (r <= 16, r % 4 == 0, I % 8 == 0, O % 8 == 0) at its plan's corners
SHAPES = [
    (1024, 1280, 1280, 16),   # SDXL attention projection: 8 waves x 16 rows, 4 column slices, the whole K in the prologue
    (1000, 328, 200, 16),     # ragged rows, K not a multiple of 32, a half column pair at the end
    (300, 1280, 1280, 8),     # rank 8: two of the four 16-byte chunks of a factor row are out of range
    (77, 2048, 640, 4),       # rank 4, 5 row tiles, long K (ring of 8 steps re-issued)
    (50, 40, 24, 12),         # rank 12, K = 40 (two k steps: six of the eight waves have none), 24 output columns
    (1, 1280, 320, 16),       # one row
    (9000, 64, 32, 16),       # M >= 8192: 4 waves x 32 rows, two workgroups per CU, ragged end
    (4100, 640, 5120, 16),    # wide output: more column slices than the chip has CUs per row tile -> two rounds
    (260, 10240, 640, 16),    # K = 10240: 320 k steps over 8 waves, ring reuse x 5
]


def _run(shape, dtype, reg):
    M, I, O, r = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x, x64 = rnd((M, I), dtype, gen)
    g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
    down, d64 = rnd((r, I), torch.float32, gen, 0.05)
    up, u64 = rnd((O, r), torch.float32, gen, 0.05)
    x.requires_grad_(True); down.requires_grad_(True); up.requires_grad_(True)
    was = ops.locon_reg_staged(reg)
    try:
        y = ops.locon_linear(x, down, up, 1.5)
        dx, dd, du = torch.autograd.grad(y, [x, down, up], g)
        torch.cuda.synchronize()
    finally:
        ops.locon_reg_staged(was)
    return (y, dx, dd, du), (x64, g64, d64, u64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_both_kernels_match_the_oracle(shape, dtype):
    (y, dx, dd, du), (x64, g64, d64, u64) = _run(shape, dtype, False)
    (y0, dx0, dd0, du0), _ = _run(shape, dtype, True)
    y_ref = oracle.locon.forward(x64, d64, u64, 1.5)
    dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, 1.5)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_ref, dtype), "d_down": err(dd, dd_ref), "d_up": err(du, du_ref),
            "y_reg": err(y0, y_ref, dtype), "dx_reg": err(dx0, dx_ref, dtype), "d_down_reg": err(dd0, dd_ref), "d_up_reg": err(du0, du_ref)}
    s, f = TOL["store_out"][dtype], TOL["f32_out"][dtype]
    check(f"locon_bneck4[{shape},{dtype}]", errs, {"y": s, "dx": s, "d_down": f, "d_up": f, "y_reg": s, "dx_reg": s, "d_down_reg": f, "d_up_reg": f})


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
# (M, I, O, r, n, fused): fused = the plan of bneck4_sum_kernel covers the set (lowrank4.h: bneck4_make_plan -- at least 256 rows, n <= 3,
# the n tiles of column pairs in the LDS); the others take lyc_locon_linear_bwd_group + lyc_sum_rows
@pytest.mark.parametrize("case", [(1024, 1280, 1280, 16, 3, True), (77, 2048, 1280, 16, 2, False), (4096, 640, 640, 8, 3, True), (300, 96, 40, 4, 2, True),
                                  (130, 96, 40, 4, 4, False), (9000, 64, 64, 16, 2, True), (1000, 328, 200, 12, 3, True)],
                         ids=lambda c: f"M{c[0]}_I{c[1]}_O{c[2]}_r{c[3]}_n{c[4]}")
def test_sibling_set_backward_sums_the_input_gradient_in_one_launch(case, dtype):
    """training configuration (factor gradients parked for the grouped launch): dx of a tensor that n projections read comes from
    lyc_locon_linear_bwd_group_sum -- one rounding of the fp32 sum -- and equals the oracle's sum; the rounds 1-5 route (n dx results +
    lyc_sum_rows) on the same inputs stays within its own, looser bound; every factor gradient equals the oracle's on both routes"""
    M, I, O, r, n, fused = case
    gen = torch.Generator().manual_seed(sum(case[:5]))
    x, x64 = rnd((M, I), dtype, gen)
    f64, gs, downs0, ups0 = [], [], [], []
    for i in range(n):
        dn, d64 = rnd((r, I), torch.float32, gen, 0.05)
        up, u64 = rnd((O, r), torch.float32, gen, 0.05)
        g, g64 = rnd((M, O), dtype, gen, 1.0 / np.sqrt(O))
        downs0.append(dn); ups0.append(up); gs.append(g); f64.append((d64, u64, g64))
    alphas = [0.5, 1.0, 2.0, 0.25][:n]
    res = {}
    for reg in (False, True):
        downs = [torch.nn.Parameter(t.clone()) for t in downs0]
        ups = [torch.nn.Parameter(t.clone()) for t in ups0]
        params = [p for pair in zip(downs, ups) for p in pair]
        for p in params:
            p.grad = torch.zeros_like(p)
        was = ops.locon_reg_staged(reg)
        ops.fused_grad_accumulation(True, callback=lambda p: None)
        try:
            xr = x.clone().requires_grad_(True)
            ys = ops.locon_linear_group(xr, downs, ups, alphas)
            torch.autograd.backward(ys, gs)
            torch.cuda.synchronize()
        finally:
            ops.fused_grad_accumulation(False, None)
            ops.locon_reg_staged(was)
        res[reg] = (ys, xr.grad, [p.grad for p in params])
    errs, bounds = {}, {}
    dx_want = 0.0
    for i in range(n):
        d64, u64, g64 = f64[i]
        dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, alphas[i], None)
        dx_want = dx_want + dx_ref
        for reg, tag in ((False, ""), (True, "_reg")):
            errs[f"y{i}{tag}"], bounds[f"y{i}{tag}"] = err(res[reg][0][i], oracle.locon.forward(x64, d64, u64, alphas[i], None), dtype), TOL["store_out"][dtype]
            errs[f"d_down{i}{tag}"], bounds[f"d_down{i}{tag}"] = err(res[reg][2][2 * i], dd_ref), TOL["f32_out"][dtype]
            errs[f"d_up{i}{tag}"], bounds[f"d_up{i}{tag}"] = err(res[reg][2][2 * i + 1], du_ref), TOL["f32_out"][dtype]
    # one rounding of the fp32 sum where the set is fused; n roundings + one more on the two-launch route
    errs["dx"], bounds["dx"] = err(res[False][1], dx_want, dtype if fused else None), (1 if fused else 3) * TOL["store_out"][dtype]
    errs["dx_reg"], bounds["dx_reg"] = err(res[True][1], dx_want), 3 * TOL["store_out"][dtype]       # n roundings + one more
    check(f"locon_sibling_sum[{case},{dtype}]", errs, bounds)


def test_the_fused_sum_refuses_what_the_kernel_does_not_cover_and_launches_nothing():
    dev = torch.device("cuda:0")
    M, I, O = 512, 64, 64
    lib = N.load()

    def items(r, n=2):
        arr = (N.LoconGroupItem * n)()
        keep = []
        for k in range(n):
            g = torch.zeros(M, O, dtype=torch.bfloat16, device=dev)
            dn = torch.zeros(r, I, device=dev)
            up = torch.zeros(O, r, device=dev)
            mid = torch.full((M, r), 7.0, device=dev)
            keep += [g, dn, up, mid]
            arr[k] = N.LoconGroupItem(g.data_ptr(), dn.data_ptr(), up.data_ptr(), mid.data_ptr(), None, M, 1.0)
        return arr, keep

    dx = torch.full((M, I), 3.0, dtype=torch.bfloat16, device=dev)
    st = N.stream_ptr(dev)
    for r, flags in ((20, 0), (6, 0), (16, 0x20000)):  # rank above 16, rank % 4, the register-staged switch
        arr, keep = items(r)
        rc = lib.lyc_locon_linear_bwd_group_sum(ctypes.cast(arr, ctypes.c_void_p), 2, I, O, r, ctypes.c_void_p(dx.data_ptr()), N.LYC_BF16 | flags, st)
        torch.cuda.synchronize()
        assert rc == 2, (r, flags, rc)  # LYC_ERR_UNSUPPORTED
        assert float(dx.float().min()) == 3.0 and float(keep[3].min()) == 7.0  # nothing was written
    arr, keep = items(16)
    rc = lib.lyc_locon_linear_bwd_group_sum(ctypes.cast(arr, ctypes.c_void_p), 2, I, O, 16, ctypes.c_void_p(dx.data_ptr()), N.LYC_BF16, st)
    torch.cuda.synchronize()
    assert rc == 0 and float(dx.float().abs().max()) == 0.0 and float(keep[3].abs().max()) == 0.0  # zeros in -> zeros out, mid written
