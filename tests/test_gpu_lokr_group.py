"""GPU: sibling projections in one launch (lyc_lokr_linear_fwd_group / _bwd_group, csrc/kron4.h kron4_group_kernel; round 4).

The reference runs one LokrModule.forward per projection (modules/lokr.py:436-486); to_q / to_k / to_v of an attention block have
the same shapes, and a single 1024-token projection fills a third of the chip.  The grouped entry points must give the SAME BITS as
the per-layer *_planes entry points (same MFMA accumulation order, whatever tile plan the group is given), match the oracle, and
leave the w1-gradient partials where lyc_lokr_wgrad_group expects them."""
import ctypes

import pytest
import torch

import oracle
from gpu_util import TOL, err, rnd
from lycoris_amd import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (list of M, a, c, d): q/k/v of a 1280-wide block, k/v of its cross-attention, ragged row counts, a 640-wide block (64-column tiles),
# five items (two launches), a single item
GROUPS = [([1024, 1024, 1024], 8, 160, 160), ([77, 77], 8, 160, 256), ([1024, 77, 333], 8, 160, 160), ([4096, 4096, 4096], 8, 80, 80),
          ([64, 200, 64, 1, 130], 4, 24, 40), ([50], 16, 16, 8)]


def _planes(w2, c, d, code):
    lib = N.load()
    pf = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, 1, 0)), dtype=torch.uint8, device=DEV)
    pb = torch.empty(int(lib.lyc_lokr_planes_bytes(c, d, 1, 1)), dtype=torch.uint8, device=DEV)
    N.call("lyc_lokr_pack_w2", N.ptr(w2), d, 1, 0, None, 0, 0, None, 0, 0, 0, 0, c, d, 1, N.ptr(pf), N.ptr(pb), code, N.stream_ptr(torch.device(DEV)))
    return pf, pb


@pytest.mark.parametrize("with_base", [False, True], ids=["delta", "base_plus_delta"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("group", GROUPS, ids=[f"M{'_'.join(map(str, g[0]))}_a{g[1]}_c{g[2]}_d{g[3]}" for g in GROUPS])
def test_grouped_projections_equal_the_per_layer_launches_and_the_oracle(group, dtype, with_base):
    Ms, a, c, d = group
    lib = N.load()
    code = N.dtype_code(dtype)
    gen = torch.Generator().manual_seed(sum(Ms) + a + c + d)
    st = N.stream_ptr(torch.device(DEV))
    alpha = 0.7
    n = len(Ms)
    fw, bw = (N.LinearGroupItem * n)(), (N.LinearGroupItem * n)()
    wg = (N.WgradItem * n)()
    keep, per = [], []
    for k, M in enumerate(Ms):
        assert lib.lyc_lokr_linear_planes_ok(M, a, a, c, d, code) == 1
        x, x64 = rnd((M, a * d), dtype, gen)
        g, g64 = rnd((M, a * c), dtype, gen, 0.1)
        base, b64 = rnd((M, a * c), dtype, gen)
        w1, w1_64 = rnd((a, a), torch.float32, gen, 0.3)
        w2, w2_64 = rnd((c, d), torch.float32, gen, 0.1)
        pf, pb = _planes(w2, c, d, code)
        y, dx = torch.empty(M, a * c, device=DEV, dtype=dtype), torch.empty(M, a * d, device=DEV, dtype=dtype)
        nws = max(int(lib.lyc_lokr_bwd_workspace_bytes(M, a, a, c, d, code)), 16)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        dw1, dw2 = torch.zeros(a, a, device=DEV), torch.zeros(c, d, device=DEV)
        fw[k] = N.LinearGroupItem(N.ptr(x), N.ptr(w1), N.ptr(pf), N.ptr(base) if with_base else None, N.ptr(y), None, M, alpha)
        bw[k] = N.LinearGroupItem(N.ptr(g), N.ptr(w1), N.ptr(pb), N.ptr(x), N.ptr(dx), N.ptr(ws), M, alpha)
        wg[k] = N.WgradItem(N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(dw1), N.ptr(dw2), N.ptr(ws), M, a, a, c, d, alpha)
        # the per-layer launches on the same operands
        y1, dx1 = torch.empty_like(y), torch.empty_like(dx)
        ws1, d1, d2 = torch.empty_like(ws), torch.zeros_like(dw1), torch.zeros_like(dw2)
        N.call("lyc_lokr_linear_fwd_planes", N.ptr(x), N.ptr(w1), N.ptr(pf), N.ptr(base) if with_base else None, N.ptr(y1), M, a, a, c, d, alpha,
               code, st)
        N.call("lyc_lokr_linear_bwd_planes", N.ptr(g), N.ptr(x), N.ptr(w1), N.ptr(pb), N.ptr(dx1), N.ptr(d1), N.ptr(d2), N.ptr(ws1), M, a, a,
               c, d, alpha, code, st)
        want = oracle.lokr.backward(x64, g64, w1=w1_64, w2=w2_64, scale=alpha)
        yo = oracle.lokr.forward(x64, w1=w1_64, w2=w2_64, scale=alpha)
        per.append((y, dx, dw1, dw2, y1, dx1, d1, d2, want, yo, b64))
        keep += [x, g, base, w1, w2, pf, pb, ws, ws1]
    N.call("lyc_lokr_linear_fwd_group", ctypes.cast(fw, ctypes.c_void_p), n, a, a, c, d, code, st)
    N.call("lyc_lokr_linear_bwd_group", ctypes.cast(bw, ctypes.c_void_p), n, a, a, c, d, code, st)
    N.call("lyc_lokr_wgrad_group", ctypes.cast(wg, ctypes.c_void_p), n, code, st)
    torch.cuda.synchronize()
    for k, (y, dx, dw1, dw2, y1, dx1, d1, d2, want, yo, b64) in enumerate(per):
        assert torch.equal(y, y1), (k, "forward differs from the per-layer launch")
        assert torch.equal(dx, dx1), (k, "dx differs from the per-layer launch")
        assert err(dw1, d1.double().cpu().numpy()) <= 1e-5 and err(dw2, d2.double().cpu().numpy()) <= 1e-5, k
        assert err(dx, want["dx"], dtype) <= TOL["store_out"][dtype], (k, err(dx, want["dx"], dtype))
        assert err(dw1, want["w1"]) <= TOL["f32_out"][dtype] and err(dw2, want["w2"]) <= TOL["f32_out"][dtype], k
        assert err(y, yo + (b64 if with_base else 0), dtype) <= TOL["store_out"][dtype], k


def test_group_calls_refuse_what_they_cannot_run():
    code = N.dtype_code(torch.bfloat16)
    x = torch.zeros(8, 64, device=DEV, dtype=torch.bfloat16)
    w1 = torch.zeros(8, 8, device=DEV)
    it = (N.LinearGroupItem * 2)()
    it[0] = N.LinearGroupItem(N.ptr(x), N.ptr(w1), N.ptr(x), None, N.ptr(x), None, 8, 1.0)
    it[1] = N.LinearGroupItem(N.ptr(x), N.ptr(w1), N.ptr(x), N.ptr(x), N.ptr(x), None, 8, 1.0)  # one item with a base, one without
    with pytest.raises(RuntimeError, match="same operands"):
        N.call("lyc_lokr_linear_fwd_group", ctypes.cast(it, ctypes.c_void_p), 2, 8, 8, 8, 8, code, None)
    with pytest.raises(RuntimeError, match="fast path"):
        N.call("lyc_lokr_linear_fwd_group", ctypes.cast(it, ctypes.c_void_p), 1, 8, 8, 6, 6, code, None)  # c, d not multiples of 8
    N.call("lyc_lokr_linear_fwd_group", None, 0, 8, 8, 8, 8, code, None)  # an empty group is fine
