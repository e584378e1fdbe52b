"""GPU: LoHa operand planes kept per parameter set and rebuilt once per optimizer step (round 6: lyc_loha_rebuild_group,
csrc/torch_ops.cpp loha_plane_for).

Reference: HadaWeight.forward (lycoris/functional/loha.py:10-16) rebuilds dW = (w1a w1b) * (w2a w2b) * scale in every layer call.  The
native forward did the same with one launch per layer; the factors are parameters, so the plane of a layer is now cached and all stale
planes are rebuilt by grouped launches at the first layer call after a step boundary.  Checked: the grouped rebuild writes the bytes the
per-layer entry point writes; results with the cache equal results without it, bit for bit; the cache follows in-place parameter updates
(optimizer steps that move the version counter and `.data` writes that do not); an explicit refresh inside a captured graph."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, err, rnd
from lycoris_amd import _native as N
from lycoris_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _factors(O, I, r, gen, param=False):
    out = []
    for shape, sc in (((O, r), 0.1), ((r, I), 1.0), ((O, r), 0.1), ((r, I), 1.0)):
        t = (torch.randn(*shape, generator=gen) * sc).to(DEV)
        out.append(torch.nn.Parameter(t) if param else t)
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_grouped_rebuild_writes_the_bytes_of_the_per_layer_rebuild(dtype):
    lib = N.load()
    code = N.dtype_code(dtype)
    gen = torch.Generator().manual_seed(3)
    st = N.stream_ptr(torch.device(DEV))
    shapes = [(1280, 1280, 32), (640, 2560, 32), (320, 2880, 16), (72, 264, 4), (1280, 640, 28)] * 12  # 60 layers: two launches
    items, keep = [], []
    for k, (O, I, r) in enumerate(shapes):
        assert lib.lyc_loha_plane_cacheable(None, None, None, None, I, O, r, code) == 1
        f = _factors(O, I, r, gen)
        nb = int(lib.lyc_loha_workspace_bytes(O, I, code))
        p_ref = torch.full((nb,), 0xCD, dtype=torch.uint8, device=DEV)
        p_grp = torch.full((nb,), 0xCD, dtype=torch.uint8, device=DEV)
        x = torch.zeros(8, I, dtype=dtype, device=DEV)
        y = torch.empty(8, O, dtype=dtype, device=DEV)
        alpha = 0.25 + 0.01 * k
        N.call("lyc_loha_linear_fwd", N.ptr(x), N.ptr(f[0]), N.ptr(f[1]), N.ptr(f[2]), N.ptr(f[3]), N.ptr(p_ref), N.ptr(y), 8, I, O, r, alpha, code, st)
        items.append(N.LohaPlaneItem(N.ptr(f[0]), N.ptr(f[1]), N.ptr(f[2]), N.ptr(f[3]), N.ptr(p_grp), O, I, r, alpha))
        keep.append((f, p_ref, p_grp, x, y))
    arr = (N.LohaPlaneItem * len(items))(*items)
    N.call("lyc_loha_rebuild_group", ctypes.cast(arr, ctypes.c_void_p), len(items), code, st)
    torch.cuda.synchronize()
    for k, (f, p_ref, p_grp, x, y) in enumerate(keep):
        assert torch.equal(p_ref, p_grp), k
    # what the cache cannot hold is refused: rank > 32, an odd width
    assert lib.lyc_loha_plane_cacheable(None, None, None, None, 1280, 1280, 40, code) == 0
    assert lib.lyc_loha_plane_cacheable(None, None, None, None, 1284, 1280, 32, code) == 0
    assert lib.lyc_loha_plane_cacheable(None, None, None, None, 1280, 1280, 32, N.LYC_F32) == 0


def _run(x, g, fs, alpha):
    y = ops.loha_linear(x, *fs, alpha)
    grads = torch.autograd.grad(y, [x] + list(fs), g)
    return y, grads


def test_results_with_the_cache_equal_results_without_it():
    gen = torch.Generator().manual_seed(5)
    dtype = torch.bfloat16
    M, I, O, r = 200, 640, 320, 32
    x, _ = rnd((M, I), dtype, gen)
    g, _ = rnd((M, O), dtype, gen, 0.1)
    x.requires_grad_(True)
    fs = _factors(O, I, r, gen, param=True)
    y1, g1 = _run(x, g, fs, 0.5)
    y1b, _ = _run(x, g, fs, 0.5)       # second call: the cached plane
    ops.lokr_planes_cache(False)
    try:
        y0, g0 = _run(x, g, fs, 0.5)
    finally:
        ops.lokr_planes_cache(True)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(y1, y1b) and torch.equal(g0[0], g1[0])
    for a, b in zip(g0[1:], g1[1:]):
        assert err(b, a.double().cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("how", ["optimizer_step", "data_write_after_backward", "other_alpha"])
def test_the_cache_follows_the_parameters(how):
    gen = torch.Generator().manual_seed(7)
    dtype = torch.bfloat16
    M, I, O, r = 96, 320, 640, 16
    layers = []
    for _ in range(5):
        x, x64 = rnd((M, I), dtype, gen)
        layers.append((x, x64, _factors(O, I, r, gen, param=True)))
    g, _ = rnd((M, O), dtype, gen, 0.1)
    alpha = 0.5
    for x, x64, fs in layers:  # step 1: planes are built
        y = ops.loha_linear(x, *fs, alpha)
        torch.autograd.grad(y, fs, g)
    delta = 0.01
    if how == "optimizer_step":
        params = [p for _, _, fs in layers for p in fs]
        for p in params:
            p.grad = torch.full_like(p, 1.0)
        torch.optim.SGD(params, lr=delta).step()   # p -= delta, in place: version counters move
        shift = -delta
    elif how == "data_write_after_backward":
        for _, _, fs in layers:                    # a write the version counter does not see; the backward above was the step boundary
            for p in fs:
                p.data.add_(delta)
        shift = delta
    else:
        shift, alpha = 0.0, 0.8
    for x, x64, fs in layers:
        y = ops.loha_linear(x, *fs, alpha)
        want = oracle.loha.forward(x64, *[p.detach().double().cpu().numpy() for p in fs], alpha)
        assert err(y, want, dtype) <= TOL["loha_store"][dtype], how
        if shift:  # and the values really moved
            old = oracle.loha.forward(x64, *[(p.detach().double().cpu().numpy() - shift) for p in fs], alpha)
            assert float(np.abs(old - want).max()) > 1e-3


def test_a_captured_step_refreshes_the_planes_of_its_own_parameters():
    gen = torch.Generator().manual_seed(11)
    dtype = torch.bfloat16
    M, I, O, r = 64, 320, 320, 32
    x, x64 = rnd((M, I), dtype, gen)
    fs = _factors(O, I, r, gen, param=True)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st), torch.no_grad():
        ops.loha_linear(x, *fs, 0.5)  # warm-up: the plane exists
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=st):
            ops.refresh_lokr_planes(force=True)
            y = ops.loha_linear(x, *fs, 0.5)
        for p in fs:
            p.add_(0.02)              # "the optimizer": between replays
        gph.replay()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    want = oracle.loha.forward(x64, *[p.detach().double().cpu().numpy() for p in fs], 0.5)
    assert err(y, want, dtype) <= TOL["loha_store"][dtype]
