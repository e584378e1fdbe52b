"""GPU parity of plug point 2, the functional API (lycoris_amd.functional.{locon,loha,lokr}.bypass_forward_diff), against
the oracle with the REFERENCE's gamma conventions:

  * locon / loha: gamma is the final multiplier (functional/locon.py:52, loha.py:14);
  * lokr: gamma is *alpha*; the function divides by the rank it infers from the low-rank factors and uses scale 1 when
    both factors are full matrices (functional/lokr.py:135-141, 171-172);
  * loha conv: the 2-D [r, I*kh*kw] factors are viewed through extra_args["_conv_shape"].
"""
import numpy as np
import pytest
import torch

import oracle
from gpu_util import TOL, check, err, rnd, loha_cast_pair

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["f32", "bf16"]


def _grads(y, ts, g):
    return torch.autograd.grad(y, ts, g)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("conv", [False, True], ids=["linear", "conv3x3"])
def test_locon_functional(dtype, conv):
    from lycoris_amd.functional import locon as F
    gen = torch.Generator().manual_seed(1)
    r, I, O, gamma = 8, 32, 48, 0.25
    if conv:
        x, x64 = rnd((2, I, 9, 8), dtype, gen)
        down, d64 = rnd((r, I, 3, 3), torch.float32, gen, 0.1)
        up, u64 = rnd((O, r, 1, 1), torch.float32, gen, 0.1)
        ea = {"stride": 1, "padding": 1, "dilation": 1, "groups": 1}
        g, g64 = rnd((2, O, 9, 8), dtype, gen, 0.2)
    else:
        x, x64 = rnd((3, 7, I), dtype, gen)
        down, d64 = rnd((r, I), torch.float32, gen, 0.1)
        up, u64 = rnd((O, r), torch.float32, gen, 0.1)
        ea = {}
        g, g64 = rnd((3, 7, O), dtype, gen, 0.2)
    for t in (x, down, up):
        t.requires_grad_(True)
    y = F.bypass_forward_diff(x, None, down, up, None, gamma=gamma, extra_args=ea)
    dx, dd, du = _grads(y, [x, down, up], g)
    ca = ea or None
    y_ref = oracle.locon.forward(x64, d64, u64, gamma, ca)
    dx_ref, dd_ref, du_ref = oracle.locon.backward(x64, g64, d64, u64, gamma, ca)
    # and the materialised form of the same API: diff_weight(*weights, gamma) is the oracle's dW
    dw = F.diff_weight(down.detach(), up.detach(), None, gamma=gamma)
    errs = {"y": err(y, y_ref, dtype), "dx": err(dx, dx_ref, dtype), "d_down": err(dd, dd_ref), "d_up": err(du, du_ref),
            "diff_weight": err(dw, oracle.locon.diff_weight(d64, u64, gamma))}
    b = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype], "d_down": TOL["f32_out"][dtype],
         "d_up": TOL["f32_out"][dtype], "diff_weight": 1e-6}
    check(f"locon_functional[{dtype},{conv}]", errs, b)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("conv", [False, True], ids=["linear", "conv3x3"])
def test_loha_functional(dtype, conv):
    from lycoris_amd.functional import loha as F
    gen = torch.Generator().manual_seed(2)
    r, I, O = 4, 24, 40
    kk = 9 if conv else 1
    gamma = torch.tensor(0.5)  # the reference modules pass a 0-d tensor (modules/loha.py:195-197)
    w1d, b1 = rnd((r, I * kk), torch.float32, gen, 1.0)
    w1u, a1 = rnd((O, r), torch.float32, gen, 0.2)
    w2d, b2 = rnd((r, I * kk), torch.float32, gen, 1.0)
    w2u, a2 = rnd((O, r), torch.float32, gen, 0.2)
    if conv:
        x, x64 = rnd((2, I, 7, 6), dtype, gen)
        g, g64 = rnd((2, O, 4, 3), dtype, gen, 0.2)
        ea = {"stride": 2, "padding": 1, "dilation": 1, "groups": 1, "_conv_shape": (O, I, 3, 3)}
        ca, shape = {"stride": 2, "padding": 1, "dilation": 1}, (O, I, 3, 3)
    else:
        x, x64 = rnd((11, I), dtype, gen)
        g, g64 = rnd((11, O), dtype, gen, 0.2)
        ea, ca, shape = {}, None, None
    ts = [x, w1u, w1d, w2u, w2d]
    for t in ts:
        t.requires_grad_(True)
    y = F.bypass_forward_diff(x, None, w1d, w1u, w2d, w2u, None, None, gamma=gamma, extra_args=ea)
    grads = _grads(y, ts, g)
    y_ref = oracle.loha.forward(x64, a1, b1, a2, b2, 0.5, shape, ca)
    ref = oracle.loha.backward(x64, g64, a1, b1, a2, b2, 0.5, shape, ca)
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[0], ref[0], dtype)}
    b = {"y": TOL["loha_store"][dtype], "dx": TOL["loha_store"][dtype]}
    for n, gr, rf in zip(["d_w1a", "d_w1b", "d_w2a", "d_w2b"], grads[1:], ref[1:]):
        errs[n], b[n] = err(gr, rf), TOL["f32_out"][dtype]
    loha_cast_pair(errs, b, dtype, y, grads[0], x64, g64, (a1, b1, a2, b2), 0.5, shape, ca)
    check(f"loha_functional[{dtype},{conv}]", errs, b)


# (name, which factors exist, alpha) -> scale the reference derives
LOKR_FORMS = [
    ("full_full", dict(w1=True, w2=True), 3.0, lambda r, al: 1.0),            # rank := gamma  => scale 1
    ("full_lowrank", dict(w1=True, w2=False), 2.0, lambda r, al: al / r),      # rank from w2a
    ("both_lowrank", dict(w1=False, w2=False), 1.0, lambda r, al: al / r),     # rank from w1a
]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("form", LOKR_FORMS, ids=[f[0] for f in LOKR_FORMS])
@pytest.mark.parametrize("conv", [False, True], ids=["linear", "conv3x3"])
def test_lokr_functional_gamma_is_alpha(dtype, form, conv):
    from lycoris_amd.functional import lokr as F
    name, has, alpha, scale_of = form
    gen = torch.Generator().manual_seed(3)
    a, b, c, d, r = 4, 4, 8, 16, 2
    I, O = b * d, a * c
    k = (3, 3) if conv else ()
    w1 = w1a = w1b = w2 = w2a = w2b = None
    n1 = n1a = n1b = n2 = n2a = n2b = None
    if has["w1"]:
        w1, n1 = rnd((a, b), torch.float32, gen, 0.3)
    else:
        w1a, n1a = rnd((a, r), torch.float32, gen, 0.5)
        w1b, n1b = rnd((r, b), torch.float32, gen, 0.5)
    if has["w2"]:
        w2, n2 = rnd((c, d, *k), torch.float32, gen, 0.2)
    else:
        w2a, n2a = rnd((c, r), torch.float32, gen, 0.5)
        w2b, n2b = rnd((r, d, *k), torch.float32, gen, 0.5)
    if conv:
        x, x64 = rnd((2, I, 6, 7), dtype, gen)
        g, g64 = rnd((2, O, 6, 7), dtype, gen, 0.2)
        ea = {"stride": 1, "padding": 1, "dilation": 1, "groups": 1}
        ca = {"stride": 1, "padding": 1, "dilation": 1}
    else:
        x, x64 = rnd((9, I), dtype, gen)
        g, g64 = rnd((9, O), dtype, gen, 0.2)
        ea, ca = {}, None
    leaves = [t for t in (x, w1, w1a, w1b, w2, w2a, w2b) if t is not None]
    for t in leaves:
        t.requires_grad_(True)
    y = F.bypass_forward_diff(x, None, w1, w1a, w1b, w2, w2a, w2b, None, gamma=alpha, extra_args=ea)
    grads = dict(zip([id(t) for t in leaves], _grads(y, leaves, g)))
    scale = scale_of(r, alpha)
    args = dict(w1=n1, w1a=n1a, w1b=n1b, w2=n2, w2a=n2a,
                w2b=None if n2b is None else n2b.reshape(r, -1), scale=scale, kshape=k, conv_args=ca)
    y_ref = oracle.lokr.forward(x64, **args)
    gr = oracle.lokr.backward(x64, g64, **args)
    errs = {"y": err(y, y_ref, dtype), "dx": err(grads[id(x)], gr["dx"], dtype)}
    bd = {"y": TOL["store_out"][dtype], "dx": TOL["store_out"][dtype]}
    for key, t in (("w1", w1), ("w1a", w1a), ("w1b", w1b), ("w2", w2), ("w2a", w2a), ("w2b", w2b)):
        if t is not None:
            errs["d_" + key] = err(grads[id(t)], gr[key].reshape(t.shape))
            bd["d_" + key] = TOL["f32_out"][dtype]
    check(f"lokr_functional[{name},{dtype},{conv}]", errs, bd)


@pytest.mark.parametrize("algo", ["locon", "loha", "lokr"])
def test_tucker_forms_through_the_functional_api(algo):
    """weights tuples with a Tucker core (mid / t1, t2 / t2): bypass_forward_diff == op(x, diff_weight) with diff_weight from the
    reference's rebuild_tucker semantics (functional/general.py:9-11), restated by oracle.general.rebuild_tucker"""
    from lycoris_amd import functional as F
    gen = torch.Generator().manual_seed(9)
    r, I, O, dtype = 4, 16, 32, torch.float32
    x, x64 = rnd((2, I, 7, 6), dtype, gen)
    ea = {"stride": 1, "padding": 1, "dilation": 1, "groups": 1}
    ca = {"stride": 1, "padding": 1, "dilation": 1}
    core = lambda: rnd((r, r, 3, 3), torch.float32, gen, 0.3)
    if algo == "locon":
        (down, d64), (up, u64), (mid, m64) = rnd((r, I, 1, 1), torch.float32, gen, 0.3), rnd((O, r, 1, 1), torch.float32, gen, 0.3), core()
        y = F.locon.bypass_forward_diff(x, None, down, up, mid, gamma=0.5, extra_args=ea)
        dw = oracle.general.rebuild_tucker(m64, u64.reshape(O, r).T, d64.reshape(r, I)) * 0.5
    elif algo == "loha":
        (w1d, a), (w1u, b), (w2d, c), (w2u, d) = (rnd((r, I), torch.float32, gen, 0.7), rnd((r, O), torch.float32, gen, 0.3),
                                                  rnd((r, I), torch.float32, gen, 0.7), rnd((r, O), torch.float32, gen, 0.3))
        (t1, t1n), (t2, t2n) = core(), core()
        y = F.loha.bypass_forward_diff(x, None, w1d, w1u, w2d, w2u, t1, t2, gamma=torch.tensor(0.5), extra_args=ea)
        dw = oracle.general.rebuild_tucker(t1n, b, a) * oracle.general.rebuild_tucker(t2n, d, c) * 0.5
        got_dw = F.loha.diff_weight(w1d, w1u, w2d, w2u, t1, t2, gamma=0.5)
        assert err(got_dw, dw) < 1e-5
    else:
        (w1, n1), (w2a, na), (w2b, nb), (t2, tn) = (rnd((4, 4), torch.float32, gen, 0.3), rnd((r, O // 4), torch.float32, gen, 0.5),
                                                    rnd((r, I // 4), torch.float32, gen, 0.5), core())
        y = F.lokr.bypass_forward_diff(x, None, w1, None, None, None, w2a, w2b, t2, gamma=2.0, extra_args=ea)
        f2 = oracle.general.rebuild_tucker(tn, na, nb)
        dw = oracle.lokr.diff_weight(w1=n1, w2=f2, scale=2.0 / r, kshape=(3, 3)).reshape(O, I, 3, 3)
    want = oracle.general.dense_forward(x64, dw, ca)
    check(f"tucker_functional[{algo}]", {"y": err(y, want)}, {"y": 2e-5})
