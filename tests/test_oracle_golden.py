"""Pin the numpy oracle against vectors produced by the real reference (CPU, no GPU needed)."""
import json
import os

import numpy as np
import torch
import pytest

import oracle
from conftest import GOLDEN, golden_case_names
from golden_util import oracle_eval


@pytest.mark.parametrize("name", golden_case_names())
def test_oracle_matches_reference_module(name, golden_cases):
    meta, a = golden_cases[name]
    delta, grads = oracle_eval(meta, a)
    assert delta.shape == a["delta"].shape
    assert oracle.general.rel_err(delta, a["delta"]) < 1e-12, name
    assert oracle.general.rel_err(grads["dx"], a["dx"]) < 1e-12, name
    checked = 0
    for k, v in a.items():
        if k.startswith("g.") and k != "g":
            assert k in grads, (name, k)
            if "gtrue." + k[2:] in a:
                # reference defect D10 (HadaWeightTucker.backward mixes up the two branches for the a-side factors): the
                # oracle follows the derivative of the reference's own FORWARD, pinned by the plain-autograd vectors
                v = a["gtrue." + k[2:]]
            assert oracle.general.rel_err(np.asarray(grads[k]).reshape(v.shape), v) < 1e-11, (name, k)
            checked += 1
    assert checked >= 1


def test_factorization_known_answers():
    with open(os.path.join(GOLDEN, "factorization.json")) as f:
        vec = json.load(f)
    for dim, factor, want in vec:
        assert list(oracle.general.factorization(dim, factor)) == want, (dim, factor)


def test_factorization_docstring_table():
    # functional/general.py:24-33 docstring rows.  Two entries of that table are stale w.r.t. the reference
    # *code* (360/-1: doc (8, 45), code (18, 20); 360/16: doc (12, 30), code (15, 24)); the code is the contract
    # and is what tests/golden/factorization.json holds, so those two rows are asserted against the code.
    f = oracle.general.factorization
    assert f(127, -1) == (1, 127) and f(127, 8) == (1, 127)
    assert f(128, -1) == (8, 16) and f(128, 2) == (2, 64) and f(128, 4) == (4, 32)
    assert f(250, -1) == (10, 25) and f(250, 2) == (2, 125) and f(250, 8) == (5, 50)
    assert f(360, -1) == (18, 20) and f(360, 4) == (4, 90) and f(360, 8) == (8, 45) and f(360, 16) == (15, 24)
    assert f(512, -1) == (16, 32) and f(512, 8) == (8, 64)
    assert f(1024, -1) == (32, 32) and f(1024, 16) == (16, 64)


def test_oracle_conv_against_torch_float64():
    """the numpy im2col oracle vs torch's own float64 CPU conv2d + autograd on a mid-size layer (oracle pin for conv)"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 48, 20, 18))
    w = rng.standard_normal((40, 48, 3, 3)) * 0.1
    for s in (1, 2):
        ca = {"stride": s, "padding": 1, "dilation": 1}
        y = oracle.general.dense_forward(x, w, ca)
        g = rng.standard_normal(y.shape)
        dx, dw = oracle.general.dense_backward(x, w, g, ca)
        xt = torch.from_numpy(x).requires_grad_(True)
        wt = torch.from_numpy(w).requires_grad_(True)
        yt = torch.nn.functional.conv2d(xt, wt, None, s, 1, 1)
        dxt, dwt = torch.autograd.grad(yt, [xt, wt], torch.from_numpy(g))
        assert oracle.general.rel_err(y, yt.detach().numpy()) < 1e-13
        assert oracle.general.rel_err(dx, dxt.numpy()) < 1e-13
        assert oracle.general.rel_err(dw, dwt.numpy()) < 1e-13


# ---- storage-rounding restatements (round 4): pinned on the real reference run in 16 bits on the CPU --------------------------
def _cast_cases():
    blob = np.load(os.path.join(GOLDEN, "cast_cases.npz"))
    with open(os.path.join(GOLDEN, "cast_cases.json")) as f:
        metas = json.load(f)
    return {n: (m, {k.split("/", 1)[1]: blob[k] for k in blob.files if k.startswith(n + "/")}) for n, m in metas.items()}


CAST = _cast_cases()


def test_round_to_is_torch_rounding():
    rng = np.random.default_rng(5)
    a = rng.standard_normal((257, 33)) * np.exp(rng.standard_normal((257, 33)) * 2)
    for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
        want = torch.from_numpy(a).float().to(dt).double().numpy()
        assert np.array_equal(oracle.general.round_to(a, name), want), name
        assert np.array_equal(oracle.general.round_to(a, str(dt)), want), name


@pytest.mark.parametrize("name", sorted(n for n in CAST if n.startswith("loha_cast")))
def test_loha_round_dw_is_the_reference_cast(name):
    """oracle.loha.diff_weight(round_dw=...) against `get_weight(shape).to(base_weight.dtype)` of the reference module
    (modules/loha.py:310): equal element for element except where the reference's fp32 rebuild sits within fp32 round-off of a
    16-bit rounding boundary (the oracle rebuilds in float64)."""
    meta, a = CAST[name]
    dw = oracle.loha.diff_weight(a["p.hada_w1_a"], a["p.hada_w1_b"], a["p.hada_w2_a"], a["p.hada_w2_b"], meta["scale"],
                                 tuple(meta["shape"]), round_dw=meta["dtype"])
    assert dw.shape == a["dw"].shape
    mism = np.mean(dw != a["dw"])
    assert mism < 2e-3, (name, mism)
    assert oracle.general.rel_err(dw, a["dw"]) < 2e-4, name
    exact = oracle.loha.diff_weight(a["p.hada_w1_a"], a["p.hada_w1_b"], a["p.hada_w2_a"], a["p.hada_w2_b"], meta["scale"],
                                    tuple(meta["shape"]))
    e = oracle.general.rel_err(exact, a["dw"])  # the cast is what costs 2^-9 (bf16) / 2^-12 (fp16), not the restatement
    assert (1e-3 < e < 3e-3) if meta["dtype"] == "bf16" else (1e-4 < e < 4e-4), (name, e)


@pytest.mark.parametrize("name", sorted(n for n in CAST if n.startswith("ia3_bypass")))
def test_ia3_bypass_restatement_matches_reference_16bit(name):
    """oracle.ia3.bypass_forward / bypass_backward(store=...) against the reference's IA3Module.bypass_forward_diff run in 16 bits
    (modules/ia3.py:114-125).  Bounds: rounding-boundary flips of the 16-bit intermediates (the CPU GEMM accumulates in another
    order than numpy's float64) -- an order of magnitude below what dropping a rounding would cost."""
    meta, a = CAST[name]
    st, on_in = meta["dtype"], meta["on_input"]
    delta, _ = oracle.ia3.bypass_forward(a["x"], a["W"], a["w"], 1.0, on_in, None, store=st)
    dx, dw = oracle.ia3.bypass_backward(a["x"], a["g"], a["W"], a["w"], 1.0, on_in, None, store=st)
    r = lambda v, dts: oracle.general.round_to(v, dts) if "float32" not in dts else v
    u = 2.0 ** -8 if st == "bf16" else 2.0 ** -11
    assert oracle.general.rel_err(r(delta, meta["delta_dtype"]), a["delta"]) < 0.2 * u, name
    assert oracle.general.rel_err(r(dx, meta["dx_dtype"]), a["dx"]) < 0.2 * u, name
    assert oracle.general.rel_err(dw, a["dw"]) < 0.2 * u, name
    # without the storage roundings the restatement is the rebuild-path oracle (same function of the inputs)
    d0, _ = oracle.ia3.bypass_forward(a["x"], a["W"], a["w"], 1.0, on_in)
    assert oracle.general.rel_err(d0, oracle.ia3.forward(a["x"], a["W"], a["w"], 1.0, on_in)) < 1e-13
    dx0, dw0 = oracle.ia3.bypass_backward(a["x"], a["g"], a["W"], a["w"], 1.0, on_in)
    tx, tw = oracle.ia3.backward(a["x"], a["g"], a["W"], a["w"], 1.0, on_in)
    assert oracle.general.rel_err(dx0, tx) < 1e-13 and oracle.general.rel_err(dw0, tw) < 1e-12
