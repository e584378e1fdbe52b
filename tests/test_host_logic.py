"""CPU tests of the host-side logic: module construction pinned against the reference, state-dict / registry
protocol, forward patching stack, dW materialisation vs the oracle, and the no-CPU-fallback contract."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
from conftest import GOLDEN, golden_case_names
from lycoris_amd.functional import general as fgen
from lycoris_amd.functional import locon as flocon
from lycoris_amd.functional import loha as floha
from lycoris_amd.functional import lokr as flokr
from lycoris_amd.modules import IA3Module, LoConModule, LohaModule, LokrModule, get_module, make_module

ALGOS = {"locon": LoConModule, "loha": LohaModule, "lokr": LokrModule, "ia3": IA3Module}


def make_layer(lk):
    lk = dict(lk)
    kind = lk.pop("kind")
    if kind == "linear":
        return nn.Linear(lk["cin"], lk["cout"], bias=lk.get("bias", True))
    return nn.Conv2d(lk["cin"], lk["cout"], lk["k"], lk["stride"], lk["padding"], lk.get("dilation", 1),
                     bias=lk.get("bias", True))


def test_factorization_matches_reference_vectors():
    with open(os.path.join(GOLDEN, "factorization.json")) as f:
        for dim, factor, want in json.load(f):
            assert list(fgen.factorization(dim, factor)) == want


def test_module_shapes_scale_alpha_match_reference():
    with open(os.path.join(GOLDEN, "shape_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) > 300
    for c in cases:
        m = ALGOS[c["algo"]]("t", make_layer(c["layer"]), 1.0, **c["mod"])
        assert {n: list(p.shape) for n, p in m.named_parameters()} == c["params"], c
        assert {n: list(v.shape) for n, v in m.state_dict().items()} == c["state_dict"], c
        assert abs(float(getattr(m, "scale", 1.0)) - c["scale"]) < 1e-12, c
        assert abs(float(getattr(m, "alpha", torch.tensor(0.0))) - c["alpha"]) < 1e-6, c
        assert list(m.shape) == c["shape"], c


@pytest.mark.parametrize("name", golden_case_names())
def test_get_diff_weight_matches_reference_forward(name, golden_cases):
    """dW materialised by get_diff_weight (merge path), pushed through the oracle's dense op, reproduces the golden
    delta of the *trained forward* (upstream's LoHa/LoKr get_diff_weight apply `scale` twice -- SURVEY D7 -- ours do not)."""
    from golden_util import conv_args_of
    meta, a = golden_cases[name]
    layer = make_layer(meta["layer"]).double()
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(a["W"]))
    mod = ALGOS[meta["algo"]]("t", layer, meta["multiplier"], **meta["mod"]).double()
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(torch.from_numpy(a["p." + n]))
    dw, _ = mod.get_diff_weight(meta["multiplier"], shape=layer.weight.shape)
    merged, _ = mod.get_merged_weight(meta["multiplier"], shape=layer.weight.shape)
    if meta["mod"].get("weight_decompose"):  # DoRA: the merged weight, not W + dW, reproduces the trained forward
        delta = oracle.general.dense_forward(a["x"], (merged - layer.weight).detach().numpy(), conv_args_of(meta))
        assert oracle.general.rel_err(delta, a["delta"]) < 1e-12
        return
    delta = oracle.general.dense_forward(a["x"], dw.detach().numpy(), conv_args_of(meta))
    assert oracle.general.rel_err(delta, a["delta"]) < 1e-12
    assert torch.allclose(merged - layer.weight, dw, atol=1e-12)


def test_functional_diff_weight_and_gamma_conventions():
    torch.manual_seed(0)
    w = torch.randn(24, 16)
    d, u, m = flocon.weight_gen(w, 4)
    assert d.shape == (4, 16) and u.shape == (24, 4) and m is None and float(u.abs().sum()) == 0.0
    u = torch.randn_like(u)
    assert np.allclose(flocon.diff_weight(d, u, None, gamma=0.5).numpy(),
                       oracle.locon.diff_weight(d.numpy(), u.numpy(), 0.5), atol=1e-6)
    ws = [torch.randn(4, 16), torch.randn(24, 4), torch.randn(4, 16), torch.randn(24, 4)]
    assert np.allclose(floha.diff_weight(*ws, None, None, gamma=torch.tensor(0.25)).numpy(),
                       oracle.loha.diff_weight(ws[1].numpy(), ws[0].numpy(), ws[3].numpy(), ws[2].numpy(), 0.25), atol=1e-5)
    # LoKr: gamma is alpha; scale = gamma / rank, and rank := gamma (scale 1) when both factors are full matrices
    w1, w1a, w1b, w2, w2a, w2b, t = flokr.weight_gen(torch.randn(32, 16), 2, factor=4, tucker=False)
    assert w1.shape == (4, 4) and w2 is None and w2a.shape == (8, 2) and w2b.shape == (2, 4) and t is None
    w2b = torch.randn_like(w2b)
    got = flokr.diff_weight(w1, None, None, None, w2a, w2b, None, gamma=3.0)
    assert np.allclose(got.numpy(), oracle.lokr.diff_weight(w1=w1.numpy(), w2a=w2a.numpy(), w2b=w2b.numpy(), scale=1.5), atol=1e-6)
    full = flokr.diff_weight(w1, None, None, torch.randn(8, 4), None, None, None, gamma=7.0)
    assert full.shape == (32, 16)  # scale 1 regardless of gamma


def test_state_dict_roundtrip_and_registry_protocol():
    torch.manual_seed(1)
    for algo, kw in (("locon", dict(lora_dim=4, alpha=2)), ("loha", dict(lora_dim=4, alpha=2)),
                     ("lokr", dict(lora_dim=4, alpha=2, factor=4)), ("lokr", dict(lora_dim=10000, factor=8)),
                     ("lokr", dict(lora_dim=2, factor=8, decompose_both=True)), ("ia3", dict(train_on_input=True))):
        layer = nn.Linear(64, 96)
        mod = ALGOS[algo]("lyc_a_b", layer, 1.0, **kw)
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn_like(p))
        sd = {"lyc_a_b." + k: v.clone() for k, v in mod.state_dict().items()}
        cls, params = get_module(sd, "lyc_a_b")
        assert cls is ALGOS[algo], (algo, cls)
        rebuilt = make_module(cls, params, "lyc_a_b", nn.Linear(64, 96))
        assert rebuilt is not None
        for k, v in mod.state_dict().items():
            if k == "alpha" and kw.get("lora_dim") == 10000:
                continue  # full-matrix LoKr: alpha is forced to lora_dim and lora_dim is not recoverable (scale == 1)
            assert torch.allclose(rebuilt.state_dict()[k].float(), v.float()), (algo, k)
        assert abs(float(getattr(rebuilt, "scale", 1.0)) - float(getattr(mod, "scale", 1.0))) < 1e-12
        fresh = ALGOS[algo]("lyc_a_b", nn.Linear(64, 96), 1.0, **kw)
        fresh.load_state_dict(mod.state_dict())
    # scalar gate: folded into the first factor on save, reset to one on load (locon.py:184-196, 262-271)
    mod = LoConModule("n", nn.Linear(8, 8), 1.0, 2, 1, use_scalar=True)
    with torch.no_grad():
        mod.scalar.fill_(0.5)
        mod.lora_up.weight.fill_(2.0)
    sd = mod.state_dict()
    assert "scalar" not in sd and float(sd["lora_up.weight"][0, 0]) == 1.0
    other = LoConModule("n", nn.Linear(8, 8), 1.0, 2, 1, use_scalar=True)
    other.load_state_dict(sd)
    assert float(other.scalar) == 1.0 and float(other.lora_up.weight[0, 0]) == 1.0


def test_apply_restore_wrapper_stack():
    layer = nn.Linear(8, 8)
    orig = layer.forward
    a = LoConModule("a", layer, 1.0, 2, 1)
    b = LokrModule("b", layer, 1.0, 2, 1, factor=2)
    c = IA3Module("c", layer, 1.0)
    a.apply_to(); b.apply_to(); c.apply_to()
    assert layer.forward == c.forward and c.org_forward == b.forward and b.org_forward == a.forward
    assert a.org_forward == orig and layer._lycoris_wrappers == [a, b, c]
    b.restore()  # middle of the stack: c must now call a
    assert c.org_forward == a.forward and layer.forward == c.forward and layer._lycoris_wrappers == [a, c]
    a.apply_to()  # re-applying moves it to the top
    assert layer._lycoris_wrappers == [c, a] and layer.forward == a.forward and a.org_forward == c.forward
    a.restore(); c.restore()
    assert layer.forward == orig and not hasattr(layer, "_lycoris_wrappers") and not hasattr(layer, "_lycoris_original_forward")


def test_merge_to_and_onfly_merge():
    torch.manual_seed(2)
    layer = nn.Linear(16, 24)
    w0 = layer.weight.detach().clone()
    mod = LokrModule("m", layer, 1.0, 10000, 1, factor=4)
    with torch.no_grad():
        mod.lokr_w2.copy_(torch.randn_like(mod.lokr_w2))
    dw = mod.get_diff_weight(0.5)[0].detach()
    mod.onfly_merge(0.5)
    assert torch.allclose(layer.weight, w0 + dw, atol=1e-6)
    mod.onfly_restore()
    assert torch.allclose(layer.weight, w0)
    mod.merge_to(0.5)
    assert torch.allclose(layer.weight, w0 + dw, atol=1e-6)


def test_host_tensors_run_and_unsupported_features_fail_loudly():
    layer = nn.Linear(16, 16)
    mod = LoConModule("m", layer, 1.0, 4, 1)
    with torch.no_grad():
        mod.lora_up.weight.normal_(0, 0.1)
    mod.apply_to()
    x = torch.randn(2, 16)
    dw = mod.get_diff_weight(1.0)[0]
    assert torch.allclose(layer(x), torch.nn.functional.linear(x, layer.weight + dw, layer.bias), atol=1e-6)  # host tensors: composite.py
    mod.restore()
    assert layer(torch.randn(2, 16)).shape == (2, 16)
    # DoRA is on the native path: the magnitude vector starts at the frozen weight's norms (locon.py:107-129)
    lin = nn.Linear(8, 6)
    for cls in (LoConModule, LohaModule, LokrModule):
        m = cls("m", lin, 1.0, 2, 1, weight_decompose=True, wd_on_out=True)
        assert m.wd and tuple(m.dora_scale.shape) == (6, 1)
        assert torch.allclose(m.dora_scale.reshape(-1), lin.weight.detach().norm(dim=1))
        assert "dora_scale" in m.state_dict()
        m = cls("m", lin, 1.0, 2, 1, weight_decompose=True, wd_on_out=False)
        assert tuple(m.dora_scale.shape) == (1, 8)
        assert torch.allclose(m.dora_scale.reshape(-1), lin.weight.detach().norm(dim=0))
    m = LoConModule("m", nn.Conv2d(4, 6, 3), 1.0, 2, 1, weight_decompose=True, wd_on_out=False)
    assert tuple(m.dora_scale.shape) == (1, 4, 1, 1)
    with pytest.raises(NotImplementedError):
        LoConModule("m", nn.Linear(8, 8), 1.0, 2, 1, weight_decompose=True, rank_dropout=0.5)
    # the dropout variants are on the native path (applied around the kernels, tests/test_gpu_dropout.py)
    m = LoConModule("m", nn.Linear(8, 8), 1.0, 2, 1, dropout=0.1, rank_dropout=0.2, module_dropout=0.3,
                    rank_dropout_scale=True)
    assert (m.dropout, m.rank_dropout, m.module_dropout, m.rank_dropout_scale) == (0.1, 0.2, 0.3, True)
    # Tucker / conv-CP forms are on the native path (core folded into the input-side factor, csrc/tucker.h)
    m = LokrModule("m", nn.Conv2d(8, 8, 3), 1.0, 1, 1, use_tucker=True, factor=2)
    assert m.tucker and tuple(m.lokr_t2.shape) == (1, 1, 3, 3) and "lokr_t2" in m.state_dict()
    m = LoConModule("m", nn.Conv2d(8, 8, 3), 1.0, 2, 1, use_tucker=True)
    assert m.tucker and tuple(m.lora_mid.weight.shape) == (2, 2, 3, 3) and tuple(m.lora_down.weight.shape) == (2, 8, 1, 1)
    assert not LoConModule("m", nn.Conv2d(8, 8, 1), 1.0, 2, 1, use_tucker=True).tucker      # 1x1: nothing to factor
    m = LohaModule("m", nn.Conv3d(8, 8, 3), 1.0, 2, 1)   # nn.Conv3d: the ATen rebuild form, reference shapes (tests/test_conv3d.py)
    assert m._aten_only and tuple(m.hada_w1_b.shape) == (2, 8 * 27)
    with pytest.raises(NotImplementedError):
        LohaModule("m", nn.Conv3d(8, 8, 3, groups=2), 1.0, 2, 1)
    # nn.Conv1d is adapted through its Conv2d twin (1 x k window; modules/base.py _Conv1dTwin, tests/test_conv1d.py)
    m = LohaModule("m", nn.Conv1d(8, 8, 3), 1.0, 2, 1)
    assert tuple(m.state_dict()["hada_w1_b"].shape) == (2, 24) and m._conv1d is not None
    with pytest.raises(ValueError):
        LoConModule("m", nn.LayerNorm(8), 1.0, 2, 1)
    # layer variants the kernels do not take fail when the network is BUILT, not at the first step
    for cls in (LoConModule, LohaModule, LokrModule):
        with pytest.raises(NotImplementedError, match="grouped"):
            cls("m", nn.Conv2d(8, 8, 3, groups=2), 1.0, 2, 1)
        assert cls("m", nn.Conv2d(8, 8, 3, padding="same", dilation=2), 1.0, 2, 1).kw_dict["padding"] == (2, 2)  # round 5: as integers
        assert cls("m", nn.Conv2d(8, 8, 3, padding="valid"), 1.0, 2, 1).kw_dict["padding"] == (0, 0)
        with pytest.raises(NotImplementedError, match="asymmetric"):
            cls("m", nn.Conv2d(8, 8, 4, padding="same"), 1.0, 2, 1)
        with pytest.raises(NotImplementedError, match="padding_mode"):
            cls("m", nn.Conv2d(8, 8, 3, padding=1, padding_mode="reflect"), 1.0, 2, 1)


def test_missing_native_library_is_an_error(monkeypatch, tmp_path):
    from lycoris_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no fallback"):
        _native.load()


def test_conv_dispatch_predicates():
    """Which Conv2d calls take the im2col-free kernels is pure shape / dtype logic (the C entry points repeat the same
    checks and answer LYC_ERR_UNSUPPORTED otherwise): pin it on CPU tensors."""
    from lycoris_amd import ops
    bf = torch.bfloat16
    x = torch.zeros(1, 320, 8, 8, dtype=bf)
    down, up = torch.zeros(8, 320, 3, 3), torch.zeros(320, 8, 1, 1)
    assert ops._locon_conv_implicit_ok(x, down, up)
    assert not ops._locon_conv_implicit_ok(x.float(), down, up)                                  # fp32 activations
    assert not ops._locon_conv_implicit_ok(torch.zeros(1, 24, 8, 8, dtype=bf), torch.zeros(8, 24, 3, 3), up)   # C % 16
    assert not ops._locon_conv_implicit_ok(x, torch.zeros(6, 320, 3, 3), torch.zeros(320, 6, 1, 1))            # rank % 4
    assert not ops._locon_conv_implicit_ok(x, torch.zeros(32, 320, 3, 3), torch.zeros(320, 32, 1, 1))          # rank > 16
    assert not ops._locon_conv_implicit_ok(x, torch.zeros(16, 320, 5, 5), torch.zeros(320, 16, 1, 1))          # taps*r > 144
    assert ops._locon_conv_implicit_ok(x, torch.zeros(4, 320, 5, 5), torch.zeros(320, 4, 1, 1))
    w1, w2 = torch.zeros(8, 8), torch.zeros(40, 40, 3, 3)
    assert ops._lokr_conv_implicit_ok(x, w1, w2)
    assert not ops._lokr_conv_implicit_ok(x, torch.zeros(8, 4), torch.zeros(40, 80, 3, 3))        # a != b
    assert not ops._lokr_conv_implicit_ok(x, torch.zeros(2, 2), torch.zeros(160, 160, 3, 3))      # factor 2
    assert not ops._lokr_conv_implicit_ok(x.float(), w1, w2)


def test_deferred_weight_gradient_controls_need_the_cpp_dispatch():
    """the park / flush machinery lives in the C++ custom-op layer: asking for it under the Python dispatch is an error, not a
    silent no-op; flush / discard without a loaded extension are harmless"""
    from lycoris_amd import ops
    ops.set_dispatch("python")
    try:
        with pytest.raises(RuntimeError, match="C\\+\\+ dispatch"):
            ops.deferred_weight_gradients(True)
    finally:
        ops.set_dispatch("cpp")
    ops.flush_deferred()                 # nothing parked: no-op (and no GPU needed)
    assert ops.discard_deferred() == 0
