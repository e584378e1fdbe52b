"""Drop-in check against the REAL reference wrapper (only where /root/reference exists, i.e. the build container):
lycoris_amd.install() must make create_lycoris / create_lycoris_from_weights build native modules with the same
names, state-dict keys and shapes the reference produces itself.  Mirrors example/standalone_example.py (BASELINE
config 0: LoCon rank 4 on a 3-layer nn.Linear MLP) -- plumbing only, no forward on CPU (there is no CPU path)."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lycoris")), reason="reference tree not present")


@pytest.fixture()
def ref():
    import tomli
    shim = types.ModuleType("toml")
    shim.load = lambda f: tomli.load(open(f, "rb")) if isinstance(f, str) else tomli.load(f)
    shim.loads = tomli.loads
    sys.modules.setdefault("toml", shim)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import lycoris
    import lycoris_amd
    yield lycoris
    lycoris_amd.uninstall()


class MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(784, 2048), nn.ReLU(), nn.Linear(2048, 784), nn.ReLU(), nn.Linear(784, 10))

    def forward(self, x):
        return self.net(x)


@pytest.mark.parametrize("algo,extra", [("lora", {}), ("loha", {}), ("lokr", {"factor": 8}), ("ia3", {})])
def test_create_lycoris_builds_native_modules(ref, algo, extra):
    import lycoris_amd
    from lycoris import LycorisNetwork, create_lycoris
    import lycoris.kohya as kohya
    import lycoris.wrapper as wrapper

    def build():
        torch.manual_seed(0)
        LycorisNetwork.apply_preset({"target_module": ["MLP"], "target_name": []})
        net = create_lycoris(MLP(), 1.0, linear_dim=4, linear_alpha=2.0, algo=algo, **extra)
        return net

    want = None
    if algo != "ia3":  # upstream cannot even build "ia3" (KeyError); install() adds the key
        want = build()
    assert lycoris_amd.install()
    assert kohya.network_module_dict is wrapper.network_module_dict
    got = build()
    assert len(got.loras) == 3
    for lora in got.loras:
        assert type(lora).__module__.startswith("lycoris_amd.modules"), type(lora)
    if want is not None:
        ws, gs = want.state_dict(), got.state_dict()
        assert list(ws.keys()) == list(gs.keys())
        for k in ws:
            assert ws[k].shape == gs[k].shape and ws[k].dtype == gs[k].dtype, k
        assert [l.lora_name for l in want.loras] == [l.lora_name for l in got.loras]
    # attach / detach through the wrapper, and the from-weights path through MODULE_LIST
    model = MLP()
    LycorisNetwork.apply_preset({"target_module": ["MLP"], "target_name": []})
    net = create_lycoris(model, 1.0, linear_dim=4, linear_alpha=2.0, algo=algo, **extra)
    net.apply_to()
    assert model.net[0].forward == net.loras[0].forward
    net.restore()
    assert model.net[0].forward.__func__ is nn.Linear.forward
    if algo != "ia3":
        from lycoris import create_lycoris_from_weights
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net2, _ = create_lycoris_from_weights(1.0, "", MLP(), weights_sd=sd)
        assert len(net2.loras) == 3 and all(type(l).__module__.startswith("lycoris_amd") for l in net2.loras)
    lycoris_amd.uninstall()
    assert wrapper.network_module_dict["lokr"].__module__.startswith("lycoris.modules")


def test_dora_builds_natively_through_the_reference_wrapper(ref):
    """weight_decompose (DoRA, `dora_wd=True` in the wrapper's kwargs) is on the native path: create_lycoris builds the
    native class with a dora_scale parameter; there is no delegation to the reference's torch modules."""
    import lycoris_amd
    from lycoris import LycorisNetwork, create_lycoris

    assert lycoris_amd.install()
    torch.manual_seed(0)
    LycorisNetwork.apply_preset({"target_module": ["MLP"], "target_name": []})
    net = create_lycoris(MLP(), 1.0, linear_dim=4, linear_alpha=2.0, algo="lora", dora_wd=True)
    assert len(net.loras) == 3 and all(type(m) is lycoris_amd.LoConModule and m.wd for m in net.loras)
    assert all(any(k.endswith("dora_scale") for k in m.state_dict()) for m in net.loras)
    assert not hasattr(lycoris_amd, "_delegating")
