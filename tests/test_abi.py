"""The C-ABI library must load without a GPU and export exactly what include/lycoris_amd.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "lycoris_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lyc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    names = declared_functions()
    assert "lyc_lokr_linear_fwd" in names and "lyc_abi_version" in names and len(names) >= 14


def test_library_loads_and_exports_every_declared_symbol():
    from lycoris_amd import _native
    lib = _native.load()
    assert lib.lyc_abi_version() == _native.ABI_VERSION
    raw = ctypes.CDLL(_native.lib_path())
    for name in declared_functions():
        assert hasattr(raw, name), f"{name} is declared in the header but not exported"


def test_python_binding_table_matches_header():
    from lycoris_amd import _native
    bound = set(_native.SIGNATURES) | set(_native.VALUE_SIGNATURES) | {"lyc_abi_version", "lyc_last_error"}
    assert bound == set(declared_functions())
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _native.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(argtypes), f"{name}: header has {nargs} parameters, binding has {len(argtypes)}"


def test_argument_errors_are_reported_not_crashed():
    """Argument checks run before any launch, so they work on the CPU-only build box."""
    from lycoris_amd import _native as N
    with pytest.raises(RuntimeError, match="128x128"):
        N.call("lyc_lokr_linear_fwd", None, None, None, None, None, 4, 200, 8, 4, 4, 1.0, N.LYC_BF16, None)
    with pytest.raises(RuntimeError, match="null pointer"):
        N.call("lyc_locon_linear_fwd", None, None, None, None, None, 4, 8, 8, 2, 1.0, N.LYC_BF16, None)
    with pytest.raises(RuntimeError, match="bad dims"):
        N.call("lyc_chan_scale", None, None, None, None, 4, 0, 1, 1.0, 1.0, N.LYC_F32, None)
    assert N.load().lyc_loha_workspace_bytes(1280, 1280, N.LYC_BF16) == 1280 * 1280 * 2  # one plane
    assert N.load().lyc_loha_workspace_bytes(1280, 640, N.LYC_F32) == 2 * 1280 * 640 * 4


def test_torch_custom_ops_are_registered_with_meta_kernels():
    """TORCH_LIBRARY(lycoris_amd): the extension loads on the CPU-only box, every public op has a schema, the Meta kernels
    give the output shapes (what FakeTensor / torch.compile tracing uses), and C++ autograd runs on meta tensors."""
    import torch
    from lycoris_amd import _native
    ext = _native.load_torch_ops()
    assert ext.abi_version() == _native.ABI_VERSION
    ns = torch.ops.lycoris_amd
    for name in ("lokr_linear", "locon_linear", "loha_linear", "chan_affine", "lokr_conv2d", "locon_conv2d",
                 "_lokr_linear_backward", "_locon_linear_forward", "_locon_linear_backward", "_loha_linear_forward",
                 "_loha_linear_backward", "_chan_reduce"):
        assert hasattr(ns, name), name
    m = dict(device="meta")
    x = torch.empty(3, 5, 64, dtype=torch.bfloat16, requires_grad=True, **m)
    w1, w2 = torch.empty(8, 8, requires_grad=True, **m), torch.empty(16, 8, requires_grad=True, **m)
    y = ns.lokr_linear(x, w1, w2, 1.0)
    assert y.shape == (3, 5, 128) and y.dtype == torch.bfloat16
    gx, g1, g2 = torch.autograd.grad(y, [x, w1, w2], torch.empty_like(y))
    assert gx.shape == x.shape and g1.shape == w1.shape and g2.shape == w2.shape
    base = torch.empty(3, 5, 128, dtype=torch.bfloat16, requires_grad=True, **m)   # fused `base + delta` form
    yb = ns.lokr_linear(x, w1, w2, 1.0, base)
    gb, g1b = torch.autograd.grad(yb, [base, w1], torch.empty_like(yb))
    assert yb.shape == base.shape and gb.shape == base.shape and g1b.shape == w1.shape
    down, up = torch.empty(4, 64, **m), torch.empty(32, 4, **m)
    y, t = ns._locon_linear_forward(x, down, up, 1.0)
    assert y.shape == (3, 5, 32) and t.shape == (15, 4) and t.dtype == torch.float32
    fs = [torch.empty(32, 4, **m), torch.empty(4, 64, **m), torch.empty(32, 4, **m), torch.empty(4, 64, **m)]
    assert ns.loha_linear(x, *fs, 1.0).shape == (3, 5, 32)
    assert ns.chan_affine(x, torch.empty(64, **m), None, 1.0, 1.0, -1).shape == x.shape
    xc = torch.empty(2, 64, 9, 7, dtype=torch.float16, **m)
    assert ns.lokr_conv2d(xc, torch.empty(8, 8, **m), torch.empty(4, 8, 3, 3, **m), 1.0, [2, 2], [1, 1], [1, 1]).shape == (2, 32, 5, 4)
    assert ns.locon_conv2d(xc, torch.empty(4, 64, 3, 3, **m), torch.empty(24, 4, 1, 1, **m), 1.0, [1, 1], [1, 1], [1, 1]).shape == (2, 24, 9, 7)
    # the Conv2d ops are differentiable under tracing: C++ autograd through the functional backward ops
    xg = torch.empty(2, 64, 9, 7, dtype=torch.bfloat16, requires_grad=True, **m)
    cw1, cw2 = torch.empty(8, 8, requires_grad=True, **m), torch.empty(4, 8, 3, 3, requires_grad=True, **m)
    yc = ns.lokr_conv2d(xg, cw1, cw2, 1.0, [1, 1], [1, 1], [1, 1])
    gs = torch.autograd.grad(yc, [xg, cw1, cw2], torch.empty_like(yc))
    assert [g.shape for g in gs] == [xg.shape, cw1.shape, cw2.shape]
    cd, cu = torch.empty(4, 64, 3, 3, requires_grad=True, **m), torch.empty(24, 4, 1, 1, requires_grad=True, **m)
    yc = ns.locon_conv2d(xg, cd, cu, 1.0, [2, 2], [1, 1], [1, 1])
    gs = torch.autograd.grad(yc, [xg, cd, cu], torch.empty_like(yc))
    assert yc.shape == (2, 24, 5, 4) and [g.shape for g in gs] == [xg.shape, cd.shape, cu.shape]
    for name in ("_lokr_conv2d_backward", "_locon_conv2d_forward", "_locon_conv2d_backward"):
        assert hasattr(ns, name), name
    # round 3: the rows-lowered Conv2d of every algorithm (LoHa / 1x1 / fp32), differentiable under tracing through its functional ops
    for name in ("adapter_conv2d", "_adapter_conv2d_forward", "_adapter_conv2d_backward", "lokr_linear_lr", "lokr_linear_lr2", "lokr_conv2d_lr"):
        assert hasattr(ns, name), name
    hf = [torch.empty(24, 4, requires_grad=True, **m), torch.empty(4, 64 * 9, requires_grad=True, **m),
          torch.empty(24, 4, requires_grad=True, **m), torch.empty(4, 64 * 9, requires_grad=True, **m)]
    yh = ns.adapter_conv2d(xg, hf[0], hf[1], hf[2], hf[3], 2, 1.0, [3, 3], [2, 2], [1, 1], [1, 1])  # algo 2 = LoHa
    gs = torch.autograd.grad(yh, [xg] + hf, torch.empty_like(yh))
    assert yh.shape == (2, 24, 5, 4) and [g.shape for g in gs] == [xg.shape] + [f.shape for f in hf]
    y1, cols, saved = ns._adapter_conv2d_forward(xg, torch.empty(4, 64, **m), torch.empty(24, 4, **m), None, None, 1, 1.0, [1, 1], [1, 1], [0, 0], [1, 1])
    assert y1.shape == (2, 24, 9, 7) and cols.numel() == 0 and saved.shape == (2 * 9 * 7, 4)  # 1x1 LoCon: rows are the NHWC view, t saved
    la, lb = torch.empty(16, 2, requires_grad=True, **m), torch.empty(2, 8, requires_grad=True, **m)
    yl = ns.lokr_linear_lr(x, w1, la, lb, 1.0)  # traced: the composite form (product by ATen, then the traceable full-matrix op)
    gs = torch.autograd.grad(yl, [x, w1, la, lb], torch.empty_like(yl))
    assert yl.shape == (3, 5, 128) and [g.shape for g in gs] == [x.shape, w1.shape, la.shape, lb.shape]
    assert ns.lokr_linear_lr2(x, torch.empty(8, 2, **m), torch.empty(2, 8, **m), torch.empty(16, 2, **m), torch.empty(2, 8, **m), 1.0).shape == (3, 5, 128)
    assert ns.lokr_conv2d_lr(xc, torch.empty(8, 8, **m), torch.empty(4, 2, **m), torch.empty(2, 8 * 9, **m), 1.0, [3, 3], [1, 1], [1, 1], [1, 1]).shape == (2, 32, 9, 7)


def test_cpu_tensors_take_the_composite_forms_not_the_custom_ops():
    """device dispatch (round 5): a host tensor is evaluated by lycoris_amd/composite.py (tests/test_cpu_composite.py pins the numbers);
    the custom ops themselves still refuse a CPU tensor -- they are the HIP path"""
    import torch
    from lycoris_amd import _native, ops
    y = ops.lokr_linear(torch.randn(4, 64), torch.randn(8, 8), torch.randn(8, 8), 1.0)
    assert y.shape == (4, 64) and y.device.type == "cpu"
    assert ops.chan_affine(torch.randn(4, 64), torch.randn(64)).shape == (4, 64)
    _native.load_torch_ops()
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.lycoris_amd.lokr_linear(torch.randn(4, 64), torch.randn(8, 8), torch.randn(8, 8), 1.0, None)


def test_loading_the_custom_op_extension_first_does_not_deadlock():
    """round-2 regression: load_torch_ops() took the loader lock and then called load(), which takes it again"""
    import subprocess
    import sys
    code = "from lycoris_amd import _native as N; N.load_torch_ops(); print('ok', N.load().lyc_abi_version())"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-500:]


def test_ctypes_item_structs_match_the_header_layout(tmp_path):
    """LycLokrWgradItem / LycLoconWgradItem / LycLohaWgradItem are passed as host arrays: the ctypes mirrors must have the C
    compiler's size and field offsets (gcc compiles a probe against include/lycoris_amd.h)."""
    import shutil
    import subprocess
    from lycoris_amd import _native as N
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    mirrors = {"LycLokrWgradItem": N.WgradItem, "LycLoconWgradItem": N.LoconWgradItem, "LycLohaWgradItem": N.LohaWgradItem,
               "LycLokrConvWgradItem": N.LokrConvWgradItem, "LycLokrPackItem": N.LokrPackItem, "LycLokrLrChainItem": N.LokrLrChainItem,
               "LycLohaPlaneItem": N.LohaPlaneItem}
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in mirrors.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in out:
        if not line.strip():
            continue
        cname, what, val = line.split()
        cls = mirrors[cname]
        want = ctypes.sizeof(cls) if what == "size" else getattr(cls, what).offset
        assert int(val) == want, (cname, what, val, want)
        seen += 1
    assert seen == sum(1 + len(c._fields_) for c in mirrors.values())
