"""ctypes binding of the C-ABI library ``liblycoris_amd.so`` (include/lycoris_amd.h).

There is deliberately NO fallback: if the shared library is missing or an entry point fails, the
caller gets a RuntimeError.  PyTorch is only used for device memory and streams here.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_LIB_NAME = "liblycoris_amd.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)
ABI_VERSION = 13

LYC_F32, LYC_F16, LYC_BF16 = 0, 1, 2
_DTYPE_CODE = {torch.float32: LYC_F32, torch.float16: LYC_F16, torch.bfloat16: LYC_BF16}

_vp, _fp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> argtypes; mirrors include/lycoris_amd.h one to one (tests/test_abi.py checks the header against this)
SIGNATURES = {
    "lyc_lokr_linear_fwd": [_vp, _fp, _fp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_lokr_linear_bwd": [_vp, _vp, _fp, _fp, _vp, _fp, _fp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_lokr_linear_fwd_group": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],  # items: pointer to an array of LinearGroupItem
    "lyc_lokr_linear_bwd_group": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lyc_lokr_wgrad_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of WgradItem
    "lyc_lokr_wgrad_group_ws": [_vp, _i32, _i32, _vp, _i64, _vp],  # + device scratch of lyc_lokr_wgrad_table_bytes(n)
    "lyc_locon_wgrad_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of LoconWgradItem
    "lyc_loha_wgrad_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of LohaWgradItem
    "lyc_lokr_conv2d_fwd": [_vp, _fp, _fp, _vp, _i64, _i64, _i64] + [_i32] * 12 + [_f32, _i32, _vp],
    "lyc_lokr_conv2d_bwd": [_vp, _vp, _fp, _fp, _fp, _vp, _fp, _fp, _vp, _i64, _i64, _i64] + [_i32] * 12 + [_f32, _i32, _vp],
    "lyc_lokr_conv_wgrad_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of LokrConvWgradItem
    "lyc_loha_rebuild_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of LohaPlaneItem
    "lyc_lokr_pack_group": [_vp, _i32, _i32, _vp],  # items: pointer to an array of LokrPackItem
    "lyc_lokr_pack_group_ws": [_vp, _i32, _i32, _vp, _i64, _i32, _vp],  # + device table of lyc_lokr_pack_table_bytes(items, n)
    "lyc_lokr_lr_chain_group": [_vp, _i32, _vp],  # items: pointer to an array of LokrLrChainItem
    "lyc_lokr_linear_fwd_planes": [_vp, _fp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_lokr_linear_bwd_planes": [_vp, _vp, _fp, _vp, _vp, _fp, _fp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_lokr_pack_w2": [_fp, _i64, _i64, _i64, _fp, _i64, _i64, _fp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "lyc_lokr_conv2d_fwd_planes": [_vp, _fp, _vp, _vp, _i64, _i64, _i64] + [_i32] * 12 + [_f32, _i32, _vp],
    "lyc_lokr_conv2d_bwd_planes": [_vp, _vp, _fp, _fp, _vp, _vp, _fp, _fp, _vp, _i64, _i64, _i64] + [_i32] * 12 + [_f32, _i32, _vp],
    "lyc_locon_linear_fwd": [_vp, _fp, _fp, _fp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_locon_linear_bwd": [_vp, _vp, _fp, _fp, _fp, _fp, _vp, _fp, _fp, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_locon_conv2d_fwd": [_vp, _fp, _fp, _fp, _vp, _i64, _i64, _i64] + [_i32] * 11 + [_f32, _i32, _vp],
    "lyc_locon_conv2d_bwd": [_vp, _vp, _fp, _fp, _fp, _fp, _vp, _fp, _fp, _i64, _i64, _i64] + [_i32] * 11 + [_f32, _i32, _vp],
    "lyc_chan_scale": [_vp, _fp, _fp, _vp, _i64, _i64, _i64, _f32, _f32, _i32, _vp],
    "lyc_chan_reduce": [_vp, _vp, _fp, _fp, _i64, _i64, _i64, _f32, _i32, _vp],
    "lyc_chan_bwd": [_vp, _vp, _fp, _fp, _vp, _fp, _i64, _i64, _i64, _f32, _f32, _i32, _vp],
    "lyc_loha_linear_fwd": [_vp, _fp, _fp, _fp, _fp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp],
    "lyc_loha_linear_bwd": [_vp, _vp, _fp, _fp, _fp, _fp, _vp, _fp, _vp, _fp, _fp, _fp, _fp, _i64, _i32, _i32, _i32,
                            _f32, _i32, _vp],
    "lyc_wspace": [_i32, _fp, _fp, _fp, _fp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _fp, _i32, _vp, _i32,
                   _f32, _fp, _f32, _vp],
    "lyc_locon_wgrad": [_fp, _fp, _fp, _fp, _fp, _i64, _i64, _i32, _f32, _vp],
    "lyc_loha_wgrad": [_fp] * 9 + [_i64, _i64, _i32, _f32, _vp],
    "lyc_lokr_wgrad": [_fp] * 5 + [_i32, _i32, _i32, _i64, _f32, _vp],
    "lyc_tucker_core_fwd": [_fp, _fp, _fp, _i32, _i32, _i64, _i32, _vp],
    "lyc_tucker_core_bwd": [_fp, _fp, _fp, _fp, _fp, _i32, _i32, _i64, _i32, _vp],
    "lyc_im2col": [_vp, _vp, _i64, _i64, _i64, _i64] + [_i32] * 8 + [_i32, _vp],
    "lyc_col2im": [_vp, _vp, _i64, _i64, _i64, _i64] + [_i32] * 8 + [_i32, _vp],
    "lyc_im2col_rows": [_vp, _vp, _i64, _i64, _i64, _i64] + [_i32] * 8 + [_i32, _vp],
    "lyc_col2im_rows": [_vp, _vp, _i64, _i64, _i64, _i64] + [_i32] * 8 + [_i32, _vp],
    "lyc_sum_rows": [_vp, _i32, _vp, _i64, _i32, _vp],
    "lyc_locon_linear_fwd_group": [_vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "lyc_locon_linear_bwd_group": [_vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "lyc_locon_linear_bwd_group_sum": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "lyc_lokr_linear_bwd_group_sum": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "lyc_nchw_to_rows": [_vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "lyc_rows_to_nchw": [_vp, _vp, _i64, _i64, _i64, _i32, _vp],
}
# entry points that return something other than a status code
VALUE_SIGNATURES = {
    "lyc_loha_workspace_bytes": ([_i32, _i32, _i32], ctypes.c_int64),
    "lyc_lokr_bwd_workspace_bytes": ([_i64, _i32, _i32, _i32, _i32, _i32], ctypes.c_int64),
    "lyc_lokr_conv2d_bwd_workspace_bytes": ([_i64, _i64, _i64, _i32, _i32, _i32], ctypes.c_int64),
    "lyc_lokr_planes_bytes": ([_i32, _i32, _i32, _i32], ctypes.c_int64),
    "lyc_lokr_linear_planes_ok": ([_i64, _i32, _i32, _i32, _i32, _i32], ctypes.c_int),
    "lyc_lokr_conv2d_dx_blocks": ([_i64, _i64, _i64] + [_i32] * 14, ctypes.c_int64),
    "lyc_lokr_conv2d_planes_ok": ([_i64, _i64, _i64] + [_i32] * 14, ctypes.c_int),
    "lyc_lokr_wgrad_table_bytes": ([_i32], ctypes.c_int64),
    "lyc_lokr_pack_table_bytes": ([_vp, _i32], ctypes.c_int64),
    "lyc_lokr_wgrad_deferrable": ([_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32], ctypes.c_int),  # 1 / 0, not an error code
    "lyc_locon_wgrad_deferrable": ([_vp, _vp, _i64, _i32, _i32, _i32, _i32], ctypes.c_int),
    "lyc_loha_wgrad_deferrable": ([_vp, _vp, _i64, _i32, _i32, _i32, _i32], ctypes.c_int),
    "lyc_loha_plane_cacheable": ([_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32], ctypes.c_int),
}


class LohaWgradItem(ctypes.Structure):
    """LycLohaWgradItem (include/lycoris_amd.h)"""
    _fields_ = [("g", _vp), ("x", _vp), ("w1a", _vp), ("w1b", _vp), ("w2a", _vp), ("w2b", _vp), ("d_w1a", _vp), ("d_w1b", _vp),
                ("d_w2a", _vp), ("d_w2b", _vp), ("gw", _vp), ("M", _i64), ("I", _i32), ("O", _i32), ("r", _i32), ("alpha", _f32)]


class LohaPlaneItem(ctypes.Structure):
    """LycLohaPlaneItem (include/lycoris_amd.h)"""
    _fields_ = [("w1a", _vp), ("w1b", _vp), ("w2a", _vp), ("w2b", _vp), ("plane", _vp), ("O", _i32), ("I", _i32), ("r", _i32), ("alpha", _f32)]


class LoconWgradItem(ctypes.Structure):
    """LycLoconWgradItem (include/lycoris_amd.h)"""
    _fields_ = [("g", _vp), ("x", _vp), ("t", _vp), ("dt", _vp), ("d_down", _vp), ("d_up", _vp), ("M", _i64),
                ("I", _i32), ("O", _i32), ("r", _i32), ("alpha", _f32)]


class LokrPackItem(ctypes.Structure):
    """LycLokrPackItem (include/lycoris_amd.h)"""
    _fields_ = [("w2", _vp), ("sq", _i64), ("sv", _i64), ("st", _i64), ("c", _i32), ("d", _i32), ("taps", _i32), ("planes_fwd", _vp),
                ("planes_bwd", _vp), ("w2a", _vp), ("w2b", _vp), ("rank", _i32)]


class LokrLrChainItem(ctypes.Structure):
    """LycLokrLrChainItem (include/lycoris_amd.h)"""
    _fields_ = [("dw2", _vp), ("w2a", _vp), ("w2b", _vp), ("d_w2a", _vp), ("d_w2b", _vp), ("c", _i32), ("d", _i32), ("r", _i32),
                ("taps", _i32)]


class LokrConvWgradItem(ctypes.Structure):
    """LycLokrConvWgradItem (include/lycoris_amd.h)"""
    _fields_ = [("g_rows", _vp), ("x_rows", _vp), ("w1", _vp), ("dw1", _vp), ("dw2p", _vp), ("ws", _vp), ("B", _i64), ("H", _i64),
                ("W", _i64), ("dw1_blocks", _i64)] + [(n, _i32) for n in ("a", "b", "c", "d", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw")] + [
                    ("alpha", _f32)]


class WgradItem(ctypes.Structure):
    """LycLokrWgradItem (include/lycoris_amd.h)"""
    _fields_ = [("g", _vp), ("x", _vp), ("w1", _vp), ("dw1", _vp), ("dw2", _vp), ("ws", _vp), ("M", _i64),
                ("a", _i32), ("b", _i32), ("c", _i32), ("d", _i32), ("alpha", _f32)]

class LoconGroupItem(ctypes.Structure):
    """LycLoconLinearGroupItem (include/lycoris_amd.h)"""
    _fields_ = [("inp", _vp), ("down", _vp), ("up", _vp), ("mid", _vp), ("out", _vp), ("M", _i64), ("alpha", _f32)]


class LinearGroupItem(ctypes.Structure):
    """LycLokrLinearGroupItem (include/lycoris_amd.h)"""
    _fields_ = [("inp", _vp), ("w1", _vp), ("planes", _vp), ("aux", _vp), ("out", _vp), ("ws", _vp), ("M", _i64), ("alpha", _f32)]


_lock = threading.RLock()  # re-entrant: load_torch_ops() calls load() while holding it
_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and return the ctypes handle.  Raises NativeLibraryError when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_NAME} not found at {_LIB_PATH}. lycoris_amd has no fallback path: build the HIP library "
                "first (python -c 'import __graft_entry__ as g; g.build()' or make -C lycoris_amd/csrc)."
            )
        try:
            lib = ctypes.CDLL(_LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise NativeLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        lib.lyc_abi_version.restype = ctypes.c_int
        lib.lyc_abi_version.argtypes = []
        lib.lyc_last_error.restype = ctypes.c_char_p
        lib.lyc_last_error.argtypes = []
        got = lib.lyc_abi_version()
        if got != ABI_VERSION:
            raise NativeLibraryError(f"{_LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild the library")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError -> missing export, fail loudly
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
        for name, (argtypes, restype) in VALUE_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return _lib


_TORCH_EXT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lyc_torch.so")
_torch_ext = None


def load_torch_ops():
    """Import the TORCH_LIBRARY(lycoris_amd) extension (csrc/torch_ops.cpp: C++ dispatch + autograd over the C ABI).
    Like load(): no fallback -- a missing / unloadable extension is an error."""
    global _torch_ext
    if _torch_ext is not None:
        return _torch_ext
    with _lock:
        if _torch_ext is not None:
            return _torch_ext
        load()  # the C-ABI library first (same ABI check, clearer error)
        if not os.path.exists(_TORCH_EXT_PATH):
            raise NativeLibraryError(
                f"{_TORCH_EXT_PATH} not found: build the custom-op extension first (make -C lycoris_amd/csrc torch_ops, or "
                "python -c 'import __graft_entry__ as g; g.build()').  lycoris_amd has no fallback path.")
        import importlib.util
        spec = importlib.util.spec_from_file_location("lycoris_amd._lyc_torch", _TORCH_EXT_PATH)
        try:
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        except (ImportError, OSError) as e:
            raise NativeLibraryError(f"cannot load {_TORCH_EXT_PATH}: {e}") from e
        if mod.abi_version() != ABI_VERSION:
            raise NativeLibraryError(f"{_TORCH_EXT_PATH}: built against ABI {mod.abi_version()}, expected {ABI_VERSION}")
        _torch_ext = mod
        return mod


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.lyc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError(f"lycoris_amd supports float32/float16/bfloat16 activations, got {dtype}") from None


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"lycoris_amd: {what} is on {t.device}; the adapter hot path only runs on the MI355X HIP device "
            "(there is no CPU fallback by design)."
        )
