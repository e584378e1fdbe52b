"""LoCon / LoRA adapter module on the native path (interface of lycoris/modules/locon.py).

Parameters live in real ``lora_down`` / ``lora_up`` submodules exactly like upstream (locon.py:74-105), so
checkpoints interchange and sd-scripts' LoRA+ grouping (substring "lora_up", kohya.py:678) keeps working.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from ..functional.general import conv_args
from . import siblings as _siblings
from .base import LycorisBaseModule, _unsupported


class LoConModule(_siblings.SiblingMixin, LycorisBaseModule):
    name = "locon"
    _ws_algo = "locon"
    support_module = {"linear", "conv1d", "conv2d", "conv3d"}
    weight_list = ["lora_up.weight", "lora_down.weight", "lora_mid.weight", "alpha", "dora_scale"]
    weight_list_det = ["lora_up.weight"]

    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1, dropout=0.0,
                 rank_dropout=0.0, module_dropout=0.0, use_tucker=False, use_scalar=False, rank_dropout_scale=False,
                 weight_decompose=False, wd_on_out=True, bypass_mode=None, rs_lora=False, **kwargs):
        super().__init__(lora_name, org_module, multiplier, dropout, rank_dropout, module_dropout,
                         rank_dropout_scale, bypass_mode)
        if self.module_type not in self.support_module:
            raise ValueError(f"{self.module_type} is not supported in LoRA/LoCon algo.")
        if self.module_type == "conv1d":  # (an nn.Conv1d layer arrives here as its Conv2d twin, base.py _TwinMeta)
            raise _unsupported(f"LoCon on {self.module_type}")
        if weight_decompose and rank_dropout:
            raise _unsupported("rank_dropout together with weight_decompose")
        self.lora_dim = lora_dim
        self.rs_lora = rs_lora
        self._init_dora(org_module, weight_decompose, wd_on_out)
        self.tucker = False
        if self.module_type in ("conv2d", "conv3d"):
            self.isconv = True
            conv = nn.Conv3d if self.module_type == "conv3d" else nn.Conv2d  # (locon.py:74-95: `self.module(...)`)
            k = org_module.kernel_size
            self.tucker = bool(use_tucker) and any(i != 1 for i in k)
            if self.tucker:  # conv-CP form (locon.py:85-90): 1x1 down, k x k core r -> r, 1x1 up
                self.lora_down = conv(org_module.in_channels, lora_dim, 1, bias=False)
                self.lora_mid = conv(lora_dim, lora_dim, k, org_module.stride, org_module.padding, bias=False)
            else:
                self.lora_down = conv(org_module.in_channels, lora_dim, k, org_module.stride, org_module.padding,
                                      bias=False)
            self.lora_up = conv(lora_dim, org_module.out_channels, 1, bias=False)
        else:
            self.isconv = False
            self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
            self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        self._init_scale(lora_dim, alpha, rs_lora, use_scalar)
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        if use_scalar:
            nn.init.kaiming_uniform_(self.lora_up.weight, a=math.sqrt(5))
        else:
            nn.init.zeros_(self.lora_up.weight)
        if self.tucker:
            nn.init.kaiming_uniform_(self.lora_mid.weight, a=math.sqrt(5))

    @classmethod
    def make_module_from_state_dict(cls, lora_name, orig_module, up, down, mid, alpha, dora_scale):
        wd_on_out = dora_scale is None or dora_scale.reshape(-1).numel() == up.size(0) and dora_scale.shape[0] == up.size(0)
        mod = cls(lora_name, orig_module, 1, down.size(0), float(alpha), use_tucker=mid is not None,
                  weight_decompose=dora_scale is not None, wd_on_out=wd_on_out)
        mod.lora_up.weight.data.copy_(up)
        mod.lora_down.weight.data.copy_(down)
        if mid is not None:
            mod.lora_mid.weight.data.copy_(mid)
        if dora_scale is not None:
            mod.dora_scale.data.copy_(dora_scale.reshape(mod.dora_scale.shape))
        return mod

    def custom_state_dict(self):
        sd = {"alpha": self.alpha, "lora_up.weight": self.lora_up.weight * self.scalar,
              "lora_down.weight": self.lora_down.weight}
        if self.wd:
            sd["dora_scale"] = self.dora_scale
        if self.tucker:
            sd["lora_mid.weight"] = self.lora_mid.weight
        return sd

    def _down_eff(self):
        """the input-side factor as the kernels take it, [r, I, kh, kw]: lora_down itself, or with use_tucker the k x k
        core folded into the 1x1 down-projection (up(mid(down(x))) == up(conv(x, mid o down)): csrc/tucker.h)"""
        if not self.tucker:
            return self.lora_down.weight
        if self.lora_mid.weight.is_cuda and not self._aten_only:
            return ops.tucker_core(self.lora_mid.weight, self.lora_down.weight)
        return torch.einsum("ij...,jq->iq...", self.lora_mid.weight, self.lora_down.weight.flatten(1))  # offline / CPU / Conv3d

    def _ws_factors(self, gated=True):
        return (self._down_eff(), self._gate(self.lora_up.weight) if gated else self.lora_up.weight)

    # ---- dW materialisation (merge / export / max-norm only) -----------------------------------------------------
    def make_weight(self, device=None):
        up = self.lora_up.weight.to(device)
        down = self._down_eff().to(device)
        w = up.reshape(up.size(0), -1) @ down.reshape(down.size(0), -1)
        return w.reshape(self.shape) * self.scalar.to(device)

    def get_diff_weight(self, multiplier=1, shape=None, device=None):
        if self._native_ws():
            diff = ops.diff_weight("locon", self._ws_factors(), shape or self.shape, self.scale * multiplier)
            return (diff if device is None else diff.to(device)), None
        diff = self.make_weight(device=device) * (self.scale * multiplier)
        if shape is not None:
            diff = diff.view(shape)
        return (diff if device is None else diff.to(device)), None

    def get_merged_weight(self, multiplier=1, shape=None, device=None):
        if self._native_ws():
            return self._merged_weight_native(multiplier), None
        diff = self.get_diff_weight(multiplier=1, shape=shape, device=device)[0]
        if self.wd:
            return self._dora_merge_host(self.org_weight + diff, multiplier), None
        return self.org_weight + diff * multiplier, None

    @torch.no_grad()
    def apply_max_norm(self, max_norm, device=None):
        if self._native_ws():  # ||dW||_F straight from the factors (locon.py:273-284 builds the full dW for it)
            scaled, ratio, orig_norm = self._max_norm_native(max_norm)
            if scaled:
                self.scalar *= ratio.to(self.scalar.device)
                self._scalar_scaled = True
            return scaled, orig_norm * ratio.to(orig_norm.device)
        orig_norm = self.make_weight(device).norm() * self.scale
        norm = torch.clamp(orig_norm, max_norm / 2)
        desired = torch.clamp(norm, max=max_norm)
        ratio = desired.cpu() / norm.cpu()
        scaled = norm != desired
        if scaled:
            self.scalar *= ratio
            self._scalar_scaled = True
        return scaled, orig_norm * ratio

    # ---- hot path --------------------------------------------------------------------------------------------------
    def _sibling_eligible(self, x):
        """on the plain `base + delta` path of a LoCon nn.Linear layer (what lyc_locon_linear_fwd_group takes)?"""
        return (not self.isconv and not getattr(self, "wd", False) and _siblings.activation_ok(x)
                and not (self.training and (self.module_dropout or self.rank_dropout or (self.bypass_mode and self.dropout))))

    def _sibling_key(self):
        return ("locon", tuple(self.lora_down.weight.shape), tuple(self.lora_up.weight.shape))

    @staticmethod
    def _sibling_launch(members, x, bases):
        deltas = ops.locon_linear_group(x, [m.lora_down.weight for m in members], [m._gate(m.lora_up.weight) for m in members],
                                        [m.scale * m.multiplier for m in members])
        return [b + d for b, d in zip(bases, deltas)]

    def bypass_forward_diff(self, x, scale=1):
        """delta = up(down(x)) * scalar * alpha/r * scale  (locon.py:286-304 and :309-332 compute the same function)."""
        alpha = self.scale * scale
        up = self._gate(self.lora_up.weight)
        if not self.isconv:
            return ops.locon_linear(x, self.lora_down.weight, up, alpha)
        if self._aten_only:  # nn.Conv3d: F.conv3d(x, dW) in ATen ops
            return self._delta_aten(x, scale)
        stride, padding, dilation = conv_args(self.kw_dict)
        return ops.locon_conv2d(x, self._down_eff(), up, alpha, stride, padding, dilation)
