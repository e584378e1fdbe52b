"""(IA)^3 adapter module on the native path (interface of lycoris/modules/ia3.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .base import LycorisBaseModule, _unsupported


class IA3Module(LycorisBaseModule):
    name = "ia3"
    support_module = {"linear", "conv1d", "conv2d", "conv3d"}
    weight_list = ["weight", "on_input"]
    weight_list_det = ["on_input"]

    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1, dropout=0.0,
                 rank_dropout=0.0, module_dropout=0.0, use_tucker=False, use_scalar=False, rank_dropout_scale=False,
                 weight_decompose=False, bypass_mode=None, rs_lora=False, train_on_input=False, **kwargs):
        super().__init__(lora_name, org_module, multiplier, dropout, rank_dropout, module_dropout,
                         rank_dropout_scale, bypass_mode)
        if self.module_type not in self.support_module:
            raise ValueError(f"{self.module_type} is not supported in IA^3 algo.")
        if self.module_type == "conv1d":  # (an nn.Conv1d layer arrives here as its Conv2d twin, base.py _TwinMeta)
            raise _unsupported(f"(IA)^3 on {self.module_type}")
        self.isconv = self.module_type in ("conv2d", "conv3d")
        train_dim = self.shape[1] if train_on_input else self.shape[0]
        if self.isconv:
            self.weight = nn.Parameter(torch.zeros(1, train_dim, *(1 for _ in self.shape[2:])))  # ia3.py:59-62
        else:
            self.weight = nn.Parameter(torch.zeros(train_dim))
        self.train_input = train_on_input
        self.register_buffer("on_input", torch.tensor(int(train_on_input)))

    @classmethod
    def make_module_from_state_dict(cls, lora_name, orig_module, weight, on_input=None):
        # upstream's signature takes one tensor but its weight_list extracts two; accept both here
        mod = cls(lora_name, orig_module, 1, train_on_input=bool(int(on_input)) if on_input is not None else False)
        mod.weight.data.copy_(weight)
        return mod

    # ---- dW materialisation (merge / export) --------------------------------------------------------------------
    def make_weight(self, multiplier=1, shape=None, device=None, diff=False):
        w = self.weight.reshape(-1) * multiplier + int(not diff)
        bshape = [1] * len(self.shape)
        bshape[1 if self.train_input else 0] = -1
        out = self.org_weight * w.reshape(bshape)
        if shape is not None:
            out = out.view(shape)
        return out if device is None else out.to(device)

    def get_diff_weight(self, multiplier=1, shape=None, device=None):
        return self.make_weight(multiplier, shape, device, diff=True), None

    def get_merged_weight(self, multiplier=1, shape=None, device=None):
        return self.make_weight(multiplier, shape, device), None

    # ---- hot path --------------------------------------------------------------------------------------------------
    def _chan_affine(self, *args):
        if self._aten_only:  # nn.Conv3d: ATen ops on any device (base.py _aten_only)
            from .. import composite
            return composite.chan_affine(*args)
        return ops.chan_affine(*args)

    def bypass_forward_diff(self, x, scale=1):
        """op(x, W * (w*scale)) -- rebuild-path semantics: the layer bias is NOT scaled (ia3.py:91-102 vs the
        upstream bypass :114-121 which scales it)."""
        chan = 1 if self.isconv else -1
        if self.train_input:
            xs = self._chan_affine(x, self.weight, None, 0.0, scale, chan)
            return self.op(xs, self._current_weight(), None, **self.kw_dict)
        base_nobias = self.op(x, self._current_weight(), None, **self.kw_dict)
        return self._chan_affine(base_nobias, self.weight, None, 0.0, scale, chan)

    def forward(self, x, *args, **kwargs):
        base = self.org_forward(x, *args, **kwargs)
        chan = 1 if self.isconv else -1
        if self.train_input:
            # the dense op on the scaled input stays with rocBLAS / MIOpen (it is the frozen layer's own GEMM shape)
            xs = self._chan_affine(x, self.weight, None, 0.0, self.multiplier, chan)
            return base + self.op(xs, self._current_weight(), None, **self.kw_dict)
        # out-side: delta = (base - bias) * w*mult, fused as  y = base * (1 + w*mult) - bias * w*mult
        return self._chan_affine(base, self.weight, self._current_bias(), 1.0, self.multiplier, chan)
