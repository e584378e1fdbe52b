"""LoHa adapter module on the native path (interface of lycoris/modules/loha.py)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from ..functional.general import conv_args
from .base import LycorisBaseModule, _unsupported


class LohaModule(LycorisBaseModule):
    name = "loha"
    _ws_algo = "loha"
    support_module = {"linear", "conv1d", "conv2d", "conv3d"}
    weight_list = ["hada_w1_a", "hada_w1_b", "hada_w2_a", "hada_w2_b", "hada_t1", "hada_t2", "alpha", "dora_scale"]
    weight_list_det = ["hada_w1_a"]

    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1, dropout=0.0,
                 rank_dropout=0.0, module_dropout=0.0, use_tucker=False, use_scalar=False, rank_dropout_scale=False,
                 weight_decompose=False, wd_on_out=True, bypass_mode=None, rs_lora=False, **kwargs):
        super().__init__(lora_name, org_module, multiplier, dropout, rank_dropout, module_dropout,
                         rank_dropout_scale, bypass_mode)
        if self.module_type not in self.support_module:
            raise ValueError(f"{self.module_type} is not supported in LoHa algo.")
        if self.module_type == "conv1d":  # (an nn.Conv1d layer arrives here as its Conv2d twin, base.py _TwinMeta)
            raise _unsupported(f"LoHa on {self.module_type}")
        if weight_decompose and rank_dropout:
            raise _unsupported("rank_dropout together with weight_decompose")
        self.lora_dim = lora_dim
        self.rs_lora = rs_lora
        self._init_dora(org_module, weight_decompose, wd_on_out)
        self.tucker = False
        out_dim, in_flat = self.shape[0], self.shape[1]
        if self.module_type in ("conv2d", "conv3d"):
            k = org_module.kernel_size
            self.tucker = bool(use_tucker) and any(i != 1 for i in k)
            in_flat = self.shape[1] * math.prod(k)  # non-Tucker conv factors are [r, I*kh*kw] (loha.py:76)
        if self.tucker:  # loha.py:78-93: cores [r, r, kh, kw], a-side [r, O] ("1-mode"), b-side [r, I] ("2-mode")
            self.hada_t1 = nn.Parameter(torch.empty(lora_dim, lora_dim, *self.shape[2:]))
            self.hada_w1_a = nn.Parameter(torch.empty(lora_dim, out_dim))
            self.hada_w1_b = nn.Parameter(torch.empty(lora_dim, self.shape[1]))
            self.hada_t2 = nn.Parameter(torch.empty(lora_dim, lora_dim, *self.shape[2:]))
            self.hada_w2_a = nn.Parameter(torch.empty(lora_dim, out_dim))
            self.hada_w2_b = nn.Parameter(torch.empty(lora_dim, self.shape[1]))
            nn.init.normal_(self.hada_t1, std=0.1)
            nn.init.normal_(self.hada_t2, std=0.1)
        else:
            self.hada_w1_a = nn.Parameter(torch.empty(out_dim, lora_dim))
            self.hada_w1_b = nn.Parameter(torch.empty(lora_dim, in_flat))
            self.hada_w2_a = nn.Parameter(torch.empty(out_dim, lora_dim))
            self.hada_w2_b = nn.Parameter(torch.empty(lora_dim, in_flat))
        self._init_scale(lora_dim, alpha, rs_lora, use_scalar)
        nn.init.normal_(self.hada_w1_b, std=1)
        nn.init.normal_(self.hada_w1_a, std=0.1)
        nn.init.normal_(self.hada_w2_b, std=1)
        if use_scalar:
            nn.init.normal_(self.hada_w2_a, std=0.1)
        else:
            nn.init.zeros_(self.hada_w2_a)

    @classmethod
    def make_module_from_state_dict(cls, lora_name, orig_module, w1a, w1b, w2a, w2b, t1, t2, alpha, dora_scale):
        wd_on_out = dora_scale is None or dora_scale.shape[0] == w1a.size(0)
        mod = cls(lora_name, orig_module, 1, w1b.size(0), float(alpha), use_tucker=t1 is not None,
                  weight_decompose=dora_scale is not None, wd_on_out=wd_on_out)
        for p, v in ((mod.hada_w1_a, w1a), (mod.hada_w1_b, w1b), (mod.hada_w2_a, w2a), (mod.hada_w2_b, w2b)):
            p.data.copy_(v)
        if t1 is not None:
            mod.hada_t1.data.copy_(t1)
            mod.hada_t2.data.copy_(t2)
        if dora_scale is not None:
            mod.dora_scale.data.copy_(dora_scale.reshape(mod.dora_scale.shape))
        return mod

    def custom_state_dict(self):
        sd = {"alpha": self.alpha, "hada_w1_a": self.hada_w1_a * self.scalar, "hada_w1_b": self.hada_w1_b,
              "hada_w2_a": self.hada_w2_a, "hada_w2_b": self.hada_w2_b}
        if self.wd:
            sd["dora_scale"] = self.dora_scale
        if self.tucker:
            sd["hada_t1"] = self.hada_t1
            sd["hada_t2"] = self.hada_t2
        return sd

    def _fold(self, t, wb):
        if t.is_cuda and not self._aten_only:
            return ops.tucker_core(t, wb).flatten(1)
        return torch.einsum("ij...,jq->iq...", t, wb).flatten(1)  # offline / CPU / Conv3d

    def _ws_factors(self, gated=True):
        """(w1a [O, r], w1b [r, I*kh*kw], w2a, w2b) as the kernels take them.  Tucker (HadaWeightTucker, functional/loha.py:
        33-75): rebuild_k = w_k_a^T @ fold(t_k, w_k_b), i.e. the plain form on the transposed a-side and the folded b-side."""
        a1 = self._gate(self.hada_w1_a) if gated else self.hada_w1_a
        if not self.tucker:
            return (a1, self.hada_w1_b, self.hada_w2_a, self.hada_w2_b)
        return (a1.t(), self._fold(self.hada_t1, self.hada_w1_b), self.hada_w2_a.t(), self._fold(self.hada_t2, self.hada_w2_b))

    # ---- dW materialisation (merge / export / max-norm only) -----------------------------------------------------
    def get_weight(self, shape):
        a1, b1, a2, b2 = self._ws_factors(gated=False)
        w = (a1 @ b1) * (a2 @ b2) * self.scale
        return w if shape is None else w.reshape(shape)

    def get_diff_weight(self, multiplier=1, shape=None, device=None):
        # NB upstream multiplies by self.scale a second time here (loha.py:228-230 with :206); the trained
        # forward uses scale once, and merging must reproduce the trained forward, so scale is applied once.
        if self._native_ws():
            diff = ops.diff_weight("loha", self._ws_factors(), shape or self.shape, self.scale * multiplier)
            return (diff if device is None else diff.to(device)), None
        diff = self.get_weight(shape) * self.scalar * multiplier
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
        if self._native_ws():  # ||dW||_F from the factors, tile by tile on chip (loha.py:281-292 builds dW)
            scaled, ratio, orig_norm = self._max_norm_native(max_norm)
            if scaled:
                self.scalar *= ratio.to(self.scalar.device)
                self._scalar_scaled = True
            return scaled, orig_norm * ratio.to(orig_norm.device)
        orig_norm = (self.get_weight(self.shape) * self.scalar).norm()
        norm = torch.clamp(orig_norm, max_norm / 2)
        desired = torch.clamp(norm, max=max_norm)
        ratio = desired.cpu() / norm.cpu()
        scaled = norm != desired
        if scaled:
            self.scalar *= ratio
            self._scalar_scaled = True
        return scaled, orig_norm * ratio

    # ---- hot path --------------------------------------------------------------------------------------------------
    def bypass_forward_diff(self, x, scale=1):
        """delta = op(x, ((w1a w1b) * (w2a w2b)) * alpha/r * scalar * scale)  (loha.py:294-299 and :301-322)."""
        alpha = self.scale * scale
        w1a, w1b, w2a, w2b = self._ws_factors()
        if self.module_type == "linear":
            return ops.loha_linear(x, w1a, w1b, w2a, w2b, alpha)
        if self._aten_only:  # nn.Conv3d: F.conv3d(x, dW) in ATen ops
            return self._delta_aten(x, scale)
        stride, padding, dilation = conv_args(self.kw_dict)
        return ops.loha_conv2d(x, w1a, w1b, w2a, w2b, alpha, tuple(self.shape), stride, padding, dilation)
