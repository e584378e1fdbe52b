"""LoKr adapter module on the native path (interface of lycoris/modules/lokr.py).

Shape logic (which of lokr_w1 | lokr_w1_a/b and lokr_w2 | lokr_w2_a/b exist, their sizes, the alpha/scale rule) is
pinned against the reference by tests/golden/shape_cases.json.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn

from .. import ops
from ..functional.general import conv_args, factorization
from ..functional.lokr import make_kron
from . import siblings as _siblings
from .base import LycorisBaseModule, _unsupported

logger = logging.getLogger("LyCORIS")
_warned = set()


def _warn_full_matrix(lora_dim, dim, factor):
    key = (lora_dim, dim, factor)
    if key not in _warned:
        _warned.add(key)
        logger.warning(f"lora_dim {lora_dim} is too large for dim={dim} and factor={factor}, using full matrix mode.")


class LokrModule(_siblings.SiblingMixin, LycorisBaseModule):
    name = "kron"
    _ws_algo = "lokr"
    support_module = {"linear", "conv1d", "conv2d", "conv3d"}
    weight_list = ["lokr_w1", "lokr_w1_a", "lokr_w1_b", "lokr_w2", "lokr_w2_a", "lokr_w2_b", "lokr_t1", "lokr_t2",
                   "alpha", "dora_scale"]
    weight_list_det = ["lokr_w1", "lokr_w1_a"]

    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1, dropout=0.0,
                 rank_dropout=0.0, module_dropout=0.0, use_tucker=False, use_scalar=False, decompose_both=False,
                 factor: int = -1, rank_dropout_scale=False, weight_decompose=False, wd_on_out=True,
                 full_matrix=False, bypass_mode=None, rs_lora=False, unbalanced_factorization=False, **kwargs):
        super().__init__(lora_name, org_module, multiplier, dropout, rank_dropout, module_dropout,
                         rank_dropout_scale, bypass_mode)
        if self.module_type not in self.support_module:
            raise ValueError(f"{self.module_type} is not supported in LoKr algo.")
        if self.module_type == "conv1d":  # (an nn.Conv1d layer arrives here as its Conv2d twin, base.py _TwinMeta)
            raise _unsupported(f"LoKr on {self.module_type}")
        if weight_decompose and rank_dropout:
            raise _unsupported("rank_dropout together with weight_decompose")
        factor = int(factor)
        self.lora_dim = lora_dim
        self.full_matrix = full_matrix
        self.rs_lora = rs_lora
        self._init_dora(org_module, weight_decompose, wd_on_out)
        self.tucker = False
        is_conv = self.module_type in ("conv2d", "conv3d")
        out_dim, in_dim = self.shape[0], self.shape[1]
        ksize = tuple(self.shape[2:])
        in_m, in_n = factorization(in_dim, factor)
        out_l, out_k = factorization(out_dim, factor)
        if unbalanced_factorization:
            out_l, out_k = out_k, out_l
        self.tucker = bool(use_tucker) and is_conv and any(i != 1 for i in ksize)

        # small factor w1: [out_l, in_m] (optionally rank-decomposed)
        self.use_w1 = not (decompose_both and lora_dim < max(out_l, in_m) / 2 and not full_matrix)
        if self.use_w1:
            self.lokr_w1 = nn.Parameter(torch.empty(out_l, in_m))
        else:
            self.lokr_w1_a = nn.Parameter(torch.empty(out_l, lora_dim))
            self.lokr_w1_b = nn.Parameter(torch.empty(lora_dim, in_m))
        # big factor w2: [out_k, in_n(, kh, kw)] (full when the rank would not save anything, lokr.py:109-136, 159-167)
        low_rank_w2 = lora_dim < max(out_k, in_n) / 2 and not full_matrix
        self.use_w2 = not low_rank_w2
        if self.use_w2:
            if not full_matrix:
                _warn_full_matrix(lora_dim, max(in_dim, out_dim), factor)
            self.lokr_w2 = nn.Parameter(torch.empty(out_k, in_n, *ksize))
        elif self.tucker:  # lokr.py:121-128: core [r, r, kh, kw], "1-mode" [r, out_k], "2-mode" [r, in_n]
            self.lokr_t2 = nn.Parameter(torch.empty(lora_dim, lora_dim, *ksize))
            self.lokr_w2_a = nn.Parameter(torch.empty(lora_dim, out_k))
            self.lokr_w2_b = nn.Parameter(torch.empty(lora_dim, in_n))
        else:
            kprod = 1
            for i in ksize:
                kprod *= i
            self.lokr_w2_a = nn.Parameter(torch.empty(out_k, lora_dim))
            self.lokr_w2_b = nn.Parameter(torch.empty(lora_dim, in_n * kprod))
        self._kron_dims = (out_l, in_m, out_k, in_n)

        self._init_scale(lora_dim, alpha, rs_lora, use_scalar, force_unit_scale=self.use_w1 and self.use_w2)
        if self.use_w2:
            if use_scalar:
                nn.init.kaiming_uniform_(self.lokr_w2, a=math.sqrt(5))
            else:
                nn.init.zeros_(self.lokr_w2)
        else:
            if self.tucker and not self.use_w2:
                nn.init.kaiming_uniform_(self.lokr_t2, a=math.sqrt(5))
            nn.init.kaiming_uniform_(self.lokr_w2_a, a=math.sqrt(5))
            if use_scalar:
                nn.init.kaiming_uniform_(self.lokr_w2_b, a=math.sqrt(5))
            else:
                nn.init.zeros_(self.lokr_w2_b)
        if self.use_w1:
            nn.init.kaiming_uniform_(self.lokr_w1, a=math.sqrt(5))
        else:
            nn.init.kaiming_uniform_(self.lokr_w1_a, a=math.sqrt(5))
            nn.init.kaiming_uniform_(self.lokr_w1_b, a=math.sqrt(5))

    @classmethod
    def make_module_from_state_dict(cls, lora_name, orig_module, w1, w1a, w1b, w2, w2a, w2b, _, t2, alpha,
                                    dora_scale):
        """Rebuild a module from checkpoint tensors: find the ``factor`` that reproduces the stored factor shapes."""
        full_matrix = w1a is None and w2a is None
        if t2 is not None:  # Tucker: w2_a is [r, out_k]
            lora_dim = t2.size(0)
        else:
            lora_dim = w1a.size(1) if w1a is not None else (w2a.size(1) if w2a is not None else 1)
        a, b = (w1.shape if w1 is not None else (w1a.size(0), w1b.size(1)))
        c = w2.size(0) if w2 is not None else (w2a.size(1) if t2 is not None else w2a.size(0))
        probe = cls.__new__(cls)  # only to read the wrapped layer's dims through the base bookkeeping
        nn.Module.__init__(probe)
        LycorisBaseModule.__init__(probe, lora_name, orig_module)
        out_dim, in_dim = probe.shape[0], probe.shape[1]
        d = in_dim // b
        candidates = [-1] + sorted({a, b, c, d, max(a, b), min(a, b)})
        factor = None
        for f in candidates:
            if f == 0:
                continue
            if factorization(out_dim, f) == (a, c) and factorization(in_dim, f) == (b, d):
                factor = f
                break
        if factor is None:
            raise ValueError(f"cannot infer LoKr factor for {lora_name}: w1 {a}x{b}, layer {out_dim}x{in_dim}")
        mod = cls(lora_name, orig_module, 1, lora_dim, float(alpha), decompose_both=w1 is None and w2 is None,
                  factor=factor, full_matrix=full_matrix, use_tucker=t2 is not None, weight_decompose=dora_scale is not None,
                  wd_on_out=dora_scale is None or dora_scale.shape[0] == out_dim)
        if dora_scale is not None:
            mod.dora_scale.data.copy_(dora_scale.reshape(mod.dora_scale.shape))
        with torch.no_grad():
            for name, val in (("lokr_w1", w1), ("lokr_w1_a", w1a), ("lokr_w1_b", w1b), ("lokr_w2", w2),
                              ("lokr_w2_a", w2a), ("lokr_w2_b", w2b), ("lokr_t2", t2)):
                if val is not None:
                    getattr(mod, name).copy_(val)
        return mod

    def custom_state_dict(self):
        sd = {"alpha": self.alpha}
        if self.wd:
            sd["dora_scale"] = self.dora_scale
        if self.use_w1:
            sd["lokr_w1"] = self.lokr_w1 * self.scalar
        else:
            sd["lokr_w1_a"] = self.lokr_w1_a * self.scalar
            sd["lokr_w1_b"] = self.lokr_w1_b
        if self.use_w2:
            sd["lokr_w2"] = self.lokr_w2
        else:
            sd["lokr_w2_a"] = self.lokr_w2_a
            sd["lokr_w2_b"] = self.lokr_w2_b
            if self.tucker:
                sd["lokr_t2"] = self.lokr_t2
        return sd

    # ---- factors -------------------------------------------------------------------------------------------------
    def _w1_full(self):
        return self.lokr_w1 if self.use_w1 else self.lokr_w1_a @ self.lokr_w1_b

    def _w2_full(self):
        if self.use_w2:
            return self.lokr_w2
        out_k, in_n = self._kron_dims[2], self._kron_dims[3]
        if self.tucker:  # rebuild_tucker(t2, w2_a, w2_b) = w2_a^T @ fold(t2, w2_b)   (general.py:9-11, csrc/tucker.h)
            if self.lokr_t2.is_cuda and not self._aten_only:
                fold = ops.tucker_core(self.lokr_t2, self.lokr_w2_b)
            else:
                fold = torch.einsum("ij...,jq->iq...", self.lokr_t2, self.lokr_w2_b)  # offline / CPU / Conv3d
            return (self.lokr_w2_a.t() @ fold.flatten(1)).reshape(out_k, in_n, *self.shape[2:])
        return (self.lokr_w2_a @ self.lokr_w2_b).reshape(out_k, in_n, *self.shape[2:])

    def _ws_factors(self, gated=True):
        w1 = self._w1_full()
        return (self._gate(w1) if gated else w1, self._w2_full())

    # ---- dW materialisation (merge / export / max-norm only) -----------------------------------------------------
    def get_weight(self, shape):
        w = make_kron(self._w1_full(), self._w2_full(), self.scale)
        return w if shape is None else w.view(shape)

    def get_diff_weight(self, multiplier=1, shape=None, device=None):
        # scale applied once (upstream applies it twice here, lokr.py:383-385 with :370)
        if self._native_ws():
            diff = ops.diff_weight("lokr", self._ws_factors(), shape or self.shape, self.scale * multiplier)
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
        if self._native_ws():  # lokr.py:442-466 takes the norm of the un-gated kron(w1, w2) * scale
            scaled, ratio, orig_norm = self._max_norm_native(max_norm, gated=False)
        else:
            orig_norm = self.get_weight(self.shape).norm()
            norm = torch.clamp(orig_norm, max_norm / 2)
            desired = torch.clamp(norm, max=max_norm)
            ratio = desired.cpu() / norm.cpu()
            scaled = norm != desired
        if scaled:
            factors = [p for n, p in self.named_parameters() if n.startswith("lokr_")]
            for p in factors:
                p *= ratio ** (1 / len(factors))
        return scaled, orig_norm * ratio.to(orig_norm.device)

    # ---- hot path --------------------------------------------------------------------------------------------------
    def _sibling_eligible(self, x):
        """on the plain `base + delta` path of a LoKr nn.Linear layer with a full w1 and a full-matrix or low-rank (not Tucker) w2 -- what
        lyc_lokr_linear_fwd_group takes?"""
        return (self.module_type == "linear" and self.use_w1 and (self.use_w2 or not self.tucker) and not self.wd and _siblings.activation_ok(x)
                and not (self.training and (self.module_dropout or self.rank_dropout or (self.bypass_mode and self.dropout))))

    def _sibling_key(self):
        w2 = (tuple(self.lokr_w2.shape),) if self.use_w2 else (tuple(self.lokr_w2_a.shape), tuple(self.lokr_w2_b.shape))
        return ("lokr", tuple(self.lokr_w1.shape), w2)

    @staticmethod
    def _sibling_launch(members, x, bases):
        w1s, alphas = [m._gate(m.lokr_w1) for m in members], [m.scale * m.multiplier for m in members]
        if members[0].use_w2:
            return ops.lokr_linear_group(x, w1s, [m.lokr_w2 for m in members], alphas, bases)
        # low-rank second factor: the pairs go to the kernels as they are (planes from the factors, grouped chain rule)
        return ops.lokr_linear_lr_group(x, w1s, [m.lokr_w2_a for m in members], [m.lokr_w2_b for m in members], alphas, bases)

    def _owned_args(self, x):
        """(W, bias, w1, w2, alpha) when ops.lokr_adapted_linear can hold this layer and its adapter in one node, else None"""
        if self.module_type != "linear" or not (self.use_w1 and self.use_w2) or not isinstance(x, torch.Tensor):
            return None
        fl = self._frozen_linear()
        if fl is None:
            return None
        w1 = self._gate(self.lokr_w1)
        if not ops.lokr_linear_ownable(x, w1, self.lokr_w2, fl[0], fl[1]):
            return None
        return fl[0], fl[1], w1, self.lokr_w2, self.scale * self.multiplier

    def _forward_owned(self, x):
        oa = self._owned_args(x)
        if oa is None:
            return None
        return ops.lokr_adapted_linear(x, [oa[0]], [oa[1]], [oa[2]], [oa[3]], [oa[4]])[0]

    @staticmethod
    def _sibling_launch_owned(members, x):
        oas = [m._owned_args(x) for m in members]
        if any(oa is None for oa in oas):
            return None
        return ops.lokr_adapted_linear(x, [oa[0] for oa in oas], [oa[1] for oa in oas], [oa[2] for oa in oas], [oa[3] for oa in oas],
                                       [oa[4] for oa in oas])

    def _forward_fused(self, x, base):
        if self.module_type != "linear":
            return None
        if self._w2_low_rank_native(x) and not self.use_w1:  # decompose_both: all four factors go to the kernels as they are
            w1s = ops._Shape2(self.lokr_w1_a.shape[0], self.lokr_w1_b.shape[1])
            if not ops.lokr_linear_fusable(x, w1s, ops._Shape2(self.lokr_w2_a.shape[0], self.lokr_w2_b.shape[1]), base):
                return None
            return ops.lokr_linear_lr2(x, self._gate(self.lokr_w1_a), self.lokr_w1_b, self.lokr_w2_a, self.lokr_w2_b,
                                       self.scale * self.multiplier, base=base)
        w1 = self._gate(self._w1_full())
        if self._w2_low_rank_native(x):  # the factors go to the kernels as they are: no w2_a @ w2_b product, no autograd mm
            if not ops.lokr_linear_fusable(x, w1, ops._Shape2(self.lokr_w2_a.shape[0], self.lokr_w2_b.shape[1]), base):
                return None
            return ops.lokr_linear_lr(x, w1, self.lokr_w2_a, self.lokr_w2_b, self.scale * self.multiplier, base=base)
        w2 = self._w2_full()
        if not ops.lokr_linear_fusable(x, w1, w2, base):
            return None
        return ops.lokr_linear(x, w1, w2, self.scale * self.multiplier, base=base)

    def _w2_low_rank_native(self, x):
        return not self.use_w2 and not self.tucker and self.module_type == "linear" and x.is_cuda

    def bypass_forward_diff(self, h, scale=1):
        """delta = (w1 (x) w2) h * alpha/r * scalar * scale, Kronecker-factored.

        Unlike upstream's bypass (lokr.py:538) this includes ``self.scale``, i.e. it equals the rebuild
        path lokr.py:543-566, which is the canonical semantics."""
        alpha = self.scale * scale
        if self._aten_only:  # nn.Conv3d: F.conv3d(x, kron(w1, w2)) in ATen ops
            return self._delta_aten(h, scale)
        if self._w2_low_rank_native(h) and not self.use_w1:
            return ops.lokr_linear_lr2(h, self._gate(self.lokr_w1_a), self.lokr_w1_b, self.lokr_w2_a, self.lokr_w2_b, alpha)
        w1 = self._gate(self._w1_full())
        if self._w2_low_rank_native(h):
            return ops.lokr_linear_lr(h, w1, self.lokr_w2_a, self.lokr_w2_b, alpha)
        if not self.use_w2 and not self.tucker and self.module_type == "conv2d" and h.is_cuda:
            stride, padding, dilation = conv_args(self.kw_dict)
            return ops.lokr_conv2d_lr(h, w1, self.lokr_w2_a, self.lokr_w2_b, alpha, self.shape[2:], stride, padding, dilation)
        w2 = self._w2_full()
        if self.module_type == "linear":
            return ops.lokr_linear(h, w1, w2, alpha)
        stride, padding, dilation = conv_args(self.kw_dict)
        return ops.lokr_conv2d(h, w1, w2, alpha, stride, padding, dilation)
