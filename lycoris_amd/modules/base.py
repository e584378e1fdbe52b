"""Adapter module base: the plug boundary between a host model's nn.Linear / nn.Conv2d and the HIP kernels.

Interface contract taken from the reference (lycoris/modules/base.py): constructor bookkeeping (:71-198),
forward patching with a per-layer wrapper stack (``apply_to`` :271-287, ``restore`` :289-324), ``merge_to``
(:326-342), ``onfly_merge`` / ``onfly_restore`` (:344-374), state-dict customisation (:11-61) and the registry
class methods ``algo_check`` / ``extract_state_dict`` / ``make_module_from_state_dict`` (:236-246).  The same
attribute names (``_lycoris_wrappers``, ``_lycoris_original_forward``) are used on the wrapped layer so native
and reference adapters can be stacked on one layer.

``forward`` on a HIP tensor always runs the HIP kernels (and fails loudly without the extension); a host tensor takes the same
entry points of ``lycoris_amd.ops`` and is evaluated there by the ATen composite forms of ``lycoris_amd/composite.py`` (device
dispatch, round 5: BASELINE configs[0] trains a small MLP on the CPU).  ``get_diff_weight`` / ``merge_to`` materialise dW with
plain tensor ops on the CPU -- that is the definition of merging, it is off the training hot path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

_STACK = "_lycoris_wrappers"
_ORIG = "_lycoris_original_forward"


def _unsupported(what: str):
    return NotImplementedError(
        f"lycoris_amd: {what} is not on the native path (see DESIGN.md, 'out of scope / next'); "
        "there is no silent fallback."
    )


class _Conv1dTwin(nn.Conv2d):
    """nn.Conv1d seen as nn.Conv2d over [B, C, 1, L] (round 3).  The adapter kernels are Conv2d kernels; a 1-D convolution with a
    window of k is the 2-D one with a 1 x k window, so a native module adapts an nn.Conv1d layer by being built on THIS object.
    No parameters of its own: `weight` / `bias` are live views of the real layer's (device / dtype moves, in-place merges and
    `.data` writes of either are seen by both), so the twin never has to be kept in sync."""

    def __init__(self, real: nn.Conv1d):
        if real.padding_mode != "zeros":  # the same construction-time rejection an nn.Conv2d layer gets
            raise _unsupported(f"nn.Conv1d with padding_mode={real.padding_mode!r}")
        nn.Module.__init__(self)  # deliberately not nn.Conv2d.__init__: nothing is allocated
        self.__dict__["_real"] = real  # not a submodule: the frozen layer must stay out of the adapter's parameters
        self.in_channels, self.out_channels = real.in_channels, real.out_channels
        self.kernel_size = (1, real.kernel_size[0])
        self.stride = (1, real.stride[0])
        self.padding = real.padding if isinstance(real.padding, str) else (0, real.padding[0])
        self.dilation = (1, real.dilation[0])
        self.groups = real.groups
        self.padding_mode = real.padding_mode
        self.transposed = False
        self.output_padding = (0, 0)

    @property
    def weight(self):
        return self._real.weight.unsqueeze(2)

    @property
    def bias(self):
        return self._real.bias

    def __setattr__(self, name, value):
        # nn.Module.__setattr__ sends an nn.Parameter to register_parameter BEFORE any property setter is looked at (and that raises
        # "attribute 'bias' already exists" because of the property): `layer.bias = nn.Parameter(b)` of the merge paths
        # (merge_to / onfly_merge on a bias-less layer) has to reach the REAL layer.  `weight` likewise, in the real layer's 3-D shape.
        if name == "bias":
            self._real.bias = value
        elif name == "weight":
            w = value.data if isinstance(value, nn.Parameter) else value
            self._real.weight = nn.Parameter(w.squeeze(2) if w.dim() == 4 else w, requires_grad=bool(getattr(value, "requires_grad", False)))
        else:
            super().__setattr__(name, value)

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class _TwinMeta(type):
    """Module construction on an nn.Conv1d layer: the class is instantiated on the layer's Conv2d twin, then told about the real
    layer (forward patching, state-dict shapes).  Everything between -- parameter shapes, kernels, weight space -- is the Conv2d
    path with a 1 x k window."""

    def __call__(cls, lora_name, org_module=None, *args, **kwargs):
        if isinstance(org_module, nn.Conv1d):
            obj = super().__call__(lora_name, _Conv1dTwin(org_module), *args, **kwargs)
            obj._attach_conv1d(org_module)
            return obj
        return super().__call__(lora_name, org_module, *args, **kwargs)


def _lift1d(t):
    """a Conv1d-shaped adapter tensor ([.., .., k]) in the native 4-D form ([.., .., 1, k]); everything else unchanged"""
    return t.unsqueeze(2) if isinstance(t, torch.Tensor) and t.dim() == 3 else t


def _drop1d(t):
    return t.squeeze(2) if isinstance(t, torch.Tensor) and t.dim() == 4 and t.shape[2] == 1 else t


class LycorisBaseModule(nn.Module, metaclass=_TwinMeta):
    name: str = "base"
    support_module: set = set()
    weight_list: list = []
    weight_list_det: list = []
    # nn.Conv3d ("Linear + Conv2d native; Conv1d / 3d -> reference fallback"): the adapter is evaluated in the
    # reference's own rebuild form, delta = F.conv3d(x, dW), with ATen ops on whatever device the tensors live on -- no HIP kernel,
    # no weight-space kernels (modules/base.py:89-158 kw_dict dispatch, functional/general.py:6 FUNC_LIST)
    _aten_only = False

    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, dropout=0.0, rank_dropout=0.0,
                 module_dropout=0.0, rank_dropout_scale=False, bypass_mode=None, **kwargs):
        super().__init__()
        self.lora_name = lora_name
        self._conv1d = None  # the real nn.Conv1d layer when this module was built on its Conv2d twin (_attach_conv1d)
        self.not_supported = False
        self.module = type(org_module)
        self.kw_dict = {}
        if isinstance(org_module, nn.Linear):
            self.module_type = "linear"
            self.shape = (org_module.out_features, org_module.in_features)
            self.op = F.linear
            self.dim = org_module.out_features
        elif isinstance(org_module, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            nd = {nn.Conv1d: 1, nn.Conv2d: 2, nn.Conv3d: 3}[next(c for c in (nn.Conv1d, nn.Conv2d, nn.Conv3d)
                                                                  if isinstance(org_module, c))]
            self.module_type = f"conv{nd}d"
            self.shape = (org_module.out_channels, org_module.in_channels, *org_module.kernel_size)
            self.op = (F.conv1d, F.conv2d, F.conv3d)[nd - 1]
            self.dim = org_module.out_channels
            self.kw_dict = {"stride": org_module.stride, "padding": self._int_padding(org_module),
                            "dilation": org_module.dilation, "groups": org_module.groups}
            if nd == 2:
                # fail when the network is BUILT (not at the first training step) for layer variants the kernels do not
                # take: grouped convolutions, padding="same"/"valid" strings, non-zero padding modes
                from ..functional.general import conv_args
                conv_args(self.kw_dict)
                if getattr(org_module, "padding_mode", "zeros") != "zeros":
                    raise _unsupported(f"Conv2d padding_mode={org_module.padding_mode!r}")
            if nd == 3:
                if org_module.groups != 1:
                    raise _unsupported("grouped nn.Conv3d layers")
                if getattr(org_module, "padding_mode", "zeros") != "zeros":
                    raise _unsupported(f"Conv3d padding_mode={org_module.padding_mode!r}")
                self._aten_only = True
        elif isinstance(org_module, nn.LayerNorm):
            self.module_type = "layernorm"
            self.shape = tuple(org_module.normalized_shape)
            self.op = F.layer_norm
            self.dim = org_module.normalized_shape[0]
            self.kw_dict = {"normalized_shape": org_module.normalized_shape, "eps": org_module.eps}
        elif isinstance(org_module, nn.GroupNorm):
            self.module_type = "groupnorm"
            self.shape = (org_module.num_channels,)
            self.op = F.group_norm
            self.group_num = org_module.num_groups
            self.dim = org_module.num_channels
            self.kw_dict = {"num_groups": org_module.num_groups, "eps": org_module.eps}
        else:
            self.not_supported = True
            self.module_type = "unknown"

        self.register_buffer("dtype_tensor", torch.tensor(0.0), persistent=False)
        # A Linear subclass that is not exactly nn.Linear is treated as quantised by the reference and forced into
        # bypass mode (base.py:162-177).  Natively both modes are the same factored computation on x.
        self.is_quant = isinstance(org_module, nn.Linear) and type(org_module).__name__ != "Linear"
        if self.is_quant and bypass_mode is None:
            bypass_mode = True
        self.bypass_mode = bypass_mode
        self.dropout = dropout
        self.rank_dropout = rank_dropout
        self.rank_dropout_scale = rank_dropout_scale
        self.module_dropout = module_dropout
        # dropout variants (reference: locon.py:210-217,292-304,310-312, loha.py:220-225, lokr.py:375-380,544-546), all
        # applied around the kernels without materialising dW -- see forward() below
        self.drop = nn.Dropout(dropout) if dropout else nn.Identity()
        self.rank_drop = nn.Identity()
        self.multiplier = multiplier
        self.org_forward = org_module.forward
        self.org_module = [org_module]  # list: keeps the frozen layer out of this module's parameters

    @staticmethod
    def _int_padding(org_module):
        """padding="valid" / "same" of the frozen layer as the integer padding the kernels take (round 5; the reference hands the
        string through to F.conv*, modules/base.py:101-121).  "same" needs stride 1 (torch enforces it) and pads dilation * (k - 1) in
        total per dimension; when that is odd torch pads one more element on the right, which no symmetric integer expresses."""
        pad = org_module.padding
        if not isinstance(pad, str):
            return pad
        if pad == "valid":
            return tuple(0 for _ in org_module.kernel_size)
        if pad == "same":
            total = [d * (k - 1) for k, d in zip(org_module.kernel_size, org_module.dilation)]
            if any(t % 2 for t in total):
                raise _unsupported(f'padding="same" with an asymmetric pad (kernel {tuple(org_module.kernel_size)}, dilation '
                                   f"{tuple(org_module.dilation)})")
            return tuple(t // 2 for t in total)
        raise _unsupported(f"padding={pad!r}")

    # ---- registry protocol (lycoris/modules/__init__.py:33-46) -------------------------------------------------
    @classmethod
    def algo_check(cls, state_dict, lora_name):
        return any(f"{lora_name}.{k}" in state_dict for k in cls.weight_list_det)

    @classmethod
    def extract_state_dict(cls, state_dict, lora_name):
        return [state_dict.get(f"{lora_name}.{k}", None) for k in cls.weight_list]

    @classmethod
    def make_module_from_state_dict(cls, lora_name, orig_module, *weights):
        raise NotImplementedError

    @classmethod
    def parametrize(cls, org_module, attr, *args, **kwargs):
        """Register the adapter as a torch parametrization of `org_module.<attr>` (reference modules/base.py:199-234, docs/API.md): from
        then on reading the attribute yields `W + dW * multiplier` (with weight_decompose: the decomposed weight), differentiable
        with respect to the adapter's parameters.  This is a weight-space use by definition -- the merged weight is materialised at
        every access -- so it runs as tensor math on whatever device the weight lives on, not through the activation kernels.

        The proxy layer is built with the weight's own (out, in) orientation; the reference hands `shape[0]` to the constructor's
        `in_features` / `in_channels` slot (base.py:209-229), which is the same thing only for square weights (reference defect:
        a non-square target fails there with a shape error in `make_weight`)."""
        import torch.nn.utils.parametrize as P
        target = getattr(org_module, attr)
        kwargs["bypass_mode"] = False
        if target.dim() == 2:
            proxy = nn.Linear(target.shape[1], target.shape[0], bias=False)
        elif 3 <= target.dim() <= 5:
            conv = {3: nn.Conv1d, 4: nn.Conv2d, 5: nn.Conv3d}[target.dim()]
            proxy = conv(target.shape[1], target.shape[0], tuple(target.shape[2:]), bias=False)
        else:
            raise _unsupported(f"parametrize on a {target.dim()}-D tensor")
        proxy.weight = target
        mod = cls("", proxy, *args, **kwargs)
        object.__setattr__(mod, "_force_tensor_math", True)  # weight-space functions stay autograd-visible (no in-place / no_grad kernels)
        mod.forward = mod.parametrize_forward
        mod.to(target)
        P.register_parametrization(org_module, attr, mod)
        return mod

    def parametrize_forward(self, weight, *args, **kwargs):
        """the parametrization: original weight -> adapted weight (reference base.py:392-395 via get_merged_weight)"""
        shape = tuple(weight.shape)
        if self._conv1d is not None:  # built on the Conv2d twin: factors are [.., 1, k]
            shape = (shape[0], shape[1], 1, shape[2])
        dw = self.get_diff_weight(1.0, shape)[0]
        w = weight.reshape(shape).to(torch.promote_types(weight.dtype, dw.dtype))
        if getattr(self, "wd", False):
            merged = self._dora_merge_host(w + dw, self.multiplier)
        elif self.name == "ia3":
            merged = self.get_merged_weight(self.multiplier, shape)[0]
        else:
            merged = w + dw * self.multiplier
        return merged.reshape(weight.shape).to(weight.dtype)

    # ---- small accessors ----------------------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.dtype_tensor.dtype

    @property
    def device(self):
        return self.dtype_tensor.device

    @property
    def org_weight(self):
        return self.org_module[0].weight

    @org_weight.setter
    def org_weight(self, value):
        self.org_module[0].weight.data.copy_(value)

    def _current_weight(self):
        return self.org_module[0].weight.detach()

    def _current_bias(self):
        b = self.org_module[0].bias
        return None if b is None else b.detach()

    # ---- state dict -----------------------------------------------------------------------------------------------
    def custom_state_dict(self):
        return None

    # ---- nn.Conv1d (round 3) ------------------------------------------------------------------------------------------
    def _attach_conv1d(self, real: nn.Conv1d):
        """called by the metaclass after construction on the twin: checkpoints keep the reference's Conv1d shapes ([.., .., k]; the
        native parameters are [.., .., 1, k]), the real layer is the one whose forward gets patched, and weights handed OUT
        (get_diff_weight / get_merged_weight without an explicit shape) have the Conv1d shape."""
        object.__setattr__(self, "_conv1d", real)  # NOT a submodule: the frozen layer stays out of the adapter's parameters
        self.module = type(real)
        self._register_load_state_dict_pre_hook(self._lift1d_state_dict)

    def __init_subclass__(cls, **kwargs):
        """get_diff_weight / get_merged_weight of every algorithm hand a Conv1d-shaped weight OUT when the module adapts an nn.Conv1d
        and the caller gave no explicit shape (class-level wrappers: nothing per instance, so copies / pickles stay self-contained)"""
        super().__init_subclass__(**kwargs)
        for name in ("get_diff_weight", "get_merged_weight"):
            fn = cls.__dict__.get(name)
            if fn is None or getattr(fn, "_conv1d_aware", False):
                continue

            def wrapped(self, *a, _fn=fn, **k):
                w, b = _fn(self, *a, **k)
                if self._conv1d is None:
                    return w, b
                explicit = k.get("shape") is not None or len(a) >= 2 and a[1] is not None
                return (w if explicit else _drop1d(w)), b

            wrapped._conv1d_aware = True
            wrapped.__name__, wrapped.__doc__ = name, fn.__doc__
            setattr(cls, name, wrapped)

    def _lift1d_state_dict(self, state_dict, prefix, *_):
        for k in [k for k in state_dict if k.startswith(prefix)]:
            state_dict[k] = _lift1d(state_dict[k])

    def _forward_1d(self, x, *args, **kwargs):
        return self.forward(x.unsqueeze(2), *args, **kwargs).squeeze(2)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if self._conv1d is not None:
            pre = args[1] if len(args) > 1 and prefix == "" else prefix
            real = self._conv1d
            object.__setattr__(self, "_conv1d", None)  # plain call below, then the Conv1d shapes
            try:
                out = self.state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
            finally:
                object.__setattr__(self, "_conv1d", real)
            for k in [k for k in out if k.startswith(pre)]:
                out[k] = _drop1d(out[k])
            return out
        custom = self.custom_state_dict()
        if custom is None:
            return super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if args:  # legacy positional form (destination, prefix, keep_vars)
            destination = args[0] if destination is None else destination
            if len(args) > 1 and prefix == "":
                prefix = args[1]
        if destination is None:
            destination = OrderedDict()
            destination._metadata = OrderedDict()
        if hasattr(destination, "_metadata"):
            destination._metadata[prefix[:-1]] = dict(version=self._version)
        for key, value in custom.items():
            destination[prefix + key] = value
        return destination

    def _reset_scalar_after_load(self, module, incompatible_keys):
        # checkpoints store `first factor * scalar`; the gate itself restarts at 1 (locon.py:184-196)
        incompatible_keys.missing_keys[:] = [k for k in incompatible_keys.missing_keys if "scalar" not in k]
        scalar = getattr(self, "scalar", None)
        if scalar is not None:
            with torch.no_grad():
                scalar.fill_(1.0)
            self._scalar_scaled = False

    # ---- attach / detach -----------------------------------------------------------------------------------------
    def apply_to(self, **kwargs):
        if self.not_supported:
            return
        layer = self._patched_layer()
        if not hasattr(layer, _ORIG):
            setattr(layer, _ORIG, layer.forward)
        stack = [w for w in getattr(layer, _STACK, []) if w is not self]
        self._below = layer.forward  # whatever is currently on top (the bare layer or another adapter)
        self.org_forward = self._lifted(self._below)
        stack.append(self)
        setattr(layer, _STACK, stack)
        layer.forward = self.forward if self._conv1d is None else self._forward_1d

    def _patched_layer(self):
        return self.org_module[0] if self._conv1d is None else self._conv1d

    def _lifted(self, fwd):
        """the forward below this adapter as the module sees it: itself, or (Conv1d) [B, C, 1, L] -> [B, C', 1, L']"""
        if self._conv1d is None:
            return fwd
        return lambda x, *a, **k: fwd(x.squeeze(2), *a, **k).unsqueeze(2)

    def restore(self):
        if self.not_supported:
            return
        layer = self._patched_layer()
        stack = list(getattr(layer, _STACK, []))
        below = getattr(self, "_below", None)
        original = getattr(layer, _ORIG, below if below is not None else self.org_forward)
        if self in stack:
            pos = stack.index(self)
            stack.pop(pos)
            if pos < len(stack):  # the adapter that sat on top of us now calls what we used to call
                if below is not None and hasattr(stack[pos], "_lifted"):
                    stack[pos]._below = below
                    stack[pos].org_forward = stack[pos]._lifted(below)
                else:
                    stack[pos].org_forward = self.org_forward
        if stack:
            setattr(layer, _STACK, stack)
            top = stack[-1]
            layer.forward = top._forward_1d if getattr(top, "_conv1d", None) is not None else top.forward
        else:
            layer.forward = original
            layer.__dict__.pop(_STACK, None)
            layer.__dict__.pop(_ORIG, None)

    # ---- weight space: merge / export / max-norm / DoRA -----------------------------------------------------------
    # On the HIP device these run the tile-rebuild kernels of csrc/wspace.h (dW is never written to HBM unless it IS
    # the requested result); CPU tensors (offline tools) use plain tensor math.
    _ws_algo = None  # "locon" | "loha" | "lokr": set by the algorithms that have weight-space kernels

    def _ws_factors(self, gated=True):
        raise NotImplementedError

    def _native_ws(self):
        return (self._ws_algo is not None and not self._aten_only and not getattr(self, "_force_tensor_math", False)
                and self.org_weight.is_cuda and next(self.parameters()).is_cuda)

    def get_diff_weight(self, multiplier=1.0, shape=None, device=None):
        raise NotImplementedError

    def get_merged_weight(self, multiplier=1.0, shape=None, device=None):
        raise NotImplementedError

    def _init_dora(self, org_module, weight_decompose, wd_on_out):
        """DoRA magnitude vector, initialised to the norms of the frozen weight (locon.py:107-129)."""
        self.wd = bool(weight_decompose)
        self.wd_on_out = bool(wd_on_out)
        if not self.wd:
            return
        w = org_module.weight.detach().float().cpu()
        if self.wd_on_out:
            n = w.reshape(w.shape[0], -1).norm(dim=1).reshape(w.shape[0], *([1] * (w.dim() - 1)))
        else:
            n = w.transpose(0, 1).reshape(w.shape[1], -1).norm(dim=1).reshape(1, w.shape[1], *([1] * (w.dim() - 2)))
        self.dora_scale = nn.Parameter(n.clone())

    def _dora_s(self, W, multiplier):
        """per-channel factor of the decomposed weight: multiplier * (dora_scale / ||W + dW|| - 1) + 1
        (apply_weight_decompose, locon.py:239-260); the norms come from the fused rebuild + norm kernel."""
        from .. import ops
        mode = ops.CH_ROW if self.wd_on_out else ops.CH_COL
        if W.is_cuda:
            norm2 = ops.weight_norm2(self._ws_algo, W, self._ws_factors(), self.scale, mode)
            s = self.dora_scale.reshape(-1).float() / (norm2.sqrt() + torch.finfo(self.dora_scale.dtype).eps)
        else:  # host tensors (BASELINE configs[0]-style plumbing runs): the norms of W + dW from the rebuilt weight, autograd-visible
            dw = self.get_diff_weight(1.0, tuple(W.shape))[0]
            merged = W.to(torch.promote_types(W.dtype, dw.dtype)) + dw.reshape(W.shape)
            norm2 = merged.reshape(W.shape[0], -1).pow(2).sum(1) if self.wd_on_out else \
                merged.transpose(0, 1).reshape(W.shape[1], -1).pow(2).sum(1)
            s = self.dora_scale.reshape(-1).to(norm2.dtype) / (norm2.sqrt() + torch.finfo(self.dora_scale.dtype).eps)
        if multiplier != 1:
            s = multiplier * (s - 1) + 1
        return s

    def _forward_dora(self, x, *args, **kwargs):
        base = self.org_forward(x, *args, **kwargs)
        return base + self._dora_delta(x, base)

    def _dora_delta(self, x, base):
        """delta of the forward with weight_decompose: the reference evaluates op(x, (W + dW) s - W) with a rebuilt
        weight (locon.py:320-332).  Natively the same function in factored form:
          wd_on_out : delta = (s - 1) * (x W^T) + s * (x dW^T)             per output channel, no extra GEMM
          otherwise : delta = ((s - 1) * x) W^T + (s * x) dW^T             per input channel: one frozen-layer GEMM more"""
        from .. import ops
        layer = self.org_module[0]
        W = self._current_weight()
        s = self._dora_s(W, self.multiplier)
        chan = 1 if self.module_type.startswith("conv") else -1
        if self.wd_on_out:
            delta = self.bypass_forward_diff(x, scale=1)
            if self.org_forward == getattr(layer, _ORIG, None):  # `base` is x W^T + bias: reuse it
                plain = ops.chan_affine(base, s - 1, self._current_bias(), 0.0, 1.0, chan)  # (s - 1) * (base - bias)
            else:  # another adapter sits below: its delta must not be rescaled
                plain = ops.chan_affine(self.op(x, W, None, **self.kw_dict), s - 1, None, 0.0, 1.0, chan)
            return plain + ops.chan_affine(delta, s, None, 0.0, 1.0, chan)
        xs1 = ops.chan_affine(x, s - 1, None, 0.0, 1.0, chan)
        xs = ops.chan_affine(x, s, None, 0.0, 1.0, chan)
        return self.op(xs1, W, None, **self.kw_dict) + self.bypass_forward_diff(xs, scale=1)

    def _dora_merge_host(self, weight, multiplier):
        """apply_weight_decompose (locon.py:239-260) in plain tensor math -- the offline / CPU merge path only"""
        weight = weight.to(self.dora_scale.dtype)
        eps = torch.finfo(weight.dtype).eps
        if self.wd_on_out:
            n = weight.reshape(weight.shape[0], -1).norm(dim=1).reshape(weight.shape[0], *([1] * (weight.dim() - 1))) + eps
        else:
            n = (weight.transpose(0, 1).reshape(weight.shape[1], -1).norm(dim=1)
                 .reshape(1, weight.shape[1], *([1] * (weight.dim() - 2)))) + eps
        s = self.dora_scale.to(weight.device) / n
        if multiplier != 1:
            s = multiplier * (s - 1) + 1
        return weight * s

    @torch.no_grad()
    def _merged_weight_native(self, multiplier):
        """(W + dW * mult) or, with DoRA, (W + dW) * s -- one pass, written once (get_merged_weight, locon.py:229-237)"""
        from .. import ops
        W = self.org_weight
        if getattr(self, "wd", False):
            s = self._dora_s(W, multiplier)
            return ops.diff_weight(self._ws_algo, self._ws_factors(), W.shape, self.scale, W.dtype, W=W, coef=s,
                                   chan_mode=ops.CH_ROW if self.wd_on_out else ops.CH_COL)
        return ops.diff_weight(self._ws_algo, self._ws_factors(), W.shape, self.scale * multiplier, W.dtype, W=W)

    @torch.no_grad()
    def _max_norm_native(self, max_norm, gated=True):
        """(scaled?, ratio, orig_norm) with ||dW||_F from the factors (no dW tensor); the reference's clamp logic."""
        from .. import ops
        orig_norm = ops.sq_norm(self._ws_algo, self._ws_factors(gated), self.shape, self.scale).sqrt()
        norm = torch.clamp(orig_norm, max_norm / 2)
        desired = torch.clamp(norm, max=max_norm)
        ratio = desired.cpu() / norm.cpu()
        return bool(norm != desired), ratio, orig_norm

    def _own_device_dtype(self):
        p = next(self.parameters())
        return p.device, p.dtype

    def merge_to(self, multiplier=1.0):
        if self.not_supported:
            return
        if self._native_ws() and not getattr(self, "wd", False) and self.org_weight.is_contiguous():
            from .. import ops
            with torch.no_grad():  # W += dW * mult in place, dW tiles rebuilt on chip (no [O, I] temporary)
                ops.merge_into(self._ws_algo, self._ws_factors(), self.org_module[0].weight.data, self.scale * multiplier)
            return
        dev, dt = self._own_device_dtype()
        self.to(self.org_weight)
        weight, bias = self.get_merged_weight(multiplier, self.org_weight.shape, self.org_weight.device)
        self.org_weight = weight.to(self.org_weight)
        if bias is not None:
            layer = self.org_module[0]
            bias = bias.to(self.org_weight)
            if layer.bias is not None:
                layer.bias.data.copy_(bias)
            else:
                layer.bias = nn.Parameter(bias)
        self.to(dev, dt)

    def onfly_merge(self, multiplier=1.0):
        if self.not_supported:
            return
        dev, dt = self._own_device_dtype()
        self.to(self.org_weight)
        layer = self.org_module[0]
        self.cached_org_weight = self.org_weight.data.cpu().clone()  # clone: .cpu() aliases a CPU weight
        self.cached_org_bias = None if layer.bias is None else layer.bias.data.cpu().clone()
        weight, bias = self.get_merged_weight(multiplier, self.org_weight.shape, self.org_weight.device)
        self.org_weight = weight
        if bias is not None:
            bias = bias.to(self.org_weight)
            if layer.bias is not None:
                layer.bias.data.copy_(bias)
            else:
                layer.bias = nn.Parameter(bias)
        self.to(dev, dt)

    def onfly_restore(self):
        if self.not_supported:
            return
        self.org_weight = self.cached_org_weight.to(self.org_weight)
        if self.cached_org_bias is not None:
            self.org_module[0].bias.data.copy_(self.cached_org_bias.to(self.org_weight))
        del self.cached_org_weight
        del self.cached_org_bias

    @torch.no_grad()
    def apply_max_norm(self, max_norm, device=None):
        return None, None

    # ---- hot path (implemented by the algorithms) -------------------------------------------------------------
    def bypass_forward_diff(self, x, scale=1):
        raise NotImplementedError

    def bypass_forward(self, x, scale=1):
        return self.org_forward(x) + self.bypass_forward_diff(x, scale=scale)

    def _rank_dropout_mask(self, delta):
        """The reference's rebuild path drops ROWS of dW (`torch.rand(weight.size(0)) > rank_dropout`, optionally
        rescaled by the keep rate): a per-output-channel factor on the delta, so nothing has to be rebuilt."""
        drop = (torch.rand(self.dim, device=delta.device) > self.rank_dropout).to(torch.float32)
        if self.rank_dropout_scale:
            drop = drop / drop.mean()
        return drop

    def forward(self, x, *args, **kwargs):
        """base + delta.  The rebuild path and the bypass path of the reference are the same mathematical function
        (rebuild semantics are canonical); natively both are the factored evaluation on x.

        Dropout variants (training mode only): `module_dropout` skips the adapter for the whole call; `rank_dropout`
        scales the delta per output channel (rebuild-path semantics, see _rank_dropout_mask) with the native
        per-channel kernel; plain `dropout` acts on the delta in bypass mode, as upstream's LoCon does (upstream
        ignores it on the rebuild path and for LoHa / LoKr)."""
        if self.module_dropout and self.training and float(torch.rand(1)) < self.module_dropout:
            return self.org_forward(x, *args, **kwargs)
        if self._aten_only:
            return self._forward_aten(x, *args, **kwargs)
        # upstream consults bypass_mode first (modules/lokr.py:548-549, locon.py / loha.py alike): in bypass mode -- forced for
        # quantised base layers -- weight_decompose is ignored and the base weight is never read
        if getattr(self, "wd", False) and not self.bypass_mode:
            return self._forward_dora(x, *args, **kwargs)
        plain = not (self.training and (self.rank_dropout or (self.bypass_mode and self.dropout)))
        # (is_compiling() first: under torch.compile the branch folds away and dynamo never traces the ownership test)
        if not torch.compiler.is_compiling() and plain and not args and not kwargs:
            owned = self._forward_owned(x)  # frozen layer + adapter as ONE autograd node, where the algorithm offers that (round 6)
            if owned is not None:
                return owned
        base = self.org_forward(x, *args, **kwargs)
        if plain:
            fused = self._forward_fused(x, base)  # `base + delta` formed in the adapter kernel's epilogue, where it can be
            if fused is not None:
                return fused
        delta = self.bypass_forward_diff(x, scale=self.multiplier)
        if self.rank_dropout and self.training:
            from .. import ops
            chan_dim = 1 if self.module_type.startswith("conv") else -1
            delta = ops.chan_affine(delta.contiguous(), self._rank_dropout_mask(delta), None, 0.0, 1.0, chan_dim)
        if self.bypass_mode and self.training and self.name in ("locon", "lora"):
            delta = self.drop(delta)
        return base + delta

    # ---- nn.Conv3d: the reference's rebuild form in ATen ops (any device) -------------------------------------------------------
    def _delta_aten(self, x, scale=1.0, dw=None):
        """op(x, dW * scale) in the promoted dtype, rounded once to x's (locon.py:321-331 casts dW to the base weight's dtype first;
        composite.py's rule -- one rounding -- is the tighter of the two)"""
        if dw is None:
            dw = self.get_diff_weight(scale, tuple(self.shape))[0]
        ct = torch.promote_types(x.dtype, dw.dtype)
        return self.op(x.to(ct), dw.to(ct), None, **self.kw_dict).to(x.dtype)

    def _forward_aten(self, x, *args, **kwargs):
        from .. import composite
        base = self.org_forward(x, *args, **kwargs)
        if getattr(self, "wd", False) and not self.bypass_mode:  # op(x, decompose(W + dW) - W)   (locon.py:318-332, apply_weight_decompose)
            W = self._current_weight()
            dw = self.get_diff_weight(1.0, tuple(W.shape))[0]
            merged = self._dora_merge_host(W.to(torch.promote_types(W.dtype, dw.dtype)) + dw, self.multiplier)
            return base + self._delta_aten(x, dw=merged - W)
        delta = self.bypass_forward_diff(x, scale=self.multiplier)
        if self.rank_dropout and self.training:
            delta = composite.chan_affine(delta, self._rank_dropout_mask(delta), None, 0.0, 1.0, 1)
        if self.bypass_mode and self.training and self.name in ("locon", "lora"):
            delta = self.drop(delta)
        return base + delta

    def _frozen_linear(self):
        """(weight, bias) of the bare nn.Linear right below this adapter when its forward is the stock `F.linear(x, W, b)` and it is frozen
        -- what an op that owns the layer's forward and input gradient may replace -- else None (another adapter below, a quantised or
        otherwise subclassed layer, a trainable base weight, a re-parametrised weight)"""
        if self.module_type != "linear" or self._conv1d is not None:
            return None
        layer = self.org_module[0]
        if type(layer) is not nn.Linear or getattr(type(layer), "forward", None) is not nn.Linear.forward:
            return None
        if "forward" in vars(layer) and getattr(layer, _ORIG, None) is None:  # an instance-level forward that is not ours
            return None
        if self.org_forward != getattr(layer, _ORIG, None) and self.org_forward != layer.forward:
            return None
        if getattr(self.org_forward, "__func__", None) is not nn.Linear.forward or getattr(self.org_forward, "__self__", None) is not layer:
            return None
        if "weight" not in layer._parameters or getattr(layer, "parametrizations", None):
            return None
        W, b = layer.weight, layer.bias
        if W is None or W.requires_grad or (b is not None and b.requires_grad):
            return None
        return W, b

    def _forward_owned(self, x):
        """`base + delta` with the frozen layer's forward and input gradient inside the adapter's autograd node; None = not available"""
        return None

    def _forward_fused(self, x, base):
        """Algorithms whose kernels can add the frozen layer's output in their epilogue return `base + delta` here
        (modules/lokr.py:566 `return base + delta` without the separate elementwise pass); None = not available."""
        return None

    # ---- helpers for subclasses --------------------------------------------------------------------------------
    def _conv_geometry(self):
        if self.module_type != "conv2d":
            raise _unsupported(f"{self.module_type} layers")
        return self.kw_dict

    def _gate(self, first_factor: torch.Tensor) -> torch.Tensor:
        """Fold the learnable `scalar` gate into the first factor (its gradient then flows through autograd)."""
        if isinstance(self.scalar, nn.Parameter) or getattr(self, "_scalar_scaled", False):
            return first_factor * self.scalar  # learnable gate, or a fixed gate that apply_max_norm has moved off 1
        return first_factor

    def _init_scale(self, lora_dim, alpha, rs_lora, use_scalar, force_unit_scale=False):
        """alpha / rank bookkeeping shared by LoCon / LoHa / LoKr (e.g. locon.py:131-152)."""
        if isinstance(alpha, torch.Tensor):
            alpha = float(alpha.detach().float())
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        if force_unit_scale:
            alpha = lora_dim
        r_factor = lora_dim ** 0.5 if rs_lora else lora_dim
        self.scale = alpha / r_factor
        self.register_buffer("alpha", torch.tensor(alpha * (lora_dim / r_factor)))
        if use_scalar:
            self.scalar = nn.Parameter(torch.tensor(0.0))
        else:
            self.register_buffer("scalar", torch.tensor(1.0), persistent=False)
        self.register_load_state_dict_post_hook(self._reset_scalar_after_load)
