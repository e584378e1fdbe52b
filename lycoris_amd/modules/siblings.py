"""Sibling projections behind the module API (round 5; VERDICT r4 missing #2 / next #4a).

The reference calls one ``LokrModule.forward`` per projection (lycoris/modules/lokr.py:543-566), so the to_q / to_k / to_v layers of a
self-attention block -- three adapters with equal shapes reading the SAME tensor -- cost three ~5 us launches that each fill a third
of the MI355X, and three more in backward.  ``lyc_lokr_linear_fwd_group`` / ``_bwd_group`` run such a set as one launch
(bit-identical; 16.0 -> 9.8 us forward, 18.8 -> 12.4 us dx per q / k / v set), but a drop-in user never calls them: the host model
calls ``layer(x)`` three times.  This file makes the modules find their siblings themselves:

* learning (first forward pass): every eligible module remembers the tensor OBJECT it was called with; a module called with the very
  tensor the previous eligible module saw (same object, same version counter) and with equal factor shapes joins that module's set.
  Nothing is assumed about the host model -- diffusers' ``Attention`` (``to_q(h)``, ``to_k(ctx)``, ``to_v(ctx)``), sd-scripts' own
  attention classes and text encoders all end up with the sets their call pattern implies (q / k / v, or k / v for cross-attention).
* steady state: the set's first member (the leader) runs the frozen forwards of ALL members on its input, hands the lot to
  ``ops.lokr_linear_group`` (one forward launch with the fused ``base + delta`` epilogues, one autograd node, one backward dx launch)
  and parks the siblings' results; a sibling called with the same tensor object returns its parked result.
* any surprise dissolves the set: a sibling called with a different tensor (or the same one modified in place), a member that has
  left the plain path (dropout variants, DoRA, restore()), a member that died.  The per-layer path then runs as if nothing happened,
  so the numbers can never depend on the grouping -- only the launch count does.

Scope: LoKr on nn.Linear with a full `lokr_w1` and either a full-matrix `lokr_w2` (the headline configuration) or the low-rank pair
`lokr_w2_a @ lokr_w2_b` (BASELINE configs[3]), and LoCon on nn.Linear (`ops.locon_linear_group`); a set holds one kind only.  A module
class takes part by deriving from ``SiblingMixin`` and supplying `_sibling_eligible(x)`, `_sibling_key()` and `_sibling_launch(members,
x, bases)`.  ``enable(False)`` switches the mechanism off.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from .. import ops

_STATE = {"enabled": True, "autocast": True, "prev": None, "sets": 0, "launches": 0, "hits": 0, "dissolved": 0}
MAX_SET = 4  # K4_GROUP_MAX of csrc/kron4.h


def enable(on: bool = True, autocast=None):
    """switch sibling grouping on / off (off: existing sets stay but are not used or extended).
    `autocast` (default ON since round 6): sets also form on fp32 activations under a 16-bit torch.autocast (see activation_ok) -- the
    configuration sd-scripts' mixed_precision=bf16 runs; `autocast=False` restricts the sets to 16-bit activations."""
    _STATE["enabled"] = bool(on)
    if autocast is not None:
        _STATE["autocast"] = bool(autocast)
    _STATE["prev"] = None


def stats():
    return {k: _STATE[k] for k in ("sets", "launches", "hits", "dissolved")}


class SiblingSet:
    __slots__ = ("members", "pending", "key", "__weakref__")

    def __init__(self, key):
        self.members = []   # weakrefs, in call order; [0] is the leader
        self.pending = {}   # id(module) -> (x, x._version, y): results the leader computed for the siblings
        self.key = key

    def alive(self):
        out = []
        for r in self.members:
            m = r()
            if m is None or getattr(m, "_sib", None) is not self:
                return None
            out.append(m)
        return out or None  # (an emptied set -- dissolved through another reference -- is dead too)

    def dissolve(self):
        for r in self.members:
            m = r()
            if m is not None and getattr(m, "_sib", None) is self:
                object.__setattr__(m, "_sib", None)
        self.members, self.pending = [], {}
        _STATE["dissolved"] += 1


_SIXTEEN = (torch.bfloat16, torch.float16)


def activation_ok(x) -> bool:
    """a 16-bit HIP activation -- or, with enable(autocast=True), an fp32 one under a 16-bit torch.autocast: sd-scripts' mixed-precision
    training hands the fp32 output of a LayerNorm to to_q / to_k / to_v, and the reference's F.linear runs in the autocast dtype
    (modules/lokr.py:543-566 under autocast).  The grouped ops cast it ONCE for the whole set (csrc/torch_ops.cpp amp(x)), the
    per-layer ops once per projection."""
    if not x.is_cuda or x.is_inference():
        return False
    if x.dtype in _SIXTEEN:
        return True
    return (_STATE["autocast"] and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") in _SIXTEEN)


def _key(mod, x):
    k = mod.__dict__.get("_sib_key_cached")
    if k is None:  # (factor shapes never change after construction: computed once per module)
        k = mod._sibling_key()
        object.__setattr__(mod, "_sib_key_cached", k)
    # (no x.shape: the kernels take any row count, and aspect-ratio buckets / a last partial batch must not dissolve the sets)
    return (k, x.dtype, x.device, torch.get_autocast_dtype("cuda") if (x.dtype == torch.float32 and x.is_cuda) else None)


def forget(mod):
    """apply_to() / restore(): the module's place in the host model changed"""
    st = getattr(mod, "_sib", None)
    if st is not None:
        st.dissolve()
    prev = _STATE["prev"]
    if prev is not None and prev[0]() is mod:
        _STATE["prev"] = None


def forward(mod, x) -> Optional[torch.Tensor]:
    """`base + delta` of `mod` on `x` through its sibling set, or None: run the per-layer path"""
    if not _STATE["enabled"]:
        return None
    st = mod._sib
    if st is not None:
        pend = st.pending.pop(id(mod), None)
        if pend is not None:
            px, pv, y = pend
            if px is x and pv == x._version:
                _STATE["hits"] += 1
                return y
            st.dissolve()  # the host model's call pattern is not what was learned
            st = None
        else:
            members = st.alive()
            if members is None or st.key != _key(mod, x):
                st.dissolve()
                st = None
            elif members[0] is mod and len(members) > 1:
                if all(m._sibling_eligible(x) for m in members[1:]):
                    st.pending.clear()  # (results nobody fetched: a sibling skipped its call in the last pass)
                    owned = getattr(mod, "_sibling_launch_owned", None)
                    ys = owned(members, x) if owned is not None else None  # frozen layers + adapters of the set as ONE node (round 6)
                    if ys is None:
                        bases = [m.org_forward(x) for m in members]
                        ys = mod._sibling_launch(members, x, bases)  # [base_i + delta_i]: one grouped launch of the algorithm's kernels
                    ver = x._version
                    for m, y in zip(members[1:], ys[1:]):
                        st.pending[id(m)] = (x, ver, y)
                    _STATE["launches"] += 1
                    return ys[0]
                st.dissolve()
                st = None
            else:
                return None  # a member that is called without a parked result (the leader was skipped): per-layer path, set kept
    # (after a dissolve the module learns again right away, so the whole q / k / v set re-forms in the next pass)
    # ---- learning: did the previous eligible module see this very tensor? -------------------------------------------------------
    prev, _STATE["prev"] = _STATE["prev"], (weakref.ref(mod), weakref.ref(x), x._version, _key(mod, x))
    if prev is None:
        return None
    pm, px = prev[0](), prev[1]()
    if pm is None or pm is mod or px is not x or prev[2] != x._version or prev[3] != _key(mod, x):
        return None
    pst = getattr(pm, "_sib", None)
    if pst is None:
        pst = SiblingSet(prev[3])
        pst.members.append(weakref.ref(pm))
        object.__setattr__(pm, "_sib", pst)
        _STATE["sets"] += 1
    if len(pst.members) < MAX_SET and pst.key == prev[3]:
        pst.members.append(weakref.ref(mod))
        object.__setattr__(mod, "_sib", pst)
    return None


class SiblingMixin:
    """forward / apply_to / restore of a module class whose instances can form sibling sets (put in front of LycorisBaseModule)"""
    _sib = None  # the set this module was found in; per instance once it has joined one (not state, not a submodule)

    def _sibling_eligible(self, x) -> bool:
        raise NotImplementedError

    def _sibling_key(self):
        raise NotImplementedError

    @staticmethod
    def _sibling_launch(members, x, bases):
        raise NotImplementedError

    def __getstate__(self):
        # copy.deepcopy / pickle (torch.save(module), spawn): set membership is a fact about the LIVE host model's call pattern --
        # weakrefs and parked tensors -- and is learned again by the copy on its first forward pass
        state = self.__dict__.copy()
        state.pop("_sib", None)
        state.pop("_sib_key_cached", None)
        return state

    def forward(self, x, *args, **kwargs):
        # projections called with one tensor run as ONE launch; everything else is the per-layer path
        # (is_compiling() first: under torch.compile the whole branch folds away and dynamo never sees the eligibility test)
        if not torch.compiler.is_compiling() and not args and not kwargs and isinstance(x, torch.Tensor) and self._sibling_eligible(x):
            y = forward(self, x)
            if y is not None:
                return y
        return super().forward(x, *args, **kwargs)

    def apply_to(self, **kwargs):
        forget(self)
        return super().apply_to(**kwargs)

    def restore(self):
        forget(self)
        return super().restore()
