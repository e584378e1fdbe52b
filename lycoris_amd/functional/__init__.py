"""Functional mirror of ``lycoris.functional`` for the native hot path (same module / function names)."""
from . import general, locon, loha, lokr  # noqa: F401
from .general import (FUNC_LIST, apply_dora_scale, factorization, power2factorization, rebuild_tucker, tucker_weight,  # noqa: F401
                      tucker_weight_from_conv)
