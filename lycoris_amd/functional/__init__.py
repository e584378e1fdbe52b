"""Functional mirror of ``lycoris.functional`` for the native hot path (same module / function names)."""
from . import general, locon, loha, lokr  # noqa: F401
from .general import factorization, rebuild_tucker  # noqa: F401
