"""CPU oracle for the LyCORIS adapter hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy (float64) restatement of the reference's
*rebuild path* semantics for LoCon / LoHa / LoKr / (IA)^3 on nn.Linear and
nn.Conv2d.  Each function cites the reference file:line it follows.

Rules:
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here.  The product package
    ``lycoris_amd`` never imports it; there is no CPU fallback in the product.
  * Parity pin: every function here is checked against golden vectors that
    were produced by importing the real reference (``/root/reference``) with
    ``tests/golden/make_golden.py``; the vectors are committed under
    ``tests/golden/*.npz`` (``tests/test_oracle_golden.py``).
"""
from . import dora, general, ia3, locon, loha, lokr  # noqa: F401
