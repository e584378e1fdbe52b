"""Pin the numpy oracle against vectors produced by the real reference (CPU, no GPU needed)."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, golden_case_names
from golden_util import oracle_eval


@pytest.mark.parametrize("name", golden_case_names())
def test_oracle_matches_reference_module(name, golden_cases):
    meta, a = golden_cases[name]
    delta, grads = oracle_eval(meta, a)
    assert delta.shape == a["delta"].shape
    assert oracle.general.rel_err(delta, a["delta"]) < 1e-12, name
    assert oracle.general.rel_err(grads["dx"], a["dx"]) < 1e-12, name
    checked = 0
    for k, v in a.items():
        if k.startswith("g.") and k != "g":
            assert k in grads, (name, k)
            assert oracle.general.rel_err(np.asarray(grads[k]).reshape(v.shape), v) < 1e-11, (name, k)
            checked += 1
    assert checked >= 1


def test_factorization_known_answers():
    with open(os.path.join(GOLDEN, "factorization.json")) as f:
        vec = json.load(f)
    for dim, factor, want in vec:
        assert list(oracle.general.factorization(dim, factor)) == want, (dim, factor)


def test_factorization_docstring_table():
    # functional/general.py:24-33 docstring rows.  Two entries of that table are stale w.r.t. the reference
    # *code* (360/-1: doc (8, 45), code (18, 20); 360/16: doc (12, 30), code (15, 24)); the code is the contract
    # and is what tests/golden/factorization.json holds, so those two rows are asserted against the code.
    f = oracle.general.factorization
    assert f(127, -1) == (1, 127) and f(127, 8) == (1, 127)
    assert f(128, -1) == (8, 16) and f(128, 2) == (2, 64) and f(128, 4) == (4, 32)
    assert f(250, -1) == (10, 25) and f(250, 2) == (2, 125) and f(250, 8) == (5, 50)
    assert f(360, -1) == (18, 20) and f(360, 4) == (4, 90) and f(360, 8) == (8, 45) and f(360, 16) == (15, 24)
    assert f(512, -1) == (16, 32) and f(512, 8) == (8, 64)
    assert f(1024, -1) == (32, 32) and f(1024, 16) == (16, 64)
