"""GPU: the forward path has no atomics, so every op must return the same bits call after call.  (A parity test sees an
intermittent hazard / race once in a few runs; this one repeats each op 120 times — tools/determinism_check.py.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_forward_ops_are_bit_reproducible():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import determinism_check
    res = determinism_check.run(reps=120, full=False, verbose=False)
    bad = {k: v for k, v in res.items() if v[0]}
    assert not bad, bad
