"""GPU: a short, fixed-seed run of tools/soak.py — random shapes, tilings, orders, stop rules, image side,
CG, edges-first and row blocks against the independent implementation / the oracle (see the tool)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2])
def test_short_soak(seed):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import soak
    assert soak.run(16, seed) == 0
