"""Repeat-determinism of the three networks (scripts/soak_determinism.py): the same inputs, 25 / 6 / 10 times, bit-identical - a
race in an LDS pipeline or a missing barrier flips bits long before it changes a picture."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_repeated_runs_are_bit_identical():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_determinism
    soak_determinism.main()
