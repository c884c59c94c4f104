import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    z = np.load(GOLDEN / f"{name}.npz")
    return {k: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(z[k]) if z[k].ndim else z[k][()]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden
