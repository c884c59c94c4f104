"""ctypes wrapper of oracle/quantile.c (test infrastructure only).  Built by oracle.build() / __graft_entry__.build()."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
SO = HERE / "libmaua_oracle.so"
_lib = None


def build(force=False):
    src = HERE / "quantile.c"
    if force or not SO.exists() or SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(SO), str(src), "-lm"], check=True)
    return SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(SO))
        _lib.maua_oracle_quantile_mid.restype = C.c_float
        _lib.maua_oracle_quantile_mid.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.POINTER(C.c_int64),
                                                  C.POINTER(C.c_int64)]
    return _lib


def quantile_with_indices(t, q):
    a = np.ascontiguousarray(t.detach().cpu().flatten().numpy(), dtype=np.float32)
    lo, hi = C.c_int64(), C.c_int64()
    v = _load().maua_oracle_quantile_mid(a.ctypes.data, a.size, C.c_float(q), C.byref(lo), C.byref(hi))
    return float(v), lo.value, hi.value


def quantile(t, q):
    """efficient_quantile/__init__.py:6-7 -> 0-dim float32 tensor."""
    return torch.tensor(quantile_with_indices(t, q)[0], dtype=torch.float32)
