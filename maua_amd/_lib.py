"""ctypes binding of libmaua_hip.so — the only thing the Python host code talks to.

There is NO CPU fallback: if the shared library is missing or no HIP device is visible every
operator raises.  (The CPU oracle lives under /oracle and is test infrastructure only.)
"""
import ctypes as C
import os
import re
import threading
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MAUA_HIP_LIB", _HERE / "csrc" / "libmaua_hip.so"))
HEADER = _HERE.parent / "include" / "maua_hip.h"

F32, BF16, F16 = 0, 1, 2
F32_SPLIT = 3   # float32 tensors, products as bf16 split products (maua_secondary_create only)
ACTS = {"linear": 0, "relu": 1, "lrelu": 2, "tanh": 3, "sigmoid": 4, "elu": 5, "selu": 6, "softplus": 7, "swish": 8}
PAD_MODES = {"circular": 0, "reflect": 1, "replicate": 2, "constant": 3}


class MauaHipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()
_ctxs = {}


def declared_symbols():
    """Every function the C header declares (used by the export test)."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(maua_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not LIB_PATH.exists():
                    raise MauaHipError(
                        f"{LIB_PATH} not found: build it with `python -m maua_amd.build` (hipcc, gfx950). "
                        "maua_amd has no CPU fallback.")
                l = C.CDLL(str(LIB_PATH))
                l.maua_last_error.restype = C.c_char_p
                l.maua_version.restype = C.c_char_p
                _lib = l
    return _lib


class host_threads:
    """Context manager / decorator: run host-side tensor prep on at most ``n`` intra-op threads.  The host work of this
    package is small tensors (512 x 512 matrices, filter banks, seeds); on a 128-thread box torch's pool costs ~100 ms to
    spin up and milliseconds of fork/join per tiny operator (measured: MappingNetwork init 100 -> 7 ms, its forward 13 .. 71
    -> 3 ms, get_z_latents 95 -> 6 ms, a CQT's filter banks 290 -> 40 ms); results are identical."""

    def __init__(self, n=1):
        self.n = n

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)

    def __call__(self, fn):
        import functools

        @functools.wraps(fn)
        def wrapped(*a, **k):
            with host_threads(self.n):
                return fn(*a, **k)
        return wrapped


def check(rc):
    if rc != 0:
        raise MauaHipError(lib().maua_last_error().decode())


def require_device():
    if not torch.cuda.is_available():
        raise MauaHipError("no HIP device visible (torch.cuda.is_available() is False); maua_amd has no CPU fallback")


def ctx(device=None):
    """Per-device context bound to torch's current HIP stream."""
    require_device()
    if device is None:
        dev = torch.cuda.current_device()
    elif isinstance(device, int):
        dev = device
    else:
        dev = torch.device(device).index
        if dev is None:
            dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream(dev).cuda_stream
    c = _ctxs.get(dev)
    if c is None:
        p = C.c_void_p()
        check(lib().maua_ctx_create(C.c_int(dev), C.c_void_p(stream), C.byref(p)))
        c = _ctxs[dev] = p
    else:
        check(lib().maua_ctx_set_stream(c, C.c_void_p(stream)))
    return c


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def dtype_id(t):
    dt = t if isinstance(t, torch.dtype) else t.dtype
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:   # the reference's own render dtype (render/ffmpeg.py:45): operator layer + synthesis network
        return F16
    raise MauaHipError(f"unsupported dtype {dt}: use float32, bfloat16 or float16")


def dev_tensor(t, dtype=None):
    """contiguous tensor on the current HIP device (uploading if the caller handed a CPU tensor)."""
    require_device()
    if dtype is None:
        dtype = t.dtype
    if t.device.type == "cuda":
        return t.to(dtype=dtype).contiguous()
    return t.to(device="cuda", dtype=dtype).contiguous()
