"""The build-owned counter RNG on the device: Philox4x32-10 streams addressed by (seed, stream, element) -
``csrc/rng.hip`` through ``maua_philox_*``.  Same numbers on every rank and device, and in the oracle twin (oracle/rng.py, pinned
to the published known-answer vectors), without anything being exchanged or uploaded.  No reference counterpart: the reference
draws its random-init weights and noise planes from torch's host generator (inference/stylegan2.py:216-227, noise.py:42-53)."""
import ctypes as C

import torch

from . import _lib as L


def philox_u32(seed, stream, n, offset=0, device=None):
    L.require_device()
    out = torch.empty((int(n),), dtype=torch.int32, device="cuda" if device is None else device)
    L.check(L.lib().maua_philox_u32(L.ctx(out.device), C.c_ulonglong(int(seed)), C.c_ulonglong(int(stream)), C.c_ulonglong(int(offset)),
                                    L.ptr(out), C.c_long(int(n))))
    return out


def philox_normal(shape, seed, stream, offset=0, mean=0.0, std=1.0, device=None):
    """float32 N(mean, std^2) tensor of ``shape`` on the device: elements offset .. of stream (seed, stream), row-major."""
    L.require_device()
    out = torch.empty(tuple(int(s) for s in shape), dtype=torch.float32, device="cuda" if device is None else device)
    L.check(L.lib().maua_philox_normal(L.ctx(out.device), C.c_ulonglong(int(seed)), C.c_ulonglong(int(stream)), C.c_ulonglong(int(offset)),
                                       L.ptr(out), C.c_long(out.numel()), C.c_float(mean), C.c_float(std)))
    return out


def clip_audio(n_samples, sr, seed=1234, device=None):
    """SURVEY 8(d)'s synthetic clip (tone + 2 Hz clicks + noise floor) drawn on the device: float32 [n_samples]."""
    L.require_device()
    out = torch.empty((int(n_samples),), dtype=torch.float32, device="cuda" if device is None else device)
    L.check(L.lib().maua_philox_clip_audio(L.ctx(out.device), C.c_ulonglong(int(seed)), C.c_long(int(n_samples)), C.c_double(float(sr)),
                                           L.ptr(out)))
    return out


class PhiloxStreams:
    """A callable for ``init_synthesis_params(generator=...)`` / the noise modules: the k-th tensor it is asked for is stream
    ``first_stream + k`` of ``seed`` (so a tensor's numbers depend on its position in the construction order only)."""

    def __init__(self, seed, first_stream=0, device=None):
        self.seed, self.next_stream, self.device = int(seed), int(first_stream), device

    def __call__(self, shape):
        t = philox_normal(shape, self.seed, self.next_stream, device=self.device)
        self.next_stream += 1
        return t
