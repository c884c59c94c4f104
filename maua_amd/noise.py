"""Stateless, index-addressed noise modules (drop-in for
maua/audiovisual/audioreactive/selfsupervised/noise.py: Noise :4-8, Blend :11-24, Multiply :27-39, Loop :42-53,
Average :56-63, Modulate :66-75, ScaleBias :78-86, noise_patch :89-140).

``forward(i, b)`` returns frames i..i+b-1 as a [b, h, w] float32 tensor on the HIP device, computed by
libmaua_hip.so.  Random planes are drawn from the caller's ``torch.Generator`` exactly like the reference
(``torch.randn(..., generator=rng)``), on the generator's device, then kept resident in HBM.
"""
import ctypes as C

import torch

from . import _lib as L


def _randn(shape, rng):
    dev = rng.device if rng is not None else "cpu"
    return torch.randn(shape, generator=rng, device=dev)


class Noise(torch.nn.Module):
    def __init__(self, length, size):
        super().__init__()
        self.length = length
        self.size = tuple(int(s) for s in size)

    def _out(self, b):
        L.require_device()
        return torch.empty((b, *self.size), dtype=torch.float32, device="cuda")


class Loop(Noise):
    def __init__(self, rng, length, size, n_loops=1, sigma=5, noise=None):
        super().__init__(length, size)
        self.sigma = sigma
        planes = _randn((3, self.size[0], self.size[1]), rng) if noise is None else noise
        self.register_buffer("noise", planes.float())
        self.register_buffer("idx", torch.linspace(0, n_loops * 2 * torch.pi, length))
        self._dev = None

    def _resident(self):
        if self._dev is None:
            self._dev = (L.dev_tensor(self.noise, torch.float32), L.dev_tensor(self.idx, torch.float32))
        return self._dev

    def forward(self, i, b):
        b = max(0, min(b, self.length - i))
        planes, idx = self._resident()
        out = self._out(b)
        L.check(L.lib().maua_noise_loop(L.ctx(), L.ptr(planes), L.ptr(idx), int(i), int(b), self.size[0], self.size[1],
                                        C.c_float(float(self.sigma)), L.ptr(out)))
        return out


class RawNoise:
    """Un-normalised Loop maps of one batch (``maps``: [b, h, w] per layer) + ``scales`` [n_layers, b]: the factor 1 / (rms + eps)
    of every (layer, sample) that noise.py:52 divides by.  ``SynthesisNetwork.forward(noise=<RawNoise>)`` hands both to the
    library, whose convolution epilogues apply the factor.  Deliberately NOT a list: whatever treats it as a sequence of tensors -
    iteration, indexing, ``list(nz)``, ``[m[:, None] for m in nz]``, the wrappers' ``noise{j}=`` keywords - gets the NORMALISED
    maps (the tensors the reference's modules return, one multiply per map, computed once), so the factors cannot be dropped
    silently on the way to the network; only code that knows about them reads ``maps`` / ``scales``."""

    def __init__(self, maps):
        self.maps, self.scales, self._norm = list(maps), None, None

    def normalised(self):
        if self._norm is None:
            self._norm = [m * self.scales[l, : m.shape[0], None, None] for l, m in enumerate(self.maps)]
        return self._norm

    def __len__(self):
        return len(self.maps)

    def __iter__(self):
        return iter(self.normalised())

    def __getitem__(self, k):
        return self.normalised()[k]


def loop_batch(modules, i, b, raw=False):
    """forward(i, b) of a list of Loop modules in one C call (two launches for all layers) -> list of [b,h,w].
    ``raw=True``: the maps un-normalised, written in ONE pass, as a ``RawNoise`` carrying the per-sample factors (what the
    render loops use: the normalised form evaluates sin(cos(.)) twice per value)."""
    n = len(modules)
    if not all(isinstance(m, Loop) for m in modules):
        return [m.forward(i, b) for m in modules]
    if raw:
        res = [m._resident() for m in modules]
        outs = RawNoise(m._out(max(0, min(b, m.length - i))) for m in modules)
        nb = int(outs.maps[0].shape[0])
        outs.scales = torch.empty((n, max(nb, 1)), dtype=torch.float32, device=outs.maps[0].device)
        P = (C.c_void_p * n)(*[r[0].data_ptr() for r in res])
        I = (C.c_void_p * n)(*[r[1].data_ptr() for r in res])
        O = (C.c_void_p * n)(*[o.data_ptr() for o in outs.maps])
        H = (C.c_int * n)(*[m.size[0] for m in modules])
        W = (C.c_int * n)(*[m.size[1] for m in modules])
        S = (C.c_float * n)(*[float(m.sigma) for m in modules])
        L.check(L.lib().maua_noise_loop_batch_raw(L.ctx(), n, P, I, H, W, S, int(i), nb, O, L.ptr(outs.scales)))
        return outs
    res = [m._resident() for m in modules]
    outs = [m._out(max(0, min(b, m.length - i))) for m in modules]
    P = (C.c_void_p * n)(*[r[0].data_ptr() for r in res])
    I = (C.c_void_p * n)(*[r[1].data_ptr() for r in res])
    O = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    H = (C.c_int * n)(*[m.size[0] for m in modules])
    W = (C.c_int * n)(*[m.size[1] for m in modules])
    S = (C.c_float * n)(*[float(m.sigma) for m in modules])
    L.check(L.lib().maua_noise_loop_batch(L.ctx(), n, P, I, H, W, S, int(i), int(outs[0].shape[0]), O))
    return outs


class _Mix(Noise):
    n_banks = 1

    def __init__(self, rng, length, size, modulator, noise=None):
        super().__init__(length, size)
        m = modulator.shape[1]
        shape = (m, *self.size) if self.n_banks == 1 else (2, m, *self.size)
        self.register_buffer("noise", (_randn(shape, rng) if noise is None else noise).float())
        self.register_buffer("modulator", modulator.float())
        self._dev = None

    def forward(self, i, b):
        if self._dev is None:
            self._dev = (L.dev_tensor(self.noise, torch.float32),
                         L.dev_tensor(self.modulator.reshape(len(self.modulator), -1), torch.float32))
        noise, mod = self._dev
        b = max(0, min(b, self.length - i))
        rows = mod[i:i + b].contiguous()
        out = self._out(b)
        if self.n_banks == 2:
            n1, n2 = noise[0], noise[1]
        else:
            n1, n2 = noise, None
        L.check(L.lib().maua_noise_mix(L.ctx(), L.ptr(n1), L.ptr(n2), L.ptr(rows), rows.shape[1], int(b),
                                       self.size[0], self.size[1], L.ptr(out)))
        return out


class Blend(_Mix):
    n_banks = 2


class Multiply(_Mix):
    n_banks = 1


def _combine(x, y, mod, mode, scale=1.0, bias=0.0):
    out = torch.empty_like(x)
    b, h, w = x.shape
    L.check(L.lib().maua_noise_combine(L.ctx(), L.ptr(x), L.ptr(y), L.ptr(mod), mode, C.c_float(scale),
                                       C.c_float(bias), b, h, w, L.ptr(out)))
    return out


class Average(Noise):
    def __init__(self, left, right):
        super().__init__(left.length, left.size)
        self.left, self.right = left, right

    def forward(self, i, b):
        return _combine(self.left(i, b), self.right(i, b), None, 0)


class Modulate(Noise):
    def __init__(self, left, right, modulator):
        super().__init__(left.length, left.size)
        self.left, self.right = left, right
        self.register_buffer("modulator", modulator.float().mean(1))
        self._dev = None

    def forward(self, i, b):
        if self._dev is None:
            self._dev = L.dev_tensor(self.modulator, torch.float32)
        b = max(0, min(b, self.length - i))
        return _combine(self.left(i, b), self.right(i, b), self._dev[i:i + b].contiguous(), 1)


class ScaleBias(Noise):
    def __init__(self, base, scale, bias):
        super().__init__(base.length, base.size)
        self.base, self.scale, self.bias = base, scale, bias

    def forward(self, i, b):
        return _combine(self.base(i, b), None, None, 2, float(self.scale), float(self.bias))


_DEPTHS = {"low": range(0, 6), "mid": range(6, 12), "high": range(12, 17), "lowmid": range(0, 12),
           "midhigh": range(6, 17), "all": range(0, 17)}


def noise_patch(rng, noise, features, tempo, fps, patch_type, loop_bars, seq_feat, seq_feat_weight, mod_feat,
                mod_feat_weight, merge_type, merge_depth, noise_mean, noise_std):
    """noise.py:89-140"""
    feature = seq_feat_weight * features[seq_feat]
    for n in _DEPTHS[merge_depth]:
        if patch_type == "blend":
            new = Blend(rng=rng, length=len(feature), size=noise[n].size, modulator=feature)
        elif patch_type == "multiply":
            new = Multiply(rng=rng, length=len(feature), size=noise[n].size, modulator=feature)
        elif patch_type == "loop":
            n_loops = len(feature) / fps / 60 / tempo / 4 / loop_bars
            new = Loop(rng=rng, length=len(feature), size=noise[n].size, n_loops=n_loops)
        else:
            raise ValueError(patch_type)
        if merge_type == "average":
            noise[n] = Average(left=noise[n], right=new)
        elif merge_type == "modulate":
            noise[n] = Modulate(left=noise[n], right=new, modulator=mod_feat_weight * features[mod_feat])
        else:
            noise[n] = new
        noise[n] = ScaleBias(noise[n], scale=noise_std, bias=noise_mean)
    return noise
