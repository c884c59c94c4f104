"""Latent schedule builders on the HIP device (drop-in for maua/audiovisual/audioreactive/latent.py:
single_weighted :12-18, multi_weighted :21-31, select_modulo :34-43, slerp :54-65, slerp_loops :68-80,
spline_loops :83-92, tempo_loops :95-102; and .../selfsupervised/latent.py: spline_loop_latents :7-13,
latent_patch :16-80)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import audio as A
from . import signal as S


def _rows(t):
    t = A._f32(t)
    return t, t.shape[0], t.numel() // max(t.shape[0], 1)


def _spline(knots, t_in, t_out):
    y, n, Cn = _rows(knots)
    ti = np.ascontiguousarray(t_in.double().numpy())
    to = np.ascontiguousarray(t_out.double().numpy())
    out = torch.empty((len(to), *y.shape[1:]), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_spline_natural(L.ctx(y.device), ti.ctypes.data_as(C.c_void_p), n, L.ptr(y), C.c_long(Cn),
                                        to.ctypes.data_as(C.c_void_p), len(to), L.ptr(out)))
    return out


def spline_loops(y, size, n_loops):
    y = A._f32(y)
    Y = torch.cat([y] * n_loops + [y[[0]]])
    return _spline(Y, torch.linspace(0, 1, len(Y)), torch.linspace(0, 1, size))


def spline_loop_latents(y, size, n_loops=1):
    y = A._f32(y)
    Y = torch.cat((y, y[[0]]))
    return _spline(Y, torch.linspace(0, 1, len(Y)), torch.linspace(0, n_loops, size) % 1)


def _blend(a, a_ts, b, b_ts, env, T, Cn, shape):
    out = torch.empty((T, *shape), dtype=torch.float32, device=env.device)
    L.check(L.lib().maua_latent_blend(L.ctx(env.device), L.ptr(a), C.c_long(a_ts), L.ptr(b), C.c_long(b_ts), L.ptr(env),
                                      T, C.c_long(Cn), L.ptr(out)))
    return out


def single_weighted(low_latent, high_latent, envelope):
    low, high, env = A._f32(low_latent), A._f32(high_latent), A._f32(envelope).reshape(-1)
    return _blend(low, 0, high, 0, env, env.numel(), low.numel(), low.shape)


def sequence_weighted(low_seq, high_seq, envelope):
    """single_weighted over two [T, ...] sequences (what the benchmark schedule uses)."""
    low, high, env = A._f32(low_seq), A._f32(high_seq), A._f32(envelope).reshape(-1)
    Cn = low.numel() // low.shape[0]
    return _blend(low, Cn, high, Cn, env, env.numel(), Cn, low.shape[1:])


def multi_weighted(latents, envelopes):
    lat, env = A._f32(latents), A._f32(envelopes)
    T, An = env.shape
    Cn = lat.numel() // lat.shape[0]
    out = torch.empty((T, *lat.shape[1:]), dtype=torch.float32, device=lat.device)
    L.check(L.lib().maua_weighted_sum(L.ctx(lat.device), L.ptr(env), L.ptr(lat), T, An, lat.shape[0], C.c_long(Cn),
                                      L.ptr(out)))
    return out


def select_modulo_indices(n_latents, envelope):
    """latent.py:37-40 -> int64 [T] on device (the onset-bin assignment)."""
    env = A._f32(envelope).reshape(-1)
    lo, _ = A.order_stat(env, 2, q=0.25)
    hi, _ = A.order_stat(env, 2, q=0.75)
    y = torch.empty_like(env)
    L.check(L.lib().maua_clamp(L.ctx(env.device), L.ptr(env), L.ptr(lo), L.ptr(hi), C.c_float(0), C.c_float(0),
                               C.c_long(env.numel()), L.ptr(y)))
    y = S.normalize(y)
    idx = torch.empty((env.numel(),), dtype=torch.int64, device=env.device)
    L.check(L.lib().maua_scale_round_index(L.ctx(env.device), L.ptr(y), C.c_float(float(n_latents - 1)),
                                           C.c_long(env.numel()), L.ptr(idx)))
    return idx


def select_modulo(latents, envelope, smooth=2):
    lat = A._f32(latents)
    idx = select_modulo_indices(len(lat), envelope)
    Cn = lat.numel() // lat.shape[0]
    out = torch.empty((idx.numel(), *lat.shape[1:]), dtype=torch.float32, device=lat.device)
    L.check(L.lib().maua_gather_rows(L.ctx(lat.device), L.ptr(lat), L.ptr(idx), lat.shape[0], idx.numel(), C.c_long(Cn),
                                     L.ptr(out)))
    return S.gaussian_filter(out, smooth, causal=0)


def slerp(a, b, t):
    """latent.py:54-65 for a = y[:-1], b = y[1:] stacks: returns [k, n_seg, L, D]."""
    a, b, t = A._f32(a), A._f32(b), A._f32(t).reshape(-1)
    y = torch.cat([a, b[-1:]])
    if not torch.equal(y[1:], b):
        raise NotImplementedError("slerp expects consecutive segments (a = y[:-1], b = y[1:])")
    n_seg, Ln, D = a.shape
    out = torch.empty((t.numel(), n_seg, Ln, D), dtype=torch.float32, device=a.device)
    L.check(L.lib().maua_slerp(L.ctx(a.device), L.ptr(y.contiguous()), L.ptr(t), t.numel(), n_seg, Ln, D, L.ptr(out)))
    return out


def slerp_loops(y, size, n_loops):
    y = A._f32(y)
    y = torch.cat([y] * n_loops + [y[[0]]])
    t = torch.linspace(0, 1, round(size / len(y)))
    out = slerp(y[:-1], y[1:], t)
    out = out.reshape(-1, *out.shape[2:])  # t-major flattening of the reference
    return S.resample(out, size).reshape(size, *y.shape[1:])


def tempo_loops(latents, n_frames, fps, tempo, type="spline"):
    bars_per_sec = tempo / 4 / 60
    n_loops = round(n_frames / fps * bars_per_sec)
    return spline_loops(latents, n_frames, n_loops) if type == "spline" else slerp_loops(latents, n_frames, n_loops)


LAYER_SLICES = {"low": (0, 6), "mid": (6, 12), "high": (12, 18), "lowmid": (0, 12), "midhigh": (6, 18), "all": (0, 18)}


def merge(latents, sequence, merge_type, merge_depth, modulation=None):
    """selfsupervised/latent.py:57-78, in place on `latents` [T, L, D] (device)."""
    T, Ln, D = latents.shape
    l0, l1 = LAYER_SLICES[merge_depth]
    mode = {"average": 0, "modulate": 1}.get(merge_type, 2)
    mod = A._f32(modulation).reshape(-1) if mode == 1 else None
    L.check(L.lib().maua_latent_merge(L.ctx(latents.device), L.ptr(latents), L.ptr(A._f32(sequence)), L.ptr(mod), mode, T,
                                      Ln, D, l0, min(l1, Ln)))
    return latents


def latent_patch(rng, latents, palette, segmentations, features, tempo, fps, patch_type, segments, loop_bars, seq_feat,
                 seq_feat_weight, mod_feat, mod_feat_weight, merge_type, merge_depth):
    """selfsupervised/latent.py:16-80"""
    feature = seq_feat_weight * features[seq_feat]
    permutation = torch.randperm(len(palette), generator=rng, device=rng.device).to(palette.device)
    if patch_type == "segmentation" and (seq_feat, segments) not in segmentations:
        # a saved patch asks for a cut this clip has no segmentation for (short clips drop the k they cannot be cut
        # into, sample.retrieve_music_information): the nearest available k of the same feature, else the
        # segmentation-free "feature" form - the reference raises KeyError here
        have = sorted(k for (name, k) in segmentations if name == seq_feat)
        if have:
            segments = min(have, key=lambda k: (abs(k - segments), k))
        else:
            patch_type = "feature"
    if patch_type == "segmentation":
        segmentation = segmentations[(seq_feat, segments)]
        selection = permutation[:segments]
        sequence = palette[selection[segmentation.to(selection.device)]]
        sequence = A.gaussian_filter(sequence, 5)
    elif patch_type == "feature":
        n_select = feature.shape[1]
        if n_select == 1:
            sel = permutation[:2]
            sequence = single_weighted(palette[sel][1], palette[sel][0], feature[:, 0])
        else:
            # einsum("TN,NWL->TWL") without the normalisation of multi_weighted
            sel = permutation[:n_select]
            f = A._f32(feature)
            sequence = multi_weighted(palette[sel], f) * f.sum(1)[:, None, None]
    elif patch_type == "loop":
        selection = permutation[:segments]
        n_loops = len(latents) / fps / 60 / tempo / 4 / loop_bars
        sequence = spline_loop_latents(palette[selection], len(latents), n_loops=n_loops)
    else:
        raise ValueError(patch_type)
    sequence = A.gaussian_filter(sequence, 1)
    mod = (mod_feat_weight * features[mod_feat]).reshape(len(latents), -1)[:, 0] if merge_type == "modulate" else None
    return merge(latents, sequence, merge_type, merge_depth, mod)


def _eerp(a, b, t, mode):
    a, b, t = (A._f32(torch.as_tensor(v)) for v in (a, b, t))
    a, b, t = torch.broadcast_tensors(a, b, t)
    a, b, t = a.contiguous(), b.contiguous(), t.contiguous()
    y = torch.empty_like(a)
    L.check(L.lib().maua_eerp(L.ctx(a.device), L.ptr(a), L.ptr(b), L.ptr(t), C.c_long(a.numel()), mode, L.ptr(y)))
    return y


def eerp(a, b, t):
    """latent.py:46-47 exponential interpolation a^(1-t) * b^t"""
    return _eerp(a, b, t, 0)


def copeerp(a, b, t):
    """latent.py:50-51"""
    return _eerp(a, b, t, 1)
