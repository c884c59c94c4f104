"""Oracle restatement of the latent schedule builders (test infrastructure only).
Follows maua/audiovisual/audioreactive/latent.py (single_weighted :12-18, multi_weighted :21-31,
select_modulo :34-43, slerp :54-65, slerp_loops :68-80, spline_loops :83-92) and
.../selfsupervised/latent.py (spline_loop_latents :7-13, latent_patch merges :57-78).

The cubic spline arithmetic lives in the un-vendored dependency torchcubicspline (setup.py:104, unpinned
git URL).  Its published algorithm: the natural cubic spline through the knots (second derivative zero at
both ends), evaluated piecewise; that interpolant is unique, so it is restated here as the standard
tridiagonal solve (float64) and pinned against scipy's CubicSpline(bc_type="natural") fixture."""
import numpy as np
import torch
import torch.nn.functional as F

from . import signal as S


def natural_cubic_spline(t_knots, y_knots, t_eval):
    """y_knots [n, ...]; returns [len(t_eval), ...] (float64 numpy)."""
    t = np.asarray(t_knots, dtype=np.float64)
    y = np.asarray(y_knots, dtype=np.float64)
    n = len(t)
    shape = y.shape[1:]
    y = y.reshape(n, -1)
    h = np.diff(t)
    # second derivatives M: natural end conditions M0 = Mn-1 = 0
    A = np.zeros((n, n))
    rhs = np.zeros((n, y.shape[1]))
    A[0, 0] = A[-1, -1] = 1.0
    for i in range(1, n - 1):
        A[i, i - 1] = h[i - 1]
        A[i, i] = 2 * (h[i - 1] + h[i])
        A[i, i + 1] = h[i]
        rhs[i] = 6 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
    M = np.linalg.solve(A, rhs)
    te = np.asarray(t_eval, dtype=np.float64)
    idx = np.clip(np.searchsorted(t, te, side="right") - 1, 0, n - 2)
    hi = h[idx][:, None]
    a = (t[idx + 1] - te)[:, None]
    b = (te - t[idx])[:, None]
    out = (M[idx] * a ** 3 + M[idx + 1] * b ** 3) / (6 * hi) + (y[idx] / hi - M[idx] * hi / 6) * a + \
          (y[idx + 1] / hi - M[idx + 1] * hi / 6) * b
    return out.reshape((len(te),) + shape)


def spline_loops(y, size, n_loops):
    """latent.py:83-92"""
    Y = torch.cat([y] * n_loops + [y[[0]]])
    t_in = torch.linspace(0, 1, len(Y)).double().numpy()
    t_out = torch.linspace(0, 1, size).double().numpy()
    return torch.from_numpy(natural_cubic_spline(t_in, Y.double().numpy(), t_out)).float()


def spline_loop_latents(y, size, n_loops=1):
    """selfsupervised/latent.py:7-13"""
    Y = torch.cat((y, y[[0]]))
    t_in = torch.linspace(0, 1, len(Y)).double().numpy()
    t_out = (torch.linspace(0, n_loops, size) % 1).double().numpy()
    return torch.from_numpy(natural_cubic_spline(t_in, Y.double().numpy(), t_out)).float()


def single_weighted(low, high, env):
    """latent.py:12-18"""
    return low[None] * (1 - env[:, None, None]) + high[None] * env[:, None, None]


def multi_weighted(latents, envelopes):
    """latent.py:21-31 (the reference normalises its argument in place, Q8; this copy does not)."""
    e = envelopes / envelopes.sum(dim=1, keepdim=True)
    sel = latents[torch.arange(e.shape[1]) % len(latents)]
    return torch.einsum("ta,awl->twl", e, sel)


def select_modulo_indices(n_latents, env):
    """latent.py:37-40 — the int64 'onset-bin assignment' (bit-exact requirement)."""
    low, high = torch.quantile(env, 0.25), torch.quantile(env, 0.75)
    idx = S.normalize(env.clamp(low, high)) * (n_latents - 1)
    return idx.round().long()


def select_modulo(latents, env, smooth=2):
    """latent.py:34-43"""
    out = latents[select_modulo_indices(len(latents), env)]
    return S.gaussian_filter(out, smooth, causal=0)


def slerp(a, b, t):
    """latent.py:54-65 — returns [k, n_seg, layers, dim], unit-normalised (Q9)."""
    a = a / a.norm(dim=-1, keepdim=True)
    b = b / b.norm(dim=-1, keepdim=True)
    d = (a * b).sum(dim=-1, keepdim=True)
    p = t * torch.acos(d)
    p = p.permute(2, 0, 1)[..., None]
    c = b - d * a
    c = c / c.norm(dim=-1, keepdim=True)
    d = a[None] * torch.cos(p) + c[None] * torch.sin(p)
    return d / d.norm(dim=-1, keepdim=True)


def slerp_loops(y, size, n_loops):
    """latent.py:68-80 — incl. the t-major flattening quirk (Q9)."""
    y = torch.cat([y] * n_loops + [y[[0]]])
    t = torch.linspace(0, 1, round(size / len(y)))
    ya, yb = y[:-1], y[1:]
    out = slerp(ya, yb, t)
    out = out.reshape(-1, *out.shape[2:])
    out = F.interpolate(out.permute(1, 2, 0), size=size, mode="linear", align_corners=False)
    return out.permute(2, 0, 1)


LAYER_SLICES = {"low": slice(0, 6), "mid": slice(6, 12), "high": slice(12, 18), "lowmid": slice(0, 12),
                "midhigh": slice(6, 18), "all": slice(0, 18)}


def merge(latents, sequence, merge_type, merge_depth, modulation=None):
    """selfsupervised/latent.py:57-78 on a copy."""
    lat = latents.clone()
    lays = LAYER_SLICES[merge_depth]
    if merge_type == "average":
        lat[:, lays] = (lat[:, lays] + sequence[:, lays]) / 2
    elif merge_type == "modulate":
        m = modulation[..., None]
        lat[:, lays] = lat[:, lays] * (1 - m) + m * sequence[:, lays]
    else:
        lat[:, lays] = sequence[:, lays]
    return lat


def latent_patch(permutation, latents, palette, segmentations, features, tempo, fps, patch_type, segments, loop_bars,
                 seq_feat, seq_feat_weight, mod_feat, mod_feat_weight, merge_type, merge_depth):
    """selfsupervised/latent.py:16-80 with the permutation the sub-patch draws (``torch.randperm(len(palette),
    generator=rng)``, :36) passed in explicitly; returns a new tensor."""
    from .audio import gaussian_filter
    feature = seq_feat_weight * features[seq_feat]
    if patch_type == "segmentation":                                   # :38-42
        selection = permutation[:segments]
        sequence = gaussian_filter(palette[selection[segmentations[(seq_feat, segments)]]], 5)
    elif patch_type == "feature":                                      # :43-50
        n_select = feature.shape[1]
        if n_select == 1:
            sel = palette[permutation[:2]]
            sequence = feature[..., None] * sel[[0]] + (1 - feature[..., None]) * sel[[1]]
        else:
            sequence = torch.einsum("TN,NWL->TWL", feature, palette[permutation[:n_select]])
    elif patch_type == "loop":                                         # :51-54
        n_loops = len(latents) / fps / 60 / tempo / 4 / loop_bars
        sequence = spline_loop_latents(palette[permutation[:segments]], len(latents), n_loops=n_loops)
    else:
        raise ValueError(patch_type)
    sequence = gaussian_filter(sequence, 1)                            # :55
    modulation = mod_feat_weight * features[mod_feat] if merge_type == "modulate" else None
    return merge(latents, sequence, merge_type, merge_depth, modulation)
