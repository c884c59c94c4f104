"""Oracle restatement of the selfsupervised noise modules (test infrastructure only).
Follows maua/audiovisual/audioreactive/selfsupervised/noise.py; plain PyTorch-CPU fp32 functions."""
import torch

EPS = torch.finfo(torch.float32).eps


def loop(planes, idx, i, b, sigma):
    """noise.py:49-53 — planes [3,h,w], idx [T]."""
    freqs = torch.cos(idx[i:i + b, None, None] + planes[[0]]).div(sigma / 50)
    out = torch.sin(freqs + planes[[1]]) * planes[[2]]
    return out / (out.square().mean(dim=(1, 2), keepdim=True).sqrt() + EPS)


def blend(noise, modulator, i, b):
    """noise.py:19-24 — noise [2,M,h,w]."""
    mod = modulator[i:i + b].reshape(-1, noise.shape[1])
    return torch.einsum("MHW,BM->BHW", noise[0], mod) + torch.einsum("MHW,BM->BHW", noise[1], 1 - mod)


def multiply(noise, modulator, i, b):
    """noise.py:35-39 — noise [M,h,w]."""
    mod = modulator[i:i + b].reshape(-1, noise.shape[0])
    return torch.einsum("MHW,BM->BHW", noise, mod)


def average(left, right):
    return (left + right) / 2  # noise.py:62-63


def modulate(left, right, modulator, i, b):
    mod = modulator.mean(1)[i:i + b, None, None]  # noise.py:71-75
    return left * mod + right * (1 - mod)


def scale_bias(base, scale, bias):
    return scale * base + bias  # noise.py:85-86


DEPTHS = {"low": range(0, 6), "mid": range(6, 12), "high": range(12, 17), "lowmid": range(0, 12),
          "midhigh": range(6, 17), "all": range(0, 17)}


def noise_patch(planes, noise, sizes, features, tempo, fps, patch_type, loop_bars, seq_feat, seq_feat_weight, mod_feat,
                mod_feat_weight, merge_type, merge_depth, noise_mean, noise_std, only=None):
    """noise.py:89-140 on closures: ``noise`` is a list of functions (i, b) -> [b,h,w]; ``planes[n]`` is an iterator
    that yields, for layer n, the random tensor the module created here draws (Blend [2,M,h,w], Multiply [M,h,w],
    Loop [3,h,w]) - the reference draws them from its generator in this order.  Returns the new list.  ``only``: restrict the
    work to these layers (the others keep their entry)."""
    feature = seq_feat_weight * features[seq_feat]
    out = list(noise)
    for n in DEPTHS[merge_depth]:
        if only is not None and n not in only:
            continue
        pl = next(planes[n])
        if patch_type == "blend":
            new = (lambda pl: lambda i, b: blend(pl, feature, i, b))(pl)
        elif patch_type == "multiply":
            new = (lambda pl: lambda i, b: multiply(pl, feature, i, b))(pl)
        elif patch_type == "loop":
            n_loops = len(feature) / fps / 60 / tempo / 4 / loop_bars
            idx = torch.linspace(0, n_loops * 2 * torch.pi, len(feature))
            new = (lambda pl, idx: lambda i, b: loop(pl, idx, i, b, 5))(pl, idx)
        else:
            raise ValueError(patch_type)
        old = out[n]
        if merge_type == "average":
            merged = (lambda old, new: lambda i, b: average(old(i, b), new(i, b)))(old, new)
        elif merge_type == "modulate":
            mod = mod_feat_weight * features[mod_feat]
            merged = (lambda old, new, mod: lambda i, b: modulate(old(i, b), new(i, b), mod, i, b))(old, new, mod)
        else:
            merged = new
        out[n] = (lambda m: lambda i, b: scale_bias(m(i, b), noise_std, noise_mean))(merged)
    return out
