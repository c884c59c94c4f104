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
