"""Render back-ends (drop-in for maua/audiovisual/render/__init__.py:1-18)."""
import torch


class Renderer:
    def __init__(self):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def get_output_class(renderer):
    if renderer == "memmap":
        from .memmap import MemMap
        return MemMap
    if renderer == "ffmpeg":
        from .ffmpeg import FFMPEG
        return FFMPEG
    raise NotImplementedError


def batch_inputs(inputs, i, b):
    """Slice a dict of per-frame inputs: tensors by index, index-addressed noise modules via forward(i, b)."""
    out = {}
    for k, v in inputs.items():
        if hasattr(v, "forward") and hasattr(v, "length"):
            out[k] = v.forward(i, b)[:, None]
        else:
            out[k] = v[i:i + b]
    return out


def n_frames_of(inputs):
    v = next(iter(inputs.values()))
    return v.length if hasattr(v, "length") else len(v)
