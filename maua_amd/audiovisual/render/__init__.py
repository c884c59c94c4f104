"""Render back-ends.  Interface of maua/audiovisual/render/__init__.py:1-18 (``Renderer`` base with a ``device``
attribute, ``get_output_class(name)``), plus the two helpers the back-ends share for slicing per-frame inputs."""
import importlib

import torch

_BACKENDS = {"memmap": (".memmap", "MemMap"), "ffmpeg": (".ffmpeg", "FFMPEG")}


class Renderer:
    """Base of the back-ends; ``device`` is where they run the synthesizer."""

    def __init__(self):
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def get_output_class(renderer):
    """Name -> back-end class (imported on demand); unknown names raise NotImplementedError like the reference."""
    try:
        module, cls = _BACKENDS[renderer]
    except KeyError:
        raise NotImplementedError(f"unknown renderer {renderer!r}; choose from {sorted(_BACKENDS)}") from None
    return getattr(importlib.import_module(module, __name__), cls)


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
