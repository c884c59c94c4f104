"""MauaPatch protocol (drop-in for maua/audiovisual/patches/base/__init__.py:7-44)."""
import importlib
import importlib.util
import inspect
from pathlib import Path

import torch

from ... import audioreactive as ar


class MauaPatch:
    """Base of user patches.  Attributes the reference's patches rely on: ``audio`` (numpy, mono), ``sr``,
    ``duration`` (s), ``fps``, ``n_frames``, ``audio_file``, ``device``; subclasses provide ``mapper`` /
    ``synthesizer`` and the four ``process_*`` stages."""

    def __init__(self, audio_file, fps=24, offset=0, duration=-1) -> None:
        waveform, self.sr, self.duration = ar.load_audio(audio_file, offset, duration)
        self.audio = waveform.numpy()
        self.audio_file, self.fps = audio_file, fps
        # Python's round: half to even, like the reference (base/__init__.py:16)
        self.n_frames = round(self.duration * self.fps)
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def process_audio(self):
        """hook: derive envelopes / features from self.audio (default: nothing)"""

    def force_output_size(self, video):
        t, c, h, w = video.shape
        if (w, h) != tuple(self.synthesizer.output_size):  # lanczos + bicubic (maua/ops/image.py:214-240)
            from ....ops import resample
            size = tuple(reversed(self.synthesizer.output_size))
            if isinstance(video, torch.Tensor):
                return resample(video, size)
            # uint8 frame stacks (the memmap renderer hands its whole [T,3,H,W] array over): resample in chunks
            import numpy as np
            out = np.empty((t, c, size[0], size[1]), dtype=np.uint8)
            for i in range(0, t, 64):
                chunk = torch.from_numpy(np.ascontiguousarray(video[i:i + 64])).float()
                out[i:i + 64] = resample(chunk, size).clamp(0, 255).round().byte().cpu().numpy()
            return out
        return video


def get_patch_from_file(filepath, class_name=None):
    """base/__init__.py:28-44: first class in the file that extends MauaPatch (optionally by name).  Accepts a
    dotted-module-style path like the reference or a real file path."""
    module_name = filepath.replace(".py", "").replace("/", ".").lstrip(".")
    try:  # the reference's way: the path doubles as a dotted module name relative to the working directory
        module = importlib.import_module(module_name)
    except ImportError:
        p = Path(filepath)
        if not p.exists():
            raise
        spec = importlib.util.spec_from_file_location(p.stem, p)  # a stand-alone user file (absolute imports)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        module_name = module.__name__
    for _, cls in inspect.getmembers(module, inspect.isclass):
        if cls.__module__ == module_name and issubclass(cls, MauaPatch) and (class_name is None or cls.__name__ == class_name):
            return cls
    raise Exception("Patch not found! Are you sure there is a class that extends MauaPatch in the file you specified "
                    "and that the name you (might have) specified is correct?")
