"""MemMap renderer (drop-in for maua/audiovisual/render/memmap.py:10-34): frames accumulate in
workspace/frames_memmap.npy as uint8 [T,3,H,W] and are returned memory-mapped."""
import os

import numpy as np
import torch

from . import Renderer, batch_inputs, n_frames_of


class MemMap(Renderer):
    def __init__(self, batch_size=8, cache_file="workspace/frames_memmap.npy", **_):
        super().__init__()
        self.batch_size, self.cache_file = batch_size, cache_file

    def __call__(self, synthesizer, inputs, postprocess=lambda x: x):
        T = n_frames_of(inputs)
        W, H = synthesizer.output_size
        if hasattr(synthesizer, "G_synth"):  # the size actually rendered (output_size rounded to the resize layer's
            H, W = synthesizer.G_synth.output_hw  # multiple; force_output_size resamples afterwards)
        os.makedirs(os.path.dirname(self.cache_file) or ".", exist_ok=True)
        frames = np.lib.format.open_memmap(self.cache_file, mode="w+", dtype=np.uint8, shape=(T, 3, H, W))
        for i in range(0, T, self.batch_size):
            b = min(self.batch_size, T - i)
            img = synthesizer(**batch_inputs(inputs, i, b))
            # memmap.py:32: .add(1).div(2).clamp(0,1).mul(255) ... astype(uint8) (truncation, as the reference)
            frames[i:i + b] = img.add(1).div(2).clamp(0, 1).mul(255).cpu().numpy().astype(np.uint8)
        frames.flush()
        return postprocess(np.load(self.cache_file, mmap_mode="r"))
