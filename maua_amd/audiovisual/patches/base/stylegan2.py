"""StyleGAN2Patch (drop-in for maua/audiovisual/patches/base/stylegan2.py:7-53)."""
import torch

from ....stylegan2 import StyleGAN2, StyleGAN2Mapper, StyleGAN2Synthesizer  # noqa: F401
from . import MauaPatch


class StyleGAN2Patch(MauaPatch):
    """Patch around a StyleGAN2 generator: builds it (checkpoint or random init, output size through the feature-space
    resize) and exposes its two halves as ``mapper`` / ``synthesizer``.  The default stages render one random latent."""

    def __init__(self, model_file, audio_file, fps=24, offset=0, duration=-1, output_size=(1024, 1024),
                 resize_strategy="pad-zero", resize_layer=0, inference=False, **generator_kwargs):
        MauaPatch.__init__(self, audio_file, fps, offset, duration)
        G = StyleGAN2(model_file, inference, output_size, resize_strategy, resize_layer, **generator_kwargs)
        self.stylegan2, self.mapper, self.synthesizer = G, G.mapper, G.synthesizer

    def process_mapper_inputs(self):
        z = torch.randn((1, 512))
        return dict(latent_z=z)

    def process_synthesizer_inputs(self, latent_w):  # identity: the mapped latents are the synthesizer's inputs
        return latent_w

    def process_outputs(self, video):  # identity post-process
        return video
