"""Example audio-reactive patch in the spirit of maua/audiovisual/patches/examples/stylegan2.py:13-68, restricted to
the feature functions that exist in-tree in the reference (its own example calls an un-vendored ``ar`` API):
onsets drive a blend between two spline-loop latent schedules; noise loops per layer."""
import torch

from ... import audioreactive as ar
from ....noise import Loop
from ..base.stylegan2 import StyleGAN2Patch


class ExampleSG2Patch(StyleGAN2Patch):
    def process_audio(self):
        self.onsets = ar.onsets(self.audio, self.sr, type="rosa")
        self.onsets = ar.gaussian_filter(ar.resample(self.onsets, self.n_frames), 2, causal=0.2)

    def process_mapper_inputs(self):
        return {"latent_z": self.stylegan2.get_z_latents("0-12").float(), "truncation": 1.0}

    def process_synthesizer_inputs(self, latent_w):
        n = len(latent_w) // 2
        low = ar.spline_loops(latent_w[:n], self.n_frames, n_loops=2)
        high = ar.spline_loops(latent_w[n:2 * n], self.n_frames, n_loops=2)
        from ....latent import sequence_weighted
        latents = ar.gaussian_filter(sequence_weighted(low, high, ar.normalize(self.onsets)), 2, causal=0.2)
        inputs = {"latents": latents}
        rng = torch.Generator().manual_seed(42)
        for l, (_, _, _, res, _) in enumerate(self.synthesizer.G_synth.layer_shapes()):
            inputs[f"noise{l}"] = Loop(rng, self.n_frames, (res, res), n_loops=4, sigma=5)
        return inputs
