"""Hop-aligned audio-reactive sampler (drop-in for
maua/audiovisual/audioreactive/selfsupervised/sample.py:36-107 ``generate`` and patch.py:34-197 ``Patch``).

Audio is resampled to sr = 1024*fps so that one STFT hop == one video frame.  Features: the reference's AFEATFNS (chromagram, tonnetz, mfcc,
spectral_contrast, spectral_flatness, rms, drop_strength, onsets).  The tempo is estimated from the onset envelope (the autocorrelation-tempogram estimate the
reference takes from librosa), beats by librosa's published dynamic program and the beat-synchronous Laplacian segmentations
by maua_amd.segment (device kernels; librosa / torch_geometric / sklearn are un-vendored).  Frames are sharded by contiguous range over the ranks of the current process group and gathered to
rank 0 chunk by chunk while the next batch renders (distributed.StreamingGather: RCCL point-to-point over xGMI).

    python -m maua_amd.audiovisual.sample --audio_file clip.wav --stylegan2_checkpoint None --downscale_factor 4
"""
import argparse
import json
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .. import audio as A
from .. import latent as LT
from .. import noise as N
from .. import segment as SG
from ..audio_io import load_audio
from ..distributed import StreamingGather, world_info
from ..pipeline import frame_range
from ..stylegan2 import StyleGAN2
from ..video import VideoWriter, muxable_audio

# selfsupervised/mir.py:9-11
AFEATFNS = [A.chromagram, A.tonnetz, A.mfcc, A.spectral_contrast, A.spectral_flatness, A.rms, A.drop_strength, A.onsets]
UNITFEATS = ["rms", "drop_strength", "onsets", "spectral_flatness"]
ALLFEATS = ["chromagram", "tonnetz", "mfcc", "spectral_contrast"] + UNITFEATS


def retrieve_music_information(audio, sr, ks=(2, 4, 6, 8, 12, 16), device="cuda"):
    """selfsupervised/mir.py:24-45 -> (features, segmentations, tempo).  Every feature function on the clip; tempo from the
    onset envelope (audio.tempo: the autocorrelation estimate the reference takes from librosa, :27-30); beats at that
    tempo (segment.beat_track, :31-33, a leading beat at frame 0 dropped); per feature and per k the argmax of the
    Laplacian segmentation (:35-38) plus the CQT / MFCC "rosa" segmentation (:40-41); then gaussian(2) -> salience ->
    normalize on the features (:43)."""
    raw = {fn.__name__: fn(audio, sr) for fn in AFEATFNS}
    raw = {k: (v if v.dim() > 1 else v.unsqueeze(-1)) for k, v in raw.items()}
    onset_env = raw["onsets"].squeeze()
    tempo = A.tempo(onset_env)
    beats = [int(b) for b in SG.beat_track(onset_env, tempo)]
    if beats and beats[0] == 0:
        del beats[0]
    # (a clip with fewer beat-synchronous frames than segments cannot be cut that finely - the reference fails on such
    #  clips; here the affected k are left out, and a clip of <= 7 beats gets no segmentations at all)
    n_sync = len(beats) + 1
    ks = [k for k in ks if k <= n_sync] if n_sync > 7 else []
    segmentations = {}
    if ks:  # (both calls raise on an empty k list / on <= 7 beat-synchronous frames)
        for name, feature in raw.items():
            for k, s in zip(ks, SG.laplacian_segmentation(feature, beats, ks=ks)):
                segmentations[(name, k)] = s.argmax(1)
        n_frames = raw[AFEATFNS[0].__name__].shape[0]
        for k, seg in zip(ks, SG.laplacian_segmentation_rosa(audio, sr, n_frames, ks=ks, beats=beats).unbind(1)):
            segmentations[("rosa", k)] = seg
    feats = {k: A.normalize(A.salience_weighted(A.gaussian_filter(v, sigma=2))) for k, v in raw.items()}
    return feats, segmentations, tempo


def random_choice(rng, options, weights=None):
    p = torch.ones(len(options)) / len(options) if weights is None else torch.tensor(weights, dtype=torch.float) / np.sum(weights)
    return options[p.multinomial(num_samples=1, generator=rng)]


class Patch(torch.nn.Module):
    """patch.py:34-197.  The generator lives on the host (the reference's is a device generator whose stream is
    device / version specific, SURVEY L6: parity is on explicit selections, not on torch's stream)."""

    def __init__(self, features, segmentations, tempo, fps=24, seed=42, min_subpatches=2, max_subpatches=20, device="cuda"):
        super().__init__()
        rng = torch.Generator("cpu").manual_seed(seed)
        self.seed, self.rng, self.fps, self.tempo = seed, rng, fps, tempo
        self.features, self.segmentations = features, segmentations
        self.ks = np.unique([k for (_, k) in segmentations]).tolist()
        self.length = features[list(features.keys())[0]].shape[0]
        self.n_base_latents = torch.randint(3, 15, size=(), generator=rng).item()
        self.sigma_base_noise = 1 + 9 * torch.rand((), generator=rng).item()
        self.loops_base_noise = random_choice(rng, [1, 2, 4, 8, 16, 32, 64])
        n = lambda: int(torch.randint(min_subpatches, max_subpatches, size=(), generator=rng))
        self.latent_patches = [self.random_latent_patch() for _ in range(n())]
        self.noise_patches = [self.random_noise_patch() for _ in range(n())]

    def _common(self):
        r = self.rng
        return dict(loop_bars=random_choice(r, [4, 8, 16, 32], weights=[2, 2, 2, 1]), seq_feat=random_choice(r, ALLFEATS),
                    seq_feat_weight=1, mod_feat=random_choice(r, UNITFEATS), mod_feat_weight=1,
                    merge_type=random_choice(r, ["average", "modulate"], weights=[1, 3]),
                    merge_depth=random_choice(r, ["low", "mid", "high", "lowmid", "midhigh", "all"], weights=[3, 3, 3, 2, 2, 1]))

    def random_latent_patch(self):
        if not self.ks:  # no segmentations (clip too short to segment): only the sub-patch types that need none
            return dict(patch_type=random_choice(self.rng, ["feature", "loop"]),
                        segments=random_choice(self.rng, [2, 4, 6, 8, 12, 16]), **self._common())
        return dict(patch_type=random_choice(self.rng, ["segmentation", "feature", "loop"]),
                    segments=random_choice(self.rng, self.ks), **self._common())

    def random_noise_patch(self):
        return dict(patch_type=random_choice(self.rng, ["blend", "multiply", "loop"]), **self._common(), noise_mean=0, noise_std=1)

    def forward(self, latent_palette, downscale_factor=1, aspect_ratio=1):
        self.rng.manual_seed(self.seed)
        base = torch.randperm(len(latent_palette), generator=self.rng)[: self.n_base_latents]
        latents = LT.spline_loop_latents(latent_palette[base.to(latent_palette.device)], self.length).contiguous()
        for sub in self.latent_patches:
            latents = LT.latent_patch(self.rng, latents, latent_palette, self.segmentations, self.features, self.tempo, self.fps, **sub)
        sizes = [4, 8, 8, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 512, 1024, 1024]
        noise = [N.Loop(rng=self.rng, length=self.length,
                        size=(round(aspect_ratio * s / downscale_factor), round(s / downscale_factor)),
                        n_loops=self.loops_base_noise, sigma=self.sigma_base_noise) for s in sizes]
        for sub in self.noise_patches:
            noise = N.noise_patch(self.rng, noise, self.features, self.tempo, self.fps, **sub)
        return latents, noise

    def __repr__(self):
        """patch.py:157-178: the two sub-patch tables."""
        reprs = []
        for patches in [self.latent_patches, self.noise_patches]:
            header = [""] + [k for k in patches[0]]
            values = [[str(i + 1)] + [(f"{v:.4f}" if isinstance(v, float) else f"{v}").replace("spectral_", "")
                                      for v in p.values()] for i, p in enumerate(patches)]
            widths = [max(len(row[n]) for row in [header] + values) for n in range(len(header))]
            rows = [header, ["-" * w for w in widths]] + values
            reprs.append([" | ".join(row[c].ljust(widths[c]) for c in range(len(row))) for row in rows])
        return ("Patch(\n  Latent(\n    " + "\n    ".join(reprs[0]) + "\n  ),\n  Noise(\n    " + "\n    ".join(reprs[1])
                + "\n  )\n)")

    @staticmethod
    def load(path, features, segmentations, tempo, fps, device="cuda"):
        """patch.py:190-197: a fresh Patch whose attributes are overwritten by the saved JSON (seed, sub-patch lists,
        base-noise parameters), so the same file reproduces the same latent / noise sequences."""
        patch = Patch(features=features, segmentations=segmentations, tempo=tempo, fps=fps, device=device)
        for key, val in json.loads(Path(path).read_text()).items():
            setattr(patch, key, val)
        return patch

    def save(self, path):
        Path(path).write_text(json.dumps(dict(seed=self.seed, latent_patches=self.latent_patches,
                                              noise_patches=self.noise_patches, n_base_latents=self.n_base_latents,
                                              sigma_base_noise=self.sigma_base_noise,
                                              loops_base_noise=self.loops_base_noise), default=str))


@torch.inference_mode()
def generate(audio_file: str, stylegan2_checkpoint: Optional[str] = None, patch_file: Optional[str] = None,
             seed: Optional[int] = None, latent_seeds: Optional[str] = None, fps: float = 30, audio_offset: float = 0,
             audio_duration: Optional[float] = None, downscale_factor: float = 4, aspect_ratio: float = 1,
             batch_size: int = 32, device: str = "cuda", tempo: Optional[float] = None, out_dir: str = "output",
             reference_tail: bool = False, dtype=torch.bfloat16, upscale: Optional[str] = None, upscale_batch: int = 4,
             upscale_random_init: bool = False):
    """sample.py:36-101.  ``reference_tail=True`` reproduces the reference loop's dropped tail.
    ``tempo``: BPM for the "loop" sub-patches; None = estimated from the onset envelope like the reference (mir.py:27-30).

    ``upscale`` = one of the reference's RealESRGAN model names (super/image/models/realesrgan.py:13-19) fuses BASELINE
    configs[4] into the render: every batch of frames goes render -> x4 (``upscale_batch`` frames per network call,
    RealESRGANer.enhance's arithmetic per frame) -> writer without leaving the device - what the reference does in two passes
    through a video file (generate, then super/video/frame_by_frame.py:22-33).  A 4096^2 frame is 48 MiB (3600 frames: 169 GB),
    so there is NO gather in this mode: every rank encodes its own contiguous frame range into ``<stem>_partRRR.mp4`` and rank 0
    joins the parts in order (ffmpeg concat demuxer, stream copy; without ffmpeg the raw parts and the list stay) -
    returns (joined file | list file, None)."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31, size=()).item())
    rank, world = world_info()
    res = round(1024 / downscale_factor)
    out_size = (round(aspect_ratio * 1024 / downscale_factor), res)   # (width, height), sample.py:53
    out_file = f"{out_dir}/{Path(audio_file).stem}_RandomPatches++_seed{seed}_{out_size[0]}x{out_size[1]}.mp4"
    audio, sr = load_audio(audio_file, audio_offset, audio_duration, fps)
    features, segmentations, est_tempo = retrieve_music_information(audio, sr)
    tempo = est_tempo if tempo is None else tempo
    if patch_file is None:
        patch = Patch(features=features, segmentations=segmentations, tempo=tempo, seed=seed, fps=fps)
    else:  # sample.py:62-66
        patch = Patch.load(patch_file, features=features, segmentations=segmentations, tempo=tempo, fps=fps)
    G = StyleGAN2(model_file=stylegan2_checkpoint, output_size=out_size, dtype=dtype,
                  generator=torch.Generator().manual_seed(seed))
    if latent_seeds is None:
        z = torch.randn((180, 512), generator=torch.Generator().manual_seed(seed))
        palette = G.mapper(z)
    else:
        palette = G.get_w_latents(latent_seeds)
    latents, noise = patch.forward(palette, downscale_factor=1024 / res, aspect_ratio=aspect_ratio)
    noise = noise[: G.synthesizer.G_synth.num_layers]
    T = len(latents) - (len(latents) % batch_size if reference_tail else 0)
    if reference_tail and len(latents) % batch_size == 0:
        T -= batch_size  # range(0, len - B, B) never reaches the last full batch either
    lo, hi = frame_range(T, rank, world)
    # the network renders output_size rounded to the resize layer's multiple (wrappers/stylegan2.py:115-120); the
    # reference writes those frames into a writer opened at out_size — here they are resampled to out_size first
    rh, rw = G.synthesizer.G_synth.output_hw
    if upscale is not None:
        return _generate_upscaled(G, latents, noise, T, lo, hi, rank, world, batch_size, upscale, upscale_batch, upscale_random_init,
                                  dtype, out_file, fps, audio_file, audio_offset, audio_duration, patch, (rw, rh))
    # frames travel to rank 0 chunk by chunk while the next batch renders (RCCL point-to-point over xGMI; rank 0 renders
    # straight into the clip buffer)
    sg = StreamingGather(T, (rh, rw, 3), batch_size, dtype=torch.uint8, device="cuda", rank=rank, world=world)
    assert (sg.lo, sg.hi) == (lo, hi)
    for off, b in sg.chunks():
        i = lo + off
        nz = {f"noise{j}": m.forward(i, b)[:, None] for j, m in enumerate(noise)}
        G.synthesizer(latents=latents[i:i + b], rgb8_out=sg.local[off:off + b], **nz)
        sg.chunk_done()
    frames = sg.finish()
    if rank == 0:
        wav = muxable_audio(audio_file)
        with VideoWriter(out_file, out_size, fps, wav, audio_offset, audio_duration) as video:
            for i in range(0, T, 64):
                chunk = frames[i:i + 64]
                if (rw, rh) != tuple(out_size):
                    from ..ops import resample
                    f = resample(chunk.permute(0, 3, 1, 2).float(), (out_size[1], out_size[0]))
                    chunk = f.clamp(0, 255).round().byte().permute(0, 2, 3, 1).contiguous()
                video.write(chunk)
        patch.save(out_file.replace(".mp4", ".json"))
    return out_file, frames


def _generate_upscaled(G, latents, noise, T, lo, hi, rank, world, batch_size, model_name, upscale_batch, random_init, dtype,
                       out_file, fps, audio_file, audio_offset, audio_duration, patch, render_wh):
    """configs[4]: this rank's frames lo .. hi as render -> RealESRGAN x4 -> its own writer; rank 0 joins the parts."""
    from ..super import load_model
    up = load_model(model_name, dtype=dtype, allow_random_init=random_init)
    rw, rh = render_wh
    s = up.scale
    stem = out_file[: -len(".mp4")] + f"_{model_name}_{s * rw}x{s * rh}"
    u8 = torch.empty((batch_size, rh, rw, 3), dtype=torch.uint8, device="cuda")

    def write_part(path, lo_, hi_):
        n_written = 0
        with VideoWriter(path, (s * rw, s * rh), fps) as video:
            for i in range(lo_, hi_, batch_size):
                b = min(batch_size, hi_ - i)
                nz = {f"noise{j}": m.forward(i, b)[:, None] for j, m in enumerate(noise)}
                G.synthesizer(latents=latents[i:i + b], rgb8_out=u8[:b], **nz)
                for k in range(0, b, upscale_batch):
                    video.write(up.enhance_frames(u8[k:min(b, k + upscale_batch)]))
                    n_written += min(b, k + upscale_batch) - k
        return n_written
    from ..distributed import write_parts_and_join
    out = write_parts_and_join(stem, T, rank, world, write_part, audio_file=muxable_audio(audio_file), audio_offset=audio_offset,
                               audio_duration=audio_duration, on_rank0=lambda: patch.save(stem + ".json"))
    return out, None


def main(argv=None):
    torch.set_num_threads(min(torch.get_num_threads(), 8))   # host side = small tensors (see _lib.host_threads)
    ap = argparse.ArgumentParser(description="argparse shim for the reference's fire.Fire(generate): same names/defaults")
    ap.add_argument("--audio_file", required=True)
    ap.add_argument("--stylegan2_checkpoint", default=None)
    ap.add_argument("--patch_file", default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--latent_seeds", default=None)
    ap.add_argument("--fps", type=float, default=30)
    ap.add_argument("--audio_offset", type=float, default=0)
    ap.add_argument("--audio_duration", type=float, default=None)
    ap.add_argument("--downscale_factor", type=float, default=4)
    ap.add_argument("--aspect_ratio", type=float, default=1)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--tempo", type=float, default=None, help="BPM of the loop sub-patches (default: estimated from the onset envelope)")
    ap.add_argument("--reference_tail", action="store_true")
    ap.add_argument("--upscale", default=None, help="RealESRGAN model name: render -> x4 per frame, one part file per rank (configs[4])")
    ap.add_argument("--upscale_batch", type=int, default=4)
    ap.add_argument("--upscale_random_init", action="store_true")
    a = ap.parse_args(argv)
    from ..distributed import maybe_init_process_group
    maybe_init_process_group()
    return generate(**vars(a))[0]


if __name__ == "__main__":
    main()
