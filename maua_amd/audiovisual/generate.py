"""Patch-driven audio-reactive render entry point (drop-in for maua/audiovisual/generate.py:16-98:
same function name, arguments and CLI flags).

    python -m maua.audiovisual.generate --audio_file clip.wav --model_file None \
        --patch_file maua/audiovisual/patches/examples/stylegan2.py
(`maua/` is the reference's import surface for this path; it re-exports `maua_amd`)
"""
import argparse
from pathlib import Path
from typing import Tuple
from uuid import uuid4

import torch

from ..video import muxable_audio
from .patches.base import get_patch_from_file
from .render import get_output_class


@torch.inference_mode()
def generate_audiovisal_from_patch(audio_file: str, model_file: str, patch_file: str, patch_name: str, renderer: str,
                                   renderer_kwargs: dict, fps: float, out_size: Tuple[int], resize_strategy: str,
                                   resize_layer: int):
    """generate.py:16-54 (the function keeps the reference's spelling): instantiate the patch class found in
    ``patch_file``, run its four stages and hand the synthesizer inputs to the chosen renderer.
    Returns (video, (audio, sr))."""
    PatchCls = get_patch_from_file(patch_file, patch_name)
    patch = PatchCls(model_file, audio_file, fps=fps, offset=0, duration=-1, output_size=out_size,
                     resize_strategy=resize_strategy, resize_layer=resize_layer)
    patch.process_audio()
    latents = patch.mapper(**patch.process_mapper_inputs())
    inputs = patch.process_synthesizer_inputs(latents)

    def postprocess(video):
        return patch.force_output_size(patch.process_outputs(video))

    # the renderers may pack u8 frames inside the synthesis call only when this closure changes nothing: the patch keeps
    # the base classes' identity process_outputs (force_output_size is a no-op at the rendered size, which the
    # renderer checks itself)
    from .patches.base.stylegan2 import StyleGAN2Patch
    po = getattr(type(patch), "process_outputs", None)
    postprocess.identity_at_native_size = po is None or po is getattr(StyleGAN2Patch, "process_outputs")

    kwargs = dict(renderer_kwargs)
    if renderer == "ffmpeg":  # the writer muxes the clip's audio back in (anything the ffmpeg executable decodes)
        kwargs.update(fps=patch.fps,
                      audio_file=muxable_audio(patch.audio_file))
    video = get_output_class(renderer)(**kwargs)(patch.synthesizer, inputs, postprocess)
    return video, (patch.audio, patch.sr)


# CLI flags of generate.py:57-98: (name, argparse keyword arguments)
_FLAGS = [
    ("audio_file", dict(required=True, type=str, help="Path to audio file")),
    ("model_file", dict(required=True, type=str, help="Checkpoint (rosinality / NVIDIA state dict), or 'None' for random init")),
    ("patch_file", dict(default="maua/audiovisual/patches/examples/stylegan2.py", type=str,
                        help="Python file defining the MauaPatch that modulates the generator's inputs")),
    ("patch_name", dict(default=None, type=str, help="Patch class to use when the file defines several")),
    ("renderer", dict(default="ffmpeg", type=str, help="'ffmpeg' (video file) or 'memmap' (uint8 array)")),
    ("ffmpeg_preset", dict(default="fast", type=str, help="x264 preset of the ffmpeg renderer")),
    ("fps", dict(default=24, type=float, help="Frames per second of the output video")),
    ("out_size", dict(default="1024,1024", type=str, help="Output width,height")),
    ("resize_strategy", dict(default="pad-zero", type=str, help="Feature-space resize: 'stretch' or 'pad-<how>-<where>'")),
    ("resize_layer", dict(default=0, choices=list(range(18)), type=int, help="Layer at which the features are resized")),
    ("out_dir", dict(default="./output/", type=str, help="Directory of the output video")),
    ("unique", dict(action="store_true", help="Append a short random id to the file name")),
]


def main(argv=None):
    torch.set_num_threads(min(torch.get_num_threads(), 8))   # host side = small tensors (see _lib.host_threads)
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for name, kw in _FLAGS:
        parser.add_argument(f"--{name}", **kw)
    args = parser.parse_args(argv)
    stem = Path(args.model_file.replace("/network-snapshot", "")).stem
    size_tag = args.out_size.replace(",", "x")
    output_file = f"{args.out_dir}/{Path(args.audio_file).stem}_{stem}_{args.resize_strategy}_{size_tag}.mp4"
    if args.unique:
        output_file = output_file[:-4] + f"-{str(uuid4())[:6]}.mp4"
    renderer_kwargs = {}
    if args.renderer == "ffmpeg":
        renderer_kwargs = dict(output_file=output_file, ffmpeg_preset=args.ffmpeg_preset)
    video, _ = generate_audiovisal_from_patch(
        audio_file=args.audio_file, model_file=args.model_file, patch_file=args.patch_file, patch_name=args.patch_name,
        renderer=args.renderer, renderer_kwargs=renderer_kwargs, fps=args.fps,
        out_size=tuple(int(v) for v in args.out_size.split(",")), resize_strategy=args.resize_strategy,
        resize_layer=args.resize_layer)
    return video


if __name__ == "__main__":
    main()
