"""FFMPEG renderer (drop-in for maua/audiovisual/render/ffmpeg.py:21-75): batch loop -> synthesizer -> packed u8
frames -> VideoWriter.  The (x+1)/2 -> clamp -> u8 HWC conversion the reference does per frame on its writer thread
(ops/io.py:47-70) happens inside the same C-ABI call as the synthesis (maua_synth_render_rgb8)."""
import torch

from ...pipeline import frame_range
from ... import _lib as L
from ...video import VideoWriter
from . import Renderer, batch_inputs, n_frames_of


class FFMPEG(Renderer):
    def __init__(self, output_file, fps=24, audio_file=None, audio_offset=0, audio_duration=None, ffmpeg_preset="medium",
                 batch_size=16):
        super().__init__()
        self.output_file, self.fps, self.ffmpeg_preset = output_file, fps, ffmpeg_preset
        self.audio_file, self.audio_offset, self.audio_duration = audio_file, audio_offset, audio_duration
        self.batch_size = batch_size

    def __call__(self, synthesizer, inputs, postprocess=None, fp16=True, rank=0, world=1):
        T = n_frames_of(inputs)
        lo, hi = frame_range(T, rank, world)
        W, H = synthesizer.output_size
        with VideoWriter(self.output_file if world == 1 else f"{self.output_file}.part{rank}", (W, H), self.fps,
                         self.audio_file, self.audio_offset, self.audio_duration, self.ffmpeg_preset) as video:
            rh, rw = synthesizer.G_synth.output_hw if hasattr(synthesizer, "G_synth") else (H, W)
            # fp16: 16-bit compute is a property of the synthesizer's dtype here (bf16 by default, torch.float16 = IEEE half); see
            # MauaGenerator.render for what the flag does in the reference and why inputs are not rounded to float16
            from ...stylegan2 import _warn_fp16_flag
            _warn_fp16_flag(synthesizer, fp16)
            for i in range(lo, hi, self.batch_size):
                b = min(self.batch_size, hi - i)
                u8 = torch.empty((b, H, W, 3), dtype=torch.uint8, device="cuda")
                # the u8 frame is packed inside the synthesis call only when the caller's postprocess is known to be
                # the identity at this size; the reference always runs postprocess(frame_batch) first (:72-73)
                identity = postprocess is None or getattr(postprocess, "identity_at_native_size", False)
                if (rh, rw) == (H, W) and identity:
                    synthesizer(**batch_inputs(inputs, i, b), rgb8_out=u8)
                else:
                    # output_size was rounded to the resize layer's multiple: render/ffmpeg.py:72-73 — frames in
                    # [0, 1] go through the patch's postprocess (force_output_size resamples), then the u8 pack
                    frames = synthesizer(**batch_inputs(inputs, i, b)).add(1).div(2)
                    frames = postprocess(frames) if postprocess is not None else frames
                    frames = frames.mul(2).sub(1).contiguous()
                    if tuple(frames.shape[-2:]) != (H, W):
                        raise ValueError(f"postprocess returned {tuple(frames.shape[-2:])}, the writer expects {(H, W)}")
                    L.check(L.lib().maua_pack_rgb8(L.ctx(frames.device), L.ptr(frames), L.ptr(u8), b, H, W))
                video.write(u8)
        return self.output_file
