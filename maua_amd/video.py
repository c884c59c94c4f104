"""Frame sink: producer -> queue -> writer thread -> ffmpeg subprocess (rawvideo rgb24 on stdin), like the
reference's maua/ops/video.py:15-128 (VideoWriter / WriteWorker).  Frames arrive already packed as uint8 HWC on the
device (maua_pack_rgb8), so the writer thread only does the D2H copy and the pipe write; float [B,C,H,W] frames are
packed on the device first (maua_tensor2bytes with the writer's value_range - the reference's worker runs
ops/io.py:47-70 tensor2bytes on the host for every frame, ops/video.py:66).
When no ``ffmpeg`` binary is on PATH the frames go to ``<output>.rgb24`` (raw) + ``<output>.json`` (geometry) so a
render can still be inspected / encoded elsewhere."""
import json
import queue
import shutil
import subprocess
import threading
from pathlib import Path

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def tensor2bytes_device(tensor, value_range=(0, 1)):
    """ops/io.py:47-70 as one launch: float [B,C,H,W] (or [C,H,W]) -> uint8 [B,H,W,C] on the device,
    round_half_even((clamp(x, mn, mx) - mn) / (mx - mn) * 255) in the reference's operation order."""
    x = L.dev_tensor(tensor, torch.float32)
    if x.ndim == 3:
        x = x[None]
    if x.ndim != 4:
        raise ValueError("tensor2bytes expects a [B,C,H,W] or [C,H,W] tensor")
    mn, mx = (float(v) for v in value_range)
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=torch.uint8, device=x.device)
    L.check(L.lib().maua_tensor2bytes(L.ctx(x.device), L.ptr(x), L.ptr(out), b, c, h, w, C.c_double(mn), C.c_double(mx)))
    return out


def tensor2bytes(tensor, value_range=(0, 1)):
    """ops/io.py:47-70: a [1,C,H,W] image tensor as HWC uint8 bytes (e.g. for ffmpeg's stdin)."""
    out = tensor2bytes_device(tensor, value_range)
    if out.shape[0] != 1:
        raise ValueError("tensor2bytes converts one [1,C,H,W] image (the reference squeezes dim 0)")
    return out[0].cpu().numpy().tobytes()


def muxable_audio(audio_file):
    """The clip's audio file if the ``ffmpeg`` executable can be handed it for muxing (it decodes wav / mp3 / flac / ... itself,
    as the reference's writer relies on, ops/video.py:44-52), None for the tensor dumps this package also accepts as "audio"
    (.npy / .npz / .pt) and for no file at all."""
    if not audio_file:
        return None
    ext = Path(str(audio_file)).suffix.lower()
    return None if ext in (".npy", ".npz", ".pt", ".pth") else str(audio_file)


class VideoWriter:
    def __init__(self, output_file, output_size, fps, audio_file=None, audio_offset=0, audio_duration=None,
                 ffmpeg_preset="slow", debug=False, value_range=(0, 1), max_queue=64):
        self.output_file, self.output_size, self.fps = str(output_file), tuple(output_size), fps
        self.audio_file, self.audio_offset, self.audio_duration = audio_file, audio_offset, audio_duration
        self.ffmpeg_preset, self.debug, self.value_range = ffmpeg_preset, debug, tuple(value_range)
        self.q = queue.Queue(maxsize=max_queue)
        self.frames_written = 0
        self._err = None

    def __enter__(self):
        w, h = self.output_size
        Path(self.output_file).parent.mkdir(parents=True, exist_ok=True)
        self.proc, self.raw = None, None
        if shutil.which("ffmpeg"):
            cmd = ["ffmpeg", "-y", "-loglevel", "info" if self.debug else "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{w}x{h}", "-r",
                   str(self.fps), "-i", "-"]
            if self.audio_file:
                cmd += ["-ss", str(self.audio_offset)] + (["-t", str(self.audio_duration)] if self.audio_duration else [])
                cmd += ["-i", self.audio_file]
            cmd += ["-c:v", "libx264", "-preset", self.ffmpeg_preset, "-pix_fmt", "yuv420p", self.output_file]
            self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE)
            self.sink = self.proc.stdin
        else:
            self.raw = open(self.output_file + ".rgb24", "wb")
            self.sink = self.raw
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        return self

    def _run(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    break
                if isinstance(item, torch.Tensor):
                    item = item.cpu().numpy()
                self.sink.write(np.ascontiguousarray(item).tobytes())
                self.frames_written += item.shape[0] if item.ndim == 4 else 1
        except Exception as e:  # surface writer failures to the producer instead of truncating silently
            self._err = e

    def write(self, tensor):
        """Packed uint8 [H,W,3] / [B,H,W,3] frames (device or host) go to the sink as they are; a float [B,C,H,W]
        tensor (what the reference's VideoWriter.write takes, ops/video.py:112-115) is converted on the device with
        the writer's value_range first."""
        if self._err:
            raise self._err
        if isinstance(tensor, np.ndarray):
            tensor = torch.from_numpy(tensor)
        if tensor.dtype != torch.uint8:
            if not tensor.is_floating_point():
                raise TypeError("VideoWriter.write expects packed uint8 HWC frames or a float [B,C,H,W] tensor")
            tensor = tensor2bytes_device(tensor, self.value_range)
        self.q.put(tensor)

    def __exit__(self, *exc):
        self.q.put(None)
        self.thread.join()
        if self.proc is not None:
            self.proc.stdin.close()
            self.proc.wait()
        if self.raw is not None:
            self.raw.close()
            w, h = self.output_size
            Path(self.output_file + ".json").write_text(json.dumps(
                {"width": w, "height": h, "fps": self.fps, "pix_fmt": "rgb24", "frames": self.frames_written,
                 "audio_file": self.audio_file}))
        if self._err:
            raise self._err
        return False


def write_video(tensor, output_file, fps=24, audio_file=None, audio_offset=0, audio_duration=None, ffmpeg_preset="slow",
                debug=False, value_range=(0, 1)):
    """ops/video.py:131-155: a [T,C,H,W] sequence (tensor or ndarray) to one video file.  The reference converts frame
    by frame on the host; here the sequence is packed on the device in slabs and handed to the writer thread."""
    if isinstance(tensor, np.ndarray):
        tensor = torch.from_numpy(tensor.copy())
    _, _, h, w = tensor[[0]].shape
    with VideoWriter(output_file, (w, h), fps, audio_file, audio_offset, audio_duration, ffmpeg_preset, debug,
                     value_range) as video:
        for i in range(0, tensor.shape[0], 32):
            video.write(tensor[i:i + 32])
