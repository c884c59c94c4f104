"""Frame sink: producer -> queue -> writer thread -> ffmpeg subprocess (rawvideo rgb24 on stdin), like the
reference's maua/ops/video.py:15-128 (VideoWriter / WriteWorker).  Frames arrive already packed as uint8 HWC on the
device (maua_pack_rgb8), so the writer thread only does the D2H copy and the pipe write.
When no ``ffmpeg`` binary is on PATH the frames go to ``<output>.rgb24`` (raw) + ``<output>.json`` (geometry) so a
render can still be inspected / encoded elsewhere."""
import json
import queue
import shutil
import subprocess
import threading
from pathlib import Path

import numpy as np
import torch


class VideoWriter:
    def __init__(self, output_file, output_size, fps, audio_file=None, audio_offset=0, audio_duration=None,
                 ffmpeg_preset="medium", max_queue=64):
        self.output_file, self.output_size, self.fps = str(output_file), tuple(output_size), fps
        self.audio_file, self.audio_offset, self.audio_duration = audio_file, audio_offset, audio_duration
        self.ffmpeg_preset = ffmpeg_preset
        self.q = queue.Queue(maxsize=max_queue)
        self.frames_written = 0
        self._err = None

    def __enter__(self):
        w, h = self.output_size
        Path(self.output_file).parent.mkdir(parents=True, exist_ok=True)
        self.proc, self.raw = None, None
        if shutil.which("ffmpeg"):
            cmd = ["ffmpeg", "-y", "-loglevel", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{w}x{h}", "-r",
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

    def write(self, frames_u8):
        """uint8 [H,W,3] or [B,H,W,3] (device or host)."""
        if self._err:
            raise self._err
        if frames_u8.dtype != torch.uint8:
            raise TypeError("VideoWriter.write expects packed uint8 HWC frames")
        self.q.put(frames_u8)

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
