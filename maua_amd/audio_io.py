"""Audio decode (host plumbing) + resampling (device).  The reference decodes with torchaudio / librosa and resamples with
torchaudio.functional.resample (sample.py:16-32, audioreactive/audio.py:15-48) — un-vendored, so "parity unpinned" (SURVEY A1).
Here: PCM/float WAV via scipy, ``.npy`` / ``.pt`` float arrays, mono mean, slice on the host; the resampler is torchaudio's
published windowed-sinc polyphase algorithm as one GEMM on the HIP device (``audio.resample_sinc``; no host fallback).  MP3 /
other compressed formats are decoded by the ``ffmpeg`` executable when one is on PATH (the decoder behind torchaudio.load
too, and the program the reference's VideoWriter already requires, ops/video.py:15-63); the build image has none, so that
route is exercised with a stand-in executable only."""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import torch


def read_audio(path):
    p = Path(path)
    if p.suffix == ".npy":
        a = np.load(p)
        return torch.from_numpy(np.asarray(a, dtype=np.float32)), None
    if p.suffix in (".pt", ".pth"):
        d = torch.load(p)
        if isinstance(d, dict):
            return d["audio"].float(), int(d["sr"])
        return d.float(), None
    if p.suffix.lower() == ".wav":
        from scipy.io import wavfile
        sr, a = wavfile.read(p)
        if a.dtype.kind == "i":
            a = a.astype(np.float32) / float(np.iinfo(a.dtype).max + 1)
        elif a.dtype.kind == "u":
            a = (a.astype(np.float32) - 128.0) / 128.0
        a = torch.from_numpy(np.asarray(a, dtype=np.float32))
        return (a.T if a.ndim == 2 else a), int(sr)
    return _read_with_ffmpeg(p)


def _read_with_ffmpeg(p):
    """Compressed audio (configs[0] names an .mp3; sample.py:17 `torchaudio.load`): ffprobe for rate / channel count, ffmpeg
    for interleaved f32 PCM on a pipe.  -> ([channels, n] float32, sr)."""
    ffmpeg, ffprobe = shutil.which("ffmpeg"), shutil.which("ffprobe")
    if ffmpeg is None or ffprobe is None:
        raise NotImplementedError(f"{p.suffix}: .wav / .npy / .pt are read directly; any other format needs the ffmpeg and "
                                  "ffprobe executables on PATH (none found)")
    if not p.is_file():
        raise FileNotFoundError(str(p))
    probe = subprocess.run([ffprobe, "-v", "error", "-select_streams", "a:0", "-show_entries", "stream=sample_rate,channels",
                            "-of", "json", str(p)], capture_output=True, check=False)
    if probe.returncode != 0:
        raise RuntimeError(f"ffprobe failed on {p}: {probe.stderr.decode(errors='replace').strip()}")
    st = json.loads(probe.stdout.decode())["streams"]
    if not st:
        raise RuntimeError(f"{p}: no audio stream")
    sr, ch = int(st[0]["sample_rate"]), int(st[0]["channels"])
    dec = subprocess.run([ffmpeg, "-v", "error", "-nostdin", "-i", str(p), "-map", "a:0", "-f", "f32le", "-acodec", "pcm_f32le",
                          "-ac", str(ch), "-ar", str(sr), "-"], capture_output=True, check=False)
    if dec.returncode != 0:
        raise RuntimeError(f"ffmpeg failed on {p}: {dec.stderr.decode(errors='replace').strip()}")
    a = np.frombuffer(dec.stdout, dtype="<f4")
    if a.size % ch:
        raise RuntimeError(f"{p}: {a.size} samples do not divide into {ch} channels")
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr


def load_audio(audio_file, offset=0, duration=None, fps=None, sr=None):
    """selfsupervised/sample.py:16-32 semantics: mono mean, [offset, offset+duration) seconds, resample to
    1024*fps when fps is given.  Returns (float32 mono tensor, sample rate)."""
    audio, file_sr = read_audio(audio_file)
    file_sr = file_sr or sr or (1024 * fps if fps else 44100)
    if audio.ndim == 2:
        audio = audio.mean(0)
    start = int(offset * file_sr)
    audio = audio[start: start + int(duration * file_sr)] if duration not in (None, -1) else audio[start:]
    if fps is not None:
        new_sr = int(1024 * fps)
        if new_sr != file_sr:
            from .audio import resample_sinc
            audio = resample_sinc(audio, file_sr, new_sr).cpu()
        file_sr = new_sr
    return audio.contiguous(), file_sr
