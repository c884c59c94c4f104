"""Audio decode (host plumbing) + resampling (device).  The reference decodes with torchaudio / librosa and resamples with
torchaudio.functional.resample (sample.py:16-32, audioreactive/audio.py:15-48) — un-vendored, so "parity unpinned" (SURVEY A1).
Here: PCM/float WAV via scipy, ``.npy`` / ``.pt`` float arrays, mono mean, slice on the host; the resampler is torchaudio's
published windowed-sinc polyphase algorithm as one GEMM on the HIP device (``audio.resample_sinc``; no host fallback).  MP3 /
other compressed formats need a decoder the image does not have."""
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
    raise NotImplementedError(f"{p.suffix}: only .wav / .npy / .pt audio can be decoded without ffmpeg/torchaudio")


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
