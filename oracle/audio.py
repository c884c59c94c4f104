"""Oracle restatement of the audio feature chain (test infrastructure only).

Follows the reference's pure-torch librosa re-implementation:
  maua/audiovisual/audioreactive/selfsupervised/features/rosa/spectral.py (stft :10-21, istft :24-32,
  spectrogram :59-62, melspectrogram :65-70, mel_frequencies :73-78, mel :81-110, magphase :113-117,
  softmask :120-142, hpss :145-161), rosa/convert.py (power_to_db :7-12, hz_to_mel :15-40, mel_to_hz :43-66),
  rosa/beat.py (onset_strength :10-23), rosa/helpers.py (sync_agg :4-21),
  features/processing.py (gaussian_filter :11-49, normalize :53-56, standardize :59-62, median_filter2d :75-85),
  features/audio.py (harmonic :13-17, percussive :20-24, onsets :27-28, rms :31-37),
  selfsupervised/mir.py (salience_weighted :13-21).
STFT framing / overlap-add are written out explicitly (numpy-style) instead of calling torch.stft, so that the
frame <-> sample mapping the HIP kernels must reproduce is stated once, here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import quantile as Q

N_FFT, HOP = 2048, 1024


def hann(n=N_FFT):
    """torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi k / n)."""
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * np.pi * k / n)).float()


def frame_signal(y, n_fft=N_FFT, hop=HOP):
    """center=True, reflect padding of n_fft//2 each side; frame f covers samples [f*hop - n_fft/2, f*hop + n_fft/2)."""
    yp = F.pad(y[None, None], (n_fft // 2, n_fft // 2), mode="reflect")[0, 0]
    n_frames = 1 + (yp.numel() - n_fft) // hop
    return yp.unfold(0, n_fft, hop)[:n_frames]  # [frames, n_fft]


def stft(y, n_fft=N_FFT, hop=HOP):
    """spectral.py:10-21 -> complex64 [n_fft//2+1, 1 + len(y)//hop], unnormalised, periodic Hann."""
    fr = frame_signal(y, n_fft, hop) * hann(n_fft)
    return torch.fft.rfft(fr, dim=1).T.contiguous()


def istft(spec, n_fft=N_FFT, hop=HOP, length=None):
    """spectral.py:24-32 (torch.istft, center=True): windowed inverse frames overlap-added, divided by the
    overlap-added squared window, centre padding removed, cut/padded to `length`."""
    w = hann(n_fft)
    frames = torch.fft.irfft(spec.T, n=n_fft, dim=1) * w  # [frames, n_fft]
    n_frames = frames.shape[0]
    total = n_fft + hop * (n_frames - 1)
    y = torch.zeros(total)
    env = torch.zeros(total)
    w2 = w * w
    for f in range(n_frames):
        y[f * hop: f * hop + n_fft] += frames[f]
        env[f * hop: f * hop + n_fft] += w2
    start = n_fft // 2
    end = total - n_fft // 2 if length is None else start + length
    y, env = y[start:end], env[start:end]
    y = y / env
    if length is not None and y.numel() < length:
        y = F.pad(y, (0, length - y.numel()))
    return y


def spectrogram(y, power=1, hop=HOP, keep_last=False):
    """spectral.py:59-62 — drops the LAST stft column (keep_last: librosa's own frame count, the classic onsets)."""
    D = stft(y, hop=hop)
    return (D if keep_last else D[:, :-1]).abs() ** power


def hz_to_mel(f):
    """convert.py:15-40 (Slaney)."""
    f = torch.as_tensor(f, dtype=torch.float32)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        t = f >= min_log_hz
        mels[t] = min_log_mel + torch.log(f[t] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + torch.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(mels):
    """convert.py:43-66."""
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        t = mels >= min_log_mel
        freqs[t] = min_log_hz * torch.exp(logstep * (mels[t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * torch.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0):
    """spectral.py:73-78."""
    return mel_to_hz(torch.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_basis(sr, n_fft=N_FFT, n_mels=128, fmin=0.0, fmax=None):
    """spectral.py:81-110 — Slaney-normalised triangular filters [n_mels, n_fft//2+1]."""
    if fmax is None:
        fmax = float(sr) / 2
    fftfreqs = torch.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = torch.diff(mel_f)
    ramps = mel_f.reshape(-1, 1) - fftfreqs
    w = torch.zeros(n_mels, 1 + n_fft // 2)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = torch.maximum(torch.zeros(()), torch.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def melspectrogram(y, sr, power=2.0, fmax=None, hop=HOP, keep_last=False):
    """spectral.py:65-70."""
    return mel_basis(sr, fmax=fmax) @ spectrogram(y, power=power, hop=hop, keep_last=keep_last)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """convert.py:7-12 (ref_value = 1)."""
    log_spec = 10.0 * torch.log10(torch.clamp(S, min=amin))
    log_spec = log_spec - 10.0 * torch.log10(torch.tensor(1.0))
    return torch.maximum(log_spec, log_spec.max() - top_db)


def onset_strength(y, sr, n_fft=N_FFT, hop=HOP, aggregate="mean", fmax=11025.0, keep_last=False):
    """beat.py:10-23 — mean (or, for plp :44, torch.median = lower middle) over mels of the rectified lag-1 dB
    difference, left-padded by 1 + n_fft // (2*hop) zeros and cropped to the spectrogram length."""
    S = power_to_db(melspectrogram(y, sr, fmax=fmax, hop=hop, keep_last=keep_last).abs())
    d = torch.clamp(S[:, 1:] - S[:, :-1], min=0)
    d = d.mean(0) if aggregate == "mean" else torch.median(d, dim=0).values
    pad_width = 1 + n_fft // (2 * hop)
    return F.pad(d, (pad_width, 0))[: S.shape[1]]


def median_filter2d(x, k, p):
    """processing.py:75-85 — x [1,1,H,W]; reflect pad p = (l, r, t, b); median over a k[0] x k[1] window
    (torch.median: lower of the two middle values for even counts; all windows here are odd)."""
    x = F.pad(x, p, mode="reflect")
    x = x.unfold(2, k[0], 1).unfold(3, k[1], 1)
    return x.contiguous().view(x.size()[:4] + (-1,)).median(dim=-1)[0]


def softmask(X, X_ref, power=2.0, split_zeros=False):
    """spectral.py:120-142."""
    Z = torch.maximum(X, X_ref)
    bad = Z < torch.finfo(torch.float32).tiny
    Z = torch.where(bad, torch.ones_like(Z), Z)
    mask = (X / Z) ** power
    ref = (X_ref / Z) ** power
    mask = torch.where(bad, torch.full_like(mask, 0.5 if split_zeros else 0.0), mask / (mask + ref))
    return mask


def hpss(D, ks=31, power=2.0, margin=1.0):
    """spectral.py:145-161 on a complex STFT: (harmonic, percussive) complex spectra."""
    S = D.abs()
    phase = torch.exp(1.0j * torch.angle(D))
    harm = median_filter2d(S[None, None], (1, ks), (ks // 2, ks // 2, 0, 0))[0, 0]
    perc = median_filter2d(S[None, None], (ks, 1), (0, 0, ks // 2, ks // 2))[0, 0]
    split = margin == 1
    mh = softmask(harm, perc * margin, power, split)
    mp = softmask(perc, harm * margin, power, split)
    return (S * mh) * phase, (S * mp) * phase


def harmonic(audio, margin=8.0, hop=HOP):
    """audio.py:13-17"""
    return istft(hpss(stft(audio, hop=hop), margin=margin)[0], hop=hop, length=len(audio))


def percussive(audio, margin=8.0, hop=HOP):
    """audio.py:20-24"""
    return istft(hpss(stft(audio, hop=hop), margin=margin)[1], hop=hop, length=len(audio))


def classic_onsets(audio, sr, prepercussive=4, hop=512):
    """audioreactive/mir.py:16-61 with type="rosa": librosa's percussive separation and onset_strength (un-vendored;
    published algorithm, librosa's default framing n_fft 2048 / hop 512 / fmax sr/2), then percentile_clip(95)."""
    from .signal import percentile_clip
    a = percussive(audio, 8.0, hop) if prepercussive else audio  # mir.py:29-30: a flag; audio.py:91 margin=8
    return percentile_clip(onset_strength(a, sr, hop=hop, fmax=sr / 2, keep_last=True), 95).squeeze()


def normalize(x):
    """processing.py:53-56"""
    x = x - x.min()
    return x / (x.max() + 1e-8)


def onsets(audio, sr):
    """audio.py:27-28 -> [T, 1]"""
    return normalize(onset_strength(percussive(audio), sr).unsqueeze(-1))


def rms(y, frame_length=N_FFT, hop=HOP):
    """audio.py:31-37 -> [T, 1] (drops the last frame)."""
    fr = frame_signal(y, frame_length, hop)[:-1]
    return torch.sqrt(torch.mean(fr.abs() ** 2, dim=1)).unsqueeze(-1)


def gaussian_filter(x, sigma, mode="circular", causal=None, classic=False):
    """processing.py:11-49 (selfsupervised; ignores `causal`) and, with classic=True, signal.py:108-157
    (right half of the kernel multiplied by `causal` if it is a Python float, by 0 for any other non-None value).
    Filters along dim 0 independently per element."""
    dim = x.ndim
    n = x.shape[0]
    while x.ndim < 3:
        x = x[:, None]
    radius = min(int(sigma * 4), 3 * len(x))
    ch = x.shape[1]
    k = torch.arange(-radius, radius + 1, dtype=torch.float32)
    k = torch.exp(-0.5 / sigma ** 2 * k ** 2)
    if classic and causal is not None:
        k[radius + 1:] *= causal if isinstance(causal, float) else 0
    k = k / k.sum()
    k = k.view(1, 1, -1).repeat(ch, 1, 1)
    if dim == 4:
        t, c, h, w = x.shape
        x = x.reshape(t, c, h * w)
    x = x.transpose(0, 2)
    if radius > n:
        x = F.pad(x, (n, n), mode=mode)
        x = F.pad(x, (radius - n, radius - n), mode="replicate")
    else:
        x = F.pad(x, (radius, radius), mode=mode)
    x = F.conv1d(x, k, groups=ch).transpose(0, 2)
    if dim == 4:
        x = x.reshape(t, c, h, w)
    if x.ndim > dim:
        x = x.squeeze()
    return x


def standardize(x):
    """processing.py:59-62 (C++ midpoint quantile with float32 q)."""
    return normalize(torch.clamp(x, Q.quantile(x, 0.25), Q.quantile(x, 0.75) + 1e-10))


def salience_weighted(env, short_sigma=5, long_sigma=80):
    """selfsupervised/mir.py:13-21"""
    if env.dim() > 1:
        env = env.squeeze(1)
    short = gaussian_filter(env, short_sigma, mode="reflect")
    long = gaussian_filter(env, long_sigma, mode="reflect")
    w = (short / long) ** 2 * env
    return w.unsqueeze(1) if w.dim() < 2 else w


# ----------------------------------------------------------------------------- SURVEY 8(f) N3 (first batch)
def emphasize(envs, strength, percentile):
    """processing.py:133-139"""
    mn = envs.min(dim=0).values
    x = envs - mn
    mx = x.max(dim=0).values
    x = x / mx
    x = x * (1 + torch.tanh(strength * (x - torch.quantile(x, q=percentile / 100, dim=0))))
    return (x * mx) + mn


def clamp_upper_percentile(signal, percentile):
    """processing.py:125-126"""
    return torch.clamp(signal, max=torch.quantile(signal, percentile / 100, dim=0))


def clamp_lower_percentile(signal, percentile):
    """processing.py:129-130"""
    return torch.clamp(signal, min=torch.quantile(signal, percentile / 100, dim=0))


def clamp_peaks_percentile(signal, percent):
    """processing.py:102-122: upper clamp of every column at the quantile of its local peaks."""
    from .signal import peak_mask
    if signal.ndim < 2:
        signal = signal.unsqueeze(1)
    return torch.stack([c.clamp(max=torch.quantile(c[peak_mask(c)], percent / 100)) for c in signal.unbind(1)], dim=1)


def sosfilt(sos, x):
    """audioreactive/audio.py:96-112 run scipy.signal.sosfilt (a listed dependency of the reference, present in this image): the
    oracle of the device recurrence is scipy itself, float64."""
    from scipy import signal
    return signal.sosfilt(np.asarray(sos, dtype=np.float64), np.asarray(x, dtype=np.float64))


def butter_pass(audio, sr, cutoff, kind, db_per_octave=12):
    """audioreactive/audio.py:96-112 low_pass / high_pass / band_pass as written there."""
    from scipy import signal
    return signal.sosfilt(signal.butter(db_per_octave, cutoff, kind, fs=sr, output="sos"), audio)


def _biquad(x, b, a):
    """torchaudio.functional.biquad (un-vendored: lfilter with clamp=True as published; parity unpinned).  Computed in float64 and
    rounded once - torchaudio's own recurrence runs in the waveform's float32."""
    from scipy import signal
    y = signal.lfilter(np.asarray(b, dtype=np.float64) / a[0], np.asarray(a, dtype=np.float64) / a[0], x.double().numpy())
    return torch.from_numpy(y).clamp(-1, 1).float()


def low_pass(audio, sr, fmax=200, q=0.707):
    """processing.py:142-143 lowpass_biquad (RBJ cookbook)"""
    w0 = 2 * math.pi * fmax / sr
    al, cs = math.sin(w0) / 2 / q, math.cos(w0)
    return _biquad(audio, [(1 - cs) / 2, 1 - cs, (1 - cs) / 2], [1 + al, -2 * cs, 1 - al])


def high_pass(audio, sr, fmin=4000, q=0.707):
    """processing.py:150-151 highpass_biquad"""
    w0 = 2 * math.pi * fmin / sr
    al, cs = math.sin(w0) / 2 / q, math.cos(w0)
    return _biquad(audio, [(1 + cs) / 2, -1 - cs, (1 + cs) / 2], [1 + al, -2 * cs, 1 - al])


def mid_pass(audio, sr, fmin=200, fmax=4000):
    """processing.py:146-147 (high pass at fmax, then low pass at fmin - as written)"""
    return low_pass(high_pass(audio, sr, fmax), sr, fmin)


def contrast_enhance(audio, strength=75):
    """processing.py:154-155 torchaudio.functional.contrast as published"""
    t1 = audio * (math.pi / 2)
    return torch.sin(t1 + (strength / 750.0) * torch.sin(t1 * 4))


def drop_strength(audio):
    """features/audio.py:40-41"""
    return emphasize(gaussian_filter(rms(audio), 10), strength=10, percentile=50).unsqueeze(1)


def dct(x, norm=None):
    """rosa/spectral.py:35-56 (Makhoul's FFT form of the DCT-II, as the reference computes it)."""
    shape = x.shape
    N = shape[-1]
    x = x.contiguous().view(-1, N)
    v = torch.cat([x[:, ::2], x[:, 1::2].flip([1])], dim=1)
    Vc = torch.view_as_real(torch.fft.fft(v, dim=1))
    k = -torch.arange(N, dtype=x.dtype)[None, :] * np.pi / (2 * N)
    V = Vc[:, :, 0] * torch.cos(k) - Vc[:, :, 1] * torch.sin(k)
    if norm == "ortho":
        V[:, 0] /= np.sqrt(N) * 2
        V[:, 1:] /= np.sqrt(N / 2) * 2
    return 2 * V.view(*shape)


def mfcc(y, sr, n_mfcc=20):
    """features/audio.py:65-70"""
    S = power_to_db(melspectrogram(y, sr))
    return dct(S.permute(1, 0), norm="ortho").permute(1, 0)[:n_mfcc].T


def tonnetz_from_chroma(chroma):
    """features/audio.py:50-62 after the chromagram"""
    dim_map = torch.linspace(0, 12, chroma.shape[0])
    scale = torch.tensor([7.0 / 6, 7.0 / 6, 3.0 / 2, 3.0 / 2, 2.0 / 3, 2.0 / 3])
    V = scale.reshape(-1, 1) * dim_map
    V[::2] -= 0.5
    R = torch.tensor([1, 1, 1, 1, 0.5, 0.5])
    phi = R[:, None] * torch.cos(torch.pi * V)
    return (phi @ (chroma / chroma.norm(p=1, dim=0))).T


def spectral_flatness(y, amin=1e-10, power=2.0):
    """features/audio.py:118-126"""
    S = spectrogram(y, power=1.0)
    S_thresh = torch.maximum(torch.tensor(amin), S ** power)
    gmean = torch.exp(torch.mean(torch.log(S_thresh), axis=0))
    return (gmean / torch.mean(S_thresh, axis=0)).unsqueeze(-1)


def spectral_contrast(y, sr, fmin=200.0, n_bands=6, quantile=0.02, linear=False):
    """features/audio.py:76-115"""
    S = spectrogram(y, power=1)
    freq = torch.linspace(0, float(sr) / 2, int(1 + N_FFT // 2))
    octa = torch.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** torch.arange(0, n_bands + 1))
    valley = torch.zeros((n_bands + 1, S.shape[1]))
    peak = torch.zeros_like(valley)
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        band = torch.logical_and(freq >= f_low, freq <= f_high)
        idx = band.flatten().nonzero()
        if k > 0:
            band[idx[0] - 1] = True
        if k == n_bands:
            band[idx[-1] + 1:] = True
        sub = S[band]
        if k < n_bands:
            sub = sub[:-1]
        n = int(max(torch.round(quantile * torch.sum(band)), torch.ones(())))
        srt = torch.sort(sub, dim=0).values
        valley[k] = torch.mean(srt[:n], dim=0)
        peak[k] = torch.mean(srt[-n:], dim=0)
    if linear:
        return (peak - valley).T
    return (power_to_db(peak) - power_to_db(valley)).T


def plp(y, sr, hop=HOP, win_length=1024, tempo_min=60, tempo_max=180):
    """rosa/beat.py:42-75 predominant local pulse."""
    env = onset_strength(y, sr, aggregate="median")
    W = min(len(env), win_length)
    ft = stft(env, n_fft=W, hop=1)
    rate = sr * 60 / float(hop)
    freqs = torch.linspace(0, float(rate) / 2, int(1 + W // 2))
    if tempo_min is not None:
        ft[freqs < tempo_min] = 0
    if tempo_max is not None:
        ft[freqs > tempo_max] = 0
    mag = torch.log1p(1e6 * torch.abs(ft))
    ft[mag < mag.max(axis=0, keepdims=True).values] = 0
    ft = ft / (torch.finfo(ft.dtype).tiny ** 0.5 + torch.abs(ft.abs().max(axis=0, keepdim=True).values))
    pulse = istft(ft, n_fft=W, hop=1, length=len(env))
    pulse = torch.clamp(pulse, torch.zeros(()), pulse.max())
    return normalize(pulse)


def pulse(audio, sr):
    """features/audio.py:72-73"""
    return plp(percussive(audio), sr).unsqueeze(-1)


def tempo(onset_envelope, sr=22050, hop=1024, max_tempo=240.0, ac_size=120.0, prior_scale=400.0, prior_s=1.0):
    """selfsupervised/mir.py:27-30 -> librosa.beat.tempo (un-vendored, unpinned: published algorithm restated with
    numpy, the autocorrelation by FFT as librosa.autocorrelate does).  Parity unpinned."""
    import numpy as np
    from scipy import stats
    env = np.asarray(onset_envelope, dtype=np.float64).reshape(-1)
    W = int(np.floor(ac_size * sr / hop))
    W = max(2, min(W, 16384))
    padded = np.pad(env, (W // 2, W // 2), mode="linear_ramp", end_values=0)
    padded = np.concatenate([padded, np.zeros(W)])
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(W) / W)
    frames = np.lib.stride_tricks.sliding_window_view(padded, W)[: len(env)] * win
    nfft = 2 * W - 1
    spec = np.fft.rfft(frames, n=nfft, axis=1)
    ac = np.fft.irfft(np.abs(spec) ** 2, n=nfft, axis=1)[:, :W]
    tg = (ac / np.maximum(np.abs(ac).max(1, keepdims=True), np.finfo(np.float32).tiny)).mean(0)
    bpms = np.full(W, np.inf)
    bpms[1:] = 60.0 * sr / (hop * np.arange(1, W))
    logprior = stats.lognorm(loc=0, scale=prior_scale, s=prior_s).logpdf(bpms)
    logprior[: int(np.argmax(bpms < max_tempo))] = -np.inf
    return float(bpms[int(np.argmax(np.log1p(1e6 * tg) + logprior))])


def sinc_resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann",
                  beta=None):
    """torchaudio.functional.resample as the reference calls it (selfsupervised/sample.py:7,29; audio.py:53-59) - torchaudio
    is un-vendored: its published algorithm (functional.py _get_sinc_resample_kernel / _apply_sinc_resample_kernel, index
    arithmetic in float64 = the ``dtype=None`` branch) restated; parity unpinned.  1-D float tensor in, 1-D out."""
    import math
    import torch.nn.functional as F
    x = torch.as_tensor(waveform, dtype=torch.float32).reshape(-1)
    if int(orig_freq) == int(new_freq):
        return x
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    if resampling_method in ("sinc_interp_hann", "sinc_interpolation"):
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    else:
        beta = 14.769656459379492 if beta is None else beta
        window = torch.i0(beta * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(torch.tensor(float(beta), dtype=torch.float64))
    t = t * math.pi
    kernel = (torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)).float()
    xp = F.pad(x[None, None], (width, width + orig))
    out = F.conv1d(xp, kernel, stride=orig).transpose(1, 2).reshape(-1)
    return out[: int(math.ceil(new * x.numel() / orig))]
