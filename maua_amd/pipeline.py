"""Clip-level plumbing shared by bench.py, the entry points and the tests: frame-range sharding and the synthetic
BASELINE clip."""
import math

import torch


def frame_range(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of `rank` (frames are independent after the clip-global pre-pass;
    SURVEY 8e).  Ranges differ by at most one frame and cover [0, n_frames) exactly."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def synthetic_audio(n_samples, sr, seed=1234, fast=False):
    """SURVEY 8(d) synthetic clip: 220 Hz tone + 2 Hz click train + noise, float32 mono.
    Default: everything from ONE generator in float64 (the waveform the committed fixtures were made from - do not change).
    ``fast=True`` (the benchmark clip): the click amplitudes from one generator (``seed``), drawn only where a click sounds, the
    noise floor from two more (``seed + 1``, ``seed + 2``: one half of the clip each, on helper threads), all float32 draws -
    torch's CPU samplers are single-threaded, and the float64 normal draws alone were 0.05 s of the 3600-frame clip's 0.09 s of
    set-up.  Same signal model (tone + 2 Hz clicks of 5 % duty + noise floor), different random numbers."""
    t = torch.arange(n_samples, dtype=torch.float64).div_(sr)
    if not fast:
        g = torch.Generator().manual_seed(seed)
        u = torch.rand(n_samples, generator=g, dtype=torch.float64)
        nz = torch.randn(n_samples, generator=g, dtype=torch.float64)
        # 0.3 sin(2 pi 220 t) + 0.2 (u - 0.5) [2t mod 1 < 0.05] + 0.01 nz, in place (eight 29 MB temporaries cost the
        # 3600-frame clip 0.25 s of page faults)
        u.sub_(0.5).mul_(0.2).mul_(t.mul(2).remainder_(1).lt_(0.05))
        t.mul_(2 * math.pi * 220).sin_().mul_(0.3).add_(u).add_(nz.mul_(0.01))
        return t.float()
    import threading
    half = n_samples // 2
    box = {}

    def noise(k, lo, hi):   # the noise floor in two halves, each from its own generator (seed + 1, seed + 2), each on its own thread
        box[k] = torch.randn(hi - lo, generator=torch.Generator().manual_seed(seed + 1 + k), dtype=torch.float32)
    ths = [threading.Thread(target=noise, args=(0, 0, half)), threading.Thread(target=noise, args=(1, half, n_samples))]
    for th in ths:
        th.start()
    out = t.mul_(2 * math.pi * 220).sin_().mul_(0.3).float()              # (the phase in float64)
    # click train: 2 Hz, each click = the first 5 % of its half-second, amplitudes 0.2 (u - 0.5) drawn only where a click sounds
    period = sr / 2.0
    n_clicks = int(math.ceil(n_samples / period))
    width = int(math.floor(0.05 * period))
    u = torch.rand((n_clicks, max(width, 1)), generator=torch.Generator().manual_seed(seed), dtype=torch.float32).sub_(0.5).mul_(0.2)
    if period == int(period) and n_samples % int(period) == 0 and width > 0:
        out.view(n_clicks, int(period))[:, :width].add_(u[:, :width])
    else:
        for k in range(n_clicks):
            lo = int(math.ceil(k * period))
            hi = min(n_samples, lo + width)
            if hi > lo:
                out[lo:hi].add_(u[k, : hi - lo])
    for th in ths:
        th.join()
    out[:half].add_(box[0], alpha=0.01)
    out[half:].add_(box[1], alpha=0.01)
    return out


def synthetic_clip_latents(n_frames, fps, num_ws, w_dim, seeds="0-60", n_loops=4, fast_audio=True, device_rng=False, keep=None):
    """Audio-reactive latent schedule of the BASELINE clip: onset envelope of the synthetic audio blends two
    spline-loop schedules (latent.py:12-18 single_weighted over latent.py:83-92 spline_loops), sigma=2 smoothing.
    Returns ([T, num_ws, w_dim] f32 on the HIP device, description).
    ``device_rng`` (round 5, what bench.py times): the waveform and the mapper's random init come from the build-owned counter RNG
    ON THE DEVICE (rng.clip_audio: streams 0 / 1 of seed 1234; the mapper's matrices: streams 2^20 .. of seed 0) - two kernels and
    no upload instead of 5.8 M host draws; same signal model and initialisation law, other numbers.  ``keep``: a dict that receives
    the waveform and the mapper (tests rebuild the schedule from them with the oracle)."""
    from . import audio, latent
    from .stylegan2 import MappingNetwork, get_z_latents
    sr = 1024 * fps
    if device_rng:
        from .rng import PhiloxStreams, clip_audio
        wav = clip_audio(n_frames * 1024, sr, seed=1234)
        env = audio.onsets(wav, sr).squeeze(-1)                      # [T], on device
        mapper = MappingNetwork(w_dim, 0, w_dim, num_ws, generator=PhiloxStreams(0, first_stream=1 << 20))
    else:
        # the mapper's random init (2.1 M host draws, its own generator) does not depend on the clip: drawn on a
        # helper thread while the waveform is synthesised (torch's CPU samplers release the GIL)
        import threading
        box = {}

        def make_mapper():
            box["mapper"] = MappingNetwork(w_dim, 0, w_dim, num_ws, generator=torch.Generator().manual_seed(0))
        th = threading.Thread(target=make_mapper)
        th.start()
        wav = synthetic_audio(n_frames * 1024, sr, fast=fast_audio)
        env = audio.onsets(wav, sr).squeeze(-1)                      # [T], on device
        th.join()
        mapper = box["mapper"]
    if keep is not None:
        keep.update(wav=wav, mapper=mapper)
    palette = mapper(get_z_latents(seeds, w_dim).float())   # [P, num_ws, w_dim]
    half = palette.shape[0] // 2
    low = latent.spline_loops(palette[:half], n_frames, n_loops)
    high = latent.spline_loops(palette[half:2 * half], n_frames, n_loops)
    lat = latent.sequence_weighted(low, high, env)
    lat = audio.gaussian_filter(lat, 2)
    return lat.contiguous(), {"seeds": seeds, "schedule": f"spline_loops(n_loops={n_loops}) x2 blended by onsets, gaussian sigma=2",
                              "audio_and_mapper": "device counter RNG" if device_rng else "torch host generators"}


def warm_up(device="cuda"):
    """One tiny clip (16 frames, 64 x 64, 64-channel generator) through every step a render takes - audio pre-pass, mapper, spline
    schedule, blend, noise planes, synthesis, u8 pack - so that what the HIP runtime and the library set up ONCE PER PROCESS (the
    runtime's staging buffers for pageable copies, the caching allocator's first pools, the code objects of the library's
    translation units, the context's scratch arena) exists before the first real clip.  Returns the u8 frames (tests)."""
    from . import audio, latent
    from .noise import Loop, loop_batch
    from .stylegan2 import MappingNetwork, SynthesisNetwork, get_z_latents
    n, fps, w_dim, res = 16, 30, 64, 64
    from .rng import PhiloxStreams, clip_audio
    wav = clip_audio(n * 1024, 1024 * fps, seed=7, device=device)
    env = audio.onsets(wav, 1024 * fps).squeeze(-1)
    net = SynthesisNetwork(w_dim, res, 3, channel_base=4096, channel_max=64, dtype=torch.bfloat16,
                           generator=torch.Generator().manual_seed(0))
    # (device-resident mapper weights: the scale / transpose of its matrices are PyTorch device kernels - loaded here once)
    mapper = MappingNetwork(w_dim, 0, w_dim, net.num_ws, generator=PhiloxStreams(0, first_stream=1 << 20, device=device))
    pal = mapper(get_z_latents("0-4", w_dim).float())
    lat = audio.gaussian_filter(latent.sequence_weighted(latent.spline_loops(pal[:2], n, 2), latent.spline_loops(pal[2:4], n, 2), env), 2)
    rng = torch.Generator().manual_seed(1)
    noise = [Loop(rng, n, (s[3], s[3]), n_loops=2, sigma=5) for s in net.layer_shapes()]
    u8 = torch.empty((n, res, res, 3), dtype=torch.uint8, device=device)
    net(lat.contiguous(), noise=loop_batch(noise, 0, n, raw=True), rgb8_out=u8)
    return u8
