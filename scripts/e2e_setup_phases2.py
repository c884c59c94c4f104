"""Where bench.py's e2e set-up goes with the device counter RNG (round 5): each phase alone (synchronised, best of 3), then
bench.build_inputs.  python scripts/e2e_setup_phases2.py"""
import os, sys, time, torch
torch.set_num_threads(8)
sys.path.insert(0, ".")
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from maua_amd import _lib as L, audio, latent, pipeline
from maua_amd.noise import Loop
from maua_amd.rng import philox_normal
from maua_amd.stylegan2 import SynthesisNetwork, MappingNetwork, get_z_latents, init_synthesis_params_device
import bench
pipeline.warm_up("cuda"); torch.cuda.synchronize()
dev = torch.device("cuda", 0)


def best(msg, fn, n=3):
    b = None
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); d = time.perf_counter() - t0
        b = d if b is None else min(b, d)
    print(f"{msg:52s} {b * 1e3:7.2f} ms", flush=True)
    return r
p = best("weights: device RNG (init_synthesis_params_device)", lambda: init_synthesis_params_device(1024, 512, seed=0))
def mk():
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, _params=p); net._handle(); return net
net = best("weights: create + load + prep (net._handle)", mk)
best("17 noise modules (device RNG)", lambda: [Loop(None, 3600, (s, s), n_loops=4, sigma=5, noise=philox_normal((3, s, s), 42, j, device=dev)) for j, s in enumerate(bench.NOISE_SIZES)])
wav = best("synthetic audio (host, fast)", lambda: pipeline.synthetic_audio(3600 * 1024, 30720, fast=True))
env = best("onsets (upload, STFT, HPSS, mel, ...)", lambda: audio.onsets(wav, 30720).squeeze(-1))
wd = wav.cuda()
best("onsets from a device waveform", lambda: audio.onsets(wd, 30720).squeeze(-1))
mapper = best("mapper init (host RNG)", lambda: MappingNetwork(512, 0, 512, 18, generator=torch.Generator().manual_seed(0)))
pal = best("mapper forward (incl. weight upload)", lambda: mapper(get_z_latents("0-60", 512).float()))
half = pal.shape[0] // 2
def sp():
    return latent.spline_loops(pal[:half], 3600, 4), latent.spline_loops(pal[half:2 * half], 3600, 4)
low, high = best("spline loops x2", sp)
best("blend + gaussian", lambda: audio.gaussian_filter(latent.sequence_weighted(low, high, env), 2))
best("clip chain (synthetic_clip_latents)", lambda: pipeline.synthetic_clip_latents(3600, 30, 18, 512))
best("bench.build_inputs", lambda: bench.build_inputs(dev, 0, 1))
