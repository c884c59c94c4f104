"""Where the clip's pre-pass time goes (first-use costs included): python scripts/e2e_phases.py"""
import sys, time, torch
sys.path.insert(0, ".")
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def lap(msg, t=[time.perf_counter()]):
    torch.cuda.synchronize(); n = time.perf_counter(); print(f"{msg:34s} {n - t[0]:.3f} s"); t[0] = n
from maua_amd import _lib as L, audio, latent, pipeline
from maua_amd.stylegan2 import MappingNetwork, get_z_latents
lap("imports")
L.ctx(); lap("library handle (code object load)")
wav = pipeline.synthetic_audio(3600 * 1024, 30720); lap("synthetic audio (host)")
env = audio.onsets(wav, 30720).squeeze(-1); lap("onsets (STFT, HPSS, mel, ...)")
env2 = audio.onsets(wav, 30720).squeeze(-1); lap("onsets again (warm)")
mapper = MappingNetwork(512, 0, 512, 18, generator=torch.Generator().manual_seed(0)); lap("mapper init (host RNG)")
pal = mapper(get_z_latents("0-60", 512).float()); lap("mapper forward (first GEMMs)")
pal = mapper(get_z_latents("0-60", 512).float()); lap("mapper forward (warm)")
half = pal.shape[0] // 2
low = latent.spline_loops(pal[:half], 3600, 4); high = latent.spline_loops(pal[half:2 * half], 3600, 4); lap("spline loops x2")
lat = latent.sequence_weighted(low, high, env); lat = audio.gaussian_filter(lat, 2); lap("blend + gaussian")
