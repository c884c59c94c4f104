"""SURVEY 8(d) metric (ii): end-to-end frames/s of the whole C1 clip on one GPU — synthetic 120 s audio -> audio
pre-pass (HPSS onsets) -> latent schedule -> weight init/upload -> 3600 frames of noise + synthesis + u8 pack,
everything timed from a cold library handle (the HIP context itself is already up)."""
import sys, time, torch
sys.path.insert(0, ".")
from maua_amd import pipeline
from maua_amd.noise import Loop, loop_batch
from maua_amd.stylegan2 import SynthesisNetwork

T, FPS, RES, B = 3600, 30, 1024, 32
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
net = SynthesisNetwork(512, RES, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
t1 = time.perf_counter()
latents, info = pipeline.synthetic_clip_latents(T, FPS, net.num_ws, 512)
latents = latents.cuda(); torch.cuda.synchronize()
t2 = time.perf_counter()
rng = torch.Generator().manual_seed(42)
sizes = [s[3] for s in net.layer_shapes()]
noise = [Loop(rng, T, (s, s), n_loops=4, sigma=5) for s in sizes]
net._handle(); torch.cuda.synchronize()
t3 = time.perf_counter()
frames = torch.empty((T, RES, RES, 3), dtype=torch.uint8, device="cuda")
for i in range(0, T, B):
    b = min(B, T - i)
    net(latents[i:i + b], noise=loop_batch(noise, i, b), rgb8_out=frames[i:i + b])
torch.cuda.synchronize()
t4 = time.perf_counter()
print(f"weights init (host RNG) {t1-t0:.2f} s | audio pre-pass + latent schedule {t2-t1:.2f} s | noise planes + upload {t3-t2:.2f} s | "
      f"render {t4-t3:.2f} s = {T/(t4-t3):.0f} frames/s | end-to-end {T/(t4-t0):.0f} frames/s ({t4-t0:.2f} s for {T} frames)")
