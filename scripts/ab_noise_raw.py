"""GPU box: the step with normalised (two-pass) and raw (one-pass + per-sample factors) Loop maps, alternating, + the noise
launches alone.   python scripts/ab_noise_raw.py [B] [steps]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from maua_amd.noise import loop_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
net, latents, noise, info = bench.build_inputs(dev, 0, 1)
u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device=dev)


def run(raw, n, synth=True):
    for k in range(n):
        i = (k * B) % (3600 - B + 1)
        nz = loop_batch(noise, i, B, raw=raw)
        if synth:
            net(latents[i:i + B], noise=nz, rgb8_out=u8)


for synth in (False, True):
    for rnd in range(2):
        for raw in (False, True):
            run(raw, 3, synth)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(raw, steps, synth)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(f"{'step' if synth else 'noise only'} raw={raw}: {dt * 1e3:.3f} ms")
