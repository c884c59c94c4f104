"""1920x1080 from the 1024^2 random-init net through the feature-space resize: timing + sanity."""
import sys, time, torch
sys.path.insert(0, ".")
from maua_amd.stylegan2 import StyleGAN2Synthesizer
B = 8
for strategy, layer in (("stretch", 11), ("pad-reflect-out", 11), ("stretch", 3)):
    gen = torch.Generator().manual_seed(0)
    syn = StyleGAN2Synthesizer(None, False, (1920, 1080), strategy, layer, generator=gen)
    ws = torch.randn(B, syn.num_ws, 512, generator=gen).cuda()
    h, w = syn.G_synth.output_hw
    u8 = torch.empty((B, h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        syn.forward(ws, rgb8_out=u8)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        syn.forward(ws, rgb8_out=u8)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    img = syn.forward(ws[:1])
    print(f"{strategy:16s} layer {layer:2d}: output {w}x{h}  {B/dt:7.1f} frames/s  finite={bool(torch.isfinite(img).all())} "
          f"u8 mean {float(u8.float().mean()):.1f}")
    del syn
