"""(Needs modconv_tconv_pc.hip wired into the build: synth options `tconv_pc` / `fir_mfma`, see README.md here.)
GPU box: the fused up-layer's kernels against each other on SEPARATE network objects (own workspaces: no stale data can pass):
arm 0 = tconv_fir_kernel, FIR on the vector ALUs (bit-identical to the two-launch path); arm 1 = the form named on the command
line: "fir" = tconv_fir_kernel's MFMA-FIR form, "pc" (default) = the persistent producer / consumer kernel.  Prints max
difference / PSNR / u8 agreement on a 256^2 network whose 32^2..128^2 up-layers all take the fused kernel (overhanging tiles,
several channel blocks, noise, biases), on a resized 40 x 96 grid, and on the 1024^2 network.   python scripts/check_fir_mfma.py [pc|fir]"""
import sys

import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork

PC = "fir" not in sys.argv[1:]


def psnr(a, b):
    rng = float(b.max() - b.min())
    return 10 * torch.log10(torch.tensor(rng * rng / max(float(((a - b) ** 2).mean()), 1e-30))).item()


def make(res, small):
    if small:
        net = SynthesisNetwork(64, res, 3, channel_base=8192, channel_max=128, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(3))
        p = net.state_dict()
        for k in p:   # non-trivial biases
            if k.endswith(".bias") and "affine" not in k:
                p[k] = torch.randn(p[k].shape, generator=torch.Generator().manual_seed(4)) * 0.1
        net.load_state_dict(p)
        return net
    return SynthesisNetwork(512, res, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))


def cmp(nets, ws, noise, fir_from, tag):
    out = {}
    for v in (1, 0):
        h = nets[v]._handle()
        L.check(L.lib().maua_synth_set_option(h, b"tconv_fir", fir_from))
        L.check(L.lib().maua_synth_set_option(h, b"fir_mfma", v))
        L.check(L.lib().maua_synth_set_option(h, b"tconv_pc", v if PC else 0))
        img = nets[v](ws, noise=noise).float().cpu()
        u8 = torch.empty((ws.shape[0], img.shape[2], img.shape[3], 3), dtype=torch.uint8, device="cuda")
        nets[v](ws, noise=noise, rgb8_out=u8)
        out[v] = (img, u8.cpu())
    d = (out[0][0] - out[1][0]).abs()
    rng = float(out[0][0].max() - out[0][0].min())
    print(f"{tag}: max |diff| {float(d.max()):.3e} of range {rng:.2f}, PSNR {psnr(out[1][0], out[0][0]):.1f} dB, "
          f"u8 bytes differing {float((out[0][1] != out[1][1]).float().mean()) * 100:.4f} %, finite {bool(torch.isfinite(out[1][0]).all())}", flush=True)
    return out


g = torch.Generator().manual_seed(21)
nets = {v: make(256, True) for v in (0, 1)}
B = 3
ws = torch.randn(B, nets[0].num_ws, 64, generator=g)
noise = [torch.randn(B, 1, s[3], s[3], generator=g) for s in nets[0].layer_shapes()]
cmp(nets, ws, noise, 32, "256^2 net, fused from 32^2")
for v in (0, 1):
    nets[v].set_resize(7, target=(40, 96), noise_generator=torch.Generator().manual_seed(5))
cmp(nets, ws, None, 32, "resized 40 x 96 grid")
nets = {v: make(1024, False) for v in (0, 1)}
ws = torch.randn(4, nets[0].num_ws, 512, generator=g)
noise = [torch.randn(4, 1, s[3], s[3], generator=g) for s in nets[0].layer_shapes()]
cmp(nets, ws, noise, 256, "1024^2 net, fused at 256^2")
cmp(nets, ws, noise, 64, "1024^2 net, fused from 64^2")
