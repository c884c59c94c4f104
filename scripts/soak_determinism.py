"""Soak: the same inputs through the three networks many times - every repeat must be bit-identical (a race in an LDS pipeline
or a missing barrier shows up as a flipped bit long before it shows up as a wrong picture).   python scripts/soak_determinism.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from maua_amd.noise import Loop, loop_batch
    from maua_amd.stylegan2 import SynthesisNetwork
    t0 = time.time()
    net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    for B in (32, 7):
        ws = (torch.randn(B, net.num_ws, 512, generator=g) * 0.5).cuda()
        rng = torch.Generator().manual_seed(42)
        mods = [Loop(rng, 64, (s[3], s[3]), n_loops=2, sigma=5) for s in net.layer_shapes()]
        u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
        ref = None
        for rep in range(25):
            net(ws, noise=loop_batch(mods, 3, B), rgb8_out=u8)
            h = u8.reshape(-1).view(torch.int32).sum(dtype=torch.int64).item(), u8[::3, ::7, ::5].clone()
            if ref is None:
                ref = h
            assert h[0] == ref[0] and torch.equal(h[1], ref[1]), f"StyleGAN2 B={B}: repeat {rep} differs"
        print(f"StyleGAN2 1024^2 B={B}: 25 identical renders", flush=True)
    del net, u8
    torch.cuda.empty_cache()
    from maua_amd.diffusion import create_models
    model, diffusion, _ = create_models("uncondImageNet256", "ddim20", allow_random_init=True, generator=torch.Generator().manual_seed(0))
    x0 = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(2)).cuda()
    ref = None
    for rep in range(6):
        x = x0.clone()
        _, pred = diffusion.ddim_sample_loop(model, x)
        s = (pred.double().sum().item(), pred[:, :, ::9, ::11].clone())
        if ref is None:
            ref = s
        assert s[0] == ref[0] and torch.equal(s[1], ref[1]), f"diffusion: repeat {rep} differs"
    print("guided-diffusion 256^2 B=4, 20 DDIM steps: 6 identical loops", flush=True)
    del model
    torch.cuda.empty_cache()
    from maua_amd.super import RRDBNet
    up = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32, scale=4, dtype=torch.bfloat16)
    img = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(3)).cuda()
    ref = None
    for rep in range(10):
        s = up(img).double().sum().item()
        ref = s if ref is None else ref
        assert s == ref, f"RRDBNet: repeat {rep} differs"
    print("RRDBNet x4 256^2: 10 identical forwards", flush=True)
    print(f"soak ok in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
