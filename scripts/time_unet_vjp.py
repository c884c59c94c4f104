"""GPU box: time the diffusion UNet's kept forward + input gradient (guidance speed "regular") against the plain forward at the
configs[3] network (256 x 256, 552.8 M parameters, random init).  python scripts/time_unet_vjp.py [--batch 8] [--reps 5]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--size", type=int, default=256)
    a = ap.parse_args()
    from maua_amd.diffusion import create_models
    model, diffusion, _ = create_models("uncondImageNet256", timestep_respacing="ddim100", allow_random_init=True,
                                        generator=torch.Generator().manual_seed(0))
    model.enable_vjp()
    B, S = a.batch, a.size
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    t = torch.full((B,), 500.0).cuda()
    g_out = torch.randn(B, 6, S, S, generator=g).cuda()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e3
    fwd = timed(lambda: model(x, t))
    keep = timed(lambda: model.forward_keep(x, t))

    def both():
        model.forward_keep(x, t)
        return model.vjp(g_out)
    fb = timed(both)
    gx = both()
    print(json.dumps({"batch": B, "size": S, "forward_ms": fwd, "forward_keep_ms": keep, "forward_keep_plus_vjp_ms": fb, "vjp_ms": fb - keep,
                      "vjp_over_forward": (fb - keep) / fwd, "finite": bool(torch.isfinite(gx).all()),
                      "peak_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
