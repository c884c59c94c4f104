"""Times the image-prompt grad modules (VGGGrads, ColorMatchGrads, LPIPSGrads; maua/grad.py:50-93, 178-196) alone: one call = the loss
and its gradient with respect to a batch of 256 x 256 images, random-init perceptors.  `python scripts/bench_grads.py [batch] [reps]`;
under `rocprofv3 --kernel-trace --stats` the same command gives the per-kernel split (profiles/r06_grads_kernel_stats.csv)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from maua_amd.grad import ColorMatchGrads, ContentPrompt, LPIPSGrads, StylePrompt, VGGGrads  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    g = torch.Generator().manual_seed(0)
    mods = {"vgg": VGGGrads(scale=100.0, allow_random_init=True, generator=g), "colormatch": ColorMatchGrads(scale=1e4),
            "lpips": LPIPSGrads(scale=10.0, allow_random_init=True, generator=g)}
    pr = [StylePrompt(img=torch.rand(1, 3, 256, 256, generator=g)).to("cuda"), ContentPrompt(img=torch.rand(1, 3, 256, 256, generator=g)).to("cuda")]
    img = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
    for name, m in mods.items():
        m.set_targets(pr)
        m(img, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m(img, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name}: {dt * 1e3:.3f} ms per call at batch {B} ({dt / B * 1e3:.3f} ms per image)")


if __name__ == "__main__":
    main()
