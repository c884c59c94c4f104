"""configs[3] with image prompts alone: the `diffusion.guided_image_prompts` arm of bench.py (VGGGrads + ColorMatchGrads + LPIPSGrads, default
"fast" conditioning, 100-step DDIM at 256^2) - `python scripts/bench_image_prompts.py [batch] [steps] [graph 0/1]`."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from maua_amd.diffusion import GuidedDiffusion, create_models  # noqa: E402
from maua_amd.grad import ColorMatchGrads, ContentPrompt, LPIPSGrads, StylePrompt, VGGGrads  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    size = 256
    model, diffusion, secondary = create_models("uncondImageNet256", f"ddim{steps}", allow_random_init=True, use_secondary=True,
                                                generator=torch.Generator().manual_seed(0))
    gr = torch.Generator().manual_seed(9)
    mods = [VGGGrads(scale=100.0, allow_random_init=True, generator=gr), ColorMatchGrads(scale=1e4), LPIPSGrads(scale=10.0, allow_random_init=True, generator=gr)]
    gd = GuidedDiffusion(mods, timesteps=steps, model=model, diffusion=diffusion, secondary_model=secondary)
    gd.use_graph = graph
    pr = [StylePrompt(img=torch.rand(1, 3, size, size, generator=gr)), ContentPrompt(img=torch.rand(1, 3, size, size, generator=gr))]
    x0, nz = (torch.randn(B, 3, size, size, generator=gr).cuda() for _ in range(2))
    n = diffusion.num_timesteps
    t0 = time.perf_counter()
    gd.run(x0, pr, n - 1, n, noise=nz)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = gd.run(x0, pr, n - 1, n, noise=nz)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"batch": B, "steps": steps, "graph_requested": graph, "hipgraph": model.guided_graph_active(), "first_run_s": first,
                      "seconds_per_batch": dt, "ms_per_step": dt / steps * 1e3, "samples_per_s": B / dt, "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
