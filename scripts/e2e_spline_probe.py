"""GPU box: what the first call into a translation unit of libmaua_hip.so costs (code-object load) - latent.hip as the example."""
import sys, time, torch
sys.path.insert(0, ".")
torch.set_num_threads(8)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from maua_amd import latent, audio
from maua_amd.stylegan2 import MappingNetwork, get_z_latents
mapper = MappingNetwork(512, 0, 512, 18, generator=torch.Generator().manual_seed(0))
pal = mapper(get_z_latents("0-60", 512).float())
torch.cuda.synchronize()
half = pal.shape[0] // 2
def timed(tag, f):
    t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{tag}: host {1e3 * (t1 - t0):.2f} ms, + device drain {1e3 * (t2 - t1):.2f} ms"); return r
a = torch.randn(4, 8).cuda(); env = torch.rand(4).cuda()
if "blend" in sys.argv:
    timed("tiny blend (latent.hip, first kernel of the TU)", lambda: latent.single_weighted(a[0], a[1], env))
big = timed("torch.empty 133 MB", lambda: torch.empty((3600, 18, 512), device="cuda"))
del big
for k in range(2):
    timed(f"spline_loops call {k}", lambda: latent.spline_loops(pal[:half], 3600, 4))
