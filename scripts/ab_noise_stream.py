"""GPU box: the per-step noise maps generated on a side stream one step ahead (overlapping the previous step's synthesis) against
the in-order form.   python scripts/ab_noise_stream.py [B] [steps]"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from maua_amd.noise import loop_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
net, latents, noise, info = bench.build_inputs(dev, 0, 1)
u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device=dev)
hi = 3600


def start(k):
    return (k * B) % (hi - B + 1)


def in_order(n):
    for k in range(n):
        i = start(k)
        net(latents[i:i + B], noise=loop_batch(noise, i, B), rgb8_out=u8)


side = torch.cuda.Stream()


def ahead(n):
    main = torch.cuda.current_stream()
    with torch.cuda.stream(side):
        nz = loop_batch(noise, start(0), B)
        ev = torch.cuda.Event(); ev.record(side)
    for k in range(n):
        main.wait_event(ev)
        cur = nz
        done = torch.cuda.Event()
        if k + 1 < n:
            with torch.cuda.stream(side):
                nz = loop_batch(noise, start(k + 1), B)
                ev = torch.cuda.Event(); ev.record(side)
        net(latents[start(k):start(k) + B], noise=cur, rgb8_out=u8)
        for t in cur:
            t.record_stream(main)


for name, fn in (("in order", in_order), ("noise one step ahead on a side stream", ahead), ("in order", in_order), ("noise one step ahead on a side stream", ahead)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:40s} {dt * 1e3:.3f} ms per step -> {B / dt:.0f} frames/s")
