"""Achievable HBM copy bandwidth on this box (torch D2D copy of 1 GiB = read + write)."""
import time, torch
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"copy 1 GiB: {dt*1e3:.3f} ms -> {2*n/dt/1e12:.2f} TB/s (read+write)")
x = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for _ in range(3): x.fill_(1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): x.fill_(1.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"fill 1 GiB: {dt*1e3:.3f} ms -> {n/dt/1e12:.2f} TB/s (write only)")
for _ in range(3): s = x.sum()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): s = x.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"sum 1 GiB: {dt*1e3:.3f} ms -> {n/dt/1e12:.2f} TB/s (read only)")
