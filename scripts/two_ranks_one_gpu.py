"""Try the N = 2 code paths on a ONE-GPU box: two processes, both on cuda:0, RCCL between them (RCCL normally refuses two ranks
on one device - then this exercises the failure handling instead).   python scripts/two_ranks_one_gpu.py [bench|gather]"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    """one rank: RCCL process group on cuda:0, the one-shot and the streamed gather of a small sharded render"""
    import datetime
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0),
                            timeout=datetime.timedelta(seconds=90))
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce over torch's RCCL group:", t.tolist(), flush=True)
    from test_gpu_distributed import _render
    from maua_amd.distributed import StreamingGather, gather_frames
    from maua_amd.pipeline import frame_range
    T = 13
    lo, hi = frame_range(T, rank, 2)
    shard = _render(lo, hi, T)
    full = gather_frames(shard, T, rank, 2)
    g = StreamingGather(T, (64, 64, 3), 3)
    for off, n in g.chunks():
        g.local[off:off + n] = shard[off:off + n]
        g.chunk_done()
    full3 = g.finish()
    torch.cuda.synchronize()
    if rank == 0:
        want = _render(0, T, T)
        print("rank 0: one-shot gather == single-GPU render:", torch.equal(full, want), " streamed:", torch.equal(full3, want), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "done", flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gather"
    if what == "_worker":
        return worker()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        if what == "bench":
            env["MAUA_BENCH_BACKEND"] = "gloo"   # torch's control plane over gloo; the library's RCCL communicator will refuse the shared GPU
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16"]
        else:
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "two_ranks_one_gpu.py"), "_worker"]
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[TIMEOUT]"
        lines = [l for l in out.strip().splitlines() if "NCCL WARN" not in l and l.strip()]
        js = [l[l.index('{"metric"'):] for l in out.splitlines() if '{"metric"' in l]
        print(f"---- rank {rank} rc={p.returncode}")
        if js:
            import json
            d = json.loads(js[-1])
            print("JSON line:", {k: d.get(k) for k in ("value", "scaling", "n_gpus", "steps", "ms_per_step", "sustained", "gather_ms", "clip_leg")})
            print("  weak (the K timed steps, `value`):", d.get("value"), " strong (the clip rendered + gathered once):", d.get("strong"))
        else:
            print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()
