"""Frame-parallel multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  Frames are independent after the clip-global pre-pass, so the render is sharded
by contiguous frame range with NO data-path collective; the only exchange is one gather of the packed u8 frames to
rank 0 at the end (precedent for the process layout: reference maua/super/image/bulk.py:31-109)."""
import os

import torch
import torch.distributed as dist

from .pipeline import frame_range


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def maybe_init_process_group(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) if present."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return world_info()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world_info()


_comm = {}


def _cabi_comm(rank, world, device):
    """RCCL communicator of the C-ABI (maua_comm_*), one per process; the 128-byte id travels by a torch.distributed
    broadcast (any initialised backend)."""
    import ctypes as C
    from . import _lib as L
    key = (rank, world, torch.device(device).index)
    if key in _comm:
        return _comm[key]
    lib = L.lib()
    idbuf = (C.c_char * 128)()
    if rank == 0:
        L.check(lib.maua_comm_unique_id(idbuf))
    t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
    if world > 1:
        dev = device if dist.get_backend() == "nccl" else "cpu"
        t = t.to(dev)
        dist.broadcast(t, src=0)
        t = t.cpu()
    idbuf.raw = bytes(t.tolist())
    comm = C.c_void_p()
    L.check(lib.maua_comm_init(L.ctx(device), idbuf, rank, world, C.byref(comm)))
    _comm[key] = comm
    return comm


def gather_frames_cabi(local, n_frames, rank=None, world=None, dst=0):
    """The same gather through the library's own entry point (``maua_gather_frames``: grouped RCCL send / recv on the
    render stream, exact shard sizes, no padding); ``local`` is this rank's packed u8 shard on the device."""
    import ctypes as C
    from . import _lib as L
    if rank is None or world is None:
        rank, world = world_info()
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    per = local[0].numel() * local.element_size() if local.shape[0] else 0
    if world > 1 and per == 0:   # an empty shard does not know the frame size: ask the ranks that have frames
        t = torch.tensor([per], dtype=torch.int64, device=local.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per = int(t.item())
    nbytes = (C.c_long * world)(*[(hi - lo) * per for lo, hi in sizes])
    local = local.contiguous()
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "shard does not match frame_range"
    out = torch.empty((n_frames, *local.shape[1:]), dtype=local.dtype, device=local.device) if rank == dst else None
    comm = _cabi_comm(rank, world, local.device)
    L.check(L.lib().maua_gather_frames(comm, L.ptr(local.view(torch.uint8).reshape(-1)) if local.numel() else None, nbytes,
                                       L.ptr(out.view(torch.uint8).reshape(-1)) if out is not None else None, dst))
    return out


def gather_frames(local, n_frames, rank=None, world=None, dst=0):
    """Gather per-rank frame shards [n_r, ...] (contiguous ranges from ``frame_range``) into [n_frames, ...] on
    ``dst``; other ranks get None.  Shards may differ by one frame: they are padded to the largest shard so that a
    single gather call moves them, then trimmed."""
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return local
    if os.environ.get("MAUA_GATHER") == "cabi" and local.is_cuda:
        return gather_frames_cabi(local, n_frames, rank, world, dst)
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxn:
        pad = torch.zeros((maxn, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)])
