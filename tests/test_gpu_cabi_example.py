"""The C ABI driven from plain C (examples/render_frame.c: no Python, no PyTorch in the process): compiled with gcc against
include/maua_hip.h + libmaua_hip.so, run as its own process, its frame compared byte for byte with the same network rendered
through the Python host layer and, in exact-f32 mode, with the CPU oracle."""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
W_DIM, CHANNEL_BASE, CHANNEL_MAX = 64, 2048, 64


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def _fill(n, seed, scale, offset):
    with np.errstate(over="ignore"):
        k = np.arange(n, dtype=np.uint32) + np.uint32((seed * 0x9E3779B9) & 0xFFFFFFFF)
        v = (_mix32(k) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(2.0) - np.float32(1.0)
    return torch.from_numpy(v * np.float32(scale) + np.float32(offset))


def _state_dict(res):
    """the parameter streams of examples/render_frame.c, in its order"""
    s3, sd, idx, prev, blk, r = 1.7320508, {}, 0, 0, 0, 4

    def put(name, shape, scale, offset):
        nonlocal idx
        idx += 1
        sd[name] = _fill(int(np.prod(shape)), idx, scale, offset).reshape(shape)
    while r <= res:
        c = min(CHANNEL_BASE // r, CHANNEL_MAX)
        if blk == 0:
            put("bs.0.const", (c, 4, 4), s3, 0.0)
        for which in ((1,) if blk == 0 else (0, 1)):
            ci = prev if which == 0 else c
            pfx = f"bs.{blk}.conv{which}"
            put(pfx + ".affine.weight", (ci, W_DIM), s3, 0.0)
            put(pfx + ".affine.bias", (ci,), 0.0, 1.0)
            put(pfx + ".weight", (c, ci, 3, 3), s3, 0.0)
            put(pfx + ".noise_const", (r, r), s3, 0.0)
            put(pfx + ".bias", (c,), 0.1, 0.0)
        put(f"bs.{blk}.torgb.affine.weight", (c, W_DIM), s3, 0.0)
        put(f"bs.{blk}.torgb.affine.bias", (c,), 0.0, 1.0)
        put(f"bs.{blk}.torgb.weight", (3, c, 1, 1), s3, 0.0)
        put(f"bs.{blk}.torgb.bias", (3,), 0.1, 0.0)
        prev, blk, r = c, blk + 1, r * 2
    return sd


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
@pytest.mark.parametrize("dtype_id,dt", [(1, torch.bfloat16), (0, torch.float32)], ids=["bf16", "f32"])
def test_plain_c_host_renders_the_same_frame(tmp_path, dtype_id, dt):
    from maua_amd.stylegan2 import SynthesisNetwork
    exe = tmp_path / "render_frame"
    cmd = ["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
           str(ROOT / "examples" / "render_frame.c"), f"-L{ROOT / 'maua_amd' / 'csrc'}", "-lmaua_hip", "-L/opt/rocm/lib",
           "-lamdhip64", f"-Wl,-rpath,{ROOT / 'maua_amd' / 'csrc'}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True)
    res = 64
    ppm = tmp_path / "frame.ppm"
    run = subprocess.run([str(exe), str(ppm), str(res), str(dtype_id)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    raw = ppm.read_bytes()
    header = f"P6\n{res} {res}\n255\n".encode()
    assert raw.startswith(header) and len(raw) == len(header) + res * res * 3
    got = torch.frombuffer(bytearray(raw[len(header):]), dtype=torch.uint8).reshape(res, res, 3)
    # the same network through the Python host layer
    net = SynthesisNetwork(W_DIM, res, 3, channel_base=CHANNEL_BASE, channel_max=CHANNEL_MAX, dtype=dt)
    sd = net.state_dict()
    new = _state_dict(res)
    assert set(new) <= set(sd) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in new.items())
    sd.update(new)
    net.load_state_dict(sd)
    ws = _fill(net.num_ws * W_DIM, 999, 1.0, 0.0).reshape(1, net.num_ws, W_DIM)
    u8 = torch.empty((1, res, res, 3), dtype=torch.uint8, device="cuda")
    img = net(ws, rgb8_out=u8)
    assert torch.equal(u8[0].cpu(), got), int((u8[0].cpu().int() - got.int()).abs().max())
    assert int(got.min()) < 64 and int(got.max()) > 192 and 20 < float(got.float().mean()) < 235     # a real picture, not a constant
    if dt == torch.float32:   # ... and the CPU oracle agrees with what the C program wrote
        from oracle import stylegan2 as OS
        ref = OS.synthesis_network(net.state_dict(), ws)
        want = ((ref[0] + 1) / 2).clamp(0, 1).mul(255).round().byte().permute(1, 2, 0)
        d = (want.int() - got.int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 0.005
