"""CPU twin of the library's counter RNG (csrc/rng.hip) - TEST INFRASTRUCTURE, like everything under oracle/.

Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123): there is
no reference file to follow - the reference draws from torch's host generator - so this module is pinned to the paper's
known-answer vectors instead (tests/test_cabi_and_host.py).  Integers are bit-exact against the device; normals agree to the
last ulp or two of the float32 log / sin / cos implementations (tests state 4e-6 absolute)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]"""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [(hi1 ^ c[1] ^ k0) & MASK, lo1, (hi0 ^ c[3] ^ k1) & MASK, lo0]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, -1).astype(np.uint32)


def _words(seed, stream, offset, n):
    first = offset >> 2
    last = (offset + n + 3) >> 2
    c = np.arange(first, last, dtype=np.uint64)
    ctr = np.stack([c & MASK, c >> np.uint64(32), np.full_like(c, stream & 0xFFFFFFFF), np.full_like(c, (stream >> 32) & 0xFFFFFFFF)], -1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return philox4x32_10(ctr.astype(np.uint32), key.astype(np.uint32)[None]), int(offset - 4 * first)


def u32(seed, stream, n, offset=0):
    """elements offset .. offset + n of stream (seed, stream) as uint32 (what maua_philox_u32 writes)"""
    w, skip = _words(seed, stream, offset, n)
    return w.reshape(-1)[skip:skip + n]


def normal(seed, stream, n, offset=0, mean=0.0, std=1.0):
    """... as float32 normals (maua_philox_normal): Box-Muller on the word pairs (x0, x1), (x2, x3) of every counter"""
    w, skip = _words(seed, stream, offset, n)
    u = ((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    r = np.sqrt(np.float32(-2.0) * np.log(u[:, 0::2]))
    th = np.float32(6.283185307179586) * u[:, 1::2]
    z = np.stack([r * np.cos(th), r * np.sin(th)], -1).astype(np.float32)     # [counters, 2 pairs, 2]
    z = z.reshape(-1)[skip:skip + n]
    return (z * np.float32(std) + np.float32(mean)).astype(np.float32)


def clip_audio(seed, n, sr):
    """maua_philox_clip_audio: 0.3 sin(2 pi 220 t) + 0.2 (u - 0.5) [(2t mod 1) < 0.05] + 0.01 n with u = stream 0's words, n = stream
    1's normals (SURVEY 8(d)'s synthetic clip); float64 phase, float32 sum in the kernel's order."""
    t = np.arange(n, dtype=np.float64) / float(sr)
    ph = 2.0 * t
    click = (ph - np.floor(ph)) < 0.05
    tone = (0.3 * np.sin(6.283185307179586 * 220.0 * t)).astype(np.float32)
    u = ((u32(seed, 0, n) >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    amp = np.where(click, np.float32(0.2) * (u - np.float32(0.5)), np.float32(0.0)).astype(np.float32)
    return ((tone + amp) + np.float32(0.01) * normal(seed, 1, n)).astype(np.float32)
