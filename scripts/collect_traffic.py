"""Measured HBM traffic per kernel launch from rocprofv3 PMC counters (GPU box).

Two separate passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md §rocprofv3 PMC
slots), each `rocprofv3 --kernel-trace --pmc <counter>` over `bench.py --steps 4 --warmup 1 --no-cpu-baseline`.
Corrections prescribed by MI355X_MICROARCH.md §HBM: counters are in KiB (x1024); on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken
as reported (uncalibrated).  Infinity-Cache hits are included in both (fabric-side counters).
Writes profiles/traffic.json: {bench kernel name: {"bytes_per_launch": ..., "fetch_bytes": ..., "write_bytes": ...}}.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "traffic")
BATCH = int(os.environ.get("BATCH", "128"))  # frames per step (bench.py default)


def run(counter, batch=None, steps=4, timeout=None, out=None):
    out = out or OUT
    os.makedirs(out, exist_ok=True)
    # (the child must not start its own counter passes: bench.py runs these two by itself at the end of a default run)
    env = dict(os.environ, TMPDIR="/tmp", MAUA_BENCH_NO_LIVE_TRAFFIC="1")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", counter, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-extras",
           "--batch", str(batch or BATCH)]
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=timeout)
    f = glob.glob(os.path.join(out, "**", f"{counter}_counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def bench_name(k):
    m = re.search(r"modconv3x3_kernel<unsigned short, (\d), (\d), (\d), (\d), (\d), (\d+)>", k)
    if m:
        return "modconv3x3_kernel<bf16,%s,%s,%s,%s,%s,%s>" % m.groups()
    if "tconv_fir_kernel" in k:
        return "tconv_fir_kernel"
    if "tconv_dma_kernel" in k:
        return "tconv_dma_kernel (+edges, +premod)"
    if "tconv_edges_kernel" in k:
        return "tconv_edges_kernel"
    m = re.search(r"modconv_dma_kernel<(\d), (\d), (\d), (\d), (\d), (\d+)(?:, (?:true|false))*>", k)
    if m:
        return "modconv_dma_kernel<%s,%s,%s,%s,%s,%s>" % m.groups()
    if "tconv2_kernel<unsigned short>" in k:
        return "tconv2_kernel<bf16> (edges)"
    if "upfir_epilogue_kernel<unsigned short" in k:
        return "upfir_epilogue_kernel<bf16>"
    if "upwalk_fused_kernel" in k:
        return "upwalk_fused_kernel<64,32>"
    if "upwalk_kernel" in k:
        return "upwalk_kernel<64,32>"
    m = re.search(r"modconv_hires_kernel<(\d+), (\d+), (\d)>", k)
    if m:
        return "modconv_hires_kernel<%s,%s,%s>" % m.groups()
    for n in ("torgb_kernel", "pack_rgb8_kernel", "styles_affine_kernel", "styles_demod_kernel",
              "noise_loop_batch_sumsq_kernel", "noise_loop_batch_write_kernel", "noise_loop_batch_raw_kernel"):
        if n in k:
            return n
    return None


def main():
    fetch, write = run("FETCH_SIZE"), run("WRITE_SIZE")
    out = {}
    for k in fetch:
        n = bench_name(k)
        if n is None:
            continue
        f = 2.0 * 1024 * sum(fetch[k]) / len(fetch[k])
        w = 1024 * sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        out[n] = {"bytes_per_launch": f + w, "fetch_bytes": f, "write_bytes": w, "launches_sampled": len(fetch[k]),
                  "batch": BATCH,
                  "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane coalesced), KiB -> B; WRITE_SIZE as reported"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
    for n, v in sorted(out.items()):
        print(f"{n:44s} {v['bytes_per_launch'] / 1e6:10.1f} MB/launch (fetch {v['fetch_bytes'] / 1e6:.1f}, write {v['write_bytes'] / 1e6:.1f})")


if __name__ == "__main__":
    main()
