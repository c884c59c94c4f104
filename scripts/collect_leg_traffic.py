"""HBM traffic of the configs[3] / configs[4] bench legs from rocprofv3 PMC counters (GPU box) - the whole leg, summed over its
kernels, per unit of work (one UNet forward of the batch / one up-scaled frame).

    python scripts/collect_leg_traffic.py diffusion|upscale

Two separate passes (FETCH_SIZE, WRITE_SIZE: they do not fit the TCC counter slots together), `rocprofv3 --kernel-trace --pmc <counter>`
only (no other trace domains), corrections of MI355X_MICROARCH.md section HBM as in collect_traffic.py (FETCH_SIZE x 2 on gfx950 for
16-byte-per-lane streaming reads, KiB -> B; WRITE_SIZE as reported; Infinity-Cache hits are included in both).
Writes gpurun_out/<leg>_traffic.json: totals per unit + the ten largest kernels."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
leg = sys.argv[1] if len(sys.argv) > 1 else "diffusion"
if leg == "diffusion":
    B, STEPS = 8, 6
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_diffusion.py"), "--batch", str(B), "--steps", str(STEPS), "--reps", "1", "--no-graph"]
    units, unit = None, f"one UNet forward at batch {B}, 256 x 256"
    COUNT, PER = "nchw_to_nhwc_kernel<float", 1      # one input conversion per forward: the run's forwards are counted from the trace
else:
    FR, UB = 4, 4
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_upscale_quick.py"), str(FR), str(UB)]
    units, unit = None, "one 1024^2 frame rendered and up-scaled x4 to 4096^2 u8 (SynthesisNetwork + RealESRGANer.enhance_frames)"
    COUNT, PER = "upwalk_fused_kernel", FR           # one fused last-block walk per synthesis call of FR frames (every frame is up-scaled)
OUT = os.path.join(ROOT, "gpurun_out", f"{leg}_traffic")


def run(counter):
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", OUT, "-o", counter, "--", *cmd],
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    f = glob.glob(os.path.join(OUT, f"{counter}_counter_collection.csv"))[0]
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]] += float(r["Counter_Value"])
    return per


fetch, write = run("FETCH_SIZE"), run("WRITE_SIZE")
trace = glob.glob(os.path.join(OUT, "FETCH_SIZE_kernel_trace.csv"))[0]
units = PER * sum(1 for r in csv.DictReader(open(trace)) if COUNT in r["Kernel_Name"])
tot_f = 2.0 * 1024 * sum(fetch.values()) / units
tot_w = 1024 * sum(write.values()) / units
ker = sorted(((2.0 * 1024 * fetch[k] + 1024 * write.get(k, 0.0)) / units, k) for k in fetch)[::-1][:10]
out = {"leg": leg, "unit": unit, "units_in_run": units, "bytes_per_unit": tot_f + tot_w, "fetch_bytes": tot_f, "write_bytes": tot_w,
       "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane coalesced), KiB -> B; WRITE_SIZE as reported; all kernels of the run summed "
                     "(set-up uploads and weight preparation included: they are part of the counters' run, a few % of the total)",
       "largest_kernels": [{"kernel": k[:120], "bytes_per_unit": v} for v, k in ker]}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{leg}_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("leg", "unit", "bytes_per_unit", "fetch_bytes", "write_bytes")}))
