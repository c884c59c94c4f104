"""GPU box: the per-instruction energy price list behind DESIGN 4b (round 5).
Runs scripts/ubench/_bin/energy (built here by hipcc when missing) for every instruction class on all CUs and on half of them,
samples rocm-smi socket power / sclk while each runs, and fits  P = P_fixed + rate x E  per class:
    E = (P_all - P_half) / (rate_all - rate_half)        joules per wave-instruction (per byte for hbm)
python scripts/energy_prices.py [seconds-per-run] > gpurun_out/energy_prices.json"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "scripts/ubench/_bin/energy")
SRC = os.path.join(ROOT, "scripts/ubench/energy.hip")
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\):\s*([0-9.]+)", out)
    s = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)


def run(mode, wgs):
    pr = subprocess.Popen([BIN, mode, str(wgs), str(SECS)], stdout=subprocess.PIPE, text=True)
    time.sleep(min(2.0, SECS * 0.4))
    samples = []
    while pr.poll() is None and len(samples) < 6:
        samples.append(smi())
        time.sleep(0.5)
    out = pr.communicate()[0]
    rec = json.loads(out.strip().split("\n")[-1])
    pw = [s[0] for s in samples if s[0] is not None]
    ck = [s[1] for s in samples if s[1] is not None]
    rec["power_w"] = sum(pw) / len(pw) if pw else None
    rec["sclk_mhz"] = sum(ck) / len(ck) if ck else None
    rec["power_samples"] = pw
    return rec


def main():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", SRC, "-o", BIN], check=True)
    res = {"runs": [], "prices": {}}
    res["runs"].append(run("idle", 0))
    for mode in ("mfma", "lds", "mix", "valu", "pkvalu", "dma", "hbm"):
        full = run(mode, 256)
        half = run(mode, 128)
        res["runs"] += [full, half]
        if full["power_w"] and half["power_w"] and full["rate"] > half["rate"]:
            e = (full["power_w"] - half["power_w"]) / (full["rate"] - half["rate"])
            res["prices"][mode] = {"joule_per_unit": e, "fixed_w": full["power_w"] - full["rate"] * e, "rate_all": full["rate"],
                                   "power_all": full["power_w"], "sclk_all": full["sclk_mhz"]}
    print(json.dumps(res, indent=1))
    # human-readable lines on stderr: nJ per wave-instruction, and the clock each saturated pipe runs at
    for m, p in res["prices"].items():
        unit = "pJ/byte" if m == "hbm" else "nJ/wave-instr"
        val = p["joule_per_unit"] * (1e12 if m == "hbm" else 1e9)
        per_cu = p["rate_all"] / 256 / 1e9
        print(f"{m:7s} {val:9.3f} {unit:14s} rate {p['rate_all']:.3e}/s ({per_cu:.3f} G/s/CU)  P {p['power_all']:.0f} W  sclk {p['sclk_all']}  fixed {p['fixed_w']:.0f} W",
              file=sys.stderr)


if __name__ == "__main__":
    main()
