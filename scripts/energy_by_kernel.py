"""Where a step's joules go, PER KERNEL, from counters (GPU box).

    python scripts/energy_by_kernel.py [B = 128] [out = gpurun_out/energy_by_kernel.txt]

Five counter passes over scripts/profile_layers.py (one synthesis forward repeated; `rocprofv3 --kernel-trace --pmc <set>`, PMC only) give,
per kernel and launch: wave-instruction counts by class (SQ_INSTS_MFMA / VALU / SALU / LDS / VMEM_RD / VMEM_WR), matrix-pipe busy cycles,
GRBM_GUI_ACTIVE, FETCH_SIZE / WRITE_SIZE and the trace's durations.  A socket-power sample (rocm-smi, 1 Hz) over an un-profiled loop
of the same forward gives the power the step draws.  Energy model of a launch:

    E = n_mfma e_mfma + n_lds e_lds + (n_valu - n_mfma) e_valu + n_salu e_salu + (n_vmem_rd + n_vmem_wr) e_kb + bytes_hbm e_hbm
        + (P_idle + k_busy * mfma_busy) * t

with the round-5 price list (profiles/r05_energy_prices.txt: instruction classes measured one at a time, all CUs vs half of them):
e_mfma 11.47, e_lds 4.10, e_valu 1.1 (plain 0.86 / packed 1.87), e_kb 20.39 nJ per wave-instruction (a 1 KB LDS-direct piece; a 16-byte-
per-lane load or store is priced the same), e_hbm 0.11 nJ per byte, P_idle 270 W (clocks up, nothing issuing), and the two entries that
list leaves open: e_salu 0.3 nJ (assumed: a scalar instruction moves 1/64 of a vector one's data) and k_busy = 380 W - the price
list's own MFMA row draws 649 W that its per-instruction price does not explain (1288 W at 5.57e10 MFMA/s x 11.47 nJ = 639 W), i.e. 380 W
more than the idle floor with the matrix pipes saturated; that share is charged in proportion to the kernel's measured matrix-pipe
occupancy.  Residual = (P_step * t - E) / (P_step * t): what the model does not explain of the energy the kernel's time costs at the
step's measured power."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "128"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "energy_by_kernel.txt")
E_MFMA, E_LDS, E_VALU, E_SALU, E_KB, E_HBM = 11.47e-9, 4.10e-9, 1.1e-9, 0.3e-9, 20.39e-9, 0.11e-9
P_IDLE, K_BUSY = 270.0, 380.0
SETS = [
    ["SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]
env = dict(os.environ, TMPDIR="/tmp")
target = os.environ.get("PMC_TARGET", f"scripts/profile_layers.py {B}").split()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for si, cs in enumerate(SETS):
    d = f"/tmp/ebk_{si}"
    subprocess.run(["rm", "-rf", d])
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *cs, "--output-format", "csv", "-d", d, "-o", "k", "--", sys.executable, *target],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    f = glob.glob(d + "/**/k_counter_collection.csv", recursive=True)
    if not f:
        print("pass", si, "failed:", r.stderr[-400:])
        continue
    disp = collections.OrderedDict()
    for row in csv.DictReader(open(f[0])):
        e = disp.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"]})
        e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    seen, per = collections.Counter(e["name"] for e in disp.values()), collections.Counter()
    for e in disp.values():   # (skip the first half of every kernel's launches: cold caches, first-use loads)
        per[e["name"]] += 1
        if seen[e["name"]] > 6 and per[e["name"]] <= seen[e["name"]] // 2:
            continue
        for c in cs:
            if c in e:
                agg[e["name"]][c].append(e[c])
    if si == 0:
        t = glob.glob(d + "/**/k_kernel_trace.csv", recursive=True)
        seen2, per2 = collections.Counter(), collections.Counter()
        rows = list(csv.DictReader(open(t[0])))
        for row in rows:
            seen2[row["Kernel_Name"]] += 1
        for row in rows:
            per2[row["Kernel_Name"]] += 1
            if seen2[row["Kernel_Name"]] > 6 and per2[row["Kernel_Name"]] <= seen2[row["Kernel_Name"]] // 2:
                continue
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)

# the step's socket power: an un-profiled loop of the same forward, rocm-smi once a second
samples = []


def sample():
    while not stop.is_set():
        o = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
        for line in o.splitlines():
            if "Power (W)" in line:
                try:
                    samples.append(float(line.split(":")[-1]))
                except ValueError:
                    pass
        time.sleep(1.0)


stop = threading.Event()
code = ("import sys, torch; sys.path.insert(0, '.'); from maua_amd.stylegan2 import SynthesisNetwork\n"
        f"B = {B}\n"
        "net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))\n"
        "ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()\n"
        "u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device='cuda')\n"
        "import time\n"
        "net(ws, rgb8_out=u8); torch.cuda.synchronize(); t0 = time.time()\n"
        "while time.time() - t0 < 9: net(ws, rgb8_out=u8); torch.cuda.synchronize()\n")
p = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env)
time.sleep(6)   # (import + weight init)
th = threading.Thread(target=sample)
th.start()
p.wait()
stop.set()
th.join()
good = [s for s in samples if s >= 0.9 * max(samples)] if samples else []   # (the loop's own samples: not its start / end)
P_STEP = sum(good) / len(good) if good else 1290.0

lines = [f"# per-kernel energy account of one 1024^2 synthesis forward at B = {B} (scripts/energy_by_kernel.py; prices: profiles/r05_energy_prices.txt)",
         f"# socket power over an un-profiled loop of the forward: {P_STEP:.0f} W (mean of the {len(good)} rocm-smi samples within 10 % of the largest: {[round(s) for s in good]})",
         f"# model: E = 11.47 n_mfma + 4.10 n_lds + 1.1 (n_valu - n_mfma) + 0.3 n_salu + 20.39 (n_vmem_rd + n_vmem_wr) nJ + 0.11 nJ/B x HBM bytes + (270 + 380 x mfma_busy) W x t",
         "# columns: ms per launch | launches in the averaged half | mfma_busy | J: mfma / lds / valu / salu / vmem / hbm / idle / busy-share | model J | P_step x t J | residual"]
tot_m = tot_t = tot_ms = 0.0
rows = []
for name, m in agg.items():
    if name not in dur or not dur[name]:
        continue
    t = sum(dur[name]) / len(dur[name])
    if t < 20e-6:
        continue
    a = {c: (sum(v) / len(v) if v else 0.0) for c, v in m.items()}
    g = a.get("GRBM_GUI_ACTIVE", 0.0)
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * g / 8) if g else 0.0
    n_mfma, n_valu = a.get("SQ_INSTS_MFMA", 0.0), a.get("SQ_INSTS_VALU", 0.0)
    hbm = 2.0 * 1024 * a.get("FETCH_SIZE", 0.0) + 1024 * a.get("WRITE_SIZE", 0.0)
    parts = (n_mfma * E_MFMA, a.get("SQ_INSTS_LDS", 0.0) * E_LDS, max(0.0, n_valu - n_mfma) * E_VALU, a.get("SQ_INSTS_SALU", 0.0) * E_SALU,
             (a.get("SQ_INSTS_VMEM_RD", 0.0) + a.get("SQ_INSTS_VMEM_WR", 0.0)) * E_KB, hbm * E_HBM, P_IDLE * t, K_BUSY * busy * t)
    model, meas = sum(parts), P_STEP * t
    rows.append((t, name, len(dur[name]), busy, parts, model, meas))
for t, name, n, busy, parts, model, meas in sorted(rows, reverse=True):
    short = name.replace("maua::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    lines.append(f"{short:72s} | {t * 1e3:7.3f} | {n:3d} | {busy:.3f} | " + " / ".join(f"{p:.3f}" for p in parts) +
                 f" | {model:.3f} | {meas:.3f} | {(meas - model) / meas * 100:+.1f} %")
    if t >= 0.4e-3:
        tot_m += model
        tot_t += meas
        tot_ms += t * 1e3
lines.append(f"# kernels of >= 0.4 ms per launch: model {tot_m:.2f} J of {tot_t:.2f} J (P_step x t), residual {(tot_t - tot_m) / tot_t * 100:+.1f} % over {tot_ms:.1f} ms")
open(OUT, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
json.dump({"P_step": P_STEP, "samples": samples}, open(OUT.replace(".txt", ".json"), "w"))
