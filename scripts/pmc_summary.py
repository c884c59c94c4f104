"""Per-kernel SQ / GRBM counter summary of one synthesis forward (GPU box).

    python scripts/pmc_summary.py [B] [out.json]

Runs `rocprofv3 --kernel-trace --pmc <set>` once per counter set (PMC only, no other trace domains) over
scripts/profile_layers.py, averages every counter per kernel name over the warmed-up launches, and derives
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)   (share of matrix-pipe cycles in
              use; rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs - checked against a kernel of known MFMA
              count and duration: 18.9 M MFMAs x 32 cycles / (1024 SIMDs x 1.04 M cycles) = 0.57)
  clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration (needs the kernel-trace durations; printed when available)
  valu_busy = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CU_CYCLES-like denominators where available
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

B = sys.argv[1] if len(sys.argv) > 1 else "32"
out_json = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc_summary.json"
SETS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
     "SQ_ACTIVE_INST_VALU", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VMEM", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS",
     "SQ_BUSY_CU_CYCLES", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "GRBM_GUI_ACTIVE"],
]
env = dict(os.environ, TMPDIR="/tmp")
avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, env=env).stdout
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for si, cs in enumerate(SETS):
    cs = [c for c in cs if c in avail]
    d = f"/tmp/pmcsum_{si}"
    # PMC_TARGET: another workload than one synthesis forward, e.g. "scripts/bench_upscale_quick.py 4 4" (the configs[4] leg)
    target = os.environ.get("PMC_TARGET", f"scripts/profile_layers.py {B}").split()
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *cs, "--output-format", "csv", "-d", d, "-o", "k", "--",
           sys.executable, *target]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    f = glob.glob(d + "/**/k_counter_collection.csv", recursive=True)
    if not f:
        print("pass", si, "failed:", r.stderr[-500:])
        continue
    disp = collections.OrderedDict()
    for row in csv.DictReader(open(f[0])):
        e = disp.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"]})
        e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    seen = collections.Counter()
    for e in disp.values():
        seen[e["name"]] += 1
    per = collections.Counter()
    for e in disp.values():  # skip the first launches of every kernel (cold caches, first-use loads)
        per[e["name"]] += 1
        if seen[e["name"]] > 6 and per[e["name"]] <= seen[e["name"]] // 2:
            continue
        for c in cs:
            if c in e:
                agg[e["name"]][c].append(e[c])
res = {}
for name, cd in agg.items():
    m = {c: sum(v) / len(v) for c, v in cd.items()}
    g = m.get("GRBM_GUI_ACTIVE", 0.0)
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        m["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * g / 8)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if g:
            m["lds_busy"] = m["SQ_LDS_IDX_ACTIVE"] / (256 * g / 8)
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                m[c.lower() + "_share"] = m[c] / m["SQ_WAVE_CYCLES"]
    res[name] = m
json.dump(res, open(out_json, "w"), indent=1)
for name, m in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if m.get("GRBM_GUI_ACTIVE", 0) < 20000:
        continue
    print(f"{name[:90]:90s} gui {m.get('GRBM_GUI_ACTIVE', 0):9.0f}  mfma_busy {m.get('mfma_busy', 0):.3f}  "
          f"wait_any {m.get('sq_wait_any_share', 0):.2f} wait_inst {m.get('sq_wait_inst_any_share', 0):.2f} "
          f"active {m.get('sq_active_inst_any_share', 0):.2f}  lds_busy {m.get('lds_busy', 0):.2f} "
          f"lds_conf {m.get('lds_conflict_share', 0):.3f}")
