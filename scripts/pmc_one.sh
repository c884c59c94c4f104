# PMC counters for kernels matching $1 in scripts/profile_layers.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc1 -o k1 -- python scripts/profile_layers.py 16 > gpurun_out/pmc1/k1.log 2>&1
python - "$1" <<'PY'
import csv, collections, glob, sys
pat = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pmc1/k1_counter_collection.csv")[0])))
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    d = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size"), "lds": r.get("LDS_Block_Size"), "vgpr": r.get("VGPR_Count")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
names = ['SQ_WAVES','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_INSTS_VALU','SQ_INSTS_LDS']
print(names)
n = 0
for k, d in disp.items():
    if pat in d["name"]:
        print(d["name"][6:40], d["grid"], d["lds"], d["vgpr"], " ".join(f"{d.get(c,0):.3g}" for c in names))
        n += 1
        if n >= 4: break
PY
