cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc -o k1 -- python scripts/profile_layers.py 16 > gpurun_out/pmc/k1.log 2>&1
python - <<'PY'
import csv, collections, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pmc/k1_counter_collection.csv")[0])))
disp = collections.OrderedDict()
for r in rows:
    k = int(r["Dispatch_Id"])
    d = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size"), "vgpr": r.get("VGPR_Count")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
ks = list(disp)[-60:]
names = ['SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_INSTS_VALU','SQ_INSTS_VMEM_RD','SQ_INSTS_LDS']
print(names)
for k in ks:
    d = disp[k]
    if any(s in d["name"] for s in ("upfir","tconv2","hires","modconv3x3")):
        print(d["name"][11:52], d["grid"], d["vgpr"], " ".join(f"{d.get(c,0):.3g}" for c in names))
PY
