# rocprofv3 PMC pass over one profiled forward (B from $1); prints per-dispatch counters for the conv kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${1:-16}
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc -o p1 -- python scripts/profile_layers.py $B > gpurun_out/pmc/run1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc -o p2 -- python scripts/profile_layers.py $B > gpurun_out/pmc/run2.log 2>&1
ls gpurun_out/pmc
python - <<'PY'
import csv, collections, glob
for tag in ["p1","p2"]:
    f = glob.glob(f"gpurun_out/pmc/{tag}_counter_collection.csv")
    if not f: print("no", tag); continue
    rows = list(csv.DictReader(open(f[0])))
    # last forward only: group by dispatch id
    disp = collections.OrderedDict()
    for r in rows:
        k = int(r["Dispatch_Id"])
        d = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size"), "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size")})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ks = [k for k in disp if "modconv3x3" in disp[k]["name"]]
    ks = ks[-17:]
    names = [c for c in disp[ks[0]] if c not in ("name","grid","vgpr","lds")]
    print(tag, names)
    for k in ks:
        d = disp[k]
        print(d["name"][28:60], d["grid"], d["vgpr"], d["lds"], " ".join(f"{d.get(c,0):.3g}" for c in names))
PY
