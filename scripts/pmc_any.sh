# PMC counters for kernels matching $1 (scripts/profile_layers.py 16); remaining args = counter names (one pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pat=$1; shift
out=gpurun_out/pmc_any_$$
mkdir -p $out
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o k1 -- python scripts/profile_layers.py ${BATCH:-16} > $out/k1.log 2>&1
python - "$pat" "$out" "$@" <<'PY'
import csv, collections, glob, sys
pat, out, names = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = list(csv.DictReader(open(glob.glob(out + "/k1_counter_collection.csv")[0])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(names)
seen = collections.Counter()
for k, d in disp.items():
    if pat in d["name"]:
        seen[d["name"]] += 1
        if seen[d["name"]] == 3:  # a warmed-up launch of each matching kernel
            print(d["name"][6:60], " ".join(f"{d.get(c, 0):.4g}" for c in names))
PY
