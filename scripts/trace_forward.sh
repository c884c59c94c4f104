cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2 -o p -- python scripts/profile_layers.py 16 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
rows=list(csv.DictReader(open(glob.glob("gpurun_out/prof2/p_kernel_trace.csv")[0])))
# last forward: take last N dispatches after the final 'styles_affine'
idx=[i for i,r in enumerate(rows) if 'styles_affine' in r['Kernel_Name']][-1]
for r in rows[idx:]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{d:9.1f} us  grid {r['Grid_Size']:>9s} wg {r['Workgroup_Size']:>4s} {r['Kernel_Name'][:70]}")
PY
