cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
tail -3 gpurun_out/r6_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","sustained","e2e","gather_ms","cpu_baseline")})
print(d["roofline"])
for k,v in d["kernels"].items(): print(k, {a:round(b,3) for a,b in v.items()})
PY
