# same-box ablation timing: bench.py kernel table for each build_ab/libmaua_abl*.so (timing only, results are wrong)
cd $GRAFT_REPO_ROOT
for lib in "" $(ls build_ab/libmaua_abl*.so); do
  if [ -n "$lib" ]; then export MAUA_HIP_LIB=$PWD/$lib; fi
  python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read())
print(sys.argv[1] or "base", round(d["value"]), {k.replace("modconv_hires_kernel", "hires"): round(v["ms_per_launch"], 3) for k, v in d["kernels"].items() if "hires" in k})
PY
done
