#!/bin/bash
# round 5, GPU call 2: guided loop parity + gn_apply rewrite (all diffusion tests), bench with the guided leg, diffusion kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5c2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_diffusion.py -x -q > $O/pytest_diffusion.log 2>&1; echo "pytest rc $?" >> $O/pytest_diffusion.log
tail -5 $O/pytest_diffusion.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c2/bench.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"])
x=d.get("diffusion", {})
for k in ("value","seconds_per_batch","hipgraph","finite","guided_over_unguided","guided_bf16_secondary_over_unguided"):
    print("diffusion", k, x.get(k))
print("unguided", x.get("unguided")); print("bf16sec", x.get("guided_bf16_secondary")); print("roof", x.get("roofline",{}).get("achieved"))
print("upscale", d.get("upscale",{}).get("value"))
PY
