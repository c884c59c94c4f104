#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c5; export TMPDIR=/tmp
O=gpurun_out/r5c5
python scripts/e2e_setup_phases.py 2>&1 | grep -v amdgpu > $O/setup_phases.txt; cat $O/setup_phases.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd -o d -- python scripts/bench_diffusion.py --batch 8 --steps 10 --reps 1 --no-graph > $O/diffusion_prof.log 2>&1
f=$(find $O/profd -name "*kernel_stats.csv" | head -1); cp "$f" $O/diffusion_kernel_stats.csv; head -12 $O/diffusion_kernel_stats.csv | cut -c1-170; rm -rf $O/profd
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c5/bench.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"], "e2e", {k: d["e2e"][k] for k in ("setup_s","process_warmup_s","fps","fps_cold")})
x=d.get("diffusion", {})
for k in ("value","seconds_per_batch","hipgraph","finite","guided_over_unguided","guided_bf16_secondary_over_unguided","prompt_switches_in_timed_frames"):
    print("diffusion", k, x.get(k))
print("unguided", x.get("unguided")); print("bf16sec", x.get("guided_bf16_secondary")); print("roof", x.get("roofline",{}).get("achieved"))
print("upscale", d.get("upscale",{}).get("value"))
PY
