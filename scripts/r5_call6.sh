#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c6; export TMPDIR=/tmp
O=gpurun_out/r5c6
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_all.log 2>&1; echo "rc $?" >> $O/pytest_gpu_all.log; tail -6 $O/pytest_gpu_all.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-extras > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c6/bench.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"], "e2e", {k: d["e2e"][k] for k in ("setup_s","process_warmup_s","fps","fps_cold")}, "ok", d["ok"])
PY
python - <<'PY'
import time, torch, sys
sys.path.insert(0, ".")
import bench
from maua_amd import pipeline
torch.set_num_threads(8)
torch.zeros(1, device="cuda"); pipeline.warm_up("cuda"); torch.cuda.synchronize()
for host in (True, False, True, False):
    t0 = time.perf_counter(); r = bench.build_inputs(torch.device("cuda", 0), 0, 1, host_rng=host); r[0]._handle(); torch.cuda.synchronize()
    print("build_inputs host_rng", host, f"{time.perf_counter() - t0:.4f} s"); del r
t0 = time.perf_counter(); pipeline.synthetic_clip_latents(3600, 30, 18, 512); torch.cuda.synchronize(); print("clip chain alone", f"{time.perf_counter() - t0:.4f}")
PY
