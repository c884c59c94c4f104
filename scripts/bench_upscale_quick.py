"""GPU box: the configs[4] leg of bench.py on its own (frames / upscaler batch from argv)."""
import json, sys
sys.path.insert(0, ".")
import bench
fr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ub = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print(json.dumps(bench.extra_upscale(steps=2, frames=fr, upscale_batch=ub))[:400])
