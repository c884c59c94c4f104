# round 6: the default bench line + a short summary of its extra legs (gpurun_out/bench_r6.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
python -X faulthandler bench.py "$@" > gpurun_out/bench_r6.json 2> gpurun_out/bench_r6.err
echo "bench.py exit code $? wall seconds: $(( $(date +%s) - t0 ))"
tail -25 gpurun_out/bench_r6.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r6.json"))
print("frames/s", r["value"], "ms/step", r["ms_per_step"], "roofline", {k: r["roofline"].get(k) for k in ("achieved", "frac", "traffic")})
d = r.get("diffusion", {})
pick = lambda v: {k: v.get(k) for k in ("value", "error", "seconds_per_batch", "hipgraph", "finite", "batch") if k in v} if isinstance(v, dict) else v
print("diffusion", {k: pick(v) for k, v in d.items() if k in ("value", "guided_clip", "guided_regular", "unguided", "error")})
print("  clip roofline", d.get("guided_clip", {}).get("roofline"))
print("  traffic", d.get("roofline", {}).get("traffic"), d.get("roofline", {}).get("traffic_note"))
u = r.get("upscale", {})
print("upscale", u.get("value"), u.get("error"), u.get("roofline", {}).get("traffic"), u.get("roofline", {}).get("traffic_note"))
PY
