# sample clocks / power with rocm-smi while the render loop runs (is the chip power- or clock-limited?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps ${STEPS:-2500} --warmup 3 --batch 16 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '
  echo
  sleep 1
done | awk '!/\(1[0-9][0-9]Mhz\)|\(9[0-9]Mhz\)/' | tail -12
cat gpurun_out/power_bench.json | cut -c1-200
