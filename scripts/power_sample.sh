# socket power / clocks while the bench's timed loop runs (evidence for DESIGN 10.4: power-limited clock).  usage: bash scripts/power_sample.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Max Graphics" ; sleep 1; done > gpurun_out/power_samples.txt
wait $BP
cat gpurun_out/power_samples.txt | head -40
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4
