# smoke + bench + rocprofv3 kernel trace on the GPU box; copies summaries to gpurun_out/
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps ${STEPS:-20} --warmup 3 --batch ${BATCH:-32} > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 10 --warmup 2 --batch ${BATCH:-32} --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
tail -2 gpurun_out/prof.err
find gpurun_out/prof -type f | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f"
