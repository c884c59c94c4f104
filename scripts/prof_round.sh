# round-2 profile set: bench line, rocprofv3 kernel stats of the same command, HBM traffic (PMC), SQ/GRBM summary
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
tail -2 gpurun_out/r02/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_prof.json 2> gpurun_out/r02/prof.err
f=$(find gpurun_out/r02/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02/kernel_stats.csv; head -12 gpurun_out/r02/kernel_stats.csv
python scripts/collect_traffic.py > gpurun_out/r02/traffic.log 2>&1; cp gpurun_out/traffic.json gpurun_out/r02/traffic.json; cat gpurun_out/r02/traffic.log
python scripts/pmc_summary.py 32 gpurun_out/r02/pmc_summary.json > gpurun_out/r02/pmc_summary.log 2>&1; head -14 gpurun_out/r02/pmc_summary.log
rm -rf gpurun_out/r02/prof
