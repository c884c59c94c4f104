# Profile set of a round: bench line (with the configs[3] / configs[4] extras and the CPU leg), rocprofv3 kernel stats of the
# same command, HBM traffic (PMC, separate passes, at the bench's own batch), SQ / GRBM summary, and the diffusion leg's own
# kernel stats.   usage (GPU box):  TAG=r03_v1 bash scripts/prof_round.sh    -> gpurun_out/$TAG/*
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-160
BATCH=128 python scripts/collect_traffic.py > $O/traffic.log 2>&1; cp gpurun_out/traffic.json $O/traffic.json; cat $O/traffic.log
python scripts/pmc_summary.py 128 $O/pmc_summary.json > $O/pmc_summary.txt 2>&1; head -14 $O/pmc_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profd -o d -- python scripts/bench_diffusion.py --batch 8 --steps 10 --reps 1 --no-graph > $O/diffusion_prof.log 2>&1
f=$(find $O/profd -name "*kernel_stats.csv" | head -1); cp "$f" $O/diffusion_kernel_stats.csv; head -8 $O/diffusion_kernel_stats.csv | cut -c1-160
rm -rf $O/prof $O/profd
