#!/bin/bash
# round 5 profile set: prof_round.sh (bench + extras, kernel stats, traffic, PMC summary, diffusion kernel stats) + the configs[4]
# leg's PMC summary and both legs' traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${TAG:-r05_v1} bash scripts/prof_round.sh
O=gpurun_out/${TAG:-r05_v1}
PMC_TARGET="scripts/bench_upscale_quick.py 4 4" python scripts/pmc_summary.py 4 $O/upscale_pmc_summary.json > $O/upscale_pmc_summary.txt 2>&1; head -8 $O/upscale_pmc_summary.txt | cut -c1-230
bash scripts/prof_upscale.sh 8 4 > $O/prof_upscale.log 2>&1; cp gpurun_out/up_prof/*kernel_stats.csv $O/upscale_kernel_stats.csv 2>/dev/null
for leg in diffusion upscale; do python scripts/collect_leg_traffic.py $leg > $O/${leg}_traffic.log 2>&1; cp gpurun_out/${leg}_traffic.json $O/${leg}_traffic.json 2>/dev/null; tail -2 $O/${leg}_traffic.log | cut -c1-250; done
rm -rf gpurun_out/up_prof gpurun_out/diffusion_traffic gpurun_out/upscale_traffic
