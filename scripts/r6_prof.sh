#!/bin/bash
# round 6 profile set: prof_round.sh (bench line with every extra leg, rocprofv3 kernel stats of the same command, live PMC traffic,
# SQ / GRBM summary, diffusion kernel stats) + the configs[4] leg's kernel stats (no at::native kernels on the product path) + the text-guided
# grad module's kernel split + the per-kernel energy account + the CLIP tower's GEMM shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${TAG:-r06_v1} bash scripts/prof_round.sh
O=gpurun_out/${TAG:-r06_v1}
bash scripts/prof_upscale.sh 8 4 > $O/prof_upscale.log 2>&1; cp gpurun_out/up_prof/*kernel_stats.csv $O/upscale_kernel_stats.csv 2>/dev/null; head -8 $O/upscale_kernel_stats.csv | cut -c1-160
BATCHES=8 bash scripts/prof_clip.sh > $O/prof_clip.log 2>&1; cp gpurun_out/clip_kernel_stats.csv $O/clip_kernel_stats.csv; cp gpurun_out/bench_clip.txt $O/clip_bench.txt; cat $O/clip_bench.txt | cut -c1-200
python scripts/energy_by_kernel.py 128 $O/energy_by_kernel.txt > /dev/null 2>&1; tail -3 $O/energy_by_kernel.txt | cut -c1-200
python scripts/bench_gemm_dma.py > $O/gemm_dma_bench.txt 2>&1; tail -6 $O/gemm_dma_bench.txt
rm -rf gpurun_out/up_prof gpurun_out/prof_clip
