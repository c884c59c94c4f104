# per-kernel split of one CLIPGrads call (ViT-B/16, batch 32, 32 cutouts x 2 cutout batches): rocprofv3 kernel trace -> gpurun_out/
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/bench_clip.py --batches ${BATCHES:-8} --reps 2 2>&1 | tail -2 | tee gpurun_out/bench_clip.txt
python scripts/bench_clip.py --batches 2 --reps 2 --dma 0 2>&1 | tail -1 | tee -a gpurun_out/bench_clip.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_clip -o clip -- python scripts/bench_clip.py --batches 2 --reps 1 > gpurun_out/prof_clip.log 2>&1
f=$(find gpurun_out/prof_clip -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/clip_kernel_stats.csv && head -30 "$f"
