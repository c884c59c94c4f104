# configs[4] leg alone under rocprofv3 (kernel stats as csv): bash scripts/prof_upscale.sh [frames] [upscaler batch]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/up_prof
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o up -- python scripts/bench_upscale_quick.py ${1:-8} ${2:-4} > $O/run.log 2>&1 < /dev/null
tail -1 $O/run.log | cut -c1-300
f=$(ls $O/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-200
