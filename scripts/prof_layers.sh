# rocprofv3 kernel stats of scripts/profile_layers.py (single-stream synthesis only); B=${1:-16}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o p -- python scripts/profile_layers.py ${1:-16} > gpurun_out/prof3/log.txt 2>&1
tail -2 gpurun_out/prof3/log.txt
find gpurun_out/prof3 -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-150 "$f" | head -${2:-20}
