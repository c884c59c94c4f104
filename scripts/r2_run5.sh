cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r5_all.log
cat gpurun_out/r5_all.log
