cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "dma" 2>&1 | tail -5
timeout 900 python scripts/pmc_summary.py 32 gpurun_out/pmc_r2a.json > gpurun_out/pmc_r2a.log 2>&1
cat gpurun_out/pmc_r2a.log
