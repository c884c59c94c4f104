# round 2, first GPU pass for the LDS-direct-load conv kernel: parity, then the same-process A/B, then the whole suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "modconv" 2>&1 | tail -15 > gpurun_out/r1_ops.log
cat gpurun_out/r1_ops.log
timeout 600 python scripts/ab_synth.py dma_conv 32 5 > gpurun_out/r1_ab.log 2>&1
cat gpurun_out/r1_ab.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r1_all.log
cat gpurun_out/r1_all.log
