cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/ab_synth.py dma_conv 32 3 1,6 > gpurun_out/r8_ab.log 2>&1
cat gpurun_out/r8_ab.log
