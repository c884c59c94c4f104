cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/ab_synth.py tconv_dma 32 3 1,3,4 > gpurun_out/r13_ab.log 2>&1
cat gpurun_out/r13_ab.log
