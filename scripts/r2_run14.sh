cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -4
timeout 600 python scripts/ab_synth.py tconv_up 32 3 1,512 > gpurun_out/r14_ab.log 2>&1
cat gpurun_out/r14_ab.log
