#!/bin/bash
# round 5, GPU call 3: guided loop fix + F16 boundary; then the full GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_diffusion.py -x -q -k "guided_loop" > $O/pytest_guided.log 2>&1; echo "rc $?" >> $O/pytest_guided.log; tail -4 $O/pytest_guided.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_synth.py -x -q -k "fp16" > $O/pytest_fp16.log 2>&1; echo "rc $?" >> $O/pytest_fp16.log; tail -15 $O/pytest_fp16.log
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_all.log 2>&1; echo "rc $?" >> $O/pytest_gpu_all.log; tail -15 $O/pytest_gpu_all.log
