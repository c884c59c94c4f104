cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_super.py tests/test_gpu_ops.py -x -q 2>&1 | tail -15
