cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_audio.py tests/test_gpu_entry.py -x -q 2>&1 | tail -12
