set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx" | head -4
python -m pytest tests -m gpu -x -q 2>&1 | tail -40
