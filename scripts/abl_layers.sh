# same-box ablation timing per layer: scripts/profile_layers.py rows matching $1 for each build_ab/libmaua_abl*.so
cd $GRAFT_REPO_ROOT
pat=${1:-"bs\.[0-2]\."}
for lib in "" $(ls build_ab/libmaua_abl*.so); do
  if [ -n "$lib" ]; then export MAUA_HIP_LIB=$PWD/$lib; fi
  echo "== ${lib:-base}"
  python scripts/profile_layers.py ${BATCH:-32} 2>&1 | grep -E "$pat" | cut -c1-48
done
