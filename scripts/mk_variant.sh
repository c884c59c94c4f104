#!/bin/bash
# Build build_ab/lib_<name>.so = the tree's objects with ONE source file recompiled under extra flags (timing-only / experiment
# builds for same-box A/B runs):   scripts/mk_variant.sh skipfir modconv_tconv_fir.hip -DTF_SKIP=2
set -e
name=$1; f=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab
cd $root/maua_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I$root/include -I$root/maua_amd/csrc "$@" -c $f -o $root/build_ab/${name}_${f%.hip}.o
objs=$(ls *.o | grep -v "^${f%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_ab/lib_${name}.so $objs $root/build_ab/${name}_${f%.hip}.o
echo $root/build_ab/lib_${name}.so
