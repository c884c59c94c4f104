#!/bin/bash
# Build an alternative libmaua_hip.so with ONE source file taken from a git ref, for same-box A/B timing:
#   scripts/ab_lib.sh HEAD modconv_hires.hip   ->  build_ab/libmaua_hip_A.so   (use with MAUA_HIP_LIB=...)
set -e
ref=$1; f=$2; root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab/src
git -C $root show $ref:maua_amd/csrc/$f > $root/build_ab/src/$f
cd $root/maua_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I$root/include -I$root/maua_amd/csrc -c $root/build_ab/src/$f -o $root/build_ab/${f%.hip}.o
objs=$(ls *.o | grep -v "^${f%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_ab/libmaua_hip_A.so $objs $root/build_ab/${f%.hip}.o
echo $root/build_ab/libmaua_hip_A.so
