#!/bin/bash
# A/B builds of the library: tools/build_variant.sh NAME [-DSVO_...=n ...]  ->  build_ab/NAME.so  (run with SVO_HIP_LIB, tools/ab2.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../octree-tracer_amd/csrc"
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -I../../include -Wall -Wno-unused-function "$@" \
  -shared -o ../../build_ab/$name.so svo_kernels.hip svo_abi.cpp svo_comm.cpp svo_host.cpp -ldl
echo built build_ab/$name.so "$@"
