#!/bin/bash
# tools/microbench/build_loops.sh <tag> [-D...] — builds tools/microbench/loops_<tag> (gfx950) with the product's flags
root=$(cd "$(dirname "$0")/../.." && pwd); tag=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
  -fno-slp-vectorize -fno-vectorize -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-value -I $root/include -I $root/cuda-path-tracer-ss_amd/csrc \
  "$@" $root/tools/microbench/loops.hip -o $root/tools/microbench/loops_$tag
