#!/bin/bash
# tools/isa_dump.sh [<template-args, e.g. 010010> [out.s]] [extra -D flags] — gfx950 assembly of one bounceKernel instantiation
# (digits = kLast kSceneInLds kFirst kAccel kBounded kPairs) with the shipped flags; prints VGPR/SGPR/scratch and instruction-class counts.
# Runs in this container (hipcc cross-compiles); used to read the hot loops, never by the product or the tests.
args=${1:-010010}; out=${2:-/tmp/isa/k_$args.s}; shift; shift
mkdir -p /tmp/isa
root=$(cd "$(dirname "$0")/.." && pwd)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
  -fno-slp-vectorize -fno-vectorize -ffp-contract=off -fno-fast-math -I $root/include -I $root/cuda-path-tracer-ss_amd/csrc \
  --cuda-device-only -S $root/cuda-path-tracer-ss_amd/csrc/ptss_kernels.hip -o /tmp/isa/all.s "$@" 2>/dev/null || exit 1
tag="bounceKernelILb${args:0:1}ELb${args:1:1}ELb${args:2:1}ELb${args:3:1}ELb${args:4:1}ELb${args:5:1}E"
awk -v t="$tag" '$0 ~ "^_ZN4ptss12" t ".*:" {p=1} p{print} p && /s_endpgm/ {e=1} p && e && /^\.Lfunc_end/ {exit}' /tmp/isa/all.s > $out
echo "$out: $(grep -c '^\s*v_' $out) VALU, $(grep -c '^\s*s_' $out) SALU, $(grep -c '^\s*ds_' $out) LDS, $(grep -c '^\s*global_' $out) VMEM, $(grep -c scratch_ $out) scratch"
grep -A40 "amdhsa_kernel _ZN4ptss12$tag" /tmp/isa/all.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed_size" 
