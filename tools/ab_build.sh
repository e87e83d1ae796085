#!/bin/bash
# A/B timing aid: build the current tree's kernels as build/ab/libpt_hip_<tag>.so (optionally with extra -D flags);
# run variants on the GPU box with  PT_HIP_LIB=build/ab/libpt_hip_<tag>.so python3 tools/prof_run.py ...
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/ab
C="--offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -Ipathtracer_amd/csrc"
hipcc $C -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c pathtracer_amd/csrc/pt_kernels.hip -o build/ab/k_$tag.o
hipcc $C -ffp-contract=off "$@" -c pathtracer_amd/csrc/pt_kernels_fast.hip -o build/ab/kf_$tag.o
hipcc --offload-arch=gfx950 -shared -fPIC build/ab/k_$tag.o build/ab/kf_$tag.o -o build/ab/libpt_hip_$tag.so
rm -f build/ab/k_$tag.o build/ab/kf_$tag.o
echo built build/ab/libpt_hip_$tag.so
