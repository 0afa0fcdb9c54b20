#!/bin/bash
# Kernel experiments: build a variant of libmtsamd.so with extra -D flags into build/ab/ (loaded through MTSAMD_LIB).
#   scripts/ab_build.sh <name> [-DMACRO=value ...]
set -e
cd "$(dirname "$0")/../mitsuba2_amd/csrc"
name=$1; shift
out=../../build/ab
mkdir -p $out
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -fvisibility=hidden -DMTSAMD_EXPERIMENTS -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o $out/kernels_$name.o kernels.hip &
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c -o $out/api_$name.o api.cpp &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libmtsamd_$name.so $out/kernels_$name.o $out/api_$name.o bvh.o spectral_upsampling.o envmap.o
/bin/rm -f $out/kernels_$name.o $out/api_$name.o
echo built $out/libmtsamd_$name.so
