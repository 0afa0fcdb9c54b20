#!/bin/bash
# Register / scratch use of the kernels whose name matches $1 (default k_trace): device-only compile of kernels.hip with the release flags.
#   scripts/kernel_regs.sh [pattern] [-DMACRO=value ...]
cd "$(dirname "$0")/../mitsuba2_amd/csrc" || exit 1
pat=${1:-k_trace}; shift
out=$(mktemp /tmp/kernels_XXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function \
    "$@" --cuda-device-only -S -o $out kernels.hip || exit 1
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $out | paste - - - - - - | grep "$pat" | sed 's/[ \t]\+/ /g'
echo "asm: $out"
