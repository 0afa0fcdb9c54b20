#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 kernel-trace summary of the secondary benchmark on the 261 k-triangle mesh
# (split pipeline).  Output: gpurun_out/prof_mesh_<tag>/ ; copy the *_kernel_stats.csv to profiles/<tag>_mesh_kernel_stats.csv.
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mesh_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 > $OUT.log 2>&1 || exit 1
tail -n 1 $OUT.log
