#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 kernel-trace summary of the secondary benchmark on the 261 k-triangle mesh
# (split pipeline).  Output: gpurun_out/prof_mesh_<tag>/ ; copy the *_kernel_stats.csv to profiles/<tag>_mesh_kernel_stats.csv.
set -o pipefail
# VARIANT=spectral in the environment: the spectral variant (BASELINE config 3), output under gpurun_out/prof_mesh_<tag>_spectral/
TAG=${1:-r01}
V=${VARIANT:-rgb}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mesh_$TAG
[ "$V" = "rgb" ] || OUT=${OUT}_$V
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant $V > $OUT.log 2>&1 || exit 1
tail -n 1 $OUT.log
# HBM traffic and SQ counters of the split pipeline's kernels (separate --pmc passes); scripts/summarize_mesh_pmc.py -> profiles/<tag>_mesh_pmc.json
if [ "$2" = "pmc" ]; then
  for c in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
    name=${c%%:*}; ctr=${c#*:}
    mkdir -p $OUT/$name
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant $V > $OUT/$name.log 2>&1 || exit 1
  done
  echo collected $OUT
fi
# the mesh line of bench.py itself (spectral, 1920x1080@1024 spp): kernel-trace summary whose k_trace<false,false> average must agree with the
# line's k_trace_closest_avg_launch_us -> profiles/<tag>_mesh_bench_kernel_stats.csv, profiles/<tag>_mesh_bench.json
if [ "$2" = "bench" ] || [ "$3" = "bench" ]; then
  mkdir -p $OUT/bench
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --config mesh --no-cpu-baseline --no-parity --steps 1 --warmup 1 > $OUT/bench.log 2>&1 || exit 1
  echo collected $OUT/bench
fi
