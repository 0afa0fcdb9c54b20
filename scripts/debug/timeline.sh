#!/bin/bash
# on the GPU box: kernel timeline of the mesh render with the library build/ab/libmtsamd_$1.so -> gpurun_out/tl_$1.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1
export MTSAMD_LIB=$PWD/build/ab/libmtsamd_$name.so
OUT=gpurun_out/tl_$name
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 > $OUT.log 2>&1 || exit 1
f=$(find $OUT -name "*kernel_trace.csv" | head -n 1)
python3 scripts/debug/timeline.py $f 40 > gpurun_out/tl_$name.txt
rm -rf $OUT
head -n 12 gpurun_out/tl_$name.txt
