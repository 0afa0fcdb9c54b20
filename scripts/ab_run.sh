#!/bin/bash
# A/B of kernel variants built by scripts/ab_build.sh on ONE box: scripts/ab_run.sh <log> <name> [<name> ...]
# per variant: the 261 k-triangle mesh render (RGB and spectral, 1920x1080@64) and the standalone ray streams.
log=$1; shift
: > $log
for round in 1 2; do
for name in "$@"; do
    export MTSAMD_LIB=$PWD/build/ab/libmtsamd_$name.so
    echo "== $name (round $round)" >> $log
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 2>&1 | grep spp >> $log || exit 1
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant spectral 2>&1 | grep spp >> $log || exit 1
    if [ $round = 1 ]; then timeout -k 10 200 python scripts/bench_traversal.py --scene mesh 2>&1 | grep -i "gray\|mray" >> $log || exit 1; fi
done
done
