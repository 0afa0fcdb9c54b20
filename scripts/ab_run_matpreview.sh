#!/bin/bash
# A/B of kernel variants (scripts/ab_build.sh) on the general-BSDF hierarchy scene: scripts/ab_run_matpreview.sh <log> <name> ...
log=$1; shift
: > $log
for round in 1 2; do
for name in "$@"; do
    export MTSAMD_LIB=$PWD/build/ab/libmtsamd_$name.so
    echo "== $name (round $round)" >> $log
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --scene matpreview 2>&1 | grep spp >> $log || exit 1
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --scene matpreview --variant spectral 2>&1 | grep spp >> $log || exit 1
done
done
