#!/bin/bash
# A/B of environment switches of ONE experiment build on one box: scripts/ab_env.sh <log> <lib name> "<VAR=value ...>" ["<VAR=value ...>" ...]
# per setting the 261 k-triangle mesh render, RGB and spectral, 1920x1080@64, two rounds ("-" = no switch).
log=$1; shift
export MTSAMD_LIB=$PWD/build/ab/libmtsamd_$1.so; shift
: > $log
for round in 1 2; do
for setting in "$@"; do
    echo "== $setting (round $round)" >> $log
    if [ "$setting" = "-" ]; then setting=""; fi
    env $setting timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 2>&1 | grep spp >> $log || exit 1
    env $setting timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant spectral 2>&1 | grep spp >> $log || exit 1
done
done
