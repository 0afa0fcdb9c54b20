#!/bin/bash
# A/B of BVH builder settings (experiment build of the library, scripts/ab_build.sh bvhx) on ONE box:
# scripts/ab_bvh.sh <log>; per setting the 261 k-triangle mesh render, RGB and spectral, 1920x1080@64, two rounds.
log=$1
: > $log
export MTSAMD_LIB=$PWD/build/ab/libmtsamd_bvhx.so
run() {
    echo "== $1" >> $log
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 2>&1 | grep spp >> $log || exit 1
    timeout -k 10 200 python scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant spectral 2>&1 | grep spp >> $log || exit 1
}
for round in 1 2; do
    run "default (16 bins, cost 1.5, leaf default)"
    MTSAMD_BVH_BINS=32 run "bins 32"
    MTSAMD_BVH_SWEEP=64 run "sweep<=64"
    MTSAMD_BVH_SWEEP=64 MTSAMD_BVH_BINS=32 run "bins 32 + sweep<=64"
    MTSAMD_BVH_ICOST=1.0 run "icost 1.0"
    MTSAMD_BVH_ICOST=0.7 MTSAMD_BVH_SWEEP=64 run "icost 0.7 + sweep<=64"
    MTSAMD_BVH_ICOST=2.5 MTSAMD_BVH_SWEEP=64 run "icost 2.5 + sweep<=64"
done
