#!/bin/bash
# A/B of kernel variants (scripts/ab_build.sh) on the headline render: scripts/ab_headline.sh <log> <name> ...
log=$1; shift
: > $log
for round in 1 2; do
for name in "$@"; do
    export MTSAMD_LIB=$PWD/build/ab/libmtsamd_$name.so
    echo "== $name (round $round)" >> $log
    timeout -k 10 200 python bench.py --only --no-cpu-baseline --no-parity --steps 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%.1f Msample/s, %.2f ms, film %.2f ms' % (d['value'], d['ms_per_step'], d['kernel_ms']['k_film_tiles_per_step']))" >> $log || exit 1
done
done
