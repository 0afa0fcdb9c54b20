#!/bin/bash
# CPU-only: the oracle built with AddressSanitizer + UndefinedBehaviorSanitizer, and every CPU test that drives it
# (GPU sanitizers are not available on the pool).  Round 3: 197 tests, no report.
set -e
cd "$(dirname "$0")/.."
out=$(mktemp -d /tmp/oracle_asan_XXXX)
(cd oracle && gcc -std=c11 -O1 -g -fPIC -fopenmp -mfma -mavx2 -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $out/libmts_oracle.so mo_scene.c mo_render.c mo_spectral.c mo_bsdf.c mo_envmap.c mo_packet.c -lm)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) MTS_ORACLE_LIB=$out/libmts_oracle.so OMP_NUM_THREADS=4 \
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider \
    -k "oracle or bsdf_cpu or envmap_cpu or spectral_cpu or ingest_cpu or adjoint_cpu or libm_cpu or packet"
