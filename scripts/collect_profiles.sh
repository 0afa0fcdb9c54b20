#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 kernel-trace summary of the default bench command and the HBM
# traffic counters of the bounce kernel (separate --pmc passes on the full 256-spp configuration, MI355X_MICROARCH.md "HBM" section).  Outputs land in
# gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into the files committed under profiles/.
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/trace $OUT/fetch $OUT/write $OUT/sq
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --only --no-parity --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --only --no-parity --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --only --no-parity --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 bench.py --only --no-parity --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.log 2>&1 || exit 1
echo collected $OUT
