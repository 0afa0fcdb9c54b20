#!/bin/bash
# Runs on the GPU box (inside gpurun): SQ counters and HBM traffic (separate --pmc passes) of the split pipeline's kernels on the
# 261 k-triangle mesh in the SPECTRAL variant (the variant of BASELINE config 3; k_shade<PathStateS> is its largest kernel by summed time).
# Output: gpurun_out/prof_mesh_spec/{sq,fetch,write}; scripts/summarize_mesh_pmc.py spec -> profiles/spec_mesh_pmc.json (renamed by hand).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mesh_spec
rm -rf $OUT
for c in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  name=${c%%:*}; ctr=${c#*:}
  mkdir -p $OUT/$name
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant spectral > $OUT/$name.log 2>&1 || exit 1
done
echo collected $OUT
