set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mesh_spec
rm -rf $OUT; mkdir -p $OUT/sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant spectral > $OUT/sq.log 2>&1 || exit 1
echo done
