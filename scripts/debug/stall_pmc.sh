#!/bin/bash
# on the GPU box: where the waves of the split pipeline's kernels spend their cycles (SQ wait / active counters, two --pmc passes on the
# RGB mesh render at 64 spp) -> gpurun_out/stall_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stall_pmc
rm -rf $OUT; mkdir -p $OUT/a $OUT/b
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 > $OUT/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/b -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 > $OUT/b.log 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/stall_pmc.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/stall_pmc/*/*/*counter_collection.csv") + glob.glob("gpurun_out/stall_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("k_trace<false, false>", "k_trace<true, false>", "k_shade", "k_film_accum"):
            if key in name:
                tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in tot.items():
    print("==", k)
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    for c, x in sorted(v.items()):
        print("  %-26s %16.0f  %s" % (c, x, ("%.3f of SQ_WAVE_CYCLES" % (x / wc)) if wc and c != "SQ_WAVE_CYCLES" else ""))
PY
rm -rf $OUT
cat gpurun_out/stall_pmc.txt
