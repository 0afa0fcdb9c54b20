set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_mesh; mkdir -p gpurun_out/pmc_mesh
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_mesh -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 16 > gpurun_out/pmc_mesh.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_mesh/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,v in agg.items():
    if v.get("SQ_ACTIVE_INST_VALU",0)>1e6:
        print(k, len(n[k]), "util %.3f"%(v["SQ_THREAD_CYCLES_VALU"]/(v["SQ_ACTIVE_INST_VALU"]*64)), "valu %.3g"%v["SQ_INSTS_VALU"], "wave_cycles %.3g"%v["SQ_WAVE_CYCLES"], "busy %.3g"%v["SQ_BUSY_CYCLES"])
PY
