#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (scripts/collect_profiles.sh) into the summaries committed under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
  profiles/<tag>_bench.json            the bench line printed by that same run
  profiles/<tag>_pmc_bounce_kernel.json     per-launch HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the
                                            kernel that advances the paths by one segment (bench line: roofline.kernel)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))


def bench_line(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit("no bench line in " + log)


json.dump(bench_line(os.path.join(src, "trace.log")), open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)


KERNEL = bench_line(os.path.join(src, "trace.log"))["roofline"]["kernel"]
KEY = KERNEL.split("<")[0]                      # k_shade / k_bounce: the only instantiation the bench command launches


def counters(sub):
    f = max(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(float)
    launches = set()
    for r in csv.DictReader(open(f)):
        if KEY + "<" in r["Kernel_Name"] or KEY + "(" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add(r["Dispatch_Id"])
    return dict(agg), len(launches), bench_line(os.path.join(src, sub + ".log"))


fetch, n_f, b_f = counters("fetch")
write, n_w, b_w = counters("write")
sq, n_s, b_s = counters("sq")
# one scheduler iteration (the bench line's "launch") is issued as `concurrent` kernels that run side by side
conc = int(b_f["kernel_ms"].get("bounce_kernel_concurrent", 1))
n_f, n_w, n_s = n_f // conc, n_w // conc, n_s // conc
# FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE reads exactly 1/2 of a 16-B-per-lane coalesced stream
# (MI355X_MICROARCH.md, HBM section) -> corrected read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-B stores.
alg_per_launch = b_f["roofline"]["alg_bytes_per_launch"]
out = {
    "command": "rocprofv3 --pmc <counter> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (one pass per TCC counter)",
    "kernel": KERNEL,
    "launches": n_f,
    "kernels_per_launch": conc,
    "fetch_size_kib_raw": fetch.get("FETCH_SIZE", 0.0),
    "write_size_kib_raw": write.get("WRITE_SIZE", 0.0),
    "hbm_read_bytes_per_launch_corrected": 2.0 * fetch.get("FETCH_SIZE", 0.0) * 1024.0 / max(n_f, 1),
    "hbm_write_bytes_per_launch": write.get("WRITE_SIZE", 0.0) * 1024.0 / max(n_w, 1),
    "algorithmic_bytes_per_launch": alg_per_launch,
    "sq": sq,
    "sq_launches": n_s,
}
out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch_corrected"] + out["hbm_write_bytes_per_launch"]
out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / alg_per_launch
if sq.get("SQ_ACTIVE_INST_VALU"):
    out["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_ACTIVE_INST_VALU"] * 64.0)
    seg = b_s["segments_per_sample"] * b_s["value"] * 1e6 * b_s["ms_per_step"] * 1e-3 * b_s["steps"]      # segments of the profiled run
    out["valu_instructions_per_wave_segment"] = sq["SQ_INSTS_VALU"] / (seg / 64.0)
json.dump(out, open(os.path.join(dst, tag + "_pmc_bounce_kernel.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
