#!/usr/bin/env python3
"""gpurun_out/prof_mesh_<tag>/{fetch,write,sq} (scripts/collect_mesh_profile.sh <tag> pmc) -> profiles/<tag>_mesh_pmc.json:
per kernel of the split pipeline on the 261 k-triangle mesh (1920x1080 @ 64 spp, all launches of the script summed): HBM bytes
(FETCH_SIZE corrected x2 as in summarize_profiles.py, WRITE_SIZE exact, both KiB) and the SQ counters with the VALU lane
utilisation."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_mesh_" + tag)


def short(name):
    m = re.search(r"(k_\w+)(<[^(]*>)?", name)
    if not m:
        return name[:60]
    return m.group(1) + (m.group(2) or "").replace("mtsamd::", "")


def counters(sub):
    f = max(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


fetch, nf = counters("fetch")
write, _ = counters("write")
sq, _ = counters("sq")
out = {"command": "rocprofv3 --pmc <counters> -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 (one pass per counter group)",
       "bench_line": open(os.path.join(src, "sq.log")).read().strip().splitlines()[-1], "kernels": {}}
for k in sorted(sq, key=lambda k: -sq[k].get("SQ_WAVE_CYCLES", 0.0)):
    s = dict(sq[k])
    e = {"launches": nf.get(k, 0),
         "hbm_read_bytes_corrected": 2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0,
         "hbm_write_bytes": write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0, "sq": s}
    if s.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_utilisation"] = s["SQ_THREAD_CYCLES_VALU"] / (s["SQ_ACTIVE_INST_VALU"] * 64.0)
    if s.get("SQ_WAVES"):
        e["valu_instructions_per_wave"] = s.get("SQ_INSTS_VALU", 0.0) / s["SQ_WAVES"]
    out["kernels"][k] = e
# HBM traffic of the closest-hit walk against its algorithmic bytes (48 B per ray + the geometry once per launch): the
# script renders the 1920x1080@64 film twice (warm-up + timed); one closest-hit ray per path segment
for line in open(os.path.join(src, "fetch.log")):
    m = re.search(r"(\d+)x(\d+)@(\d+)spp:.* ([0-9.]+) segments/sample", line)
    if m:
        bench = [x for x in out["kernels"] if x.startswith("k_trace<false")][0]
        rays = 2.0 * int(m.group(1)) * int(m.group(2)) * int(m.group(3)) * float(m.group(4))
        geometry = 32.0 * 160979 + 48.0 * 261124         # as bench.py prices it: nodes + triangle slots of the 261 k-triangle scene (scene.info())
        e = out["kernels"][bench]
        alg = 48.0 * rays + geometry * e["launches"]
        out["k_trace_closest"] = {"rays": rays, "algorithmic_bytes": alg, "counted_bytes": e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"],
                                  "traffic_over_algorithmic": (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / alg}
        out["bench_line"] = line.strip()
json.dump(out, open(os.path.join(root, "profiles", tag + "_mesh_pmc.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print("%-60s launches %5d  read %.2f GB  write %.2f GB  lanes %.2f" % (k, e["launches"], e["hbm_read_bytes_corrected"] / 1e9, e["hbm_write_bytes"] / 1e9, e.get("valu_lane_utilisation", 0.0)))
