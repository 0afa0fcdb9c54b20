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
variant = sys.argv[2] if len(sys.argv) > 2 else "rgb"          # "spectral": gpurun_out/prof_mesh_<tag>_spectral -> profiles/<tag>_mesh_spectral_pmc.json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_mesh_" + tag + ("" if variant == "rgb" else "_" + variant))


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
out = {"command": "rocprofv3 --pmc <counters> -- python3 scripts/bench_mesh.py --width 1920 --height 1080 --spp 64 --variant %s (one pass per counter group)" % variant,
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
# HBM traffic of the walks against their algorithmic bytes: closest hit 48 B per ray (32-B ray record read + 16-B hit
# written), any hit 36 B per ray (32-B shadow ray + 4-B slot; the radiance update of an unoccluded ray is extra), + the geometry once per
# launch.  The script renders the film twice (warm-up + timed); one closest-hit ray per path segment.  k_shade: bytes per path segment
# against the record it streams (RGB 88 B in + 88 B out, spectral 100 + 100; + the 16-B hit, + 52 B per queued shadow ray).
for line in open(os.path.join(src, "fetch.log")):
    m = re.search(r"(\d+)x(\d+)@(\d+)spp: ([0-9.]+) ms, .* ([0-9.]+) Mray/s, ([0-9.]+) segments/sample", line)
    if m:
        samples = 2.0 * int(m.group(1)) * int(m.group(2)) * int(m.group(3))
        rays = samples * float(m.group(6))
        all_rays = 2.0 * float(m.group(5)) * 1e6 * float(m.group(4)) * 1e-3          # closest + any, both renders
        any_rays = max(all_rays - rays, 0.0)
        geometry = 32.0 * 160979 + 48.0 * 261124         # as bench.py prices it: nodes + triangle slots of the 261 k-triangle scene (scene.info())
        for key, name, per_ray, n in (("k_trace_closest", "k_trace<false", 48.0, rays), ("k_trace_any", "k_trace<true", 36.0, any_rays)):
            k = [x for x in out["kernels"] if x.startswith(name)]
            if not k:
                continue
            e = out["kernels"][k[0]]
            alg = per_ray * n + geometry * e["launches"]
            counted = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
            out[key] = {"rays": n, "algorithmic_bytes": alg, "counted_bytes": counted, "traffic_over_algorithmic": counted / alg,
                        "read_bytes_per_ray": e["hbm_read_bytes_corrected"] / max(n, 1.0), "write_bytes_per_ray": e["hbm_write_bytes"] / max(n, 1.0)}
        k = [x for x in out["kernels"] if x.startswith("k_shade<")]
        if k:
            e = out["kernels"][k[0]]
            rec = 100.0 if variant == "spectral" else 88.0
            alg_seg = 2.0 * rec + 16.0 + 52.0 * any_rays / max(rays, 1.0)
            out["k_shade"] = {"kernel": k[0], "segments": rays, "read_bytes_per_segment": e["hbm_read_bytes_corrected"] / rays,
                              "write_bytes_per_segment": e["hbm_write_bytes"] / rays, "bytes_per_segment": (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / rays,
                              "algorithmic_bytes_per_segment": alg_seg,
                              "traffic_over_algorithmic": (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / rays / alg_seg}
        out["bench_line"] = line.strip()
json.dump(out, open(os.path.join(root, "profiles", tag + ("_mesh_pmc.json" if variant == "rgb" else "_mesh_%s_pmc.json" % variant)), "w"), indent=1)
for k, e in out["kernels"].items():
    print("%-60s launches %5d  read %.2f GB  write %.2f GB  lanes %.2f" % (k, e["launches"], e["hbm_read_bytes_corrected"] / 1e9, e["hbm_write_bytes"] / 1e9, e.get("valu_lane_utilisation", 0.0)))
for key in ("k_trace_closest", "k_trace_any", "k_shade"):
    if key in out:
        print(key, json.dumps(out[key]))
