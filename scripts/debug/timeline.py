#!/usr/bin/env python3
"""Timeline of the split pipeline from a rocprofv3 --kernel-trace CSV (kernel_trace.csv): for the LAST render in the trace, per kernel
kind the launches, the mean / max duration, the summed duration, and how many kernels were in flight on average."""
import csv
import sys
from collections import defaultdict


def short(name):
    for key in ("k_trace<false, false>", "k_trace<true, false>", "k_shade", "k_film_accum", "k_film_merge", "k_finish"):
        if key in name:
            return key
    return name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    # the last render: everything after the last k_film_merge but one
    merges = [i for i, e in enumerate(ev) if e[2] == "k_film_merge"]
    lo = merges[-2] + 1 if len(merges) >= 2 else 0
    ev = ev[lo:]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    print("last render: %d kernels over %.2f ms" % (len(ev), (t1 - t0) * 1e-6))
    per = defaultdict(list)
    for s, e, n in ev:
        per[n].append((s, e))
    for n, v in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        d = [(e - s) * 1e-3 for s, e in v]
        print("  %-28s launches %4d  mean %8.1f us  max %8.1f us  sum %8.2f ms" % (n, len(d), sum(d) / len(d), max(d), sum(d) * 1e-3))
    # kernels in flight
    pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
    busy = defaultdict(int)
    depth, prev = 0, pts[0][0]
    for t, dlt in pts:
        busy[depth] += t - prev
        depth += dlt
        prev = t
    tot = sum(busy.values())
    print("  kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(busy.items())))
    if len(sys.argv) > 2:        # dump the first N launches relative to t0
        for s, e, n in ev[: int(sys.argv[2])]:
            print("    %9.1f .. %9.1f us  %s" % ((s - t0) * 1e-3, (e - t0) * 1e-3, n))


if __name__ == "__main__":
    main()
