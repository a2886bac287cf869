#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (kernel-trace / stats / pmc CSVs) into one small JSON."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(d, out):
    res = {"dir": d, "kernel_stats": [], "counters": {}}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res["kernel_stats"].append(row)
    # per-dispatch trace -> per-kernel avg duration
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row.get("Kernel_Name", "?")
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            agg[n][0] += 1
            agg[n][1] += dur
    res["kernel_trace_avg_us"] = {k: {"calls": v[0], "avg_us": v[1] / v[0], "total_us": v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    cnt = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            c = row.get("Counter_Name", "?")
            v = float(row.get("Counter_Value", 0))
            cnt[k][c][0] += 1
            cnt[k][c][1] += v
    res["counters"] = {k: {c: {"dispatches": v[0], "sum": v[1], "avg": v[1] / max(v[0], 1)} for c, v in cs.items()} for k, cs in cnt.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernel_trace_avg_us"], indent=1)[:3000])
    print(json.dumps(res["counters"], indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
