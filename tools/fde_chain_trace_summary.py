#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of tools/fde_chain_probe.py -> the kernel sequence of ONE request (the last complete one): name, duration,
gap to the previous kernel's end.  python tools/fde_chain_trace_summary.py <trace dir> <out.json>"""
import csv
import glob
import json
import os
import sys


def main(d, out):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a request starts at the query-side prep kernel that precedes the FDE query encode; find the last three "fde_scan" kernels
    scans = [i for i, r in enumerate(rows) if "fde_scan_rowq_kernel" in r[2] or "fde_scan_kernel" in r[2]]
    if len(scans) < 3:
        json.dump({"error": "no FDE scan kernels in the trace", "kernels": len(rows)}, open(out, "w"))
        return
    lo, hi = scans[-3], scans[-2]
    # walk back from the scan to the first kernel of its request (everything after the previous request's last kernel)
    prev_end_idx = lo
    while prev_end_idx > 0 and rows[prev_end_idx][0] - rows[prev_end_idx - 1][1] < 20_000:  # < 20 us apart: same request
        prev_end_idx -= 1
    nxt = hi
    while nxt > lo and rows[nxt][0] - rows[nxt - 1][1] < 20_000:
        nxt -= 1
    seq = rows[prev_end_idx:nxt]
    t0 = seq[0][0]
    chain = []
    prev_end = None
    for s, e, name in seq:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        chain.append({"kernel": short[:90], "start_us": round((s - t0) / 1e3, 2), "dur_us": round((e - s) / 1e3, 2),
                      "gap_before_us": None if prev_end is None else round((s - prev_end) / 1e3, 2)})
        prev_end = e
    span = (seq[-1][1] - t0) / 1e3
    busy = sum(c["dur_us"] for c in chain)
    res = {"kernels_in_request": len(chain), "span_us": round(span, 2), "sum_of_kernel_durations_us": round(busy, 2),
           "sum_of_gaps_us": round(span - busy, 2), "chain": chain}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:4000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
