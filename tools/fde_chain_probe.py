#!/usr/bin/env python3
"""One FDE_THEN_FLOAT request at a time (coarse list = the reference's min(10 k, 75)), for a rocprofv3 --kernel-trace run:
python tools/fde_chain_probe.py [pages] [requests] [stats 0|1].  stats = 1 asks for the HIP-event stage split (five event records inside
the chain); stats = 0 is what a serving request runs (no events: read its span from the kernel trace).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    stats = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
    ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=False, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, n)
    qs = [synth_rows(4321, j, 32) for j in range(8)]
    rows, wall = [], []
    for r in range(reps):
        t0 = time.perf_counter()
        res = ix.query(qs[r % 8], 10, mode="fde_then_float", want_stats=stats)
        wall.append(time.perf_counter() - t0)
        if stats and r >= reps // 4:
            st = res[2]
            rows.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms, st.total_device_ms))
    out = {"pages": n, "requests": reps, "stats_requested": stats, "wall_ms_per_request_median": round(float(np.median(wall[reps // 4:])) * 1e3, 4)}
    if rows:
        m = np.median(np.array(rows), axis=0)
        out["stage_ms"] = dict(zip(("encode", "coarse", "select", "rerank", "topk", "total_device"), [round(float(x), 4) for x in m]))
        out["everything_but_the_coarse_scan_ms"] = round(float(m[5] - m[1]), 4)
        out["coarse_GBps"] = round(n * 20480 / float(m[1]) / 1e6, 1)
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
