#!/usr/bin/env python3
"""Per-query latency of the library at small / medium corpus sizes (host wall time around mv_query_topk)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import MvIndex, synth_rows

res = {}
for n in (1000, 10_000, 125_000):
    ix = MvIndex(capacity_pages=n, stride_rows=1024, with_binary=True)
    ix.fill_synthetic(1234, 0, n)
    q = synth_rows(4321, 0, 32)
    for mode in ("float", "binary"):
        for _ in range(5):
            ix.query(q, 10, mode=mode)
        ts = []
        for _ in range(200):
            t = time.perf_counter(); ix.query(q, 10, mode=mode); ts.append(time.perf_counter() - t)
        _s, _i, st = ix.query(q, 10, mode=mode, want_stats=True)
        res[f"{mode}_{n}"] = {"wall_us_med": round(float(np.median(ts)) * 1e6, 1), "wall_us_p99": round(float(np.percentile(ts, 99)) * 1e6, 1),
                              "scan_kernel_us": round(st.score_kernel_ms * 1e3, 1), "topk_us": round(st.topk_ms * 1e3, 1)}
    ix.close()
print(json.dumps(res, indent=1))
