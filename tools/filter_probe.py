#!/usr/bin/env python3
"""Scan time under doc_ids filters of different selectivity (real deployments always pass the authorised doc set)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd.index import MvIndex, synth_rows, allow_bitmap

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
ix = MvIndex(capacity_pages=n, stride_rows=1024, with_binary=True)
ix.fill_synthetic(1234, 0, n, pages_per_doc=10)
q = synth_rows(4321, 0, 32)
n_docs = n // 10
rng = np.random.default_rng(0)
res = {}
for frac in (1.0, 0.5, 0.1, 0.01, 0.001):
    docs = np.sort(rng.choice(n_docs, size=max(1, int(n_docs * frac)), replace=False))
    allow = allow_bitmap(docs.tolist(), n_docs)
    for mode in ("float", "binary"):
        ts, ks = [], []
        for r in range(6):
            t = time.perf_counter(); s, i, st = ix.query(q, 10, mode=mode, allow=allow, want_stats=True); dt = time.perf_counter() - t
            if r: ts.append(dt); ks.append(st.score_kernel_ms)
        res[f"{mode}_{frac}"] = {"wall_ms": round(float(np.median(ts)) * 1e3, 3), "scan_kernel_ms": round(float(np.median(ks)), 3),
                                 "pages_scanned": int(st.pages_scored), "ideal_ms_at_7.2TBps": round(st.bytes_scanned / 7.2e9, 3)}
print(json.dumps(res, indent=1))
