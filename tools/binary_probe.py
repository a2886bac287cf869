#!/usr/bin/env python3
"""A/B of the sign-bit scan variants on a bits-only index (interleaved rounds, HIP-event kernel times)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd import _lib
from morphik_core_amd.index import MvIndex, synth_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5").split(",")]
qt = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=False, with_binary=True)
ix.fill_synthetic(1234, 0, n)
q = synth_rows(4321, 0, qt)
ref, times = None, {v: [] for v in variants}
for r in range(6):
    for v in variants:
        ix.set_option(_lib.MV_OPT_BINARY_VARIANT, v)
        s, i, st = ix.query(q, 10, mode="binary", want_stats=True)
        if ref is None:
            ref = (s, i)
        assert i.tolist() == ref[1].tolist() and s.tolist() == ref[0].tolist(), f"variant {v} disagrees"
        if r:
            times[v].append(st.score_kernel_ms)
res = {}
for v in variants:
    ms = float(np.median(times[v]))
    res[f"v{v}"] = {"kernel_ms": round(ms, 4), "GBps": round(n * 16384 / ms / 1e6), "Mpages_s": round(n / ms / 1e3, 1)}
    print(f"binary v{v}: {ms:.3f} ms  {n*16384/ms/1e6:.0f} GB/s  {n/ms/1e3:.0f} M pages/s", flush=True)
print(json.dumps({"pages": n, "q_rows": qt, "variants": res}))
