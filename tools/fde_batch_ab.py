#!/usr/bin/env python3
"""A/B of the batched FDE coarse pass forms (MV_OPT_FDE_BATCH_VARIANT) in ONE process (interleaved rounds, stats.coarse_ms), identical answers asserted.
   python tools/fde_batch_ab.py [pages=1250000] [forms=0,6]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphik_core_amd import _lib as L
from morphik_core_amd.index import MvIndex, synth_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
forms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,6").split(",")]
ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
ix.fill_synthetic(1234, 0, n)
out_dim = ix.fde_config.output_dim
qs = [synth_rows(4321, j, 32) for j in range(32)]
ref = {}
times = {(f, B): [] for f in forms for B in (16, 32)}
for r in range(9):
    for f in forms:
        ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, f)
        for B in (16, 32):
            res, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
            key = B
            got = [(s.tolist(), i.tolist()) for s, i in res]
            if key not in ref:
                ref[key] = got
            if f != 2:  # form 2 drops the low halves of the query split: other scores by design
                assert got == ref[key], f"form {f} B{B} differs from form {forms[0]}"
            if r >= 2:
                times[(f, B)].append(st.coarse_ms)
out = {"pages": n}
for (f, B), ts in times.items():
    ms = float(np.median(ts))
    out[f"form{f}_B{B}"] = {"coarse_ms": round(ms, 4), "GBps": round(n * out_dim * 2 / ms / 1e6, 1), "frac_8TBps": round(n * out_dim * 2 / ms / 1e6 / 8000, 4)}
ts = []
for r in range(12):
    _s, _i, st = ix.query(qs[r % 32], 10, mode="fde", want_stats=True)
    ts.append(st.coarse_ms)
ms = float(np.median(ts[3:]))
out["single"] = {"coarse_ms": round(ms, 4), "GBps": round(n * out_dim * 2 / ms / 1e6, 1)}
print(json.dumps(out))
