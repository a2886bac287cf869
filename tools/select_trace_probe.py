#!/usr/bin/env python3
"""A short run of the batched FDE pipeline (32 requests per pass, reference candidate rule and 1000 candidates) on a shard-sized
FDE + e4m3 index of SHORT pages (16 rows: the FDE slab and the score matrix have their full size, the rerank slab is small),
meant to be run under `rocprofv3 --kernel-trace --stats`: per-kernel durations of the selection chain."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, n)
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    out = {"pages": n}
    for cn in (75, 1000):
        ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
        for _ in range(3):
            ix.query_batch(qs, 10, mode="fde_then_float")
        sel = []
        for _ in range(10):
            _r, st = ix.query_batch(qs, 10, mode="fde_then_float", want_stats=True)
            sel.append((st.coarse_ms, st.select_ms, st.rerank_ms, st.total_device_ms))
        m = np.median(np.array(sel), axis=0)
        out[f"coarse{cn}"] = dict(zip(("coarse_ms", "select_ms", "rerank_ms", "total_device_ms"), (round(float(x), 4) for x in m)))
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
