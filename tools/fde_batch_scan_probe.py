#!/usr/bin/env python3
"""Coarse-stage time of the batched FDE scan (FDE_ONLY mode, stats.coarse_ms) for B = 16 / 32 at a given corpus size."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
    ix.fill_synthetic(1234, 0, n)
    out_dim = ix.fde_config.output_dim
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    form = sys.argv[2] if len(sys.argv) > 2 else "default"  # MV_OPT_FDE_BATCH_VARIANT: hi_only = 2 (query FDE rounded to bf16), single_tile = 3, separate_finish = 5 (4, 6-8 were removed in round 5)
    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, {"default": 0, "hi_only": 2, "single_tile": 3, "separate_finish": 5}[form])
    out = {"pages": n, "form": form}
    for B in (16, 32):
        for _ in range(5):
            ix.query_batch(qs[:B], 10, mode="fde")
        ts = []
        for _ in range(15):
            _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
            ts.append(st.coarse_ms)
        ms = float(np.median(ts))
        out[f"B{B}"] = {"coarse_ms": round(ms, 4), "GBps": round(n * out_dim * 2 / ms / 1e6, 1), "us_per_query": round(ms * 1e3 / B, 2)}
    ts = []
    for r in range(20):
        _s, _i, st = ix.query(qs[r % 32], 10, mode="fde", want_stats=True)
        ts.append(st.coarse_ms)
    ms = float(np.median(ts[5:]))
    out["single"] = {"coarse_ms": round(ms, 4), "GBps": round(n * out_dim * 2 / ms / 1e6, 1)}
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
