#!/usr/bin/env python3
"""Coarse-stage time (stats.coarse_ms, HIP events) of the single-query FDE scan: variant 0 (wave per page, plain nt loads into
VGPRs) against variant 5 (the same arithmetic, row quarters through the nt LDS-DMA ring), interleaved rounds on one index.

  python tools/fde_scan_probe.py [pages] [label] [e4m3]
Variant 6 = variant 5 with one 256 KiB-aligned block of the slab per workgroup.
Rows per workgroup of variant 5 (read once per process): MV_FDE_SCAN_RU."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    e4m3 = len(sys.argv) > 3 and sys.argv[3] == "e4m3"  # the coarse stage on the e4m3 copy of the slab (variant 0 = 5 there)
    ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True, with_fde_e4m3=e4m3)
    ix.fill_synthetic(1234, 0, n)
    out_dim = ix.fde_config.output_dim
    qs = [synth_rows(4321, j, 32) for j in range(8)]
    out = {"pages": n, "label": sys.argv[2] if len(sys.argv) > 2 else "", "rows_per_workgroup": os.environ.get("MV_FDE_SCAN_RU")}
    ts = {0: [], 5: [], 6: []}
    for rnd in range(4):
        for v in (0, 5, 6):
            ix.set_option(L.MV_OPT_FDE_SCAN_VARIANT, v)
            for r in range(12):
                _s, _i, st = ix.query(qs[r % 8], 10, mode="fde", want_stats=True)
                if r >= 4:
                    ts[v].append(st.coarse_ms)
    for v in (0, 5, 6):
        ms = float(np.median(ts[v]))
        bpr = out_dim * (1 if e4m3 else 2)
        out[f"variant_{v}"] = {"coarse_ms": round(ms, 4), "min_ms": round(float(np.min(ts[v])), 4), "GBps": round(n * bpr / ms / 1e6, 1),
                               "frac_hbm_8TBps": round(n * bpr / ms / 1e6 / 8000.0, 4)}
    ans = {}
    for v in (0, 5, 6):
        ix.set_option(L.MV_OPT_FDE_SCAN_VARIANT, v)
        ans[v] = ix.score_all(qs[3], mode="fde").tobytes()
    out["scores_bit_identical_in_all_variants"] = ans[0] == ans[5] == ans[6]
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
