#!/usr/bin/env python3
"""The FDE coarse stage on the e4m3 copy of the slab (MV_WITH_FDE_E4M3) against the bf16 slab, configs[3]'s shard size: 1.25 M pages, one
request (stats.coarse_ms, medians) and 16 / 32 requests per pass, interleaved rounds in one process.

  python tools/fde_e4m3_probe.py [pages=1250000] [rounds=4] [copy=e4m3|fp4]
(copy fp4: MV_WITH_FDE_FP4 -- the single-request scan on the e2m1 copy, quarter of the bytes; batches read the bf16 slab either way)
One JSON document on stdout."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    copy = sys.argv[3] if len(sys.argv) > 3 else "e4m3"
    cslab = 2 if copy == "fp4" else 1
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True, with_fde_e4m3=copy != "fp4", with_fde_fp4=copy == "fp4")
    ix.fill_synthetic(1234, 0, n)
    od = ix.fde_config.output_dim
    res = {"pages": n, "fde_width": od, "rounds": []}
    for r in range(rounds):
        row = {}
        for slab, name in ((cslab, copy), (0, "bf16")):
            ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, slab)
            for _ in range(3):
                ix.query(qs[0], 10, mode="fde")
            ts = []
            for j in range(15):
                _s, _i, st = ix.query(qs[j % 32], 10, mode="fde", want_stats=True)
                ts.append((st.coarse_ms, st.total_device_ms))
            c, t = np.median(np.array(ts), axis=0)
            bpp = od * 2 if slab == 0 else (od if slab == 1 else od // 2)
            row[name] = {"coarse_ms": round(float(c), 4), "request_device_ms": round(float(t), 4), "GBps": round(n * bpp / float(c) / 1e6, 1),
                         "frac_hbm_8TBps": round(n * bpp / float(c) / 1e6 / 8000.0, 4)}
            for B in (16, 32):
                for _ in range(2):
                    ix.query_batch(qs[:B], 10, mode="fde")
                tb = []
                for _ in range(7):
                    _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
                    tb.append(st.coarse_ms)
                row[name][f"batch{B}_coarse_ms"] = round(float(np.median(tb)), 4)
                if slab == 1:  # the queries' hi term only (MV_OPT_FDE_BATCH_VARIANT 2)
                    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, 2)
                    tb = []
                    for r2 in range(9):
                        _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
                        if r2 >= 2:
                            tb.append(st.coarse_ms)
                    row[name][f"batch{B}_one_term_coarse_ms"] = round(float(np.median(tb)), 4)
                    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, 0)
        res["rounds"].append(row)
        print(f"round {r}: {row}", file=sys.stderr, flush=True)
    same = 0
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, cslab)
    a = [set(ix.query(q, 75, mode="fde")[1].tolist()) for q in qs[:8]]
    ix.set_option(L.MV_OPT_FDE_COARSE_SLAB, 0)
    b = [set(ix.query(q, 75, mode="fde")[1].tolist()) for q in qs[:8]]
    res[f"top75_overlap_{copy}_vs_bf16_on_an_unstructured_corpus"] = round(float(np.mean([len(x & y) / 75 for x, y in zip(a, b)])), 4)
    ix.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
