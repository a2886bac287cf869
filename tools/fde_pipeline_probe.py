#!/usr/bin/env python3
"""Stage split of FDE_THEN_FLOAT (encode / coarse scan / select / rerank / top-k) for both query-encode kernels, plus the
scan kernels' steady-state rates with proper warm-up.  One JSON object."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=False, with_binary=True, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, n)
    qs = [synth_rows(4321, j, 32) for j in range(8)]
    out = {"pages": n}
    ix.set_option(L.MV_OPT_FDE_COARSE_N, 1000)
    for qe in (0, 1, 2):
        ix.set_option(L.MV_OPT_FDE_QUERY_ENCODE_VARIANT, qe)
        rows = []
        for r in range(40):
            _s, _i, st = ix.query(qs[r % 8], 10, mode="fde_then_float", want_stats=True)
            if r >= 10:
                rows.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms, st.total_device_ms))
        m = np.median(np.array(rows), axis=0)
        out[f"fde_then_fp8_query_encode_v{qe}"] = dict(zip(("encode", "coarse", "select", "rerank", "topk", "total"), [round(float(x), 4) for x in m]))
        out[f"fde_then_fp8_query_encode_v{qe}"]["overhead_over_coarse"] = round(float(m[5] - m[1]), 4)
    per = {"binary": 1024 * 16, "float_fp8": 1024 * 128}
    for mode, b in per.items():
        ts = []
        for r in range(45):
            _s, _i, st = ix.query(qs[r % 8], 10, mode=mode, want_stats=True)
            if r >= 15:
                ts.append(st.score_kernel_ms)
        ms = float(np.median(ts))
        out[mode] = {"kernel_ms_med": round(ms, 4), "kernel_ms_min": round(float(min(ts)), 4), "GBps_med": round(n * b / ms / 1e6, 1), "GBps_best": round(n * b / min(ts) / 1e6, 1)}
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
