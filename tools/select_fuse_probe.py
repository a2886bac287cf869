#!/usr/bin/env python3
"""Batched FDE pipeline at the full-shard shape: the default (the scan kernel applies the cosine rule / tombstones where it writes a
tile's scores; vectorised selection passes) against the round-2 structure (MV_OPT_FDE_BATCH_VARIANT = 5: a finish pass of its own).
Stage medians from the library's own events; one JSON object.

  python tools/select_fuse_probe.py [pages=1250000] [patches=1024]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ix = MvIndex(capacity_pages=n, stride_rows=stride, with_float=False, with_fp8=True, with_fde=True)
    ix.fill_synthetic(1234, 0, n)
    out_dim = ix.fde_config.output_dim
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    out = {"pages": n, "stride": stride, "fde_bytes_per_page": out_dim * 2, "requests_per_pass": 32}
    for coarse_n in (75, 1000):
        ix.set_option(L.MV_OPT_FDE_COARSE_N, coarse_n)
        row = {}
        ref = None
        for variant, name in ((5, "separate_finish_pass"), (0, "finish_in_the_scan_kernel"), (5, "separate_finish_pass_again"), (0, "finish_in_the_scan_kernel_again")):
            ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, variant)
            for _ in range(3):
                ix.query_batch(qs, 10, mode="fde_then_float")
            stages, devs = [], []
            for _ in range(15):
                r, st = ix.query_batch(qs, 10, mode="fde_then_float", want_stats=True)
                stages.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
                devs.append(st.total_device_ms)
            sm = np.median(np.array(stages), axis=0)
            row[name] = dict(zip(("encode_ms", "coarse_ms", "select_ms", "rerank_ms", "topk_ms"), [round(float(x), 4) for x in sm]))
            row[name]["device_ms_per_batch"] = round(float(np.median(devs)), 4)
            row[name]["coarse_plus_select_ms"] = round(float(sm[1] + sm[2]), 4)
            row[name]["coarse_GBps"] = round(n * out_dim * 2 / float(sm[1]) / 1e6, 1)
            if ref is None:
                ref = r
            else:
                row[name]["identical_to_first"] = all(a[1].tolist() == b[1].tolist() and a[0].tolist() == b[0].tolist() for a, b in zip(ref, r))
        out[f"coarse{coarse_n}"] = row
    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, 0)
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
