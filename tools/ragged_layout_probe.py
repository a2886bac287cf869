#!/usr/bin/env python3
"""The full bf16 MaxSim scan on ONE index per process (allocation history moves the figure by 2-3 %, DESIGN 3.22: compare layouts
across fresh processes on a fresh box, not across indexes of one process): ColQwen-like ragged pages (550..1024 rows) in fixed
stride_rows slots, the same pages in the packed layout, or uniform 1024-row pages.  Rates on VALID bytes.

  python tools/ragged_layout_probe.py <ragged_fixed|ragged_packed|uniform|uniform_packed> [pages=300000] [rounds=3] [variants=-1]
variants: float kernel ids measured in interleaved rounds (-1 default; packed: 8 = workgroups own aligned blocks, 6 = a workgroup per page).
One JSON line on stdout."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd.index import MvIndex, synth_ragged_rows, synth_rows

    what = sys.argv[1] if len(sys.argv) > 1 else "ragged_packed"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    variants = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "-1").split(",")]
    stride, lo, hi, seed = 1024, 550, 1024, 1234
    qs = [synth_rows(4321, j, 32) for j in range(8)]
    rows = np.array([synth_ragged_rows(seed, u, lo, hi) for u in range(n)], np.int64)
    kw = {"packed": True, "capacity_rows": int(((rows + 15) // 16 * 16).sum())} if what == "ragged_packed" else ({"packed": True} if what == "uniform_packed" else {})
    ix = MvIndex(capacity_pages=n, stride_rows=stride, **kw)
    if what.startswith("uniform"):
        ix.fill_synthetic(seed, 0, n)
        valid = n * stride * 256
    else:
        ix.fill_synthetic_ragged(seed, 0, n, lo, hi)
        valid = int(rows.sum()) * 256
    from morphik_core_amd import _lib as L

    fr = {v: [] for v in variants}
    ans = {}
    for _ in range(rounds):
        for v in variants:
            ix.set_option(L.MV_OPT_MAXSIM_VARIANT, v)
            for _ in range(3):
                ix.query(qs[0], 10)
            ms = float(np.median([ix.query(qs[j % 8], 10, want_stats=True)[2].score_kernel_ms for j in range(11)]))
            fr[v].append(round(valid / ms / 1e6 / 8000.0, 4))
            ans.setdefault(v, ix.score_all(qs[1]).tobytes())
    print(json.dumps({"corpus": what, "pages": n, "valid_GB": round(valid / 1e9, 2), "slab_GB": round(ix.capacity_rows * 256 / 1e9, 2),
                      "frac_hbm_8TBps_valid_bytes_per_round": {f"variant_{v}": fr[v] for v in variants},
                      "median": {f"variant_{v}": float(np.median(fr[v])) for v in variants},
                      "scores_bit_identical_across_variants": all(a == ans[variants[0]] for a in ans.values())}))
    ix.close()


if __name__ == "__main__":
    main()
