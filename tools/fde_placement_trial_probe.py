#!/usr/bin/env python3
"""What mv_index_fde_placement_trial buys at the configs[3] shape: per fresh index (1.25 M pages x 20 KiB FDE rows = 25.6 GB), the batched coarse
pass (stats.coarse_ms, medians) before the trial and after it, the trial's own report and wall time, and the answers compared.

  python tools/fde_placement_trial_probe.py [pages=1250000] [indexes=5] [trials=4]
One JSON document on stdout."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(ix, qs):
    out = {}
    for B in (16, 32):
        for _ in range(3):
            ix.query_batch(qs[:B], 10, mode="fde")
        ts = []
        for _ in range(11):
            _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
            ts.append(st.coarse_ms)
        out[f"B{B}_ms"] = round(float(np.median(ts)), 4)
    ts = []
    for r in range(10):
        _s, _i, st = ix.query(qs[r], 10, mode="fde", want_stats=True)
        ts.append(st.coarse_ms)
    out["single_ms"] = round(float(np.median(ts[3:])), 4)
    return out


def main():
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    trials = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    res = {"pages": n, "trials_per_index": trials, "indexes": []}
    for i in range(K):
        ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
        ix.fill_synthetic(1234, 0, n)
        before = measure(ix, qs)
        want = ix.query_batch(qs, 10, mode="fde")
        t0 = time.perf_counter()
        rep = ix.fde_placement_trial(trials)
        wall = time.perf_counter() - t0
        after = measure(ix, qs)
        got = ix.query_batch(qs, 10, mode="fde")
        same = all(a[1].tolist() == b[1].tolist() and a[0].tolist() == b[0].tolist() for a, b in zip(want, got))
        rec = {"before": before, "after": after, "trial_report_ms_before_after_moves": [round(rep[0], 4), round(rep[1], 4), rep[2]],
               "trial_wall_s": round(wall, 3), "answers_identical": same}
        res["indexes"].append(rec)
        print(f"index {i}: {rec}", file=sys.stderr, flush=True)
        ix.close()
    b = np.array([r["before"]["B32_ms"] for r in res["indexes"]])
    a = np.array([r["after"]["B32_ms"] for r in res["indexes"]])
    res["B32_ms_before_mean_max"] = [round(float(b.mean()), 4), round(float(b.max()), 4)]
    res["B32_ms_after_mean_max"] = [round(float(a.mean()), 4), round(float(a.max()), 4)]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
