#!/usr/bin/env python3
"""Does the time of the batched FDE coarse pass belong to an ALLOCATION or to the DEVICE'S STATE at the moment?  K indexes of the same content
live side by side in one process (1.25 M pages x 20 KiB = 25.6 GB each) and are measured round-robin: A B C ... A B C ...
  * every index keeps its own time over all rounds, and the indexes differ  -> the time follows the allocation (physical placement);
    building a few candidates and keeping the fastest would lift the slow mode;
  * all indexes move together from round to round                            -> device state, nothing an allocation can choose.

  python tools/fde_batch_coresident_probe.py [pages=1250000] [indexes=6] [rounds=8] [fde | float]
`float`: the same question for the headline bf16 scan (single 32-row query, 16-row pages: 4 KiB per page), `score_kernel_ms` per index.
One JSON document on stdout."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(ix, qs, B, reps=9):
    ts = []
    for _ in range(2):
        ix.query_batch(qs[:B], 10, mode="fde")
    for _ in range(reps):
        _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
        ts.append(st.coarse_ms)
    return round(float(np.median(ts)), 4)


def main():
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    if len(sys.argv) > 4 and sys.argv[4] == "float":
        return main_float(n, K, rounds, qs)
    ixs = []
    for _ in range(K):
        ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
        ix.fill_synthetic(1234, 0, n)
        ixs.append(ix)
    res = {"pages": n, "indexes": K, "rounds": rounds, "B32_ms": [[] for _ in range(K)], "B16_ms": [[] for _ in range(K)], "t_s": []}
    t0 = time.time()
    for r in range(rounds):
        res["t_s"].append(round(time.time() - t0, 2))
        for i, ix in enumerate(ixs):
            res["B32_ms"][i].append(one(ix, qs, 32))
            res["B16_ms"][i].append(one(ix, qs, 16))
        print(f"round {r}: " + " ".join(f"{res['B16_ms'][i][-1]}/{res['B32_ms'][i][-1]}" for i in range(K)), file=sys.stderr, flush=True)
    a = np.array(res["B32_ms"])
    res["B32_per_index_median"] = [round(float(x), 4) for x in np.median(a, axis=1)]
    res["B32_per_index_spread_over_rounds"] = [round(float(x), 4) for x in (a.max(axis=1) - a.min(axis=1))]
    res["B32_per_round_median"] = [round(float(x), 4) for x in np.median(a, axis=0)]
    res["B32_spread_between_indexes"] = round(float(np.median(a, axis=1).max() - np.median(a, axis=1).min()), 4)
    for ix in ixs:
        ix.close()
    print(json.dumps(res, indent=1))


def main_float(n, K, rounds, qs):
    from morphik_core_amd.index import MvIndex

    ixs = []
    for _ in range(K):
        ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=True)
        ix.fill_synthetic(1234, 0, n)
        ixs.append(ix)
    res = {"pages": n, "indexes": K, "rounds": rounds, "what": "bf16 scan, one 32-row query, stats.score_kernel_ms medians of 15", "scan_ms": [[] for _ in range(K)]}
    for r in range(rounds):
        for i, ix in enumerate(ixs):
            for _ in range(3):
                ix.query(qs[0], 10, mode="float")
            ts = []
            for j in range(15):
                _s, _i, st = ix.query(qs[j % 32], 10, mode="float", want_stats=True)
                ts.append(st.score_kernel_ms)
            res["scan_ms"][i].append(round(float(np.median(ts)), 4))
        print(f"round {r}: " + " ".join(str(res["scan_ms"][i][-1]) for i in range(K)), file=sys.stderr, flush=True)
    a = np.array(res["scan_ms"])
    res["per_index_median"] = [round(float(x), 4) for x in np.median(a, axis=1)]
    res["per_index_spread_over_rounds"] = [round(float(x), 4) for x in (a.max(axis=1) - a.min(axis=1))]
    res["spread_between_indexes_rel"] = round(float((np.median(a, axis=1).max() - np.median(a, axis=1).min()) / np.median(a)), 4)
    for ix in ixs:
        ix.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
