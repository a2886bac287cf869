#!/usr/bin/env python3
"""The batched bf16 MaxSim scan (csrc/mv_batch.hip) by itself: B queries of 32 tokens per slab pass, kernel-only HIP-event times,
TFLOP/s against 2.5 PF, one JSON line per (variant, B).  Under `rocprofv3 --pmc ...` (tools/measure.sh batch_sq) the same run yields
the SQ counters of the kernel: matrix-pipe busy cycles, issue stalls, LDS stalls and GRBM_GUI_ACTIVE (-> the clock the kernel sustained).

  python tools/batch_scan_probe.py [pages=200000] [variant:B,...=0:16,0:4] [rounds=5] [zeroq]

`zeroq`: the same launches with ALL-ZERO query rows (every product and every accumulator is zero: the instruction stream, the stalls and
the HBM / LDS traffic are unchanged, only the switching activity of the multipliers drops).  If the kernel were bound by its schedule the
time would not move; a power-bound (DVFS) kernel speeds up -- MI355X_MICROARCH.md "DVFS give-back": zero-filled inputs ran +19 % TF/s at
+0.1 % SQ_WAVE_CYCLES.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex, synth_rows

    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    combos = [(int(v), int(b)) for v, b in (x.split(":") for x in (sys.argv[2] if len(sys.argv) > 2 else "0:16,0:4").split(","))]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    zeroq = len(sys.argv) > 4 and sys.argv[4] == "zeroq"
    patches = 1024
    ix = MvIndex(capacity_pages=pages, stride_rows=patches)
    ix.fill_synthetic(1234, 0, pages)
    ref = {}
    for bv, B in combos:
        qs = [synth_rows(4321, j, 32) for j in range(B)]
        if zeroq:
            qs = [np.zeros_like(q) for q in qs]
        ix.set_option(_lib.MV_OPT_BATCH_VARIANT, bv)
        ts = []
        for r in range(rounds + 1):
            out, st = ix.query_batch(qs, 10, want_stats=True)
            if r:
                ts.append(st.score_kernel_ms)
        ids = [o[1].tolist() for o in out]
        sc = np.concatenate([o[0] for o in out])
        if B not in ref:
            ix.set_option(_lib.MV_OPT_BATCH_VARIANT, -1)
            ref[B] = [ix.query(q, 10) for q in qs]  # the single-query scan: the batched forms must return its ids / scores
        same_ids = ids == [r[1].tolist() for r in ref[B]]
        max_rel = 0.0 if zeroq else float(np.max(np.abs(sc - np.concatenate([r[0] for r in ref[B]])) / np.abs(sc)))
        ms = float(np.median(ts))
        tf = 2.0 * B * 32 * patches * 128 * pages / ms / 1e9
        print(json.dumps({"queries": "all-zero rows" if zeroq else "synthetic unit rows", "variant": bv, "B": B, "pages": pages, "kernel_ms_med": round(ms, 4), "kernel_ms_min": round(min(ts), 4), "TFLOPs": round(tf, 1),
                          "frac_mfma_bf16_2500TF": round(tf / 2500.0, 4), "GBps": round(pages * patches * 256 / ms / 1e6, 1),
                          "same_ids_as_single_query": same_ids, "max_rel_score_diff_vs_single_query": max_rel}), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
