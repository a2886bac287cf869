#!/usr/bin/env python3
"""Batched FDE pipeline (mv_query_topk_batch, FDE modes) against the query-by-query form of the same entry point:
wall and device time per query, stage split, the coarse GEMM's slab rate, agreement of the results.  One JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morphik_core_amd import _lib as L
    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ix = MvIndex(capacity_pages=n, stride_rows=stride, with_float=True, with_fde=True)
    ix.fill_synthetic(1234, 0, n)
    out_dim = ix.fde_config.output_dim
    out = {"pages": n, "stride": stride, "fde_bytes_per_page": out_dim * 2}
    qs = [synth_rows(4321, j, 32) for j in range(64)]
    for coarse_n in (1000, 75):
        ix.set_option(L.MV_OPT_FDE_COARSE_N, coarse_n)
        for mode in ("fde_then_float", "fde"):
            if mode == "fde" and coarse_n != 1000:
                continue
            key = f"{mode}_coarse{coarse_n}" if mode != "fde" else "fde_only"
            res = {}
            for B in (16, 32, 64):
                row = {}
                for variant, name in ((0, "batched"), (1, "query_by_query")):
                    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, variant)
                    for _ in range(3):
                        ix.query_batch(qs[:B], 10, mode=mode)
                    walls, devs, stages = [], [], []
                    for _ in range(7):
                        t0 = time.perf_counter()
                        r, st = ix.query_batch(qs[:B], 10, mode=mode, want_stats=True)
                        walls.append((time.perf_counter() - t0) * 1e3)
                        devs.append(st.total_device_ms)
                        stages.append((st.encode_ms, st.coarse_ms, st.select_ms, st.rerank_ms, st.topk_ms))
                    sm = np.median(np.array(stages), axis=0)
                    row[name] = {"wall_ms_per_query": round(float(np.median(walls)) / B, 4), "device_ms_per_query": round(float(np.median(devs)) / B, 4),
                                 "queries_per_s_wall": round(B / (float(np.median(walls)) / 1e3), 1)}
                    if variant == 0:
                        passes = -(-B // 32)
                        row[name]["stage_us_per_query"] = dict(zip(("encode", "coarse", "select", "rerank", "topk"), [round(float(x) * 1e3 / B, 2) for x in sm]))
                        row[name]["coarse_ms_per_slab_pass"] = round(float(sm[1]) / passes, 4)
                        row[name]["coarse_GBps"] = round(n * out_dim * 2 / (float(sm[1]) / passes) / 1e6, 1)
                        row["_res"] = r
                    else:
                        same = sum(int(a[1].tolist() == b[1].tolist()) for a, b in zip(row["_res"], r))
                        overlap = float(np.mean([len(set(a[1].tolist()) & set(b[1].tolist())) / max(len(b[1]), 1) for a, b in zip(row["_res"], r)]))
                        row["identical_id_lists"] = f"{same}/{B}"
                        row["mean_id_overlap"] = round(overlap, 4)
                del row["_res"]
                row["speedup_wall"] = round(row["query_by_query"]["wall_ms_per_query"] / row["batched"]["wall_ms_per_query"], 2)
                res[f"B{B}"] = row
            out[key] = res
    ix.set_option(L.MV_OPT_FDE_BATCH_VARIANT, 0)
    ix.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
