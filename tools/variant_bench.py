#!/usr/bin/env python3
"""A/B of the float MaxSim kernel variants (and the binary / FDE scans) in ONE process, interleaved
rounds, HIP-event kernel times from the library's own stats.  Writes JSON to stdout / a file."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=100_000)
    ap.add_argument("--patches", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="0,6,7")  # the float kernels that exist (1-5, 8-12, 14 were removed in round 5)
    ap.add_argument("--qtokens", default="32")
    ap.add_argument("--aux", action="store_true", help="also time binary and FDE scans (smaller corpus)")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-batch", action="store_true", help="skip the batched-query sweep")
    ap.add_argument("--batch-only", action="store_true", help="only the batched-query sweep (SQ counter passes)")
    ap.add_argument("--batch-variants", default="", help="variant:B pairs, e.g. 0:16,6:16 (default: the full sweep)")
    a = ap.parse_args()
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex, calibrate_read_bw, synth_rows

    res = {"pages": a.pages, "patches": a.patches, "float": {}, "calib_read_gbps": calibrate_read_bw(4 << 30, 10)}
    ix = MvIndex(capacity_pages=a.pages, stride_rows=a.patches)
    ix.fill_synthetic(1234, 0, a.pages)
    nbytes = a.pages * a.patches * 256
    variants = [int(v) for v in a.variants.split(",")]
    for qt in ([] if a.batch_only else [int(x) for x in a.qtokens.split(",")]):
        q = synth_rows(4321, 0, qt)
        times = {v: [] for v in variants}
        ref = None
        for r in range(a.rounds + 1):
            for v in variants:
                ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, v)
                s, i, st = ix.query(q, 10, want_stats=True)
                if ref is None:
                    ref = (s, i)
                assert i.tolist() == ref[1].tolist() and np.allclose(s, ref[0], rtol=1e-5), f"variant {v} disagrees"
                if r:  # round 0 = warm-up
                    times[v].append((st.score_kernel_ms, st.topk_ms))
        for v in variants:
            k = np.array([t[0] for t in times[v]])
            res["float"][f"q{qt}_v{v}"] = {
                "kernel_ms_med": float(np.median(k)), "kernel_ms_min": float(k.min()),
                "GBps_med": nbytes / np.median(k) / 1e6, "GBps_best": nbytes / k.min() / 1e6,
                "topk_ms_med": float(np.median([t[1] for t in times[v]])),
            }
            print(f"q={qt} variant {v}: {np.median(k):.3f} ms  {nbytes/np.median(k)/1e6:.0f} GB/s (best {nbytes/k.min()/1e6:.0f})  topk {np.median([t[1] for t in times[v]]):.3f} ms", flush=True)
    # batched queries: one slab pass for B queries of 32 tokens (HBM-bound -> MFMA-bound as B grows)
    res["batch"] = {}
    # 0 = auto (page-split form <= 128 rows, transposed row-split form above), 3 = row-split always, 2 = round-1 pipeline, 1 = 32x32x16
    from morphik_core_amd.index import calibrate
    if not a.no_batch:
        res["mfma_calibration_TF"] = {"16x16x32": calibrate("mfma_bf16", 0, 5), "32x32x16": calibrate("mfma_bf16_32x32", 0, 5)}
        print("mfma calibration", res["mfma_calibration_TF"], flush=True)
    combos = [(0, 1), (0, 2), (0, 4), (5, 4), (6, 4), (3, 4), (0, 8), (5, 8), (6, 8), (0, 12), (6, 12), (0, 16), (5, 16), (6, 16), (1, 16)]
    if a.batch_variants:
        combos = [(int(v), int(b)) for v, b in (x.split(":") for x in a.batch_variants.split(","))]
    for bv, B in ([] if a.no_batch else combos):
        ix.set_option(_lib.MV_OPT_BATCH_VARIANT, bv)
        qs = [synth_rows(4321, j, 32) for j in range(B)]
        ts = []
        for r in range(a.rounds + 1):
            out, st = ix.query_batch(qs, 10, want_stats=True)
            if r:
                ts.append(st.score_kernel_ms)
        ms = float(np.median(ts))
        flops = 2.0 * B * 32 * a.patches * 128 * a.pages
        res["batch"][f"v{bv}_B{B}"] = {"kernel_ms_med": ms, "GBps": nbytes / ms / 1e6, "TFLOPs": flops / ms / 1e9, "query_pages_per_s": B * a.pages / ms * 1e3}
        print(f"batch v{bv} B={B}: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s  {flops/ms/1e9:.0f} TFLOP/s  {B*a.pages/ms*1e3/1e6:.1f} M query-pages/s", flush=True)
    ix.close()
    if not a.no_batch and not a.batch_variants:
        # the e4m3 form of the batched scan: half the page bytes; two-term (parity) and single-term (coarse pass) queries
        ix = MvIndex(capacity_pages=a.pages, stride_rows=a.patches, with_float=False, with_fp8=True)
        ix.fill_synthetic(1234, 0, a.pages)
        res["batch_fp8"] = {}
        for bv, B in [(0, 1), (0, 2), (0, 4), (7, 4), (0, 8), (7, 8), (0, 16), (7, 16)]:
            ix.set_option(_lib.MV_OPT_BATCH_VARIANT, bv)
            qs = [synth_rows(4321, j, 32) for j in range(B)]
            ts = []
            for r in range(a.rounds + 1):
                out, st = ix.query_batch(qs, 10, mode="float_fp8", want_stats=True)
                if r:
                    ts.append(st.score_kernel_ms)
            ms = float(np.median(ts))
            flops = 2.0 * B * 32 * a.patches * 128 * a.pages
            res["batch_fp8"][f"v{bv}_B{B}"] = {"kernel_ms_med": ms, "GBps": a.pages * a.patches * 128 / ms / 1e6, "useful_TFLOPs": flops / ms / 1e9,
                                               "issued_TFLOPs": flops * (1 if bv == 7 else 2) / ms / 1e9, "query_pages_per_s": B * a.pages / ms * 1e3}
            print(f"batch fp8 v{bv} B={B}: {ms:.3f} ms  {a.pages*a.patches*128/ms/1e6:.0f} GB/s  {flops/ms/1e9:.0f} useful TFLOP/s  {B*a.pages/ms*1e3/1e6:.1f} M query-pages/s", flush=True)
        ix.close()
    if a.aux:
        n = min(a.pages, 200_000)
        ix = MvIndex(capacity_pages=n, stride_rows=a.patches, with_float=True, with_binary=True, with_fde=True, with_fp8=True)
        import time
        t0 = time.time()
        ix.fill_synthetic(1234, 0, n)
        res["fill_all_slabs_s"] = time.time() - t0
        res["fill_pages"] = n
        q = synth_rows(4321, 0, 32)
        runs = [("binary_v0_popcount", "binary", a.patches * 16, (_lib.MV_OPT_BINARY_VARIANT, 0)),
                ("binary_v4_fp4mfma_lean_d4", "binary", a.patches * 16, (_lib.MV_OPT_BINARY_VARIANT, 4)),
                ("fde_v0_regs", "fde", 10240 * 2, (_lib.MV_OPT_FDE_SCAN_VARIANT, 0)),
                ("fde_v5_rowq_ring", "fde", 10240 * 2, (_lib.MV_OPT_FDE_SCAN_VARIANT, 5)),
                ("float_fp8", "float_fp8", a.patches * 128, None),
                ("float_bf16", "float", a.patches * 256, None)]
        for name, mode, per_page, opt in runs:
            if opt:
                ix.set_option(*opt)
            ts = []
            for r in range(a.rounds + 1):
                s, i, st = ix.query(q, 10, mode=mode, want_stats=True)
                if r:
                    ts.append(st.score_kernel_ms)
            res[name] = {"kernel_ms_med": float(np.median(ts)), "GBps_med": n * per_page / np.median(ts) / 1e6, "pages_per_s": n / np.median(ts) * 1e3,
                         "top": i[:3].tolist()}
            print(f"{name}: {np.median(ts):.3f} ms  {n*per_page/np.median(ts)/1e6:.0f} GB/s  {n/np.median(ts)*1e3/1e6:.1f} M pages/s", flush=True)
        ix.close()
    js = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write(js)
    print(js)


if __name__ == "__main__":
    main()
