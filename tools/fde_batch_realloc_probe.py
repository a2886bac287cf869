#!/usr/bin/env python3
"""Is the mode of the batched FDE coarse pass (fast / slow, ~4-7 % apart: profiles/r5/fde_batch_pass_bimodal_across_processes_r5p.json) a property of
the PROCESS or of an ALLOCATION?  ONE process builds the index several times over -- between builds it takes a filler allocation of a different size,
so the slabs and workspaces land elsewhere -- and measures the pass each time; then it measures one build repeatedly for a few seconds.

  python tools/fde_batch_realloc_probe.py [pages=1250000] [builds=6]
One JSON document on stdout: a mode that changes between builds of one process is an allocation (placement) effect; one that only changes between
processes is not."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(ix, qs, rounds=15):
    out = {}
    for B in (16, 32):
        for _ in range(4):
            ix.query_batch(qs[:B], 10, mode="fde")
        ts = []
        for _ in range(rounds):
            _r, st = ix.query_batch(qs[:B], 10, mode="fde", want_stats=True)
            ts.append(st.coarse_ms)
        out[f"B{B}_ms"] = round(float(np.median(ts)), 4)
    ts = []
    for r in range(12):
        _s, _i, st = ix.query(qs[r % 32], 10, mode="fde", want_stats=True)
        ts.append(st.coarse_ms)
    out["single_ms"] = round(float(np.median(ts[4:])), 4)
    return out


def main():
    import torch

    from morphik_core_amd.index import MvIndex, synth_rows

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    builds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    qs = [synth_rows(4321, j, 32) for j in range(32)]
    fillers = [0, 2 << 20, 1 << 30, (3 << 30) + (64 << 10), 37 << 20, 8 << 30, 300 << 20, 5 << 30]
    res = {"pages": n, "builds": []}
    keep = []
    for b in range(builds):
        sz = fillers[b % len(fillers)]
        if sz:
            keep.append(torch.empty(sz, dtype=torch.uint8, device="cuda"))  # stays allocated: the next build lands behind it
        ix = MvIndex(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True)
        ix.fill_synthetic(1234, 0, n)
        m = measure(ix, qs)
        m["filler_bytes_before_this_build"] = sz
        res["builds"].append(m)
        print(f"build {b}: {m}", file=sys.stderr, flush=True)
        if b < builds - 1:
            ix.close()
    # the last build keeps its SLABS; only the batch workspace (score vectors, query image, selection buffers: ~0.6 GB) is dropped and
    # allocated afresh behind another filler: does the mode follow the workspace?
    res["same_slabs_fresh_batch_workspace"] = []
    for w in range(6):
        ix.set_option(1000, 0)  # diagnostic: free the lazily allocated batch workspaces
        keep.append(torch.empty((3 + 5 * w) << 20, dtype=torch.uint8, device="cuda"))
        m = measure(ix, qs)
        res["same_slabs_fresh_batch_workspace"].append(m)
        print(f"workspace {w}: {m}", file=sys.stderr, flush=True)
    # the same, but the old workspace is SET ASIDE (not freed), so the new one cannot get the same memory back -- all of it, then one part at
    # a time: the part whose replacement moves the time is the one the mode follows
    res["same_slabs_workspace_part_replaced"] = {}
    for name, val in (("all", 1), ("score_vectors_only", 3), ("query_image_only", 5), ("selection_and_query_buffers_only", 9)):
        rows = []
        for w in range(8):
            ix.set_option(1000, val)
            rows.append(measure(ix, qs, rounds=9))
        res["same_slabs_workspace_part_replaced"][name] = rows
        print(f"replaced {name}: " + " ".join(f"{r['B16_ms']}/{r['B32_ms']}" for r in rows), file=sys.stderr, flush=True)
    # the last build, measured again and again: does the mode drift inside one set of allocations?
    t0 = time.time()
    rep = []
    while time.time() - t0 < 6.0:
        rep.append(measure(ix, qs, rounds=7)["B32_ms"])
    res["last_build_B32_ms_repeated"] = rep
    ix.close()
    # the pass's TRANSPORT alone -- mv_calibrate allocates a fresh 24 GiB buffer per call and reads it in the pass's pattern (64 rows x 512 B
    # per step at a 20 480 B stride, nt LDS-DMA, nothing consumed): is the placement effect there without the kernel?  Control: whole rows.
    import ctypes as C

    from morphik_core_amd._lib import check, lib

    tr = {"dma_strided_512B_GBps": [], "plain_strided_512B_GBps": [], "whole_rows_20K_GBps": []}
    for _ in range(8):
        for key, code in (("dma_strided_512B_GBps", 11), ("plain_strided_512B_GBps", 5), ("whole_rows_20K_GBps", 9)):
            g = C.c_double()
            check(lib().mv_calibrate(0, code, 24 << 30, 6, C.byref(g)))
            tr[key].append(round(float(g.value), 1))
    res["transport_alone_fresh_24GiB_buffer_per_call"] = tr
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
