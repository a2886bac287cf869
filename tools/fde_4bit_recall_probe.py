#!/usr/bin/env python3
"""What a 4-BIT copy of the FDE slab would cost in recall -- priced WITHOUT building it (round 6, closing session).

The coarse stage of configs[3] reads 20 KiB per page (bf16 FDE slab) or 10 KiB (its e4m3 copy, MV_WITH_FDE_E4M3); the
reference's own coarse stage is an ANN index, approximate by contract (fast_multivector_store.py:526-532).  A 4-bit copy would read
5 KiB per page.  This probe answers the question that decides whether such a slab is worth its plumbing: the document vectors of a
structured corpus are read back (mv_index_read_fde), quantised OUTSIDE the library (torch) to the candidate format, imported again
(mv_index_import_fde: the slab then holds exactly the quantised values -- every candidate grid is a subset of bf16 once the scale is a
power of two) and the SHIPPED pipeline (fp32 query FDE, cosine scan, top-75 / top-1000, exact bf16 rerank) measures recall@10 against the
exact bf16 top-10 on the bench's own recall sets (hard negatives: 64 near-tied pages per query; clustered topics; planted).

  python tools/fde_4bit_recall_probe.py [pages=200000]
One JSON document on stdout."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FP4_GRID = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]  # e2m1 magnitudes
FP4_MIDS = [0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0]   # round to nearest (ties: up)


def _t(x):
    import torch

    return torch.from_numpy(x).cuda()


def pow2_ceil(x):
    """smallest power of two >= x (elementwise, x > 0)"""
    import torch

    return torch.exp2(torch.ceil(torch.log2(torch.clamp(x, min=1e-30))))


def q_int(x, bits, block):
    """symmetric integer grid, one power-of-two scale per `block` elements (block = row length: per row); quantised on the GPU with torch"""
    import torch

    n, d = x.shape
    qmax = (1 << (bits - 1)) - 1
    b = _t(x).reshape(n, d // block, block)
    s = pow2_ceil(b.abs().amax(dim=2, keepdim=True) / qmax)
    return (torch.clamp(torch.round(b / s), -qmax, qmax) * s).reshape(n, d).cpu().numpy()


def q_fp4(x, block, shrink=1.0):
    """e2m1 (the f8f6f4 MFMA's FP4) with one power-of-two scale per `block` elements (block 32 = MXFP4); shrink > 1: a scale that many
    times smaller than the one that covers the largest element -- the largest elements saturate at 6 * scale, the bulk gets a finer grid"""
    import torch

    n, d = x.shape
    b = _t(x).reshape(n, d // block, block)
    s = pow2_ceil(b.abs().amax(dim=2, keepdim=True) / 6.0) / shrink
    grid = torch.tensor(FP4_GRID, device=b.device)
    idx = torch.bucketize(b.abs() / s, torch.tensor(FP4_MIDS, device=b.device), right=True)
    return (torch.sign(b) * grid[idx] * s).reshape(n, d).cpu().numpy()


FORMATS = [
    ("bf16_slab_as_built", None, 20480),
    ("int8_per_row", lambda x: q_int(x, 8, x.shape[1]), 10240),
    ("int4_per_row", lambda x: q_int(x, 4, x.shape[1]), 5120),
    ("int4_per_block_of_128", lambda x: q_int(x, 4, 128), 5120 + 80),
    ("fp4_e2m1_per_block_of_32_mxfp4", lambda x: q_fp4(x, 32), 5120 + 320),
    ("int3_per_block_of_128", lambda x: q_int(x, 3, 128), 3840 + 80),
    ("fp4_e2m1_per_row", lambda x: q_fp4(x, x.shape[1]), 5120 + 4),
    ("fp4_e2m1_per_block_of_512", lambda x: q_fp4(x, 512), 5120 + 20),
    ("fp4_e2m1_per_block_of_128", lambda x: q_fp4(x, 128), 5120 + 80),
    ("fp4_e2m1_per_row_scale_halved_saturating", lambda x: q_fp4(x, x.shape[1], 2.0), 5120 + 4),
    ("fp4_e2m1_per_row_scale_quartered_saturating", lambda x: q_fp4(x, x.shape[1], 4.0), 5120 + 4),
]


def main():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from morphik_core_amd import _lib as L
    from morphik_core_amd import synth
    from morphik_core_amd.index import MvIndex

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    args = argparse.Namespace(qtokens=32, patches=1024)
    t0 = time.time()
    ix = MvIndex(capacity_pages=n, stride_rows=1024, with_float=True, with_fde=True)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=1024)
    sets = bench.recall_sets(args, n, 0, [])
    planted = sum(synth.plant_neighbours_any(ix, st["spec"], synth.SEED_CORPUS, 1024, 0, n) for st in sets.values())
    truths, gaps = {}, {}
    for name, st in sets.items():
        truths[name], gaps[name] = bench.exact_truth(ix, st["queries"])
    res = {"pages": n, "structured_pages_written": planted, "setup_s": round(time.time() - t0, 1), "fde_width": ix.fde_config.output_dim, "formats": {}}
    od = ix.fde_config.output_dim
    CH = 8192
    original = [ix.read_fde(p0, min(CH, n - p0)) for p0 in range(0, n, CH)]  # 8 GB of host memory at 200 k pages

    def ids_of(mode, k=bench.K, cn=None):
        def f(q, al):
            if cn is not None:
                ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
            return ix.query(q, k, mode=mode, allow=al)[1].tolist()
        return f

    modes = [("fde_top75_then_exact", ids_of("fde_then_float", cn=75)), ("fde_top1000_then_exact", ids_of("fde_then_float", cn=1000)),
             ("fde_coarse_recall_at_75", ids_of("fde", k=75)), ("fde_coarse_recall_at_1000", ids_of("fde", k=1000))]
    for name, quant, bytes_per_page in FORMATS:
        t0 = time.time()
        err = []
        if quant is not None:
            for j, blk in enumerate(original):
                qb = quant(blk)
                ix.import_fde(j * CH, qb)
                if j == 0:
                    num = np.linalg.norm(qb - blk, axis=1)
                    err = float(np.median(num / np.maximum(np.linalg.norm(blk, axis=1), 1e-30)))
        r = bench.recall_of(ix, sets, truths, gaps, modes)
        ent = {"bytes_per_page": bytes_per_page, "median_relative_l2_error_of_a_document_vector": err if quant is not None else 0.0,
               "recall_at_10_vs_exact_bf16": {s: {m: r[s][m] for m, _f in modes} for s in sets}, "seconds": round(time.time() - t0, 1)}
        res["formats"][name] = ent
        print(name, json.dumps(ent["recall_at_10_vs_exact_bf16"]), file=sys.stderr, flush=True)
    # ---- the QUERY side of a batched pass on the fp4 codes (both MFMA operands FP4): the query FDE as two e2m1 terms under one power-of-two
    # scale per query -- hi = fp4(x / s), lo = fp4(4 (x / s - hi)) / 4 -- against the fp32 query, documents in the form that was built
    import torch

    from morphik_core_amd.index import fde_encode

    def orc_bf16_to_f32(a):
        return (a.astype(np.uint32) << 16).view(np.float32)

    def two_term_fp4(qf):
        t = torch.from_numpy(qf).cuda()
        s = pow2_ceil(t.abs().amax() / 6.0)
        grid = torch.tensor(FP4_GRID, device=t.device)
        mids = torch.tensor(FP4_MIDS, device=t.device)
        def enc(v):
            return torch.sign(v) * grid[torch.bucketize(v.abs(), mids, right=True)]
        hi = enc(t / s)
        lo = enc((t / s - hi) * 4.0) / 4.0
        return ((hi + lo) * s).cpu().numpy()

    built = dict((n_, q_) for n_, q_, _b in FORMATS)["fp4_e2m1_per_row_scale_halved_saturating"]
    for j, blk in enumerate(original):
        ix.import_fde(j * CH, built(blk))
    qerr = []

    def ids_q4(mode, k=bench.K, cn=None):
        def f(q, al):
            if cn is not None:
                ix.set_option(L.MV_OPT_FDE_COARSE_N, cn)
            qa = np.asarray(q)
            qf = fde_encode(orc_bf16_to_f32(qa) if qa.dtype == np.uint16 else qa, ix.fde_config, is_query=True)  # synth rows are bf16 bit patterns
            q4 = two_term_fp4(qf)
            qerr.append(float(np.linalg.norm(q4 - qf) / max(np.linalg.norm(qf), 1e-30)))
            return ix.query(q, k, mode=mode, allow=al, q_fde=q4)[1].tolist()
        return f

    modes4 = [("fde_top75_then_exact", ids_q4("fde_then_float", cn=75)), ("fde_top1000_then_exact", ids_q4("fde_then_float", cn=1000))]
    r = bench.recall_of(ix, sets, truths, gaps, modes4)
    res["fp4_documents_and_two_term_fp4_query"] = {"recall_at_10_vs_exact_bf16": {s_: {m: r[s_][m] for m, _f in modes4} for s_ in sets},
                                                   "median_relative_l2_error_of_a_query_vector": float(np.median(qerr))}
    print("fp4 docs + two-term fp4 query", json.dumps(res["fp4_documents_and_two_term_fp4_query"]), file=sys.stderr, flush=True)
    ix.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
