#!/usr/bin/env python3
"""BASELINE.json configs[1]: ColPali-v1.2 embed 1k synthetic pages -> MaxSim top-10, bf16, 1x MI355X.

Random-init weights of the ColPali-v1.2 architecture (no checkpoints / network here): the numbers are encoder
THROUGHPUT and end-to-end plumbing (PyTorch-ROCm forward -> bf16 rows stay on the GPU -> mv_index_add_device ->
fused MaxSim scan -> top-10), not retrieval quality.  Writes one JSON object.
"""
import argparse
import asyncio
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_page(rng, size=448):
    """Noise background + a few text-like dark rectangles (SURVEY.md 8d cfg 2)."""
    img = rng.integers(200, 255, (size, size, 3), dtype=np.uint8)
    for _ in range(rng.integers(8, 20)):
        y, x = rng.integers(0, size - 12), rng.integers(0, size - 120)
        img[y : y + rng.integers(4, 10), x : x + rng.integers(40, 120)] = rng.integers(0, 60)
    return img


def run(pages=1000, batch=32, feed=16, preset="colpali-v1.2", queries=8):
    """-> result dict (also used by bench.py --workload embed and by its aux_paths)."""
    a = argparse.Namespace(pages=pages, batch=batch, feed=feed, preset=preset, queries=queries)
    import torch
    from PIL import Image

    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel
    from morphik_core_amd.models import Chunk, DocumentChunk
    from morphik_core_amd.store import MI355XMultiVectorStore

    t0 = time.time()
    emb = MI355XColpaliEmbeddingModel(preset=a.preset, device="cuda:0", batch_size=a.batch)
    n_params = sum(p.numel() for p in emb.model.parameters())
    build_s = time.time() - t0
    rng = np.random.default_rng(0)
    rows_per_page = emb.n_image_tokens + 6
    stride = ((rows_per_page + 15) // 16) * 16
    store = MI355XMultiVectorStore(capacity_pages=a.pages, stride_rows=stride, mode="float")
    assert store.initialize()

    def chunk(i):
        buf = io.BytesIO()
        Image.fromarray(synth_page(rng, emb.image_size)).save(buf, format="PNG")
        return Chunk(content="", metadata={"is_image": True, "_image_bytes": buf.getvalue()})

    # warm-up (kernel selection, allocator)
    asyncio.run(emb.embed_for_ingestion_device([chunk(-1) for _ in range(a.batch)]))
    embed_s = store_s = prep_s = 0.0
    model_s = 0.0
    done = 0
    while done < a.pages:
        n = min(a.feed, a.pages - done)  # default 16 = the worker's COLPALI_STORE_BATCH_SIZE (ingestion_worker.py:1035)
        t = time.perf_counter()
        chunks = [chunk(done + j) for j in range(n)]
        prep_s += time.perf_counter() - t
        t = time.perf_counter()
        rows, n_rows = asyncio.run(emb.embed_for_ingestion_device(chunks))
        torch.cuda.synchronize()
        embed_s += time.perf_counter() - t
        model_s += emb.latest_ingest_timing()["model"]
        t = time.perf_counter()
        o = 0
        dcs = []
        for j, r in enumerate(n_rows):
            dcs.append(DocumentChunk(document_id=f"doc{(done + j) // 10}", content=f"page {done + j}", embedding=rows[o : o + r],
                                     chunk_number=(done + j) % 10, metadata={}))
            o += r
        ok, ids, _m = asyncio.run(store.store_embeddings(dcs))
        assert ok and len(ids) == n
        store_s += time.perf_counter() - t
        done += n
    # queries: text through the encoder, then exact MaxSim top-10 over all pages
    q_embed_ms, q_search_ms, top = [], [], []
    for qi in range(a.queries):
        t = time.perf_counter()
        q = asyncio.run(emb.embed_for_query(f"synthetic query number {qi} about revenue tables and totals"))
        q_embed_ms.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        hits = asyncio.run(store.query_similar(q, k=10))
        q_search_ms.append((time.perf_counter() - t) * 1e3)
        assert len(hits) == 10 and all(hits[i].score >= hits[i + 1].score for i in range(9))
        top.append(hits[0].content)
    flops_page = 2.0 * n_params * rows_per_page  # dense forward estimate (embedding table excluded would lower it slightly)
    res = {
        "workload": f"BASELINE configs[1]: {a.preset} architecture (random init), {a.pages} synthetic {emb.image_size}x{emb.image_size} pages -> MaxSim top-10",
        "params": n_params, "rows_per_page": rows_per_page, "model_build_s": round(build_s, 1),
        "embed_pages_per_s": round(a.pages / embed_s, 2), "embed_model_only_pages_per_s": round(a.pages / model_s, 2),
        "embed_tflops_est": round(flops_page * a.pages / model_s / 1e12, 1),
        "store_device_path_pages_per_s": round(a.pages / store_s, 1), "png_synthesis_s": round(prep_s, 2),
        "query_embed_ms_med": round(float(np.median(q_embed_ms)), 2), "query_maxsim_top10_ms_med": round(float(np.median(q_search_ms)), 3),
        "model_batch": a.batch, "chunks_per_call": a.feed, "fused_encoder_ops": getattr(emb, "fused_ops", None), "tuned_gemm_selections": getattr(emb, "tuned_gemms", None), "dtype": "bf16", "data": "synthetic page images; random-init weights (no checkpoint in this environment)",
        "top1_examples": top[:3],
    }
    store.close()
    del emb
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--feed", type=int, default=16, help="chunks per embed_for_ingestion call (the worker's COLPALI_STORE_BATCH_SIZE)")
    ap.add_argument("--preset", default="colpali-v1.2")
    ap.add_argument("--queries", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = run(a.pages, a.batch, a.feed, a.preset, a.queries)
    js = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write(js)
    print(js)


if __name__ == "__main__":
    main()
