#!/usr/bin/env python3
"""Encoder forward throughput vs batch size (random-init ColPali-v1.2 architecture, bf16, model-only)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel

emb = MI355XColpaliEmbeddingModel(preset=sys.argv[1] if len(sys.argv) > 1 else "colpali-v1.2", device="cuda:0", batch_size=8)
rng = np.random.default_rng(0)
res = {}
for B in [int(x) for x in os.environ.get("MV_EMBED_PROBE_BATCHES", "1,4,8,16,32,64").split(",")]:
    imgs = [rng.integers(0, 255, (emb.image_size, emb.image_size, 3), dtype=np.uint8) for _ in range(B)]
    for it in range(2):
        emb._embed_images_device(imgs)
    torch.cuda.synchronize()
    n = max(2, 64 // B)
    t = time.perf_counter()
    for it in range(n):
        emb._embed_images_device(imgs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    res[B] = {"ms_per_batch": round(dt * 1e3, 2), "pages_per_s": round(B / dt, 1)}
    print(B, res[B], flush=True)
print(json.dumps(res))
