"""One fresh process of the encoder stress test (tests/test_gpu_store.py::test_encoder_start_ups_with_fused_ops_and_tuned_gemms_*):
build the FULL ColPali-v1.2 architecture, embed `pages` synthetic pages in one forward, embed one query, print a JSON line.
The switches come from the environment (MV_ENCODER_FUSED_OPS, MV_ENCODER_TUNED_GEMMS), as a deployment sets them."""
import asyncio
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    preset = sys.argv[2] if len(sys.argv) > 2 else "colpali-v1.2"
    from PIL import Image

    import torch

    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel
    from morphik_core_amd.models import Chunk

    t0 = time.time()
    emb = MI355XColpaliEmbeddingModel(preset=preset, device="cuda:0", batch_size=max(pages, 1))
    rng = np.random.default_rng(pages)
    chunks = []
    for _ in range(pages):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 255, (448, 448, 3), dtype=np.uint8)).save(buf, format="PNG")
        chunks.append(Chunk(content="", metadata={"is_image": True, "_image_bytes": buf.getvalue()}))
    rows, n_rows = asyncio.run(emb.embed_for_ingestion_device(chunks))
    torch.cuda.synchronize()
    q = asyncio.run(emb.embed_for_query("quarterly revenue by region"))
    rows2, _n = asyncio.run(emb.embed_for_ingestion_device(chunks))  # a second forward of the same pages: identical rows
    torch.cuda.synchronize()
    norms = rows.float().norm(dim=1)
    print(json.dumps({"pages": pages, "rows": [int(x) for x in n_rows], "finite": bool(torch.isfinite(rows.float()).all()),
                      "norm_err": float((norms - 1).abs().max()), "repeat_max_diff": float((rows.float() - rows2.float()).abs().max()), "query_rows": int(q.shape[0]),
                      "query_finite": bool(np.isfinite(q).all()), "fused_ops": emb.fused_ops, "tuned_gemms": bool(emb.tuned_gemms),
                      "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()
