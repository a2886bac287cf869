"""MI355X embed server speaking the reference's `/embeddings` protocol.

ColpaliApiEmbeddingModel (core/embedding/colpali_api_embedding_model.py) POSTs
    {"input_type": "image" | "text", "inputs": [base64 image | text, ...]}      (:287-290, Bearer token header :286)
and expects an `.npz` body {count, input_type, emb_0..emb_{count-1}} of float32 (n_tok,128) arrays (:293-310);
HTTP 413 makes the client bisect its batch (:243-268).  Pointing `colpali_mode = "api"` /
`morphik_embedding_api_domain` (morphik.toml:151-153) at N of these servers -- one per GPU -- puts the encoder
on N MI355X with zero changes inside morphik-core: the reference's own round-robin fan-out and failover
(:160-207) is the data parallelism ("replicas only").

    python -m morphik_core_amd.embed_server --port 8765 [--family colpali|colqwen2] [--model /path/to/checkpoint] [--device cuda:0]
"""
from __future__ import annotations

import argparse
import asyncio
import base64
import hmac
import logging
import os
from typing import Any, List, Optional

from .formats import encode_embeddings_npz
from .models import Chunk

logger = logging.getLogger(__name__)

from pydantic import BaseModel

MAX_INPUTS_PER_REQUEST = 256  # beyond this answer 413 so the reference client halves its batch


class EmbedRequest(BaseModel):
    """Request body of the reference client (colpali_api_embedding_model.py:287)."""

    input_type: str
    inputs: List[str]


def create_app(embedder: Any, api_key: Optional[str] = None):
    """FastAPI app around any object with `embed_for_ingestion(chunks)` / `embed_for_query(text)` coroutines."""
    from fastapi import FastAPI, Header, HTTPException
    from fastapi.responses import Response

    app = FastAPI(title="mi355x-colpali-embeddings")
    lock = asyncio.Lock()  # one forward pass at a time per GPU

    @app.get("/health")
    async def health():
        return {"status": "ok"}

    @app.post("/embeddings")
    async def embeddings(req: EmbedRequest, authorization: Optional[str] = Header(default=None)):  # noqa: B008
        if api_key and not hmac.compare_digest((authorization or "").encode(), f"Bearer {api_key}".encode()):
            raise HTTPException(status_code=401, detail="invalid api key")
        if req.input_type not in ("image", "text"):
            raise HTTPException(status_code=422, detail="input_type must be 'image' or 'text'")
        if len(req.inputs) > MAX_INPUTS_PER_REQUEST:
            raise HTTPException(status_code=413, detail="batch too large")
        if req.input_type == "image":
            chunks = []
            for s in req.inputs:
                b64 = s.split(",", 1)[1] if s.startswith("data:") else s
                try:
                    raw = base64.b64decode(b64, validate=False)
                except Exception:  # noqa: BLE001
                    raise HTTPException(status_code=422, detail="inputs must be base64 images")
                chunks.append(Chunk(content="", metadata={"is_image": True, "_image_bytes": raw}))
        else:
            chunks = [Chunk(content=s, metadata={}) for s in req.inputs]
        async with lock:
            embs = await embedder.embed_for_ingestion(chunks) if chunks else []
        return Response(content=encode_embeddings_npz(embs, req.input_type), media_type="application/octet-stream")

    return app


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8765)
    ap.add_argument("--family", choices=["colpali", "colqwen2"], default="colpali",
                    help="encoder family: colpali (ColPali-v1.2 / PaliGemma) or colqwen2 (ColQwen2 / ColQwen2.5, the reference's "
                         "colpali_embedding_model.py:47-52 family: dynamic patch counts -> ragged pages; needs --model)")
    ap.add_argument("--model", default=None, help="checkpoint directory (colpali default: random-init colpali-v1.2 architecture)")
    ap.add_argument("--preset", default="colpali-v1.2")
    ap.add_argument("--device", default=None)
    ap.add_argument("--batch-size", type=int, default=8)
    a = ap.parse_args(argv)
    import uvicorn

    if a.family == "colqwen2":
        if not a.model:
            raise SystemExit("embed_server: --family colqwen2 needs --model <checkpoint directory> (its processor ships with the checkpoint)")
        from .colqwen_embedding import MI355XColQwen2EmbeddingModel

        emb: Any = MI355XColQwen2EmbeddingModel(model_name_or_path=a.model, device=a.device, batch_size=a.batch_size)
    else:
        from .embedding import MI355XColpaliEmbeddingModel

        emb = MI355XColpaliEmbeddingModel(model_name_or_path=a.model, preset=a.preset, device=a.device, batch_size=a.batch_size)
    uvicorn.run(create_app(emb, os.environ.get("MORPHIK_EMBEDDING_API_KEY")), host=a.host, port=a.port, log_level="info")


if __name__ == "__main__":
    main()
