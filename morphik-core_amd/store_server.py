"""One HBM slab, many processes: the store-owner server and its BaseVectorStore client.

The reference constructs its multivector store in EVERY process that needs one -- the API server
(core/services_init.py:141-165) and each ingestion worker (core/workers/ingestion_worker.py:102-142) -- which is fine
when the store is a Postgres / TurboPuffer client, and impossible when the store IS 262 GB of HBM: a second process
cannot allocate a second slab, and pages ingested by the worker must be visible to the API process's queries.  So one
process owns the MI355X store (this server, one per node) and every other process uses `MI355XRemoteMultiVectorStore`,
a BaseVectorStore whose four coroutines forward over HTTP on localhost:

    POST /store_embeddings   npz {meta: json [{document_id, chunk_number, content, metadata}], app_id, emb_0..emb_{n-1}}
    POST /query_similar      npz {meta: json {k, doc_ids, app_id, skip_image_content}, q}          -> json chunks
    POST /get_chunks_by_id   json {chunk_identifiers, app_id, skip_image_content}                  -> json chunks
    POST /delete_chunks_by_document_id   json {document_id, app_id}                                -> json {ok}
    POST /save               checkpoint now (store.save into --save-dir)                            -> json {ok, pages, seconds}
    GET  /health

The owner holds the ONLY copy of the corpus: with --save-dir it checkpoints on POST /save, every --save-every-s seconds when
pages changed, and on shutdown; --load resumes from such a directory.  Chunk payloads (page images) go to the owner's
`.storage` (--payload-dir: a directory; or the host application's storage object passed to build_store / create_app's
store) -- or, when the REMOTE client was constructed with a `storage` of its own, the client uploads them there and only
the storage keys cross the wire.

Embeddings travel as float32 (or bf16 bit patterns as uint16) arrays inside one .npz body -- the wire format the
reference already uses for multi-vectors (colpali_api_embedding_model.py:293-310).  Scoring still happens only in
libmvmaxsim.so inside the owner.

    python -m morphik_core_amd.store_server --provider mi355x_fast --capacity-pages 250000 --port 8766
"""
# (no `from __future__ import annotations` here: FastAPI resolves the handler annotations at decoration time)

import argparse
import asyncio
import hmac
import io
import json
import logging
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .models import BaseVectorStore, DocumentChunk

logger = logging.getLogger(__name__)


def _chunks_json(chunks: List[DocumentChunk], key_flags: Optional[List[bool]] = None) -> List[Dict[str, Any]]:
    if key_flags is not None:
        return [{"document_id": c.document_id, "chunk_number": int(c.chunk_number), "content": c.content, "metadata": c.metadata or {}, "score": float(c.score),
                 "content_is_key": bool(f)} for c, f in zip(chunks, key_flags)]
    return [{"document_id": c.document_id, "chunk_number": int(c.chunk_number), "content": c.content, "metadata": c.metadata or {}, "score": float(c.score)}
            for c in chunks]


def _pack(meta: Any, arrays: Dict[str, np.ndarray]) -> bytes:
    buf = io.BytesIO()
    np.savez(buf, meta=np.array(json.dumps(meta)), **arrays)
    return buf.getvalue()


def _unpack(body: bytes) -> Tuple[Any, Any]:
    z = np.load(io.BytesIO(body), allow_pickle=False)
    return json.loads(str(z["meta"])), z


class _Checkpointer:
    """Serialised store.save() calls into one directory + the dirty flag the periodic / shutdown saves look at."""

    def __init__(self, store: Any, save_dir: Optional[str]):
        self.store, self.save_dir = store, save_dir
        self.dirty = False
        self._lock = asyncio.Lock()
        self.last: Dict[str, Any] = {}

    async def save(self) -> Dict[str, Any]:
        import time

        if not self.save_dir:
            raise RuntimeError("the server was started without --save-dir")
        async with self._lock:
            t0 = time.perf_counter()
            self.dirty = False  # writes landing during the save set it again
            await asyncio.to_thread(self.store.save, self.save_dir)
            self.last = {"ok": True, "pages": len(self.store) if hasattr(self.store, "__len__") else None, "seconds": round(time.perf_counter() - t0, 3),
                         "directory": self.save_dir}
            return self.last


def create_app(store: Any, api_key: Optional[str] = None, save_dir: Optional[str] = None, save_every_s: float = 0.0):
    """FastAPI app around any BaseVectorStore (normally an MI355X store that owns the GPU)."""
    from contextlib import asynccontextmanager

    from fastapi import FastAPI, Header, HTTPException, Request

    ckpt = _Checkpointer(store, save_dir)

    @asynccontextmanager
    async def lifespan(_app):
        task = None
        if save_dir and save_every_s > 0:
            async def periodic():
                while True:
                    await asyncio.sleep(save_every_s)
                    if ckpt.dirty:
                        try:
                            await ckpt.save()
                        except Exception as e:  # noqa: BLE001 -- keep serving; the next tick tries again
                            logger.error("periodic checkpoint failed: %s", e)
            task = asyncio.ensure_future(periodic())
        try:
            yield
        finally:
            if task is not None:
                task.cancel()
            if save_dir and ckpt.dirty:  # everything ingested over HTTP since the last checkpoint
                try:
                    await ckpt.save()
                except Exception as e:  # noqa: BLE001
                    logger.error("checkpoint on shutdown failed: %s", e)

    app = FastAPI(title="mi355x-multivector-store", lifespan=lifespan)
    app.state.checkpointer = ckpt

    def auth(authorization: Optional[str]) -> None:
        if api_key and not hmac.compare_digest((authorization or "").encode(), f"Bearer {api_key}".encode()):
            raise HTTPException(status_code=401, detail="invalid api key")

    def _key_flags(chunks: List[DocumentChunk], app_id: Optional[str]) -> Optional[List[bool]]:
        if not hasattr(store, "content_key_flags"):
            return None
        return store.content_key_flags([(c.document_id, int(c.chunk_number)) for c in chunks], app_id)

    @app.get("/health")
    async def health():
        return {"status": "ok", "pages": len(store) if hasattr(store, "__len__") else None}

    @app.post("/store_embeddings")
    async def store_embeddings(request: Request, authorization: Optional[str] = Header(default=None)):  # noqa: B008
        auth(authorization)
        meta, z = _unpack(await request.body())
        chunks, flags = [], []
        for i, m in enumerate(meta["chunks"]):
            emb = z[f"emb_{i}"] if f"emb_{i}" in z.files else None
            chunks.append(DocumentChunk(document_id=m["document_id"], chunk_number=int(m["chunk_number"]), content=m["content"], embedding=emb,
                                        metadata=m.get("metadata") or {}))
            flags.append(bool(m.get("content_is_key", False)))  # the client says so explicitly; nothing is inferred from the text
        try:
            kw = {"content_is_key": flags} if any(flags) else {}
            ok, ids, metrics = await store.store_embeddings(chunks, app_id=meta.get("app_id"), **kw)
        except Exception as e:  # noqa: BLE001 -- the client re-raises it, as a local store would have raised
            raise HTTPException(status_code=500, detail=f"{type(e).__name__}: {e}")
        ckpt.dirty = True
        return {"ok": bool(ok), "ids": ids, "metrics": metrics}

    @app.post("/save")
    async def save(authorization: Optional[str] = Header(default=None)):  # noqa: B008
        auth(authorization)
        try:
            return await ckpt.save()
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=f"{type(e).__name__}: {e}")

    @app.post("/query_similar")
    async def query_similar(request: Request, authorization: Optional[str] = Header(default=None)):  # noqa: B008
        auth(authorization)
        meta, z = _unpack(await request.body())
        try:
            res = await store.query_similar(z["q"], int(meta["k"]), doc_ids=meta.get("doc_ids"), app_id=meta.get("app_id"),
                                            skip_image_content=bool(meta.get("skip_image_content", False)))
        except Exception as e:  # noqa: BLE001
            raise HTTPException(status_code=500, detail=f"{type(e).__name__}: {e}")
        return {"chunks": _chunks_json(res, _key_flags(res, meta.get("app_id")))}

    @app.post("/get_chunks_by_id")
    async def get_chunks_by_id(req: Dict[str, Any], authorization: Optional[str] = Header(default=None)):  # noqa: B008
        auth(authorization)
        res = await store.get_chunks_by_id([(d, int(c)) for d, c in req.get("chunk_identifiers", [])], app_id=req.get("app_id"),
                                           skip_image_content=bool(req.get("skip_image_content", False)))
        return {"chunks": _chunks_json(res, _key_flags(res, req.get("app_id")))}

    @app.post("/delete_chunks_by_document_id")
    async def delete_chunks(req: Dict[str, Any], authorization: Optional[str] = Header(default=None)):  # noqa: B008
        auth(authorization)
        if hasattr(store, "delete_chunks_returning_keys"):  # keys of payloads a client uploaded to ITS storage go back to it
            ok, left = await store.delete_chunks_returning_keys(req["document_id"], app_id=req.get("app_id"))
        else:
            ok, left = bool(await store.delete_chunks_by_document_id(req["document_id"], app_id=req.get("app_id"))), []
        ckpt.dirty = ckpt.dirty or ok
        return {"ok": bool(ok), "content_keys": left}

    return app


class MI355XRemoteMultiVectorStore(BaseVectorStore):
    """BaseVectorStore whose work is done by the store-owner process (see module docstring).  Same signatures, return shapes
    and error conventions as the local stores: initialize() -> bool and never raises; query errors propagate."""

    backend_name = "mi355x-remote"

    def __init__(self, url: str = "http://127.0.0.1:8766", api_key: Optional[str] = None, timeout_s: float = 600.0, storage: Any = None,
                 **_ignored: Any):
        self.url = url.rstrip("/")
        # A client constructed with the application's storage object uploads chunk payloads THERE (as the local stores do:
        # multi_vector_store.py:650-676) and sends the owner only the storage keys; hits come back as keys and are
        # resolved through the same object.  Without one the payloads travel to the owner, which keeps them in ITS storage.
        self.storage = storage
        from .payloads import PayloadStore

        self._payloads = PayloadStore(storage) if storage is not None else None
        self._headers = {"Authorization": f"Bearer {api_key}"} if api_key else {}
        self._timeout = timeout_s
        self._last_store_metrics: Dict[str, Any] = {}
        self._clients: Dict[int, Any] = {}  # one keep-alive client per event loop (a request costs no connection set-up)

    def _client(self):
        import httpx

        loop = asyncio.get_running_loop()
        c = self._clients.get(id(loop))
        if c is None or c.is_closed:
            if len(self._clients) > 8:  # loops come and go (asyncio.run per call in scripts and tests)
                self._clients.clear()
            c = self._clients[id(loop)] = httpx.AsyncClient(timeout=self._timeout, limits=httpx.Limits(max_keepalive_connections=64, max_connections=256))
        return c

    async def aclose(self) -> None:
        for c in list(self._clients.values()):
            try:
                await c.aclose()
            except Exception:  # noqa: BLE001 -- a client of a finished loop
                pass
        self._clients.clear()

    async def _post(self, path: str, *, content: Optional[bytes] = None, json_body: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        client = self._client()
        if content is not None:
            r = await client.post(self.url + path, content=content, headers={**self._headers, "Content-Type": "application/octet-stream"})
        else:
            r = await client.post(self.url + path, json=json_body, headers=self._headers)
        if r.status_code != 200:
            raise RuntimeError(f"store server {path} -> {r.status_code}: {r.text[:300]}")
        return r.json()

    def initialize(self) -> bool:
        try:
            import httpx

            r = httpx.get(self.url + "/health", headers=self._headers, timeout=10.0)
            return r.status_code == 200
        except Exception as e:  # noqa: BLE001
            logger.error("Error initializing %s: %s", type(self).__name__, e)
            return False

    @staticmethod
    def _rows(e: Any) -> np.ndarray:
        from .store import _embedding_rows

        return _embedding_rows(e)

    async def store_embeddings(self, chunks: List[DocumentChunk], app_id: Optional[str] = None) -> Tuple[bool, List[str], Dict[str, Any]]:
        meta = {"app_id": app_id, "chunks": []}
        arrays: Dict[str, np.ndarray] = {}
        contents = [c.content for c in chunks]
        if self._payloads is not None:  # payloads stay on this side of the wire: upload, send the keys
            res = await asyncio.gather(*[self._payloads.put(c.content, c.document_id, int(c.chunk_number), c.metadata or {}, app_id)
                                         if getattr(c, "embedding", None) is not None else asyncio.sleep(0, result=(None, 0)) for c in chunks])
            contents = [key if key else c.content for (key, _n), c in zip(res, chunks)]
        for i, c in enumerate(chunks):
            meta["chunks"].append({"document_id": c.document_id, "chunk_number": int(c.chunk_number), "content": contents[i], "metadata": c.metadata or {},
                                   "content_is_key": contents[i] is not c.content})  # True only for a key THIS client just uploaded
            if getattr(c, "embedding", None) is not None:
                arrays[f"emb_{i}"] = self._rows(c.embedding)
        try:
            out = await self._post("/store_embeddings", content=_pack(meta, arrays))
        except Exception:
            if self._payloads is not None:  # the owner refused the chunks: the payloads uploaded for them must not stay behind
                orphans = [k for k, c in zip(contents, chunks) if k is not c.content]
                if orphans:
                    try:
                        await self._payloads.delete(orphans, chunks[0].document_id if chunks else "")
                    except Exception as e:  # noqa: BLE001
                        logger.error(f"could not remove {len(orphans)} payloads uploaded for a failed store_embeddings: {e}")
            raise
        self._last_store_metrics = out.get("metrics", {})
        return bool(out["ok"]), list(out["ids"]), self._last_store_metrics

    async def _to_chunks(self, rows: List[Dict[str, Any]], skip_image_content: bool = False) -> List[DocumentChunk]:
        contents = [r["content"] for r in rows]
        if self._payloads is not None:  # keys this client uploaded (the owner flags them): resolve them here (images stay keys when the caller skips them)
            fetch = [j for j, r in enumerate(rows) if r.get("content_is_key") and not (skip_image_content and (r.get("metadata") or {}).get("is_image"))]
            got = await asyncio.gather(*[self._payloads.get(rows[j]["content"], rows[j].get("metadata") or {}) for j in fetch])
            for j, c in zip(fetch, got):
                contents[j] = c
        return [DocumentChunk(document_id=r["document_id"], chunk_number=int(r["chunk_number"]), content=c, embedding=[],
                              metadata=r.get("metadata") or {}, score=float(r.get("score", 0.0))) for r, c in zip(rows, contents)]

    async def query_similar(self, query_embedding: Any, k: int, doc_ids: Optional[List[str]] = None, app_id: Optional[str] = None,
                            skip_image_content: bool = False) -> List[DocumentChunk]:
        meta = {"k": int(k), "doc_ids": doc_ids, "app_id": app_id, "skip_image_content": bool(skip_image_content)}
        out = await self._post("/query_similar", content=_pack(meta, {"q": self._rows(query_embedding)}))
        return await self._to_chunks(out["chunks"], skip_image_content)

    async def get_chunks_by_id(self, chunk_identifiers: List[Tuple[str, int]], app_id: Optional[str] = None,
                               skip_image_content: bool = False) -> List[DocumentChunk]:
        if not chunk_identifiers:
            return []
        out = await self._post("/get_chunks_by_id", json_body={"chunk_identifiers": [[d, int(c)] for d, c in chunk_identifiers], "app_id": app_id,
                                                               "skip_image_content": bool(skip_image_content)})
        return await self._to_chunks(out["chunks"], skip_image_content)

    async def delete_chunks_by_document_id(self, document_id: str, app_id: Optional[str] = None) -> bool:
        try:
            out = await self._post("/delete_chunks_by_document_id", json_body={"document_id": document_id, "app_id": app_id})
            keys = list(out.get("content_keys") or [])
            if keys and self._payloads is not None:  # the payloads THIS side uploaded: the reference deletes storage objects on delete
                await self._payloads.delete(keys, document_id)
            return bool(out["ok"])
        except Exception as e:  # noqa: BLE001 -- delete returns False on error (multi_vector_store.py:944-946)
            logger.error(f"Error deleting chunks for document {document_id}: {e}")
            return False


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8766)
    ap.add_argument("--provider", default="mi355x_fast", help="create_store provider (mi355x | mi355x_fast | mi355x_float | mi355x_sharded ...)")
    ap.add_argument("--capacity-pages", type=int, default=250_000)
    ap.add_argument("--stride-rows", type=int, default=1040)
    ap.add_argument("--devices", default="", help="comma separated GPU ordinals for the sharded providers (default: all)")
    ap.add_argument("--load", default="", help="checkpoint directory (store.save) to resume from; capacity / stride / mode come from the checkpoint")
    ap.add_argument("--batch-window-ms", type=float, default=-1.0,
                    help="request coalescing: < 0 (default) = adaptive -- a request that finds the GPU idle is dispatched at once, "
                         "requests arriving while a scan is in flight share the next slab pass; > 0 = timer window in ms (a lone "
                         "request pays it); 0 = off, one scan per request")
    ap.add_argument("--max-batch", type=int, default=32, help="largest coalesced batch")
    ap.add_argument("--save-dir", default="", help="checkpoint directory: POST /save, --save-every-s and shutdown write store.save() here (resume with --load)")
    ap.add_argument("--save-every-s", type=float, default=0.0, help="checkpoint every N seconds when pages were added or deleted since the last one (0 = only /save and shutdown)")
    ap.add_argument("--payload-dir", default="", help="directory for chunk payloads (page images): the owner's `.storage`; without it payloads stay inline in the owner's memory")
    ap.add_argument("--fde-module", default="", help="importable module with the API of the reference's `fde` extension (FixedDimensionalEncodingConfig, "
                                                      "generate_document_encoding, generate_query_encoding): the fast providers use ITS vectors for the candidate stage")
    a = ap.parse_args(argv)
    import uvicorn

    store = build_store(a)
    uvicorn.run(create_app(store, os.environ.get("MORPHIK_STORE_API_KEY"), save_dir=a.save_dir or None, save_every_s=a.save_every_s), host=a.host,
                port=a.port, log_level="info")


def build_store(a: Any) -> Any:
    """The store the server owns: a fresh one of the requested provider, or the same class resumed from a checkpoint."""
    from .store import create_store

    opts: Dict[str, Any] = dict(batch_window_ms=a.batch_window_ms, max_batch=a.max_batch)
    if getattr(a, "payload_dir", ""):
        from .payloads import LocalDirStorage

        opts["storage"] = LocalDirStorage(a.payload_dir)
    if a.devices:
        opts["devices"] = [int(x) for x in a.devices.split(",")]
    if getattr(a, "fde_module", ""):
        import importlib

        opts["fde_module"] = importlib.import_module(a.fde_module)  # raises loudly when the deployment's encoder is not installed
    proto = create_store(a.provider, capacity_pages=a.capacity_pages, stride_rows=a.stride_rows, **opts)  # allocates nothing yet
    if a.load:
        if not hasattr(proto, "devices"):
            opts.pop("devices", None)  # only the sharded stores take a device list
        return type(proto).load(a.load, **opts)  # raises loudly when the checkpoint does not fit / is inconsistent
    if not proto.initialize():
        raise SystemExit("store_server: the MI355X store could not be initialised (no GPU / slab does not fit)")
    return proto


if __name__ == "__main__":
    main()
