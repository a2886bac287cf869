"""BaseVectorStore plugins backed by the MI355X index -- the drop-in boundary of SURVEY.md 8(b).

  MI355XMultiVectorStore      drop-in for core/vector_store/multi_vector_store.py:MultiVectorStore
                              (scores = SQL max_sim on sign bits; mode "binary")
  MI355XFastMultiVectorStore  drop-in for core/vector_store/fast_multivector_store.py:FastMultiVectorStore
                              (FDE coarse top min(10k,75) -> exact float MaxSim rerank; mode "fde_then_float")
  either can run mode "float": exact float MaxSim over the WHOLE corpus (no coarse stage) -- what the
  HBM-resident slab makes affordable (1 M pages in ~40 ms on one GPU).

Signatures, return shapes and error conventions follow the reference:
  store_embeddings -> (True, ["{document_id}-{chunk_number}", ...], metrics)   multi_vector_store.py:623-719
  query_similar    -> List[DocumentChunk] sorted by score desc, embedding=[]     multi_vector_store.py:721-817
  get_chunks_by_id -> score 0.0                                                  multi_vector_store.py:824-919
  delete_chunks_by_document_id -> bool, False on error                           multi_vector_store.py:921-951
  initialize() -> bool, never raises                                             multi_vector_store.py:186-327
Scoring is done by libmvmaxsim.so only; this file is bookkeeping (ids, payloads, filters).
"""
from __future__ import annotations

import asyncio
import json
import logging
import threading
import time
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from .models import BaseVectorStore, DocumentChunk, build_store_metrics

logger = logging.getLogger(__name__)


def _embedding_rows(e: Any) -> np.ndarray:
    """ndarray / torch.Tensor (any device) / list -> C-contiguous [n,128] fp32 or uint16(bf16)
    (multi_vector_store.py:334-337 and fast_multivector_store.py:515-518 accept the same inputs)."""
    from .index import _to_host

    a = _to_host(e)
    if a.dtype != np.uint16:
        a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != 128:
        raise ValueError(f"multi-vector embedding must be [n, 128]; got {a.shape}")
    return a


def _device_tensor(e: Any, device_index: int):
    """The embedding itself when it is a torch tensor resident on the index's GPU ([n,128], bf16 or fp32), else None."""
    if hasattr(e, "is_cuda") and getattr(e, "is_cuda") and e.device.index == device_index and e.dim() == 2 and e.shape[1] == 128:
        if str(e.dtype) in ("torch.bfloat16", "torch.float32"):
            return e
    return None


class MI355XMultiVectorStore(BaseVectorStore):
    backend_name = "mi355x"
    default_mode = "binary"

    def __init__(
        self,
        capacity_pages: int = 1_000_000,
        stride_rows: int = 1040,  # ColPali-v1.2: 1024 patches + 6 prompt rows -> 1030, padded to a multiple of 16
        device: int = 0,
        mode: Optional[str] = None,
        storage: Any = None,
        id_base: int = 0,
        index_factory: Optional[Callable[..., Any]] = None,
        fde_coarse_n: int = 0,
        batch_window_ms: float = 0.0,
        max_batch: int = 16,
        min_score: Optional[float] = None,
        **_ignored: Any,
    ):
        self.capacity_pages = int(capacity_pages)
        self.stride_rows = int(stride_rows)
        self.device = int(device)
        self.mode = mode or self.default_mode
        if self.mode not in ("binary", "float", "fde_then_float", "float_fp8"):
            raise ValueError(f"unknown mode {self.mode}")
        self.storage = storage  # callers reach for .storage (document_service.py:1570-1575)
        self.id_base = int(id_base)
        self.fde_coarse_n = int(fde_coarse_n)
        self._index_factory = index_factory
        # request coalescing (mode "float" only): concurrent query_similar calls arriving within batch_window_ms are
        # scored in ONE slab pass by the batched MFMA kernel, each keeping its own doc_ids filter and k
        self.batch_window_s = float(batch_window_ms) / 1e3
        self.max_batch = int(max_batch)
        # The API accepts min_score (core/models/request.py:138) and threads it through retrieve_chunks
        # (document_service.py:184) but the reference never applies it to multivector hits; None keeps that behaviour,
        # a number drops hits scoring below it
        self.min_score = None if min_score is None else float(min_score)
        self._pending: List[Tuple[np.ndarray, int, Any, Any]] = []
        self._flush_handle = None
        self.coalesced_batches: List[int] = []  # sizes of the batches actually issued (introspection / tests)
        self._index = None
        self._lock = threading.RLock()
        # payload table: page -> (document_id, chunk_number, content, metadata_json, app_id)
        self._rows: Dict[int, Tuple[str, int, str, str, Optional[str]]] = {}
        self._page_of: Dict[Tuple[str, int], int] = {}
        self._doc_ord: Dict[str, int] = {}
        self._doc_app: Dict[int, Optional[str]] = {}
        self._doc_pages: Dict[str, List[int]] = {}
        self._last_store_metrics: Dict[str, Any] = {}
        self.last_query_timing: Dict[str, float] = {}

    # ------------------------------------------------------------------ lifecycle
    def _make_index(self):
        if self._index_factory is not None:
            return self._index_factory(capacity_pages=self.capacity_pages, stride_rows=self.stride_rows, device=self.device,
                                       id_base=self.id_base, mode=self.mode)
        from .index import MvIndex  # HIP-only: raises MvError when libmvmaxsim.so or the GPU is missing

        ix = MvIndex(
            capacity_pages=self.capacity_pages, stride_rows=self.stride_rows, device=self.device, id_base=self.id_base,
            with_float=self.mode in ("float", "fde_then_float"), with_binary=self.mode == "binary",
            with_fde=self.mode == "fde_then_float", with_fp8=self.mode == "float_fp8",
        )
        if self.fde_coarse_n:
            from ._lib import MV_OPT_FDE_COARSE_N

            ix.set_option(MV_OPT_FDE_COARSE_N, self.fde_coarse_n)
        return ix

    def initialize(self) -> bool:
        """Allocate the HBM slabs. Returns False on failure, never raises (multi_vector_store.py:325-327)."""
        try:
            with self._lock:
                if self._index is None:
                    self._index = self._make_index()
            logger.info("%s initialized successfully", type(self).__name__)
            return True
        except Exception as e:  # noqa: BLE001
            logger.error("Error initializing %s: %s", type(self).__name__, e)
            return False

    def close(self) -> None:
        with self._lock:
            if self._index is not None:
                try:
                    self._index.close()
                except Exception as e:  # noqa: BLE001
                    logger.error("Error closing index: %s", e)
                self._index = None

    def _require_index(self):
        if self._index is None:
            with self._lock:
                if self._index is None:
                    self._index = self._make_index()
        return self._index

    # ------------------------------------------------------------------ store
    def _store_sync(self, valid: List[DocumentChunk], embs: List[np.ndarray], app_id: Optional[str]) -> List[str]:
        ix = self._require_index()
        with self._lock:
            ords = []
            for c in valid:
                o = self._doc_ord.get(c.document_id)
                if o is None:
                    o = len(self._doc_ord)
                    self._doc_ord[c.document_id] = o
                    self._doc_app[o] = app_id
                ords.append(o)
            # upsert: an existing (document_id, chunk_number) is replaced (FastMultiVectorStore upserts by id)
            for c in valid:
                old = self._page_of.pop((c.document_id, c.chunk_number), None)
                if old is not None:
                    ix.remove_page(old - self.id_base)
                    self._rows.pop(old, None)
                    if old in self._doc_pages.get(c.document_id, []):
                        self._doc_pages[c.document_id].remove(old)
            if embs and all(not isinstance(e, np.ndarray) for e in embs):
                # ingest-side fusion: encoder output already on this GPU -> one D2D pass fills
                # every slab (mv_index_add_device); no D2H -> fp32 -> H2D round trip
                import torch

                from ._lib import MV_BF16, MV_F32

                code = MV_BF16 if all(e.dtype == torch.bfloat16 for e in embs) else MV_F32
                flat = torch.cat([e if code == MV_BF16 else e.to(torch.float32) for e in embs], 0).contiguous()
                torch.cuda.current_stream(flat.device).synchronize()  # the library orders on its own stream
                first = ix.add_device(flat.data_ptr(), code, [int(e.shape[0]) for e in embs], ords) + self.id_base
            else:
                first = ix.add([e if isinstance(e, np.ndarray) else _embedding_rows(e) for e in embs], ords) + self.id_base
            ids = []
            for i, c in enumerate(valid):
                page = first + i
                self._rows[page] = (c.document_id, int(c.chunk_number), c.content, json.dumps(c.metadata or {}), app_id)
                self._page_of[(c.document_id, int(c.chunk_number))] = page
                self._doc_pages.setdefault(c.document_id, []).append(page)
                ids.append(f"{c.document_id}-{c.chunk_number}")
            return ids

    async def store_embeddings(self, chunks: List[DocumentChunk], app_id: Optional[str] = None) -> Tuple[bool, List[str], Dict[str, Any]]:
        valid: List[DocumentChunk] = []
        for chunk in chunks:
            if not hasattr(chunk, "embedding") or chunk.embedding is None:
                logger.error(f"Missing embeddings for chunk {chunk.document_id}-{chunk.chunk_number}")
                continue
            valid.append(chunk)
        if not valid:
            self._last_store_metrics = build_store_metrics(
                chunk_payload_backend="memory", multivector_backend=self.backend_name, vector_store_backend=self.backend_name
            )
            return True, [], self._last_store_metrics
        use_dev = self._index_factory is None
        embs = [(_device_tensor(c.embedding, self.device) if use_dev else None) for c in valid]
        embs = [e if e is not None else _embedding_rows(c.embedding) for c, e in zip(valid, embs)]
        for c, e in zip(valid, embs):
            if e.shape[0] > self.stride_rows:
                raise ValueError(
                    f"chunk {c.document_id}-{c.chunk_number} has {e.shape[0]} vectors; this store was created with "
                    f"stride_rows={self.stride_rows}"
                )
        t0 = time.perf_counter()
        ids = await asyncio.to_thread(self._store_sync, valid, embs, app_id)
        dt = time.perf_counter() - t0
        self._last_store_metrics = build_store_metrics(
            chunk_payload_backend="memory", multivector_backend=self.backend_name, vector_store_backend=self.backend_name,
            multivector_upload_s=dt, multivector_objects=len(ids), multivector_bytes=int(sum(e.shape[0] for e in embs)) * 256,
            vector_store_write_s=dt, vector_store_rows=len(ids),
        )
        return True, ids, self._last_store_metrics

    # ------------------------------------------------------------------ query
    def _allow_for(self, doc_ids: Optional[List[str]], app_id: Optional[str]):
        """doc_ids falsy => no filter (multi_vector_store.py:754). Returns (bitmap or None, empty?)."""
        from .index import allow_bitmap

        ords = None
        if doc_ids:
            ords = [self._doc_ord[d] for d in doc_ids if d in self._doc_ord]
        if app_id is not None and self._filter_by_app:
            base = range(len(self._doc_ord)) if ords is None else ords
            ords = [o for o in base if self._doc_app.get(o) == app_id]
        if ords is None:
            return None, False
        if not ords:
            return None, True
        return allow_bitmap(ords, len(self._doc_ord)), False

    _filter_by_app = False  # MultiVectorStore.query_similar ignores app_id (multi_vector_store.py:721-763)

    def _query_sync(self, q: np.ndarray, k: int, allow) -> Tuple[np.ndarray, np.ndarray]:
        ix = self._require_index()
        t0 = time.perf_counter()
        s, i = ix.query(q, k, mode=self.mode, allow=allow)
        self.last_query_timing = {"vector_search_s": time.perf_counter() - t0}
        return s, i

    # -- request coalescing
    def _batch_sync(self, items: List[Tuple[np.ndarray, int, Any, Any]]):
        ix = self._require_index()
        kmax = max(k for _q, k, _a, _f in items)
        allows = [a for _q, _k, a, _f in items]
        with self._lock:
            n_docs = len(self._doc_ord)
        t0 = time.perf_counter()
        res = ix.query_batch([q for q, _k, _a, _f in items], kmax, mode="float", allows=allows if any(a is not None for a in allows) else None,
                             n_docs=n_docs)
        self.last_query_timing = {"vector_search_s": time.perf_counter() - t0, "batched_queries": len(items)}
        return [(s[:k], i[:k]) for (s, i), (_q, k, _a, _f) in zip(res, items)]

    def _flush(self) -> None:
        items, self._pending = self._pending, []
        if self._flush_handle is not None:
            self._flush_handle.cancel()
            self._flush_handle = None
        if not items:
            return
        self.coalesced_batches.append(len(items))

        async def run():
            try:
                outs = await asyncio.to_thread(self._batch_sync, items)
                for (_q, _k, _a, fut), out in zip(items, outs):
                    if not fut.done():
                        fut.set_result(out)
            except Exception as e:  # noqa: BLE001 -- every waiter sees the failure (query errors propagate)
                for _q, _k, _a, fut in items:
                    if not fut.done():
                        fut.set_exception(e)

        asyncio.ensure_future(run())

    async def _coalesced_query(self, q: np.ndarray, k: int, allow) -> Tuple[np.ndarray, np.ndarray]:
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        self._pending.append((q, k, allow, fut))
        if len(self._pending) >= self.max_batch:
            self._flush()
        elif self._flush_handle is None:
            self._flush_handle = loop.call_later(self.batch_window_s, self._flush)
        return await fut

    async def query_similar(
        self,
        query_embedding: Any,
        k: int,
        doc_ids: Optional[List[str]] = None,
        app_id: Optional[str] = None,
        skip_image_content: bool = False,
    ) -> List[DocumentChunk]:
        q = _embedding_rows(query_embedding)
        with self._lock:
            allow, empty = self._allow_for(doc_ids, app_id)
        if empty or k <= 0:
            return []
        if self.batch_window_s > 0 and self.mode == "float":
            scores, pages = await self._coalesced_query(q, int(k), allow)
        else:
            scores, pages = await asyncio.to_thread(self._query_sync, q, int(k), allow)  # exceptions propagate (:819-822)
        out: List[DocumentChunk] = []
        with self._lock:
            for s, p in zip(scores.tolist(), pages.tolist()):
                if self.min_score is not None and s < self.min_score:
                    break  # hits are sorted by score desc
                row = self._rows.get(int(p))
                if row is None:
                    continue  # deleted between scan and lookup
                doc_id, chunk_no, content, meta_json, _app = row
                out.append(DocumentChunk(document_id=doc_id, chunk_number=chunk_no, content=content, embedding=[],
                                         metadata=json.loads(meta_json) if meta_json else {}, score=float(s)))
        return out

    async def get_chunks_by_id(self, chunk_identifiers: List[Tuple[str, int]], app_id: Optional[str] = None,
                               skip_image_content: bool = False) -> List[DocumentChunk]:
        if not chunk_identifiers:
            return []
        out = []
        with self._lock:
            for doc_id, chunk_no in dict.fromkeys((d, int(c)) for d, c in chunk_identifiers):
                page = self._page_of.get((doc_id, chunk_no))
                if page is None:
                    continue
                _d, _c, content, meta_json, _app = self._rows[page]
                out.append(DocumentChunk(document_id=doc_id, chunk_number=chunk_no, content=content, embedding=[],
                                         metadata=json.loads(meta_json) if meta_json else {}, score=0.0))
        return out

    async def delete_chunks_by_document_id(self, document_id: str, app_id: Optional[str] = None) -> bool:
        try:
            with self._lock:
                o = self._doc_ord.get(document_id)
                if o is None:
                    return True  # DELETE of nothing succeeds
                ix = self._require_index()
                ix.remove_doc(o)
                for page in self._doc_pages.pop(document_id, []):
                    row = self._rows.pop(page, None)
                    if row is not None:
                        self._page_of.pop((row[0], row[1]), None)
                # the ordinal stays reserved (its pages are tombstoned in the slab)
            logger.info(f"Deleted all chunks for document {document_id} from {self.backend_name} store")
            return True
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error deleting chunks for document {document_id}: {e}")
            return False

    # ------------------------------------------------------------------ maintenance
    def compact(self) -> int:
        """Reclaim the slab slots of deleted / replaced pages (mv_index_compact) and remap the bookkeeping.
        Returns the number of slots reclaimed.  Page ids are internal to the store, so callers see no change."""
        with self._lock:
            ix = self._require_index()
            before = len(ix)
            o2n = ix.compact()
            remap = {self.id_base + int(o): self.id_base + int(n) for o, n in enumerate(o2n.tolist()) if n >= 0}
            self._rows = {remap[p]: r for p, r in self._rows.items() if p in remap}
            self._page_of = {key: remap[p] for key, p in self._page_of.items() if p in remap}
            self._doc_pages = {d: [remap[p] for p in ps if p in remap] for d, ps in self._doc_pages.items()}
            return before - len(ix)

    # ------------------------------------------------------------------ checkpoint / resume
    def save(self, directory: str) -> None:
        """Persist the HBM index (mv_index_save: raw slabs + metadata) and the store's bookkeeping (payload rows,
        document ordinals) so a restarted process resumes without re-embedding -- the role Postgres / S3 play for the
        reference stores (checkpoint/resume)."""
        import os

        os.makedirs(directory, exist_ok=True)
        with self._lock:
            ix = self._require_index()
            ix.save(os.path.join(directory, "index.mv"))
            book = {
                "version": 1, "mode": self.mode, "capacity_pages": self.capacity_pages, "stride_rows": self.stride_rows,
                "id_base": self.id_base, "fde_coarse_n": self.fde_coarse_n,
                "rows": [[p, r[0], r[1], r[2], r[3], r[4]] for p, r in self._rows.items()],
                "doc_ord": self._doc_ord, "doc_app": {str(k): v for k, v in self._doc_app.items()},
            }
            tmp = os.path.join(directory, "store.json.tmp")
            with open(tmp, "w") as f:
                json.dump(book, f)
            os.replace(tmp, os.path.join(directory, "store.json"))

    @classmethod
    def load(cls, directory: str, device: int = 0, storage: Any = None, **kw: Any) -> "MI355XMultiVectorStore":
        import os

        from .index import MvIndex

        with open(os.path.join(directory, "store.json")) as f:
            book = json.load(f)
        self = cls(capacity_pages=book["capacity_pages"], stride_rows=book["stride_rows"], device=device, mode=book["mode"], storage=storage,
                   id_base=book["id_base"], fde_coarse_n=book.get("fde_coarse_n", 0), **kw)
        self._index = MvIndex.load(os.path.join(directory, "index.mv"), device=device)
        if self.fde_coarse_n:
            from ._lib import MV_OPT_FDE_COARSE_N

            self._index.set_option(MV_OPT_FDE_COARSE_N, self.fde_coarse_n)
        for p, doc, chunk_no, content, meta_json, app in book["rows"]:
            self._rows[int(p)] = (doc, int(chunk_no), content, meta_json, app)
            self._page_of[(doc, int(chunk_no))] = int(p)
            self._doc_pages.setdefault(doc, []).append(int(p))
        self._doc_ord = {k: int(v) for k, v in book["doc_ord"].items()}
        self._doc_app = {int(k): v for k, v in book["doc_app"].items()}
        return self

    # ------------------------------------------------------------------ introspection
    def __len__(self) -> int:
        return len(self._rows)


class MI355XFastMultiVectorStore(MI355XMultiVectorStore):
    """Drop-in for FastMultiVectorStore: FDE coarse stage + exact float rerank, per-app namespaces
    (fast_multivector_store.py:504-607; `self.ns(app_id)` :526)."""

    default_mode = "fde_then_float"
    _filter_by_app = True


def create_store(provider: str, **kw: Any) -> MI355XMultiVectorStore:
    """Factory for core/services_init.py: [multivector_store] provider = "mi355x" | "mi355x_fast" | "mi355x_float"."""
    if provider == "mi355x":
        return MI355XMultiVectorStore(**kw)
    if provider == "mi355x_fast":
        return MI355XFastMultiVectorStore(**kw)
    if provider == "mi355x_float":
        return MI355XMultiVectorStore(mode="float", **kw)
    raise ValueError(f"unknown MI355X multivector provider {provider!r}")
