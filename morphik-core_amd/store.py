"""BaseVectorStore plugins backed by the MI355X index -- the drop-in boundary of SURVEY.md 8(b).

  MI355XMultiVectorStore            drop-in for core/vector_store/multi_vector_store.py:MultiVectorStore
                                    (scores = SQL max_sim on sign bits; mode "binary")
  MI355XFastMultiVectorStore        drop-in for core/vector_store/fast_multivector_store.py:FastMultiVectorStore
                                    (FDE coarse top min(10k,75) -> exact float MaxSim rerank; mode "fde_then_float";
                                    per-app namespaces)
  MI355XSharded{,Fast}MultiVectorStore   the same stores over R shards (one per GPU, or logical shards on one GPU)
                                    behind ONE object: one payload table, writes routed to the least-full shard, queries
                                    through libmvmaxsim's communicator (mv_comm: RCCL all-gather of k pairs over xGMI)
  either can run mode "float": exact float MaxSim over the WHOLE corpus (no coarse stage) -- what the
  HBM-resident slab makes affordable (1 M pages in ~40 ms on one GPU) -- "float_fp8" (the same scan over an e4m3 slab, half
  the bytes) or "fp8_then_float" (e4m3 scan -> top-128 -> exact bf16 re-score from a pinned-host exact tier: the exact scan's
  answers at the fp8 slab's HBM footprint).

Signatures, return shapes and error conventions follow the reference:
  store_embeddings -> (True, ["{document_id}-{chunk_number}", ...], metrics)   multi_vector_store.py:623-719
  query_similar    -> List[DocumentChunk] sorted by score desc, embedding=[]     multi_vector_store.py:721-817
  get_chunks_by_id -> score 0.0                                                  multi_vector_store.py:824-919
  delete_chunks_by_document_id -> bool, False on error                           multi_vector_store.py:921-951
  initialize() -> bool, never raises                                             multi_vector_store.py:186-327
Chunk content goes to the caller's `.storage` (payloads.py) exactly as the reference stores it externally; only the
storage key stays in host memory.  Scoring is done by libmvmaxsim.so only; this file is bookkeeping (ids, keys, filters).
"""
from __future__ import annotations

import asyncio
import json
import logging
import threading
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .models import BaseVectorStore, DocumentChunk, build_store_metrics, hit_chunk_builder
from .payloads import DEFAULT_APP_ID, PayloadStore, is_storage_key, parse_metadata, storage_backend_name

logger = logging.getLogger(__name__)

# what a row's content column holds -- recorded at ingest, never guessed from the text.  Rows of checkpoints written before
# round 5 carry no such field; those builds treated every key-SHAPED content as a key: this store's own when it has a storage
# object, else one a remote client uploaded (the owner server flagged it by is_storage_key and handed it back on delete).
# ROW_LEGACY_KEY keeps exactly that for such rows, in-process and behind the owner server alike.
ROW_INLINE, ROW_OWN_KEY, ROW_CLIENT_KEY, ROW_LEGACY_KEY = 0, 1, 2, 3


def row_origin(row: Sequence[Any]) -> int:
    if len(row) > 5:
        return int(row[5])
    return ROW_LEGACY_KEY if is_storage_key(row[2]) else ROW_INLINE


def _fsync_dir(path: str) -> None:
    import os

    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
    finally:
        os.close(fd)


_HELPERS = None


def _index_helpers():
    """(index._to_host, index.allow_bitmap), imported once: an `import` statement inside a per-request function costs a
    microsecond of import-system lookups every call (index.py pulls in ctypes bindings; keep its import lazy for CPU-only hosts)."""
    global _HELPERS
    if _HELPERS is None:
        from .index import _to_host, allow_bitmap

        _HELPERS = (_to_host, allow_bitmap)
    return _HELPERS


def _embedding_rows(e: Any) -> np.ndarray:
    """ndarray / torch.Tensor (any device) / list -> C-contiguous [n,128] fp32 or uint16(bf16)
    (multi_vector_store.py:334-337 and fast_multivector_store.py:515-518 accept the same inputs)."""
    a = e if isinstance(e, np.ndarray) else _index_helpers()[0](e)
    if a.dtype != np.uint16:
        a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != 128:
        raise ValueError(f"multi-vector embedding must be [n, 128]; got {a.shape}")
    return a


def _device_tensor(e: Any, devices: Sequence[int]):
    """The embedding itself when it is a torch tensor resident on one of the index's GPUs ([n,128], bf16 or fp32), else None."""
    if hasattr(e, "is_cuda") and getattr(e, "is_cuda") and e.device.index in devices and e.dim() == 2 and e.shape[1] == 128:
        if str(e.dtype) in ("torch.bfloat16", "torch.float32"):
            return e
    return None


class MI355XMultiVectorStore(BaseVectorStore):
    backend_name = "mi355x"
    default_mode = "binary"
    _filter_by_app = False  # MultiVectorStore.query_similar ignores app_id (multi_vector_store.py:721-763)

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
        pipeline_depth: int = 2,
        min_score: Optional[float] = None,
        enable_external_storage: bool = True,
        app_id_resolver: Optional[Callable[[str], Optional[str]]] = None,
        exact_tier: str = "hbm",
        rerank_n: int = 0,
        prune_slab: bool = True,
        fde_module: Any = None,
        fp32_pages: bool = False,
        fp32_scan: str = "both_halves",
        fde_e4m3: bool = False,
        fde_fp4: bool = False,
        packed_layout: bool = False,
        capacity_rows: int = 0,
        **_ignored: Any,
    ):
        self.capacity_pages = int(capacity_pages)
        self.stride_rows = int(stride_rows)
        self.device = int(device)
        self.mode = mode or self.default_mode
        if self.mode not in ("binary", "float", "fde_then_float", "float_fp8", "fp8_then_float"):
            raise ValueError(f"unknown mode {self.mode}")
        # where the exact bf16 rows of mode "fde_then_float" live: "hbm" = the bf16 slab (262 KB / page of HBM), "host" = PINNED
        # HOST memory (262 KB / page of host RAM) with an e4m3 slab in HBM beside the FDE slab -- BASELINE configs[3]'s shard
        # shape (1.25 M pages / GPU: a bf16 slab would need 328 GB of HBM): the coarse candidates are reranked exactly out of
        # host RAM (lists longer than MV_OPT_RERANK_N through an e4m3 pruning stage first).  "fp8_then_float" always uses "host".
        if exact_tier not in ("hbm", "host", "split"):
            raise ValueError(f"unknown exact_tier {exact_tier!r} (\"hbm\", \"host\" or \"split\")")
        self.exact_tier = exact_tier
        # pages read from the exact tier per request (MV_OPT_RERANK_N; 0 = the library's 128): the candidate list of "fp8_then_float",
        # the cut of the e4m3 pruning stage of "fde_then_float" over a host tier -- n x 256 KiB over PCIe per request
        self.rerank_n = int(rerank_n)
        # "fde_then_float" over a host / split exact tier: keep the e4m3 slab (131 KB / page of HBM) whose only job there is the pruning
        # stage behind coarse lists longer than rerank_n.  False: no e4m3 slab -- every candidate goes to the exact tier (the reference's
        # pipeline with its own min(10 k, 75) rule never prunes anyway), and a split tier gets that HBM for exact rows instead: a
        # 1.25 M-page shard keeps ~1 M pages' exact rows on the device and pins the rest
        self.prune_slab = bool(prune_slab)
        # FastMultiVectorStore keeps every page as an fp32 `.npy` and reranks in fp32 (fast_multivector_store.py:676-681, :736, :774, :553-555).
        # True: the bf16 slab gets its lo half (MV_WITH_FLOAT_LO: x = hi + lo to 2^-18, the same 4 bytes per element) and the rerank /
        # the float scan return that fp32 score to ~1e-6 -- for encoders whose output is not bf16 to begin with.  (A bf16 encoder under
        # autocast -- the reference's own, colpali_embedding_model.py:251-262 -- emits bf16 values: False loses nothing there.)
        self.fp32_pages = bool(fp32_pages)
        # mode "float" on fp32 pages: "both_halves" = every page scored from hi + lo (twice the bytes per page; MV_OPT_FLOAT_LO_SCAN 1),
        # "cascade" = the hi halves scanned, the best max(rerank_n, k) pages re-scored from both halves (the speed of a bf16 index, the
        # same fp32-faithful top-k scores; a batch of requests then takes ONE pass over the slab: MV_OPT_FLOAT_LO_SCAN 2)
        if fp32_scan not in ("both_halves", "cascade"):
            raise ValueError(f"unknown fp32_scan {fp32_scan!r} (\"both_halves\" or \"cascade\")")
        self.fp32_scan = fp32_scan
        # "fde_then_float": keep an e4m3 copy of the FDE slab (10 KiB per page at the reference's FDE width) and run the COARSE stage on it --
        # half the bytes of the pass that is nine tenths of a request.  The reference's coarse stage is an ANN index (approximate by contract);
        # the rerank and its scores are untouched (MV_WITH_FDE_E4M3)
        self.fde_e4m3 = bool(fde_e4m3)
        # ... or an FP4 (e2m1) copy: out_dim / 2 bytes per page, a quarter of the bf16 slab's, read by the coarse stage of single requests and of
        # coalesced batches (one pass per 32 requests on the FP4 matrix path).  MV_WITH_FDE_FP4; not together with fde_e4m3
        self.fde_fp4 = bool(fde_fp4)
        if self.fde_fp4 and self.fde_e4m3:
            raise ValueError("fde_e4m3 and fde_fp4 are two forms of the same copy of the FDE slab: choose one")
        # MV_LAYOUT_PACKED: pages of different lengths (ColQwen2.5's dynamic token counts, colpali_embedding_model.py:47-52) lie back to back
        # in whole 16-row tiles instead of one stride_rows slot each; capacity_rows sizes the row-indexed slabs (0 = capacity_pages *
        # stride_rows: no saving, only the layout).  Not with the host / split exact tiers.
        self.packed_layout = bool(packed_layout)
        self.capacity_rows = int(capacity_rows)
        # Bring-your-own FDE ("fde_then_float"): an object with the API of the reference's `fde` extension -- FixedDimensionalEncodingConfig,
        # generate_document_encoding(emb, cfg), generate_query_encoding(q, cfg) (fast_multivector_store.py:325-331, :447-449, :521).  The
        # store then calls IT, on the host as the reference does, for every chunk and every query, imports the document vectors into the
        # FDE slab and hands the query vectors to the scan: candidate generation runs on the deployment's own encodings (the published
        # algorithm this library restates matches that extension only as far as the publication pins it), the GPU does scan and rerank.
        self.fde_module = fde_module
        self._fde_ext_cfg = None
        if fde_module is not None:
            if self.mode != "fde_then_float":
                raise ValueError("fde_module only applies to mode \"fde_then_float\"")
            self._fde_ext_cfg = fde_module.FixedDimensionalEncodingConfig(dimension=128, num_repetitions=20, num_simhash_projections=5,
                                                                          projection_dimension=16, projection_type="AMS_SKETCH")
        self.storage = storage  # callers reach for .storage (document_service.py:1570-1575)
        # multi_vector_store.py:120-160: content is stored externally when a storage object is configured
        self.enable_external_storage = bool(enable_external_storage)
        self._payloads = PayloadStore(storage) if storage is not None else None
        # store_embeddings resolves a missing app_id from the document (multi_vector_store.py:644-648 /
        # fast_multivector_store.py:440-444 -> SELECT app_id FROM documents); the callback plays that lookup's role
        self._app_id_resolver = app_id_resolver
        self.id_base = int(id_base)
        self.fde_coarse_n = int(fde_coarse_n)
        self._index_factory = index_factory
        # request coalescing (modes "float", "float_fp8", "fp8_then_float" and "fde_then_float"): concurrent query_similar calls arriving within
        # batch_window_ms are scored in ONE slab pass (batched MFMA MaxSim scan / batched FDE pipeline: up to 32 requests per
        # pass over the FDE slab, every request's candidates reranked exactly), each keeping its own doc_ids filter and k
        # batch_window_ms > 0: a timer window (a lone request pays it); batch_window_ms < 0: ADAPTIVE (group commit) -- a
        # request that finds the index idle is dispatched at once, requests arriving while a scan is in flight ride the next
        # pass together: no added latency when idle, natural batches under load
        self.batch_window_s = float(batch_window_ms) / 1e3
        self.max_batch = int(max_batch)
        # adaptive mode: a SECOND batch may be dispatched while one is on the device, but only a full one (max_batch requests
        # waiting): the library serialises the two passes, and the event loop builds the hits of pass N while pass N+1 runs
        # (ctypes releases the GIL) -- without it the GPU idles for the ~0.7 ms of Python that 32 finished requests cost
        self.pipeline_depth = max(1, int(pipeline_depth))
        self._inflight = 0
        self._flush_scheduled = False
        # The API accepts min_score (core/models/request.py:138) and threads it through retrieve_chunks
        # (document_service.py:184) but the reference never applies it to multivector hits; None keeps that behaviour,
        # a number drops hits scoring below it
        self.min_score = None if min_score is None else float(min_score)
        self.collect_device_time = False  # True: last_query_timing carries the library's device time of every request (benchmarks)
        self._pending: List[Tuple[np.ndarray, int, Any, Any]] = []
        self._flush_handle = None
        self.coalesced_batches: List[int] = []  # sizes of the batches actually issued (introspection / tests)
        self._index = None
        self._lock = threading.RLock()
        # writers (store / delete / compact) and save() exclude each other here; queries never take it (save() holds the store
        # lock only for the bookkeeping snapshot, so the event loop keeps serving while the slabs are dumped)
        self._write_gate = threading.RLock()
        # bookkeeping: page -> (document_id, chunk_number, content OR storage key, metadata_json, app_id)
        self._rows: Dict[int, Tuple[str, int, str, str, Optional[str]]] = {}
        self._page_of: Dict[Tuple[str, int], int] = {}
        self._doc_ord: Dict[str, int] = {}
        self._next_ord = 0
        self._doc_app: Dict[int, Optional[str]] = {}
        self._doc_pages: Dict[str, List[int]] = {}
        # bumped by compact() (page ids are renumbered): a query whose scan ran against the old numbering is re-run
        self._generation = 0
        # bumped whenever the document-ordinal / app tables change: cached doc_ids / app bitmaps are valid for one stamp
        self._ord_stamp = 0
        self._allow_cache: Dict[Any, Tuple[int, Any, bool]] = {}
        self._last_store_metrics: Dict[str, Any] = {}
        self.last_query_timing: Dict[str, float] = {}
        self._hit_chunk = hit_chunk_builder()

    # ------------------------------------------------------------------ lifecycle
    def _devices(self) -> List[int]:
        return [self.device]

    def _slab_flags(self) -> Dict[str, bool]:
        # "fp8_then_float": e4m3 slab in HBM (131 KB / page) + the exact bf16 rows in PINNED HOST memory (262 KB / page of host
        # RAM): every page scanned in fp8, the top candidates re-scored exactly out of host RAM by the rerank kernel itself --
        # exact-scan answers for corpora whose bf16 slab does not fit the GPU (BASELINE configs[4] with recall 1.0)
        # exact_tier "split": the host tier, with the exact rows of the leading pages in whatever HBM the other slabs leave free
        host = self.mode == "fp8_then_float" or (self.mode == "fde_then_float" and self.exact_tier in ("host", "split"))
        with_float = self.mode == "float" or (self.mode == "fde_then_float" and not host)
        if self.packed_layout and host:
            raise ValueError("packed_layout cannot be combined with a host / split exact tier")
        return dict(with_float=with_float, with_binary=self.mode == "binary", **({"with_float_lo": True} if with_float and self.fp32_pages else {}),
                    **({"packed": True, "capacity_rows": self.capacity_rows} if self.packed_layout else {}),
                    with_fde=self.mode == "fde_then_float", **({"with_fde_e4m3": True} if self.mode == "fde_then_float" and self.fde_e4m3 else {}), **({"with_fde_fp4": True} if self.mode == "fde_then_float" and self.fde_fp4 else {}), with_fp8=self.mode in ("float_fp8", "fp8_then_float") or (self.mode == "fde_then_float" and host and self.prune_slab),
                    **({"with_host_exact": True} if host else {}), **({"with_exact_split": True} if host and self.exact_tier == "split" else {}))

    def _make_index(self):
        if self._index_factory is not None:
            return self._index_factory(capacity_pages=self.capacity_pages, stride_rows=self.stride_rows, device=self.device,
                                       id_base=self.id_base, mode=self.mode)
        from .index import MvIndex  # HIP-only: raises MvError when libmvmaxsim.so or the GPU is missing

        ix = MvIndex(capacity_pages=self.capacity_pages, stride_rows=self.stride_rows, device=self.device, id_base=self.id_base,
                     **self._slab_flags())
        self._apply_options(ix)
        return ix

    def _apply_options(self, ix) -> None:
        if self.fde_coarse_n:
            from ._lib import MV_OPT_FDE_COARSE_N

            ix.set_option(MV_OPT_FDE_COARSE_N, self.fde_coarse_n)
        if self.rerank_n:
            from ._lib import MV_OPT_RERANK_N

            ix.set_option(MV_OPT_RERANK_N, self.rerank_n)
        if self.fp32_pages and self.fp32_scan == "cascade" and self.mode == "float":
            from ._lib import MV_OPT_FLOAT_LO_SCAN

            ix.set_option(MV_OPT_FLOAT_LO_SCAN, 2)

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
    def _resolve_app(self, app_id: Optional[str], chunks: List[DocumentChunk]) -> Optional[str]:
        """app_id given -> it; else, for a single-document batch, the document's app (callback) as the reference looks it
        up in the documents table; stores with per-app namespaces fall back to "default"."""
        if app_id is not None:
            return app_id
        if self._app_id_resolver is not None and chunks and all(c.document_id == chunks[0].document_id for c in chunks):
            try:
                got = self._app_id_resolver(chunks[0].document_id)
                if got:
                    return got
            except Exception as e:  # noqa: BLE001
                logger.warning(f"Failed to get app_id for document {chunks[0].document_id}: {e}")
        return DEFAULT_APP_ID if self._filter_by_app else None

    def _nk(self, document_id: str, app_id: Optional[str]) -> str:
        """Bookkeeping key of a document.  Stores with per-app namespaces key by (app, document_id): the same document_id under
        two apps is two documents with two ordinals -- FastMultiVectorStore writes to self.ns(app_id) and never touches another
        namespace (fast_multivector_store.py:440-502) -- so a re-ingest under app B can neither expose nor delete app A's chunks."""
        if not self._filter_by_app:
            return document_id
        return f"{app_id if app_id is not None else DEFAULT_APP_ID}\x1f{document_id}"

    def _store_sync(self, valid: List[DocumentChunk], embs: List[Any], contents: List[str], app_id: Optional[str],
                    origins: Optional[List[int]] = None) -> List[str]:
        ix = self._require_index()
        origins = origins if origins is not None else [ROW_INLINE] * len(valid)
        with self._write_gate, self._lock:
            ords, fresh = [], []
            for c in valid:
                key = self._nk(c.document_id, app_id)
                o = self._doc_ord.get(key)
                if o is None:
                    o = self._next_ord
                    self._next_ord += 1
                    self._doc_ord[key] = o
                    self._doc_app[o] = app_id
                    fresh.append(key)
                ords.append(o)
            try:
                first = self._add_pages(ix, embs, ords)
            except Exception:
                # ADD FIRST: the slab was full or the device call failed -- nothing was published (mv_index_add is all or
                # nothing), the previous versions of these chunks are still live, and the ordinals handed out above are taken back
                for key in fresh:
                    self._doc_app.pop(self._doc_ord.pop(key), None)
                raise
            if fresh:
                self._ord_stamp += 1
            first += 0 if self._global_ids else self.id_base
            # upsert: the previous page of an existing (document, chunk_number) is retired only now
            # (FastMultiVectorStore upserts by id)
            for c in valid:
                key = self._nk(c.document_id, app_id)
                old = self._page_of.pop((key, int(c.chunk_number)), None)
                if old is not None:
                    ix.remove_page(old if self._global_ids else old - self.id_base)
                    self._rows.pop(old, None)
                    if old in self._doc_pages.get(key, []):
                        self._doc_pages[key].remove(old)
            ids = []
            for i, c in enumerate(valid):
                page = first + i
                key = self._nk(c.document_id, app_id)
                self._rows[page] = (c.document_id, int(c.chunk_number), contents[i], json.dumps(c.metadata or {}), app_id, origins[i])
                self._page_of[(key, int(c.chunk_number))] = page
                self._doc_pages.setdefault(key, []).append(page)
                ids.append(f"{c.document_id}-{c.chunk_number}")
            return ids

    def _external_doc_fdes(self, embs: List[Any]) -> Optional[np.ndarray]:
        """The deployment's own document encodings of these pages (fde_module), or None.  Computed BEFORE anything is appended: an
        encoder failure or a malformed vector aborts the call with nothing stored."""
        if self.fde_module is None:
            return None
        out = []
        for e in embs:
            rows = e if isinstance(e, np.ndarray) else _embedding_rows(e) if not hasattr(e, "detach") else e.detach().to("cpu").float().numpy()
            v = np.asarray(self.fde_module.generate_document_encoding(np.asarray(rows, np.float32), self._fde_ext_cfg), np.float32).reshape(-1)
            out.append(v)
        want = self._fde_out_dim()
        for v in out:
            if want and v.size != want:  # before anything is appended: vectors of another width are never regrouped into pages
                raise ValueError(f"fde_module.generate_document_encoding returned {v.size} floats; the index's FDE holds {want} per page")
        docs = np.stack(out) if out else np.zeros((0, 0), np.float32)
        if out and not np.isfinite(docs).all():
            raise ValueError("fde_module.generate_document_encoding returned NaN / Inf")
        return docs

    def _fde_out_dim(self) -> int:
        """Width of one FDE vector of the index (0 if it cannot be told, e.g. a test double without an FDE config)."""
        ix = self._index
        cfg = getattr(ix, "fde_config", None) if ix is not None else None
        if cfg is None and ix is not None and getattr(ix, "shards", None):
            cfg = getattr(ix.shards[0], "fde_config", None)
        return int(cfg.output_dim) if cfg is not None else 0

    def _query_fde_kw(self, q: Any) -> Dict[str, Any]:
        if self.fde_module is None:
            return {}
        rows = q if isinstance(q, np.ndarray) else _embedding_rows(q)
        v = np.asarray(self.fde_module.generate_query_encoding(np.asarray(rows, np.float32), self._fde_ext_cfg), np.float32).reshape(-1)
        want = self._fde_out_dim()
        if want and v.size != want:  # the *_fde entry points of the C ABI read `want` floats from a bare pointer
            raise ValueError(f"fde_module.generate_query_encoding returned {v.size} floats; the index's FDE holds {want}")
        return {"q_fde": v}

    def _add_pages(self, ix, embs: List[Any], ords: List[int]) -> int:
        """Append the pages to the slab (all or nothing) -> first page id the index assigned."""
        docs = self._external_doc_fdes(embs)
        first = self._add_pages_raw(ix, embs, ords)
        if docs is not None and len(docs):
            try:
                ix.import_fde(first, docs)  # the library's own encodings of these pages are replaced before the pages get their bookkeeping
            except Exception:
                # the pages are live in the slab but will never get bookkeeping (_store_sync takes the ordinals back): retire them,
                # or they would hold slab slots and top-k positions for rows nobody can return
                for page in range(first, first + len(embs)):
                    try:
                        ix.remove_page(page)
                    except Exception:  # noqa: BLE001
                        logger.error(f"could not retire page {page} after a failed import_fde")
                raise
        return first

    def _add_pages_raw(self, ix, embs: List[Any], ords: List[int]) -> int:
        if embs and all(not isinstance(e, np.ndarray) for e in embs):
            # ingest-side fusion: encoder output already on this GPU -> one D2D pass fills
            # every slab (mv_index_add_device); no D2H -> fp32 -> H2D round trip
            import torch

            from ._lib import MV_BF16, MV_F32

            code = MV_BF16 if all(e.dtype == torch.bfloat16 for e in embs) else MV_F32
            flat = torch.cat([e if code == MV_BF16 else e.to(torch.float32) for e in embs], 0).contiguous()
            torch.cuda.current_stream(flat.device).synchronize()  # the library orders on its own stream
            kw = {"device": flat.device.index} if len(self._devices()) > 1 or flat.device.index != self.device else {}
            return ix.add_device(flat.data_ptr(), code, [int(e.shape[0]) for e in embs], ords, **kw)
        return ix.add([e if isinstance(e, np.ndarray) else _embedding_rows(e) for e in embs], ords)

    _global_ids = False  # a ShardedIndex hands out global page ids itself

    async def store_embeddings(self, chunks: List[DocumentChunk], app_id: Optional[str] = None,
                               content_is_key: Optional[Sequence[bool]] = None) -> Tuple[bool, List[str], Dict[str, Any]]:
        """content_is_key (owner-server path only, store_server.py): per chunk, True = `content` is the key of a payload the
        remote CLIENT uploaded to ITS storage.  Nothing is ever inferred from what the content looks like: without the flag the
        content is content and is uploaded (when this store has a storage) exactly as the reference does
        (multi_vector_store.py:650-676) -- a chunk whose text merely resembles another tenant's key is just text."""
        payload_backend = storage_backend_name(self.storage) if self._use_external() else "memory"
        valid: List[DocumentChunk] = []
        key_flags: List[bool] = []
        if content_is_key is not None and len(content_is_key) != len(chunks):
            raise ValueError("content_is_key must have one entry per chunk")
        for j, chunk in enumerate(chunks):
            if not hasattr(chunk, "embedding") or chunk.embedding is None:
                logger.error(f"Missing embeddings for chunk {chunk.document_id}-{chunk.chunk_number}")
                continue
            valid.append(chunk)
            key_flags.append(bool(content_is_key[j]) if content_is_key is not None else False)
        if not valid:
            self._last_store_metrics = build_store_metrics(
                chunk_payload_backend=payload_backend, multivector_backend=self.backend_name, vector_store_backend=self.backend_name
            )
            return True, [], self._last_store_metrics
        resolved_app = self._resolve_app(app_id, valid)
        use_dev = self._index_factory is None
        devs = self._devices()
        embs = [(_device_tensor(c.embedding, devs) if use_dev else None) for c in valid]
        if len({e.device.index for e in embs if e is not None}) > 1:
            embs = [None] * len(valid)  # rows spread over several GPUs: take the host path
        embs = [e if e is not None else _embedding_rows(c.embedding) for c, e in zip(valid, embs)]
        if any(isinstance(e, np.ndarray) for e in embs) and not all(isinstance(e, np.ndarray) for e in embs):
            embs = [e if isinstance(e, np.ndarray) else _embedding_rows(e) for e in embs]
        for c, e in zip(valid, embs):
            if e.shape[0] > self.stride_rows:
                raise ValueError(
                    f"chunk {c.document_id}-{c.chunk_number} has {e.shape[0]} vectors; this store was created with "
                    f"stride_rows={self.stride_rows}"
                )
        # chunk content -> external storage, keys stay here (multi_vector_store.py:650-676)
        contents = [c.content for c in valid]
        origins = [ROW_CLIENT_KEY if f else ROW_INLINE for f in key_flags]
        payload_s, payload_objects, payload_bytes = 0.0, 0, 0
        if self._use_external():
            t0 = time.perf_counter()
            mine = [i for i, f in enumerate(key_flags) if not f]  # a client's keys stay keys; everything else is uploaded
            res = await asyncio.gather(*[self._payloads.put(valid[i].content, valid[i].document_id, int(valid[i].chunk_number), valid[i].metadata or {}, resolved_app)
                                         for i in mine])
            payload_s = time.perf_counter() - t0
            for i, (key, nbytes) in zip(mine, res):
                if key:
                    contents[i] = key
                    origins[i] = ROW_OWN_KEY
                    payload_objects += 1
                    payload_bytes += nbytes
                else:
                    logger.warning(f"Failed to store chunk {valid[i].document_id}-{valid[i].chunk_number} externally, keeping it inline")
        t0 = time.perf_counter()
        try:
            ids = await asyncio.to_thread(self._store_sync, valid, embs, contents, resolved_app, origins)
        except Exception:
            own = [contents[i] for i, o in enumerate(origins) if o == ROW_OWN_KEY]
            if own:  # the slab refused the pages: the payloads uploaded for them must not stay behind
                try:
                    await self._payloads.delete(own, valid[0].document_id)
                except Exception as e:  # noqa: BLE001
                    logger.error(f"could not remove {len(own)} payloads uploaded for a failed store_embeddings: {e}")
            raise
        dt = time.perf_counter() - t0
        self._last_store_metrics = build_store_metrics(
            chunk_payload_backend=payload_backend, multivector_backend=self.backend_name, vector_store_backend=self.backend_name,
            chunk_payload_upload_s=payload_s, chunk_payload_objects=payload_objects, chunk_payload_bytes=payload_bytes,
            multivector_upload_s=dt, multivector_objects=len(ids), multivector_bytes=int(sum(e.shape[0] for e in embs)) * 256,
            vector_store_write_s=dt, vector_store_rows=len(ids),
        )
        return True, ids, self._last_store_metrics

    def _use_external(self) -> bool:
        return self.enable_external_storage and self._payloads is not None

    # ------------------------------------------------------------------ query
    def _allow_for(self, doc_ids: Optional[List[str]], app_id: Optional[str]):
        """doc_ids falsy => no doc filter (multi_vector_store.py:754).  Stores with per-app namespaces ALWAYS restrict to
        the resolved app: FastMultiVectorStore queries self.ns(app_id), so app_id=None reads the default namespace only
        (fast_multivector_store.py:526).  Returns (bitmap or None, empty?)."""
        allow_bitmap = _index_helpers()[1]
        # The bitmap of a (doc_ids, app) pair only changes when documents are added / compacted away (_ord_stamp): requests
        # repeat the same authorised document set, so it is built once per stamp, not once per request (deleted documents
        # keep their ordinal until compact(); their pages are tombstoned in the slab itself).
        want = (app_id if app_id is not None else DEFAULT_APP_ID) if self._filter_by_app else None
        key = (tuple(doc_ids) if doc_ids else None, want)
        hit = self._allow_cache.get(key)
        if hit is not None and hit[0] == self._ord_stamp:
            return hit[1], hit[2]
        ords = None
        if doc_ids:
            ords = [self._doc_ord[kd] for kd in (self._nk(d, want) for d in doc_ids) if kd in self._doc_ord]
        if self._filter_by_app:
            if ords is None and all(a == want for a in self._doc_app.values()):
                ords = None  # every document belongs to this app: the namespace filter is the identity
            else:
                base = list(self._doc_ord.values()) if ords is None else ords
                ords = [o for o in base if self._doc_app.get(o) == want]
        if ords is None:
            res = (None, False)
        elif not ords:
            res = (None, True)
        else:
            res = (allow_bitmap(ords, self._next_ord), False)
        if len(self._allow_cache) >= 512:
            self._allow_cache.clear()
        self._allow_cache[key] = (self._ord_stamp, res[0], res[1])
        return res

    def _query_sync(self, q: np.ndarray, k: int, allow) -> Tuple[np.ndarray, np.ndarray]:
        ix = self._require_index()
        t0 = time.perf_counter()
        want_stats = self._index_factory is None and (self.collect_device_time or (self.mode == "fde_then_float" and logger.isEnabledFor(logging.INFO)))
        res = ix.query(q, k, mode=self.mode, allow=allow, want_stats=want_stats, **self._query_fde_kw(q))
        dt = time.perf_counter() - t0
        self.last_query_timing = {"vector_search_s": dt}
        if want_stats and len(res) == 3 and not isinstance(res[2], list):
            st = res[2]
            self.last_query_timing["device_ms"] = st.total_device_ms
        if want_stats and len(res) == 3 and not isinstance(res[2], list) and self.mode == "fde_then_float" and logger.isEnabledFor(logging.INFO):
            st = res[2]
            # the reference's stage lines (fast_multivector_store.py:523-577), device-side: there is no network hop and no
            # multivector download -- the candidates never leave HBM
            logger.info(f"query_similar timing - encode_query: {st.encode_ms:.2f} ms")
            logger.info(f"query_similar timing - ns.query: {st.coarse_ms + st.select_ms:.2f} ms")
            logger.info("query_similar timing - load_multivectors: 0.00 ms")
            logger.info(f"query_similar timing - rerank_scoring: {st.rerank_ms + st.topk_ms:.2f} ms")
            self.last_query_timing.update(encode_query_ms=st.encode_ms, ns_query_ms=st.coarse_ms + st.select_ms,
                                          rerank_scoring_ms=st.rerank_ms + st.topk_ms, device_ms=st.total_device_ms)
        return res[0], res[1]

    # -- request coalescing
    def _batch_sync(self, items: List[Tuple[np.ndarray, int, Any, Any]]):
        ix = self._require_index()
        with self._lock:
            n_docs = self._next_ord
        t0 = time.perf_counter()
        if self.mode in ("float", "float_fp8"):
            groups = [list(range(len(items)))]  # full scan: the top-k of a larger k is a prefix, one pass serves every k
        else:
            # the FDE stage keeps min(10*k, 75) candidates (fast_multivector_store.py:529): requests share a pass only with
            # requests of the same k, so each sees exactly the candidates a lone call would have reranked
            by_k: Dict[int, List[int]] = {}
            for j, (_q, k, _a, _f) in enumerate(items):
                by_k.setdefault(int(k), []).append(j)
            groups = list(by_k.values())
        out: List[Any] = [None] * len(items)
        device_ms = 0.0
        for g in groups:
            kmax = max(items[j][1] for j in g)
            allows = [items[j][2] for j in g]
            fkw = {} if self.fde_module is None else {"q_fdes": np.stack([self._query_fde_kw(items[j][0])["q_fde"] for j in g])}
            res = ix.query_batch([items[j][0] for j in g], kmax, mode=self.mode, allows=allows if any(a is not None for a in allows) else None,
                                 n_docs=n_docs, want_stats=self._index_factory is None, **fkw)
            if isinstance(res, tuple):  # (results, QueryStats)
                res, st = res
                device_ms += float(getattr(st, "total_device_ms", 0.0))
            for j, (s, i) in zip(g, res):
                out[j] = (s[: items[j][1]], i[: items[j][1]])
        self.last_query_timing = {"vector_search_s": time.perf_counter() - t0, "batched_queries": len(items), "device_ms": device_ms}
        return out

    def _flush(self) -> None:
        self._flush_scheduled = False
        items, self._pending = self._pending[: self.max_batch], self._pending[self.max_batch :]
        if self._flush_handle is not None:
            self._flush_handle.cancel()
            self._flush_handle = None
        if not items:
            return
        self.coalesced_batches.append(len(items))
        self._inflight += 1

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
            finally:
                self._inflight -= 1
                if self._pending and self.batch_window_s < 0 and self._may_dispatch():
                    self._flush()  # adaptive: everything that arrived during this pass rides the next one

        asyncio.ensure_future(run())
        if self.batch_window_s < 0 and self._pending and self._may_dispatch():
            self._flush()  # a full batch is still waiting and the pipeline has room

    def _may_dispatch(self) -> bool:
        """Adaptive mode: an idle index takes whatever waits; behind a pass in flight only a full batch goes out."""
        return self._inflight == 0 or (self._inflight < self.pipeline_depth and len(self._pending) >= self.max_batch)

    async def _coalesced_query(self, q: np.ndarray, k: int, allow) -> Tuple[np.ndarray, np.ndarray]:
        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        self._pending.append((q, k, allow, fut))
        if self.batch_window_s < 0:  # adaptive: dispatch when idle (after the requests that are ready in this same loop tick)
            if not self._flush_scheduled and self._may_dispatch():
                self._flush_scheduled = True
                loop.call_soon(self._flush)
        elif len(self._pending) >= self.max_batch:
            self._flush()
        elif self._flush_handle is None:
            self._flush_handle = loop.call_later(self.batch_window_s, self._flush)
        return await fut

    async def _resolve_contents(self, rows: List[Tuple[str, int, str, str, Optional[str]]], skip_image_content: bool) -> Tuple[List[str], List[dict]]:
        """content column -> content: storage keys are downloaded, except image payloads the caller asked to skip (their
        key is returned as the content: multi_vector_store.py:778-790, fast_multivector_store.py:583-586)."""
        metas = [parse_metadata(r[3]) for r in rows]
        out = [r[2] for r in rows]
        if not self._use_external():
            return out, metas  # nothing to fetch: no task, no await (the common case of an in-memory payload table)
        fetch = [j for j, (r, m) in enumerate(zip(rows, metas)) if row_origin(r) in (ROW_OWN_KEY, ROW_LEGACY_KEY) and not (skip_image_content and m.get("is_image"))]
        if fetch:
            resolved = await asyncio.gather(*[self._payloads.get(rows[j][2], metas[j]) for j in fetch], return_exceptions=True)
            for j, c in zip(fetch, resolved):
                if isinstance(c, Exception):
                    logger.error("Failed to retrieve content from storage for chunk %s-%s: %s", rows[j][0], rows[j][1], c)
                else:
                    out[j] = c
        return out, metas

    async def query_similar(
        self,
        query_embedding: Any,
        k: int,
        doc_ids: Optional[List[str]] = None,
        app_id: Optional[str] = None,
        skip_image_content: bool = False,
    ) -> List[DocumentChunk]:
        t_start = time.perf_counter()
        q = _embedding_rows(query_embedding)
        if k <= 0:
            return []
        for _attempt in range(8):
            with self._lock:
                allow, empty = self._allow_for(doc_ids, app_id)
                gen = self._generation
            if empty:
                return []
            if self.batch_window_s != 0 and self.mode in ("float", "fde_then_float", "float_fp8", "fp8_then_float"):
                scores, pages = await self._coalesced_query(q, int(k), allow)
            else:
                scores, pages = await asyncio.to_thread(self._query_sync, q, int(k), allow)  # exceptions propagate (:819-822)
            hits: List[Tuple[float, Tuple[str, int, str, str, Optional[str]]]] = []
            min_score, rows_get = self.min_score, self._rows.get
            with self._lock:
                if self._generation != gen:
                    continue  # compact() renumbered the pages while the scan ran: these ids are stale, scan again
                for s, p in zip(scores.tolist(), pages.tolist()):  # tolist(): python float / int already
                    if min_score is not None and s < min_score:
                        break  # hits are sorted by score desc
                    row = rows_get(p)
                    if row is not None:  # None: deleted between scan and lookup
                        hits.append((s, row))
            break
        else:
            raise RuntimeError("query_similar: the index was compacted during every attempt")
        t_scan = time.perf_counter()
        hit = self._hit_chunk  # models.hit_chunk_builder(): the stored values were validated at ingest
        if self.enable_external_storage and self._payloads is not None:
            contents, metas = await self._resolve_contents([r for _s, r in hits], skip_image_content)
            out = [hit(r[0], r[1], c, m, s) for (s, r), c, m in zip(hits, contents, metas)]
        else:  # in-memory payload table: nothing to fetch, no coroutine, one pass over the hits
            out = [hit(r[0], r[1], r[2], parse_metadata(r[3]), s) for s, r in hits]
        if self.mode == "fde_then_float" and logger.isEnabledFor(logging.INFO):
            t_end = time.perf_counter()
            logger.info(f"query_similar timing - load_contents: {(t_end - t_scan)*1000:.2f} ms")
            logger.info(f"query_similar total time: {(t_end - t_start)*1000:.2f} ms")
        return out

    async def get_chunks_by_id(self, chunk_identifiers: List[Tuple[str, int]], app_id: Optional[str] = None,
                               skip_image_content: bool = False) -> List[DocumentChunk]:
        if not chunk_identifiers:
            return []
        rows = []
        want = (app_id if app_id is not None else DEFAULT_APP_ID) if self._filter_by_app else None
        with self._lock:
            for doc_id, chunk_no in dict.fromkeys((d, int(c)) for d, c in chunk_identifiers):
                # per-app namespaces: a chunk is visible through ITS app only (fast_multivector_store.py:615 reads self.ns(app_id))
                page = self._page_of.get((self._nk(doc_id, want), chunk_no))
                if page is not None:
                    rows.append(self._rows[page])
        contents, metas = await self._resolve_contents(rows, skip_image_content)
        return [DocumentChunk(document_id=r[0], chunk_number=r[1], content=c, embedding=[], metadata=m, score=0.0)
                for r, c, m in zip(rows, contents, metas)]

    def content_key_flags(self, chunk_identifiers: Sequence[Tuple[str, int]], app_id: Optional[str] = None) -> List[bool]:
        """Per (document_id, chunk_number): True = the chunk's content column holds a key a remote CLIENT uploaded (the owner
        server marks such chunks so that the client dereferences those and nothing else)."""
        want = (app_id if app_id is not None else DEFAULT_APP_ID) if self._filter_by_app else None
        out = []
        with self._lock:
            for doc_id, chunk_no in chunk_identifiers:
                page = self._page_of.get((self._nk(doc_id, want), int(chunk_no)))
                row = self._rows.get(page) if page is not None else None
                o = row_origin(row) if row is not None else ROW_INLINE
                out.append(o == ROW_CLIENT_KEY or (o == ROW_LEGACY_KEY and not self._use_external()))
        return out

    def _delete_sync(self, document_id: str, app_id: Optional[str], all_keys: Optional[List[str]] = None) -> List[str]:
        """Tombstone one app's copy of a document -> the storage keys of its payloads THIS store uploaded (all_keys, when given,
        collects the keys a remote client uploaded to its own storage and flagged as such at ingest: they go back to it).  Runs in a worker thread: it waits behind a
        checkpoint in progress (_write_gate) without holding up the event loop."""
        keys: List[str] = []
        with self._write_gate, self._lock:
            # per-app namespaces: only THIS app's copy of the document (fast_multivector_store.py:643 deletes from self.ns(app_id))
            key = self._nk(document_id, app_id)
            o = self._doc_ord.get(key)
            if o is None:
                return keys  # DELETE of nothing succeeds
            ix = self._require_index()
            ix.remove_doc(o)
            for page in self._doc_pages.pop(key, []):
                row = self._rows.pop(page, None)
                if row is not None:
                    self._page_of.pop((key, row[1]), None)
                    o_ = row_origin(row)
                    if o_ == ROW_LEGACY_KEY:  # pre-round-5 row: this store's key when it has storage, else a client's
                        o_ = ROW_OWN_KEY if self._use_external() else ROW_CLIENT_KEY
                    if o_ == ROW_CLIENT_KEY and all_keys is not None:
                        all_keys.append(row[2])  # a client's payload: only that client may delete it
                    elif o_ == ROW_OWN_KEY and self._use_external():
                        keys.append(row[2])
            # the ordinal stays reserved until compact() (its pages are tombstoned in the slab under that ordinal)
        return keys

    async def delete_chunks_returning_keys(self, document_id: str, app_id: Optional[str] = None) -> Tuple[bool, List[str]]:
        """delete_chunks_by_document_id that also reports the storage keys of the deleted chunks' payloads which this store could
        NOT delete itself (it has no storage object: a remote client uploaded them to its own) -- the owner server hands them back
        to that client, which removes the objects as the reference does on delete (multi_vector_store.py:921-951)."""
        all_keys: List[str] = []
        try:
            keys = await asyncio.to_thread(self._delete_sync, document_id, app_id, all_keys)
            if keys:
                await self._payloads.delete(keys, document_id)
            return True, all_keys
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error deleting chunks for document {document_id}: {e}")
            return False, []

    async def delete_chunks_by_document_id(self, document_id: str, app_id: Optional[str] = None) -> bool:
        try:
            keys = await asyncio.to_thread(self._delete_sync, document_id, app_id)
            logger.info(f"Deleted all chunks for document {document_id} from {self.backend_name} store")
            if keys:
                await self._payloads.delete(keys, document_id)
            return True
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error deleting chunks for document {document_id}: {e}")
            return False

    # ------------------------------------------------------------------ maintenance
    def compact(self) -> int:
        """Reclaim the slab slots of deleted / replaced pages (mv_index_compact) and remap the bookkeeping.
        Returns the number of slots reclaimed.  Page ids are internal to the store, so callers see no change; a query whose
        scan overlapped the renumbering notices the generation change and scans again."""
        with self._write_gate, self._lock:
            ix = self._require_index()
            before = len(ix)
            o2n = ix.compact()
            if isinstance(o2n, dict):
                remap = o2n
            else:
                remap = {self.id_base + int(o): self.id_base + int(n) for o, n in enumerate(o2n.tolist()) if n >= 0}
            self._rows = {remap[p]: r for p, r in self._rows.items() if p in remap}
            self._page_of = {key: remap[p] for key, p in self._page_of.items() if p in remap}
            self._doc_pages = {d: [remap[p] for p in ps if p in remap] for d, ps in self._doc_pages.items()}
            # documents without a live page no longer appear in the slab: forget their ordinals (a re-ingest gets a new one)
            for d in [d for d in self._doc_ord if not self._doc_pages.get(d)]:
                self._doc_app.pop(self._doc_ord.pop(d), None)
                self._doc_pages.pop(d, None)
            self._generation += 1
            self._ord_stamp += 1
            return before - len(ix)

    def rebalance_exact_tier(self, max_moves: int = 0) -> int:
        """exact_tier="split" only: move the pages the reranks read most into the HBM part of the exact tier (the library counts the
        reads per page; mv_index_exact_tier_rebalance).  -> pages moved.  Answers do not change; the PCIe share of the reranks does.
        Writers and queries wait for its duration: call it from a maintenance task (or after compact())."""
        with self._write_gate, self._lock:
            ix = self._require_index()
            f = getattr(ix, "rebalance_exact_tier", None)
            return int(f(max_moves)) if f is not None else 0

    def place_fde_slab(self, trials: int = 3):
        """FDE modes only: try up to `trials` other device allocations for the FDE slab and keep the one the batched coarse pass reads
        fastest (mv_index_fde_placement_trial, DESIGN 3.20: the pass's time follows the slab's allocation, up to 10 % apart).  ->
        [(pass ms before, pass ms after, moves)] per shard; [] when the index has no FDE slab.  Answers do not change.  Peak device memory
        three slabs; writers and queries wait for its duration (seconds): call it once after a bulk load or a checkpoint load."""
        with self._write_gate, self._lock:
            ix = self._require_index()
            f = getattr(ix, "fde_placement_trial", None)
            if f is None or "fde" not in self.mode:
                return []
            r = f(trials)
            return [tuple(r)] if r and not isinstance(r[0], tuple) else list(r)

    # ------------------------------------------------------------------ checkpoint / resume
    def _book(self) -> Dict[str, Any]:
        return {
            "version": 3, "mode": self.mode, "capacity_pages": self.capacity_pages, "stride_rows": self.stride_rows,
            "id_base": self.id_base, "fde_coarse_n": self.fde_coarse_n, "exact_tier": self.exact_tier, "rerank_n": self.rerank_n, "prune_slab": self.prune_slab, "fp32_pages": self.fp32_pages, "fp32_scan": self.fp32_scan, "fde_e4m3": self.fde_e4m3, "fde_fp4": self.fde_fp4, "packed_layout": self.packed_layout, "capacity_rows": self.capacity_rows, "fde_external": self.fde_module is not None, "next_ord": self._next_ord,
            "rows": [[p, r[0], r[1], r[2], r[3], r[4], row_origin(r)] for p, r in self._rows.items()],
            "doc_ord": self._doc_ord, "doc_app": {str(k): v for k, v in self._doc_app.items()},
        }

    @staticmethod
    def checkpoint_path(directory: str) -> str:
        """Directory of the CURRENT checkpoint generation under `directory` (`directory` itself for the flat layout of checkpoints
        written before generations existed)."""
        import os

        cur = os.path.join(directory, "CURRENT")
        if os.path.exists(cur):
            with open(cur) as f:
                return os.path.join(directory, f.read().strip())
        return directory

    def save(self, directory: str) -> None:
        """Persist the HBM index (mv_index_save: raw slabs + metadata) and the store's bookkeeping (keys, document ordinals) so
        a restarted process resumes without re-embedding -- the role Postgres / S3 play for the reference stores (SURVEY.md
        section 5, checkpoint/resume).
        Every checkpoint is a fresh GENERATION: <directory>/gen-<id>/{index.mv*, store.json} is written and fsync'ed in full while
        <directory>/CURRENT still names the previous generation; CURRENT is then replaced by one atomic rename and the older
        generations are removed; the parent directory
        is fsync'ed after the generation directory is created and again after the rename, before any removal.  A crash, OOM, kill or
        power loss at ANY point leaves CURRENT naming a complete checkpoint (the previous one until the rename, the new one after): the owner process holds the only copy of the corpus, an interrupted periodic save
        must not cost it.
        Locking: writers (store / delete / compact) are held off for the duration (_write_gate); the store lock is held only while
        the bookkeeping is snapshotted, NOT across the slab dump -- the async query paths take that lock on the event loop and
        would freeze it (and /health) for the seconds a dump takes.  Queries keep being served from the event loop; those that
        reach the library wait in their worker threads while mv_index_save holds the index (ExclusiveLock)."""
        import os
        import shutil
        import uuid

        os.makedirs(directory, exist_ok=True)
        with self._write_gate:
            with self._lock:
                ix = self._require_index()
                book = self._book()
            gen = "gen-" + uuid.uuid4().hex
            gdir = os.path.join(directory, gen)
            os.makedirs(gdir)
            _fsync_dir(directory)  # the new generation's directory ENTRY
            book["checkpoint"] = gen
            cur = os.path.join(directory, "CURRENT")
            try:
                ix.save(os.path.join(gdir, "index.mv"))  # every shard file: temp file + fsync + rename each
                with open(os.path.join(gdir, "store.json"), "w") as f:
                    json.dump(book, f)
                    f.flush()
                    os.fsync(f.fileno())
                _fsync_dir(gdir)
                tmp = os.path.join(directory, "CURRENT.tmp")
                with open(tmp, "w") as f:
                    f.write(gen)
                    f.flush()
                    os.fsync(f.fileno())
                os.replace(tmp, cur)  # the switch: one atomic rename ...
                _fsync_dir(directory)  # ... made durable BEFORE anything older is removed (a power loss must not keep the deletions and lose the rename)
            except BaseException:
                shutil.rmtree(gdir, ignore_errors=True)  # an incomplete generation is never named by CURRENT
                raise
            for name in os.listdir(directory):  # older generations (the rename that retired them is durable), and the flat files of the pre-generation layout
                pth = os.path.join(directory, name)
                if name.startswith("gen-") and name != gen:
                    shutil.rmtree(pth, ignore_errors=True)
                elif name in ("store.json", "index.id") or name.startswith("index.mv"):
                    try:
                        os.remove(pth)
                    except OSError:
                        pass

    @classmethod
    def _load_index(cls, self, directory: str, book: Dict[str, Any], device: int):
        import os

        if self._index_factory is not None and hasattr(self._index_factory, "load"):  # CPU tests: the injected index class
            return self._index_factory.load(os.path.join(directory, "index.mv"), device=device)
        from .index import MvIndex

        return MvIndex.load(os.path.join(directory, "index.mv"), device=device)

    @classmethod
    def load(cls, directory: str, device: int = 0, storage: Any = None, **kw: Any) -> "MI355XMultiVectorStore":
        import os

        directory = cls.checkpoint_path(directory)  # the generation CURRENT names (or the flat pre-generation layout)
        with open(os.path.join(directory, "store.json")) as f:
            book = json.load(f)
        idp = os.path.join(directory, "index.id")
        if str(book.get("checkpoint", "")).startswith("gen-"):
            if os.path.basename(os.path.normpath(directory)) != book["checkpoint"]:
                raise RuntimeError(f"{directory}: store.json belongs to generation {book['checkpoint']} (directory moved or mixed up?)")
        elif book.get("checkpoint") and (not os.path.exists(idp) or open(idp).read().strip() != book["checkpoint"]):
            raise RuntimeError(f"{directory}: index.mv and store.json belong to different checkpoints (crash during save?)")
        if bool(book.get("fde_external")) != (kw.get("fde_module") is not None):
            # the FDE slab holds one encoder's document vectors: queries must come from the same one
            raise RuntimeError(f"{directory}: this checkpoint " + ("holds document FDE vectors of an external encoder: pass fde_module= to load()"
                                                                     if book.get("fde_external") else "was encoded by the library itself: load it without fde_module"))
        self = cls(capacity_pages=book["capacity_pages"], stride_rows=book["stride_rows"], device=device, mode=book["mode"], storage=storage,
                   id_base=book["id_base"], fde_coarse_n=book.get("fde_coarse_n", 0), exact_tier=book.get("exact_tier", "hbm"), rerank_n=book.get("rerank_n", 0), prune_slab=book.get("prune_slab", True), **{"fp32_pages": book.get("fp32_pages", False), "fp32_scan": book.get("fp32_scan", "both_halves"), "fde_e4m3": book.get("fde_e4m3", False), "fde_fp4": book.get("fde_fp4", False), "packed_layout": book.get("packed_layout", False), "capacity_rows": book.get("capacity_rows", 0), **kw})
        self._index = cls._load_index(self, directory, book, device)
        self._apply_options(self._index)
        for p, doc, chunk_no, content, meta_json, app, *rest in book["rows"]:
            # rows written before round 5 carry no origin: they keep what those builds did (a key-shaped content is this store's key)
            self._rows[int(p)] = (doc, int(chunk_no), content, meta_json, app, int(rest[0])) if rest else (doc, int(chunk_no), content, meta_json, app)
            self._page_of[(self._nk(doc, app), int(chunk_no))] = int(p)
            self._doc_pages.setdefault(self._nk(doc, app), []).append(int(p))
        self._doc_app = {int(k): v for k, v in book["doc_app"].items()}
        self._doc_ord = {k: int(v) for k, v in book["doc_ord"].items()}
        if int(book.get("version", 2)) < 3 and self._filter_by_app:  # checkpoints before the per-app keys: one ordinal per document_id
            self._doc_ord = {self._nk(d, self._doc_app.get(o)): o for d, o in self._doc_ord.items()}
        self._next_ord = int(book.get("next_ord", max(self._doc_ord.values(), default=-1) + 1))
        return self

    # ------------------------------------------------------------------ bench / test helper
    def adopt_synthetic_corpus(self, seed: int, n_pages: int, n_rows: Optional[int] = None, pages_per_doc: int = 1,
                               app_id: Optional[str] = None) -> None:
        """Fill an EMPTY store with n_pages pages of the device generator (mv_index_fill_synthetic: no embeddings cross
        PCIe) and create the matching bookkeeping: document "synth-<d>" owns pages [d * pages_per_doc, ...), chunk j of
        it is its j-th page.  Serving benchmarks (tools/serve_bench.py) and tests build 10^5..10^6-page stores this way."""
        ix = self._require_index()
        with self._lock:
            if self._rows or self._next_ord or len(ix):
                raise RuntimeError("adopt_synthetic_corpus needs an empty store")
            ix.fill_synthetic(seed, 0, int(n_pages), n_rows=n_rows, pages_per_doc=int(pages_per_doc))
            app = app_id if app_id is not None else (DEFAULT_APP_ID if self._filter_by_app else None)
            for p in range(int(n_pages)):
                d, j = divmod(p, int(pages_per_doc))
                doc = f"synth-{d}"
                key = self._nk(doc, app)
                page = p + (0 if self._global_ids else self.id_base)
                self._rows[page] = (doc, j, f"page {p}", "{}", app)
                self._page_of[(key, j)] = page
                if j == 0:
                    self._doc_ord[key] = d
                    self._doc_app[d] = app
                    self._doc_pages[key] = []
                self._doc_pages[key].append(page)
            self._next_ord = (int(n_pages) + int(pages_per_doc) - 1) // int(pages_per_doc)
            self._ord_stamp += 1

    # ------------------------------------------------------------------ introspection
    def __len__(self) -> int:
        return len(self._rows)


class MI355XFastMultiVectorStore(MI355XMultiVectorStore):
    """Drop-in for FastMultiVectorStore: FDE coarse stage + exact float rerank, per-app namespaces
    (fast_multivector_store.py:504-607; `self.ns(app_id)` :526).  A query only ever sees the pages of ONE app:
    the given app_id, else "default"."""

    default_mode = "fde_then_float"
    _filter_by_app = True

    def __init__(self, capacity_pages: int = 250_000, **kw: Any):
        # sizing rule: fde_then_float keeps the bf16 slab (stride_rows * 256 B / page) AND the FDE slab (20 480 B / page):
        # 250 k pages of 1040 rows = 66.6 + 5.1 GB.  1 M such pages (266 + 20 GB) do not fit one MI355X; use
        # MI355XShardedFastMultiVectorStore over several GPUs, or mode="float_fp8" slabs, for corpora of that size.
        super().__init__(capacity_pages=capacity_pages, **kw)


class _ShardedMixin:
    """R shards behind one store object (SURVEY.md 8e / VERDICT r1 item 4).  `devices` names the GPU of every shard
    (default: every visible GPU once); repeating a device gives logical shards."""

    _global_ids = True

    def _init_sharding(self, devices: Optional[Sequence[int]], transport: str) -> None:
        if devices is None:
            from . import _lib

            devices = list(range(max(int(_lib.lib().mv_device_count()), 1)))
        self.devices = [int(d) for d in devices]
        self.transport = transport
        self.device = self.devices[0]

    def _devices(self) -> List[int]:
        return list(self.devices)

    def _make_index(self):
        from .shard_index import ShardedIndex

        kw = {}
        if self._index_factory is not None:  # CPU tests: oracle-backed shards + host merge
            kw = dict(index_cls=self._index_factory, comm_cls=self._comm_factory)
        ix = ShardedIndex(capacity_pages=self.capacity_pages, stride_rows=self.stride_rows, devices=self.devices, transport=self.transport,
                          id_base=self.id_base, **self._slab_flags(), **kw)
        self._apply_options(ix)
        return ix

    @classmethod
    def _load_index(cls, self, directory: str, book: Dict[str, Any], device: int):
        import os

        from .shard_index import ShardedIndex

        n_saved = int(book.get("n_shards", len(self.devices)))
        if n_saved != len(self.devices):
            # every shard file holds pages the bookkeeping refers to: opening fewer would silently lose them from queries
            raise RuntimeError(f"{directory} holds {n_saved} shards but this store was given {len(self.devices)} devices "
                               f"({self.devices}); pass devices= with {n_saved} entries (a device may repeat: logical shards)")
        return ShardedIndex.load(os.path.join(directory, "index.mv"), devices=self.devices, transport=self.transport)

    def _book(self) -> Dict[str, Any]:
        b = super()._book()
        b["n_shards"] = len(self.devices)
        return b


class MI355XShardedMultiVectorStore(_ShardedMixin, MI355XMultiVectorStore):
    def __init__(self, devices: Optional[Sequence[int]] = None, transport: str = "auto", comm_factory: Any = None, **kw: Any):
        super().__init__(**kw)
        self._comm_factory = comm_factory
        self._init_sharding(devices, transport)


class MI355XShardedFastMultiVectorStore(_ShardedMixin, MI355XFastMultiVectorStore):
    def __init__(self, devices: Optional[Sequence[int]] = None, transport: str = "auto", comm_factory: Any = None,
                 capacity_pages: int = 1_000_000, **kw: Any):
        super().__init__(capacity_pages=capacity_pages, **kw)
        self._comm_factory = comm_factory
        self._init_sharding(devices, transport)


def create_store(provider: str, **kw: Any) -> MI355XMultiVectorStore:
    """Factory for core/services_init.py: [multivector_store] provider =
    "mi355x" | "mi355x_fast" | "mi355x_fast_host_exact" | "mi355x_fast_split_exact" | "mi355x_fast_split_exact_lean" | "mi355x_float" | "mi355x_fp8_exact" | "mi355x_fp8_split_exact" | "mi355x_sharded" |
    "mi355x_sharded_fast" | "mi355x_sharded_fast_host_exact" | "mi355x_sharded_fast_split_exact" | "mi355x_sharded_fast_split_exact_lean" | "mi355x_sharded_float" |
    "mi355x_sharded_fp8_exact" | "mi355x_sharded_fp8_split_exact" | "mi355x_remote"."""
    if provider == "mi355x":
        return MI355XMultiVectorStore(**kw)
    if provider == "mi355x_fast":
        return MI355XFastMultiVectorStore(**kw)
    if provider == "mi355x_fast_host_exact":  # FDE + e4m3 slabs in HBM, exact bf16 rows in pinned host RAM (configs[3] shard shape)
        return MI355XFastMultiVectorStore(exact_tier="host", **kw)
    if provider == "mi355x_fast_split_exact":  # the same, with the exact rows of the leading pages in the HBM the slabs leave free (1.25 M pages / GPU)
        return MI355XFastMultiVectorStore(exact_tier="split", **kw)
    if provider == "mi355x_fast_split_exact_lean":  # ... without the e4m3 pruning slab: FDE slab + exact rows only, ~80 % of them in HBM at 1.25 M pages / GPU
        return MI355XFastMultiVectorStore(exact_tier="split", prune_slab=False, **kw)
    if provider == "mi355x_float":
        return MI355XMultiVectorStore(mode="float", **kw)
    if provider == "mi355x_fp8_exact":  # e4m3 slab in HBM + exact bf16 tier in pinned host RAM
        return MI355XMultiVectorStore(mode="fp8_then_float", **kw)
    if provider == "mi355x_fp8_split_exact":  # ... with the exact rows of the leading pages in the HBM the e4m3 slab leaves free
        return MI355XMultiVectorStore(mode="fp8_then_float", exact_tier="split", **kw)
    if provider == "mi355x_sharded":
        return MI355XShardedMultiVectorStore(**kw)
    if provider == "mi355x_sharded_fast":
        return MI355XShardedFastMultiVectorStore(**kw)
    if provider == "mi355x_sharded_fast_host_exact":  # configs[3]: 10 M pages over 8 GPUs, every shard with its pinned-host exact tier
        return MI355XShardedFastMultiVectorStore(exact_tier="host", **kw)
    if provider == "mi355x_sharded_fast_split_exact":  # configs[3] at full size: every shard splits its exact tier between its free HBM and pinned host memory
        return MI355XShardedFastMultiVectorStore(exact_tier="split", **kw)
    if provider == "mi355x_sharded_fast_split_exact_lean":  # ... and no e4m3 pruning slab: most of every shard's exact rows stay in its HBM
        return MI355XShardedFastMultiVectorStore(exact_tier="split", prune_slab=False, **kw)
    if provider == "mi355x_sharded_float":
        return MI355XShardedMultiVectorStore(mode="float", **kw)
    if provider == "mi355x_sharded_fp8_exact":  # configs[4]: e4m3 scan of every shard -> GLOBAL top-n -> exact re-score from the owners' host tiers
        return MI355XShardedMultiVectorStore(mode="fp8_then_float", **kw)
    if provider == "mi355x_sharded_fp8_split_exact":  # ... every shard with a split exact tier
        return MI355XShardedMultiVectorStore(mode="fp8_then_float", exact_tier="split", **kw)
    if provider == "mi355x_remote":  # every process but the one that owns the HBM slab (store_server.py)
        from .store_server import MI355XRemoteMultiVectorStore

        return MI355XRemoteMultiVectorStore(**kw)
    raise ValueError(f"unknown MI355X multivector provider {provider!r}")
