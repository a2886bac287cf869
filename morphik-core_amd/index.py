"""MvIndex -- Python face of one mv_index (one GPU's shard of the page corpus).

Thin: argument normalisation (numpy / torch / list -> contiguous host buffers) and ctypes calls.
All scoring happens in libmvmaxsim.so on the MI355X; nothing here computes a score.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (
    MV_BF16,
    MV_F32,
    MV_MODE_BINARY,
    MV_MODE_FDE_ONLY,
    MV_MODE_FDE_THEN_FLOAT,
    MV_MODE_FLOAT,
    MV_MODE_FLOAT_FP8,
    MV_MODE_FP8_THEN_FLOAT,
    MV_WITH_BINARY,
    MV_WITH_FDE,
    MV_WITH_FLOAT,
    MV_WITH_FP8,
    MV_WITH_HOST_EXACT,
    MV_WITH_EXACT_SPLIT,
    MV_WITH_FLOAT_LO,
    MV_LAYOUT_PACKED,
    MV_WITH_FDE_E4M3,
    MV_WITH_FDE_FP4,
    ConfigC,
    FdeConfigC,
    QueryStatsC,
    check,
    lib,
)

MODES = {"float": MV_MODE_FLOAT, "binary": MV_MODE_BINARY, "fde_then_float": MV_MODE_FDE_THEN_FLOAT, "fde": MV_MODE_FDE_ONLY,
         "float_fp8": MV_MODE_FLOAT_FP8, "fp8_then_float": MV_MODE_FP8_THEN_FLOAT}


@dataclass
class FdeConfig:
    """fixed_dimensional_encoding.FixedDimensionalEncodingConfig as built at
    core/vector_store/fast_multivector_store.py:325-331."""

    dimension: int = 128
    num_repetitions: int = 20
    num_simhash_projections: int = 5
    projection_dimension: int = 16
    seed: int = 1

    def to_c(self) -> FdeConfigC:
        return FdeConfigC(self.dimension, self.num_repetitions, self.num_simhash_projections, self.projection_dimension, self.seed)

    @property
    def output_dim(self) -> int:
        return self.num_repetitions * (1 << self.num_simhash_projections) * self.projection_dimension


@dataclass
class QueryStats:
    score_kernel_ms: float = 0.0
    topk_ms: float = 0.0
    total_device_ms: float = 0.0
    score_launches: int = 0
    pages_scored: int = 0
    bytes_scanned: int = 0
    encode_ms: float = 0.0   # FDE modes: query encode / coarse scan / candidate selection / exact rerank
    coarse_ms: float = 0.0
    select_ms: float = 0.0
    rerank_ms: float = 0.0

    @classmethod
    def from_c(cls, s: QueryStatsC) -> "QueryStats":
        return cls(s.score_kernel_ms, s.topk_ms, s.total_device_ms, s.score_launches, s.pages_scored, s.bytes_scanned,
                   s.encode_ms, s.coarse_ms, s.select_ms, s.rerank_ms)


def _to_host(x: Any) -> np.ndarray:
    """ndarray / torch.Tensor (any device, incl. bfloat16) / list -> numpy array on the host."""
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "detach") and hasattr(x, "cpu"):  # torch.Tensor without importing torch here
        t = x.detach()
        if str(t.dtype) == "torch.bfloat16":
            import torch

            return t.cpu().contiguous().view(torch.int16).numpy().view(np.uint16)
        return t.cpu().numpy()
    if isinstance(x, (list, tuple)) and len(x) and hasattr(x[0], "detach"):
        return np.stack([_to_host(t) for t in x])
    return np.asarray(x)


def as_rows(x: Any) -> Tuple[np.ndarray, int]:
    """-> (C-contiguous [n,128] array, dtype code).  uint16 arrays are taken as bf16 bit patterns."""
    a = _to_host(x)
    if a.dtype == np.uint16:
        a = np.ascontiguousarray(a)
        code = MV_BF16
    else:
        a = np.ascontiguousarray(a, dtype=np.float32)
        code = MV_F32
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != 128:
        raise ValueError(f"expected rows of width 128, got shape {a.shape}")
    return a, code


def allow_bitmap(allowed_ordinals: Optional[Iterable[int]], n_docs_hint: int = 0) -> Optional[np.ndarray]:
    """Bitmap over dense document ordinals for the doc_ids filter (None = no filter)."""
    if allowed_ordinals is None:
        return None
    ords = np.fromiter((int(o) for o in allowed_ordinals), dtype=np.int64)
    n = int(max(ords.max() + 1 if ords.size else 0, n_docs_hint, 1))
    bits = np.zeros((n + 31) // 32, np.uint32)
    if ords.size:
        np.bitwise_or.at(bits, ords >> 5, (np.uint32(1) << (ords & 31).astype(np.uint32)))
    return bits


def _fde_block(q_fde: Any, n_queries: int, out_dim: int) -> np.ndarray:
    """Caller-supplied query FDE vector(s) -> contiguous float32 [n_queries][out_dim].  The *_fde entry points of the C ABI take a
    bare pointer and read out_dim floats per query: a vector of another width (an fde_module configured differently from the index)
    must stop HERE, not become a host over-read."""
    a = np.ascontiguousarray(np.asarray(q_fde, dtype=np.float32))
    if a.size != int(n_queries) * int(out_dim):
        raise ValueError(f"query FDE block holds {a.size} floats; this index expects {n_queries} x {out_dim} "
                         "(is the fde_module configured like the index's FDE?)")
    return a.reshape(int(n_queries), int(out_dim))


def _stack_queries(queries: Sequence[Any], rows=None):
    """Queries of possibly different lengths -> one [n, longest, 128] block (zero rows behind the shorter ones: a zero row
    contributes exactly 0), its dtype code and the padded length."""
    rows = rows if rows is not None else [as_rows(q) for q in queries]
    nmax = max(a.shape[0] for a, _ in rows)
    code = MV_BF16 if all(c == MV_BF16 for _, c in rows) else MV_F32
    blk = np.zeros((len(rows), nmax, 128), np.uint16 if code == MV_BF16 else np.float32)
    for i, (a, c) in enumerate(rows):
        blk[i, : a.shape[0]] = a if c == code else (a.astype(np.uint32) << 16).view(np.float32)
    return blk, code, nmax


def _allow_block(n_queries: int, allow, allows, n_docs: int):
    """-> (bitmap block or None, words per bitmap, per_query flag): one doc bitmap for all queries, or one per query
    (None = everything: an all-ones bitmap covering n_docs ordinals)."""
    if allows is not None and any(a is not None for a in allows):
        if len(allows) != n_queries:
            raise ValueError("allows must have one entry per query")
        n_words = max(max(int(np.size(a)) for a in allows if a is not None), (int(n_docs) + 31) // 32, 1)
        ab = np.zeros((n_queries, n_words), np.uint32)
        for i, a in enumerate(allows):
            if a is None:
                ab[i] = 0xFFFFFFFF  # no filter for this query
            else:
                ab[i, : np.size(a)] = np.asarray(a, dtype=np.uint32)
        return ab, n_words, 1
    ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
    return ab, (0 if ab is None else ab.size), 0


class MvIndex:
    def __init__(
        self,
        capacity_pages: int,
        stride_rows: int = 1024,
        device: int = 0,
        with_float: bool = True,
        with_binary: bool = False,
        with_fde: bool = False,
        fde: Optional[FdeConfig] = None,
        id_base: int = 0,
        with_fp8: bool = False,
        with_host_exact: bool = False,
        with_exact_split: bool = False,
        with_float_lo: bool = False,
        packed: bool = False,
        capacity_rows: int = 0,
        with_fde_e4m3: bool = False,
        with_fde_fp4: bool = False,
    ):
        """with_host_exact: keep the exact bf16 rows in PINNED HOST memory (no HBM) as the exact tier of mode
        "fp8_then_float" -- for shards whose bf16 slab does not fit beside the fp8 slab.
        with_exact_split (with with_host_exact, without with_float): the exact rows of the first pages fill the HBM the other
        slabs leave free, only the rest is pinned (exact_hbm_pages tells the split).
        with_float_lo (with with_float): a second bf16 slab holding lo = bf16(x - bf16(x)) -- pages added as float32 keep 16
        significant bits (x = hi + lo to 2^-18), the candidate scorers and the float scan multiply both halves on the bf16 MFMA
        and return the reference's fp32 scores (fast_multivector_store.py:553-555 on its fp32 `.npy` pages) to ~1e-6.
        packed (MV_LAYOUT_PACKED): ragged pages lie back to back in whole 16-row tiles instead of one stride_rows slot each -- a
        ColQwen2.5-like corpus (dynamic token counts, colpali_embedding_model.py:47-52) then takes the HBM of its valid rows, not of
        its longest page; capacity_rows = rows the row-indexed slabs hold in all (0 = capacity_pages * stride_rows); stride_rows stays
        the longest page accepted.  Same scores as the fixed layout, bit for bit.
        with_fde_e4m3 (with with_fde): an e4m3 copy of the FDE slab (out_dim bytes per page) that the COARSE stage of the FDE modes reads
        instead of the bf16 slab -- half the bytes of the pass that dominates every request; set_option(MV_OPT_FDE_COARSE_SLAB, 0) goes back.
        with_fde_fp4 (with with_fde; beside or instead of with_fde_e4m3): a copy in FP4 (e2m1, one power-of-two scale per row): out_dim / 2 bytes per
        page, read by the coarse stage (single requests: a conversion scan with the fp32 query; batches: both MFMA operands FP4, the queries as two
        e2m1 terms); MV_OPT_FDE_COARSE_SLAB 2 / 0."""
        self.fde_config = fde or FdeConfig()
        flags = ((MV_WITH_FLOAT if with_float else 0) | (MV_WITH_BINARY if with_binary else 0) | (MV_WITH_FDE if with_fde else 0)
                 | (MV_WITH_FP8 if with_fp8 else 0) | (MV_WITH_HOST_EXACT if with_host_exact else 0) | (MV_WITH_EXACT_SPLIT if with_exact_split else 0)
                 | (MV_WITH_FLOAT_LO if with_float_lo else 0) | (MV_LAYOUT_PACKED if packed else 0) | (MV_WITH_FDE_E4M3 if with_fde_e4m3 else 0) | (MV_WITH_FDE_FP4 if with_fde_fp4 else 0))
        cfg = ConfigC(128, int(stride_rows), int(capacity_pages), int(device), flags, int(id_base), self.fde_config.to_c(), int(capacity_rows) if packed else 0)
        h = C.c_void_p()
        check(lib().mv_index_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.stride_rows = int(stride_rows)
        self.device = int(device)
        self.id_base = int(id_base)
        self.flags = flags

    @property
    def exact_hbm_pages(self) -> int:
        """Pages of a split exact tier (with_exact_split) whose exact rows live in HBM; 0 without a split."""
        return int(lib().mv_index_exact_hbm_pages(self._h))

    def rebalance_exact_tier(self, max_moves: int = 0) -> int:
        """Split exact tier: swap the most-read host-resident pages with the least-read HBM-resident ones (reads counted by every
        rerank since the last call). -> pages moved into HBM.  Answers are unchanged; the PCIe share of the reranks falls."""
        moved = C.c_int64()
        check(lib().mv_index_exact_tier_rebalance(self._h, int(max_moves), C.byref(moved)))
        return int(moved.value)

    def exact_tier_hits(self) -> Tuple[int, int]:
        """-> (reads served from the HBM part, reads served from pinned host memory) since the last rebalance."""
        a, b = C.c_int64(), C.c_int64()
        check(lib().mv_index_exact_tier_hits(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def fde_placement_trial(self, trials: int = 3) -> Tuple[float, float, int]:
        """Try up to `trials` other allocations for the FDE slab and keep the one the batched coarse pass reads fastest (mvmaxsim.h:
        mv_index_fde_placement_trial; peak memory three slabs).  -> (pass ms before, pass ms after, times a candidate won)."""
        a, b, m = C.c_double(), C.c_double(), C.c_int32()
        check(lib().mv_index_fde_placement_trial(self._h, int(trials), C.byref(a), C.byref(b), C.byref(m)))
        return float(a.value), float(b.value), int(m.value)

    # -- lifecycle
    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().mv_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(lib().mv_index_size(self._h))

    @property
    def capacity(self) -> int:
        return int(lib().mv_index_capacity(self._h))

    @property
    def capacity_rows(self) -> int:
        """Rows the row-indexed slabs hold (packed layout: the capacity_rows it was created with; fixed: capacity * stride_rows)."""
        return int(lib().mv_index_capacity_rows(self._h))

    @property
    def rows_used(self) -> int:
        """Slab rows taken by the pages appended so far (packed: whole 16-row tiles per page; fixed: pages * stride_rows)."""
        return int(lib().mv_index_rows_used(self._h))

    def set_option(self, option: int, value: int) -> None:
        check(lib().mv_index_set_option(self._h, option, int(value)))
        self.__dict__.setdefault("_opts", {})[int(option)] = int(value)  # what was last set through this object

    def get_option(self, option: int, default: int = 0) -> int:
        """The value last set through set_option() on this object (the library has no getter), else `default`."""
        return self.__dict__.get("_opts", {}).get(int(option), default)

    # -- build
    def add(self, pages: Sequence[Any], doc_ordinals: Optional[Sequence[int]] = None) -> int:
        """Append pages (each [n_rows_i,128]; fp32 or bf16-as-uint16). Returns the first local page id."""
        if len(pages) == 0:
            return len(self)
        arrs, codes = zip(*(as_rows(p) if np.size(p) else (np.zeros((0, 128), np.float32), MV_F32) for p in pages))
        code = MV_BF16 if all(c == MV_BF16 for c in codes) else MV_F32
        if code == MV_F32 and any(c == MV_BF16 for c in codes):
            arrs = [(a.astype(np.uint32) << 16).view(np.float32) if c == MV_BF16 else a for a, c in zip(arrs, codes)]
        n_rows = np.array([a.shape[0] for a in arrs], np.int32)
        flat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if n_rows.sum() else np.zeros((1, 128), arrs[0].dtype)
        ords = None if doc_ordinals is None else np.ascontiguousarray(doc_ordinals, dtype=np.int32)
        first = C.c_int64()
        check(
            lib().mv_index_add(
                self._h, flat.ctypes.data, code, n_rows.ctypes.data, len(arrs), None if ords is None else ords.ctypes.data, C.byref(first)
            )
        )
        return int(first.value)

    def add_bits(self, pages_bits: Sequence[np.ndarray], doc_ordinals: Optional[Sequence[int]] = None) -> int:
        """Append pages given as packed sign rows [(P_i,16) uint8] (import of BIT(128)[] rows; binary-only index)."""
        arrs = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1, 16) for p in pages_bits]
        n_rows = np.array([a.shape[0] for a in arrs], np.int32)
        flat = np.ascontiguousarray(np.concatenate(arrs, 0)) if n_rows.sum() else np.zeros((1, 16), np.uint8)
        ords = None if doc_ordinals is None else np.ascontiguousarray(doc_ordinals, dtype=np.int32)
        first = C.c_int64()
        check(lib().mv_index_add_bits(self._h, flat.ctypes.data, n_rows.ctypes.data, len(arrs), None if ords is None else ords.ctypes.data, C.byref(first)))
        return int(first.value)

    def add_device(self, d_ptr: int, dtype_code: int, n_rows: Sequence[int], doc_ordinals: Optional[Sequence[int]] = None) -> int:
        """Append pages whose rows already sit in device memory (encoder output): d_ptr = device address
        of sum(n_rows) x 128 rows."""
        nr = np.ascontiguousarray(n_rows, dtype=np.int32)
        ords = None if doc_ordinals is None else np.ascontiguousarray(doc_ordinals, dtype=np.int32)
        first = C.c_int64()
        check(lib().mv_index_add_device(self._h, C.c_void_p(d_ptr), dtype_code, nr.ctypes.data, len(nr), None if ords is None else ords.ctypes.data, C.byref(first)))
        return int(first.value)

    def fill_synthetic(self, seed: int, first_unit: int, n_pages: int, n_rows: Optional[int] = None, pages_per_doc: int = 1) -> None:
        check(lib().mv_index_fill_synthetic(self._h, seed, first_unit, n_pages, self.stride_rows if n_rows is None else n_rows, pages_per_doc))

    def fill_synthetic_ragged(self, seed: int, first_unit: int, n_pages: int, min_rows: int, max_rows: int, pages_per_doc: int = 1) -> None:
        """Pages of the device generator with a different row count each (synth_ragged_rows(seed, unit, min_rows, max_rows)): the shape of
        a ColQwen2.5 corpus.  Both layouts."""
        check(lib().mv_index_fill_synthetic_ragged(self._h, seed, first_unit, n_pages, int(min_rows), int(max_rows), pages_per_doc))

    def write_rows(self, page: int, row0: int, rows_bf16: np.ndarray) -> None:
        r = np.ascontiguousarray(rows_bf16, dtype=np.uint16)
        check(lib().mv_index_write_rows(self._h, page, row0, r.shape[0], r.ctypes.data))

    def replace_page(self, page: int, rows_bf16: np.ndarray) -> None:
        """Overwrite one whole page from host bf16 rows and refresh every enabled slab."""
        r = np.ascontiguousarray(rows_bf16, dtype=np.uint16).reshape(-1, 128)
        check(lib().mv_index_replace_page(self._h, page, r.ctypes.data, r.shape[0]))

    def import_fde(self, page0: int, fde: Any, n_pages: Optional[int] = None) -> None:
        """Replace the FDE vectors the library encoded for pages [page0, page0 + len(fde)) with the caller's own document encodings
        (float32 [n][fde_config.output_dim]; e.g. `fde.generate_document_encoding` of the reference, or vectors exported from a
        TurboPuffer namespace).  Query such an index with q_fde= vectors from the SAME encoder."""
        a = np.ascontiguousarray(np.asarray(fde, dtype=np.float32))
        od = self.fde_config.output_dim
        if a.ndim == 2 and a.shape[1] != od or a.ndim == 1 and a.size != od or a.ndim > 2 or a.size % od:
            # never regroup vectors of another width into "some number of pages"
            raise ValueError(f"document FDE block of shape {a.shape}; this index expects [n_pages][{od}]")
        a = a.reshape(-1, od)
        if n_pages is not None and a.shape[0] != int(n_pages):
            raise ValueError(f"{a.shape[0]} document FDE vectors for {int(n_pages)} pages")
        check(lib().mv_index_import_fde(self._h, int(page0), a.shape[0], a.ctypes.data))

    def read_fde(self, page0: int, n_pages: int) -> np.ndarray:
        """-> float32 [n_pages][output_dim]: the document FDE vectors the scan reads (bf16 values), own or imported."""
        out = np.empty((int(n_pages), self.fde_config.output_dim), np.float32)
        check(lib().mv_index_read_fde(self._h, int(page0), int(n_pages), out.ctypes.data))
        return out

    def read_fde_e4m3(self, page0: int, n_pages: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (e4m3 codes [n_pages][output_dim] uint8, scales [n_pages] float32) of the FDE slab's e4m3 copy: value = decode(code) * scale."""
        codes = np.empty((int(n_pages), self.fde_config.output_dim), np.uint8)
        sc = np.empty(int(n_pages), np.float32)
        check(lib().mv_index_read_fde_e4m3(self._h, int(page0), int(n_pages), codes.ctypes.data, sc.ctypes.data))
        return codes, sc

    def read_fde_fp4(self, page0: int, n_pages: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (e2m1 code bytes [n_pages][output_dim / 2] uint8 -- element 2i in the low nibble --, scales [n_pages] float32) of the FDE slab's fp4 copy."""
        codes = np.empty((int(n_pages), self.fde_config.output_dim // 2), np.uint8)
        sc = np.empty(int(n_pages), np.float32)
        check(lib().mv_index_read_fde_fp4(self._h, int(page0), int(n_pages), codes.ctypes.data, sc.ctypes.data))
        return codes, sc

    def read_fp8(self, page0: int, n_pages: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (e4m3 codes [n, stride_rows, 128] uint8, per-page inverse scales [n] float32)."""
        codes = np.empty((n_pages, self.stride_rows, 128), np.uint8)
        inv = np.empty(n_pages, np.float32)
        check(lib().mv_index_read_fp8(self._h, page0, n_pages, codes.ctypes.data, inv.ctypes.data))
        return codes, inv

    def read_pages(self, page0: int, n_pages: int) -> np.ndarray:
        out = np.empty((n_pages, self.stride_rows, 128), np.uint16)
        check(lib().mv_index_read_pages(self._h, page0, n_pages, out.ctypes.data))
        return out

    def read_pages_f32(self, page0: int, n_pages: int) -> np.ndarray:
        """-> float32 [n_pages, stride_rows, 128]: hi + lo of an index with with_float_lo (the fp32 input to 2^-18), else the bf16 rows widened."""
        out = np.empty((n_pages, self.stride_rows, 128), np.float32)
        check(lib().mv_index_read_pages_f32(self._h, page0, n_pages, out.ctypes.data))
        return out

    def remove_doc(self, doc_ordinal: int) -> int:
        n = C.c_int64()
        check(lib().mv_index_remove_doc(self._h, doc_ordinal, C.byref(n)))
        return int(n.value)

    def remove_page(self, page: int) -> None:
        check(lib().mv_index_remove_page(self._h, page))

    def compact(self) -> np.ndarray:
        """Reclaim tombstoned slots (live pages move down in order). -> old_to_new int64 [old size], -1 = removed.
        Page ids change: remap whatever was keyed by them (MI355XMultiVectorStore.compact does)."""
        old = len(self)
        o2n = np.empty(max(old, 1), np.int64)
        new_size = C.c_int64()
        check(lib().mv_index_compact(self._h, o2n.ctypes.data, C.byref(new_size)))
        return o2n[:old]

    # -- query
    def query(self, q: Any, k: int, mode: str = "float", allow: Optional[np.ndarray] = None, want_stats: bool = False, q_fde: Any = None):
        """-> (scores[n] float32, ids[n] int64 global page ids[, QueryStats]); n <= k.
        q_fde (FDE modes): the caller's own FDE of this query (float32 [fde_config.output_dim], e.g. from the reference's `fde` extension)
        instead of the encoding of the query rows on the device; the rerank still scores the rows."""
        qa, code = as_rows(q)
        k = int(k)
        scores = np.empty(max(k, 1), np.float32)
        ids = np.empty(max(k, 1), np.int64)
        n = C.c_int32()
        st = QueryStatsC()
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        if q_fde is not None:
            qf = _fde_block(q_fde, 1, self.fde_config.output_dim)
            check(lib().mv_query_topk_fde(self._h, qa.ctypes.data, code, qa.shape[0], qf.ctypes.data, k, MODES[mode], None if ab is None else ab.ctypes.data,
                                          0 if ab is None else ab.size, scores.ctypes.data, ids.ctypes.data, C.byref(n), C.byref(st) if want_stats else None))
        else:
            check(lib().mv_query_topk(self._h, qa.ctypes.data, code, qa.shape[0], k, MODES[mode], None if ab is None else ab.ctypes.data,
                                      0 if ab is None else ab.size, scores.ctypes.data, ids.ctypes.data, C.byref(n), C.byref(st) if want_stats else None))
        res = (scores[: n.value].copy(), ids[: n.value].copy())
        return res + (QueryStats.from_c(st),) if want_stats else res

    def query_batch(self, queries: Sequence[Any], k: int, mode: str = "float", allow: Optional[np.ndarray] = None,
                    want_stats: bool = False, allows: Optional[Sequence[Optional[np.ndarray]]] = None, n_docs: int = 0, q_fdes: Any = None):
        """Top-k of several queries in one slab pass.  -> list of (scores, ids) per query [, QueryStats].
        q_fdes (FDE modes): the caller's own FDE vector of every query (float32 [len(queries)][output_dim]), see query().
        mode "float": the batched MFMA MaxSim scan (<= 512 query rows per pass); "fde_then_float" / "fde": the batched FDE
        pipeline (one pass over the FDE slab per 32 queries, every query's candidates reranked exactly, same results as
        query()); other modes are served query by query inside the library.
        Queries may have different lengths: they are zero-padded to the longest (a zero row contributes 0).
        `allow` = one doc bitmap for all queries; `allows` = one bitmap (or None = everything) PER query
        (pass n_docs = number of document ordinals in use so an unfiltered query's all-ones bitmap covers them)."""
        rows = [as_rows(q) for q in queries]
        nmax = max(a.shape[0] for a, _ in rows)
        if mode == "binary" and any(a.shape[0] != nmax for a, _ in rows):
            # SQL max_sim scores a row of zero bits like any other row, so zero padding would change the answer: queries of
            # different lengths go one by one (the library serves this mode query by query anyway)
            out = []
            for j, q in enumerate(queries):
                a = allow if allows is None else allows[j]
                out.append(self.query(q, k, mode=mode, allow=None if a is None else np.ascontiguousarray(a, dtype=np.uint32)))
            return (out, QueryStats.from_c(QueryStatsC())) if want_stats else out
        blk, code, nmax = _stack_queries(queries, rows)
        k = int(k)
        scores = np.empty((len(rows), max(k, 1)), np.float32)
        ids = np.empty((len(rows), max(k, 1)), np.int64)
        n = np.zeros(len(rows), np.int32)
        st = QueryStatsC()
        ab, n_words, per_query = _allow_block(len(rows), allow, allows, n_docs)
        if q_fdes is not None:
            qf = _fde_block(q_fdes, len(rows), self.fde_config.output_dim)
            check(lib().mv_query_topk_batch_fde(self._h, blk.ctypes.data, code, len(rows), nmax, qf.ctypes.data, k, MODES[mode],
                                                None if ab is None else ab.ctypes.data, n_words, per_query, scores.ctypes.data, ids.ctypes.data,
                                                n.ctypes.data, C.byref(st) if want_stats else None))
        else:
            check(lib().mv_query_topk_batch(self._h, blk.ctypes.data, code, len(rows), nmax, k, MODES[mode], None if ab is None else ab.ctypes.data,
                                            n_words, per_query, scores.ctypes.data, ids.ctypes.data, n.ctypes.data, C.byref(st) if want_stats else None))
        res = [(scores[i, : n[i]].copy(), ids[i, : n[i]].copy()) for i in range(len(rows))]
        return (res, QueryStats.from_c(st)) if want_stats else res

    def query_device(self, q: Any, k: int, d_scores_ptr: int, d_ids_ptr: int, mode: str = "float", allow: Optional[np.ndarray] = None,
                     stream: int = 0, want_stats: bool = False) -> Optional[QueryStats]:
        """Top-k left in caller-provided DEVICE buffers (k floats / k int64), padded with (-inf, -1)."""
        qa, code = as_rows(q)
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        st = QueryStatsC()
        check(
            lib().mv_query_topk_device(
                self._h, qa.ctypes.data, code, qa.shape[0], int(k), MODES[mode], None if ab is None else ab.ctypes.data,
                0 if ab is None else ab.size, C.c_void_p(d_scores_ptr), C.c_void_p(d_ids_ptr), C.c_void_p(stream) if stream else None,
                C.byref(st) if want_stats else None,
            )
        )
        return QueryStats.from_c(st) if want_stats else None

    def query_device_async(self, q: Any, k: int, d_scores_ptr: int, d_ids_ptr: int, stream: int, mode: str = "float",
                           allow: Optional[np.ndarray] = None) -> "QueryStatsC":
        """mv_query_topk_device_async: enqueue only (the caller's `stream` is ordered behind the result); -> the pending stats
        record to hand to finish_stats() before the next query on this index."""
        qa, code = as_rows(q)
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        st = QueryStatsC()
        check(lib().mv_query_topk_device_async(self._h, qa.ctypes.data, code, qa.shape[0], int(k), MODES[mode], None if ab is None else ab.ctypes.data,
                                               0 if ab is None else ab.size, C.c_void_p(d_scores_ptr), C.c_void_p(d_ids_ptr), C.c_void_p(stream), C.byref(st)))
        return st

    def finish_stats(self, pending: "QueryStatsC") -> QueryStats:
        """Fill the HIP-event timings of the query `pending` came from (waits for that query's kernels, nothing else)."""
        check(lib().mv_query_stats_finish(self._h, C.byref(pending)))
        return QueryStats.from_c(pending)

    def score_all(self, q: Any, mode: str = "float", allow: Optional[np.ndarray] = None, want_stats: bool = False):
        qa, code = as_rows(q)
        out = np.empty(max(len(self), 1), np.float32)  # the library never writes past this, even if the corpus grows meanwhile
        n = C.c_int64()
        st = QueryStatsC()
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        check(
            lib().mv_score_all(
                self._h, qa.ctypes.data, code, qa.shape[0], MODES[mode], None if ab is None else ab.ctypes.data,
                0 if ab is None else ab.size, out.ctypes.data, out.size, C.byref(n), C.byref(st) if want_stats else None,
            )
        )
        out = out[: n.value]
        return (out, QueryStats.from_c(st)) if want_stats else out

    def score_candidates(self, q: Any, cand: Sequence[int], pad_to: int = 0, pads: Optional[Sequence[int]] = None) -> np.ndarray:
        """Exact float MaxSim of the named local pages, in list order.  pad_to = -1: the reference rule (every batch of
        128 candidates is zero-padded to its own longest page); 0: none; > 0: one length for all.  `pads` = an explicit
        pad length per candidate (row-sharded rerank: the batch is the GLOBAL list's)."""
        qa, code = as_rows(q)
        c = np.ascontiguousarray(cand, dtype=np.int32)
        out = np.empty(max(c.size, 1), np.float32)
        if pads is not None:
            pd = np.ascontiguousarray(pads, dtype=np.int32)
            if pd.size != c.size:
                raise ValueError("pads must have one entry per candidate")
            check(lib().mv_score_candidates_pads(self._h, qa.ctypes.data, code, qa.shape[0], c.ctypes.data, c.size, pd.ctypes.data, out.ctypes.data, None))
        else:
            check(lib().mv_score_candidates(self._h, qa.ctypes.data, code, qa.shape[0], c.ctypes.data, c.size, int(pad_to), out.ctypes.data, None))
        return out[: c.size]

    # -- sharded two-stage pipeline (device-resident stages; see include/mvmaxsim.h)
    def two_stage_coarse_device(self, q: Any, n_coarse: int, d_recs_ptr: int, allow: Optional[np.ndarray] = None, stream: int = 0,
                                mode: str = "fde_then_float", q_fde: Any = None) -> None:
        """Stage 1 of a staged query on this shard: coarse scan ("fde_then_float": the FDE slab; "fp8_then_float": the e4m3
        slab) + local top-n_coarse -> n_coarse 16-byte records in a device buffer.  q_fde: the caller's own FDE of the query (query())."""
        qa, code = as_rows(q)
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        if q_fde is not None:
            qf = _fde_block(q_fde, 1, self.fde_config.output_dim)
            check(lib().mv_two_stage_coarse_device_fde(self._h, qa.ctypes.data, code, qa.shape[0], qf.ctypes.data, int(n_coarse), MODES[mode],
                                                       None if ab is None else ab.ctypes.data, 0 if ab is None else ab.size, C.c_void_p(d_recs_ptr),
                                                       C.c_void_p(stream) if stream else None))
            return
        check(lib().mv_two_stage_coarse_device(self._h, qa.ctypes.data, code, qa.shape[0], int(n_coarse), MODES[mode], None if ab is None else ab.ctypes.data,
                                               0 if ab is None else ab.size, C.c_void_p(d_recs_ptr), C.c_void_p(stream) if stream else None))

    def rerank_plan(self, n_list: int, k: int, n_q_rows: int, mode: str = "fde_then_float", batched: bool = False) -> Tuple[int, str]:
        """-> (n_mid, tier): tier = "hbm" | "host" | "fp8" (what the rerank of a list of n_list candidates reads on this index);
        n_mid > 0: an e4m3 pruning stage keeps that many entries first (pinned-host tier behind a long FDE candidate list)."""
        n_mid, tier = C.c_int32(), C.c_int32()
        check(lib().mv_index_rerank_plan(self._h, MODES[mode], int(n_list), int(k), int(n_q_rows), 1 if batched else 0, C.byref(n_mid), C.byref(tier)))
        return int(n_mid.value), ("hbm", "host", "fp8")[tier.value]

    def two_stage_mid_device(self, q: Any, d_all_recs_ptr: int, world: int, n_coarse: int, d_mid_ptr: int, stream: int = 0,
                             mode: str = "fde_then_float") -> None:
        """Pruning stage: e4m3 scores of the entries of the GLOBAL coarse list this shard owns -> n_coarse floats (device)."""
        qa, code = as_rows(q)
        check(lib().mv_two_stage_mid_device(self._h, qa.ctypes.data, code, qa.shape[0], MODES[mode], C.c_void_p(d_all_recs_ptr), int(world), int(n_coarse),
                                            C.c_void_p(d_mid_ptr), C.c_void_p(stream) if stream else None))

    def two_stage_rerank_device(self, q: Any, d_all_recs_ptr: int, world: int, n_coarse: int, k: int, d_scores_ptr: int, d_ids_ptr: int,
                                stream: int = 0, mode: str = "fde_then_float", d_all_mid_ptr: int = 0, n_mid: int = 0) -> None:
        qa, code = as_rows(q)
        check(lib().mv_two_stage_rerank_device(self._h, qa.ctypes.data, code, qa.shape[0], MODES[mode], C.c_void_p(d_all_recs_ptr), int(world), int(n_coarse),
                                               C.c_void_p(d_all_mid_ptr) if d_all_mid_ptr else None, int(n_mid), int(k),
                                               C.c_void_p(d_scores_ptr), C.c_void_p(d_ids_ptr), C.c_void_p(stream) if stream else None))

    def page_rows(self, pages: Sequence[int]) -> np.ndarray:
        """Row counts of local pages (host metadata; no device work)."""
        c = np.ascontiguousarray(pages, dtype=np.int32)
        out = np.empty(max(c.size, 1), np.int32)
        check(lib().mv_index_page_rows(self._h, c.ctypes.data, c.size, out.ctypes.data))
        return out[: c.size]

    # -- persistence
    def save(self, path: str) -> None:
        check(lib().mv_index_save(self._h, path.encode()))

    @classmethod
    def load(cls, path: str, device: int = 0) -> "MvIndex":
        h = C.c_void_p()
        check(lib().mv_index_load(path.encode(), device, C.byref(h)))
        self = cls.__new__(cls)
        self._h = h
        self.device = device
        # re-read geometry through the C ABI is not exposed; the header is small, parse it here
        import struct

        with open(path, "rb") as f:
            hdr = f.read(8 + C.sizeof(ConfigC))
        cfg = ConfigC.from_buffer_copy(hdr[8:])
        self.stride_rows = cfg.stride_rows
        self.id_base = cfg.id_base
        self.flags = cfg.flags
        self.fde_config = FdeConfig(cfg.fde.dimension, cfg.fde.num_repetitions, cfg.fde.num_simhash_projections, cfg.fde.projection_dimension, cfg.fde.seed)
        del struct
        return self


# ------------------------------------------------------------------ stateless helpers
def sign_pack(x: Any, device: int = 0) -> np.ndarray:
    """fast_ops.binary_quantize_packed on the GPU: [n,d] fp32 -> [n, ceil(d/8)] uint8 (MSB first)."""
    a = np.ascontiguousarray(_to_host(x), dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    out = np.empty((a.shape[0], (a.shape[1] + 7) // 8), np.uint8)
    check(lib().mv_sign_pack(device, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data))
    return out


def hamming_batch(query: bytes, cands: Sequence[bytes], device: int = 0) -> List[int]:
    q = np.frombuffer(bytes(query), np.uint8)
    c = np.frombuffer(b"".join(bytes(x) for x in cands), np.uint8)
    if any(len(x) != q.size for x in cands):
        raise ValueError("All candidates must have the same length as query")  # binary_ops.rs:277-284
    out = np.empty(max(len(cands), 1), np.int32)
    check(lib().mv_hamming_batch(device, q.ctypes.data, c.ctypes.data if len(cands) else None, len(cands), q.size, out.ctypes.data))
    return out[: len(cands)].tolist()


def fde_encode(x: Any, cfg: Optional[FdeConfig] = None, is_query: bool = False, device: int = 0) -> np.ndarray:
    cfg = cfg or FdeConfig()
    a = np.ascontiguousarray(_to_host(x), dtype=np.float32)
    out = np.empty(cfg.output_dim, np.float32)
    cc = cfg.to_c()
    check(lib().mv_fde_encode(device, C.byref(cc), a.ctypes.data, a.shape[0], 1 if is_query else 0, out.ctypes.data))
    return out


def synth_ragged_rows(seed: int, unit: int, min_rows: int, max_rows: int) -> int:
    """Row count of unit `unit` in a ragged synthetic corpus (mv_index_fill_synthetic_ragged): min_rows + splitmix64(seed ^ golden * (unit + 1))
    % (max_rows - min_rows + 1).  Pure host arithmetic (the library computes the same on the host and uploads the counts)."""
    m = (1 << 64) - 1
    z = (int(seed) ^ (0x9E3779B97F4A7C15 * (int(unit) + 1))) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return int(min_rows) + int(z % (int(max_rows) - int(min_rows) + 1))


def synth_rows(seed: int, unit: int, n_rows: int, device: int = 0) -> np.ndarray:
    out = np.empty((n_rows, 128), np.uint16)
    check(lib().mv_synth_rows(device, seed, unit, n_rows, out.ctypes.data))
    return out


def calibrate_read_bw(bytes_: int = 8 << 30, iters: int = 5, device: int = 0) -> float:
    g = C.c_double()
    check(lib().mv_calibrate_read_bw(device, bytes_, iters, C.byref(g)))
    return float(g.value)


def calibrate(what: str, bytes_: int = 8 << 30, iters: int = 5, device: int = 0) -> float:
    """Measured peaks (same process as the measurement): "read_nt" / "read_ldsdma" -> GB/s, "mfma_bf16" (16x16x32 chains) / "mfma_bf16_32x32" (32x32x16 chains) -> TFLOP/s."""
    g = C.c_double()
    code = {"read_nt": _lib.MV_CAL_READ_NT, "mfma_bf16": _lib.MV_CAL_MFMA_BF16, "read_ldsdma": _lib.MV_CAL_READ_LDSDMA,
            "mfma_bf16_32x32": _lib.MV_CAL_MFMA_BF16_32X32, "fde_scan_regs": 14, "fde_scan_rowq": 15, "stream_probe": 19}[what]
    check(lib().mv_calibrate(device, code, bytes_, iters, C.byref(g)))
    return float(g.value)


class ShardComm:
    """mv_comm: R MvIndex shards driven from ONE process (one GPU each, or logical shards on one GPU).  Shard i must own
    the global ids [id_base_i, id_base_i + len_i), ascending with i.  query() returns what ONE index holding every page
    would return (same order, same tie rule)."""

    TRANSPORTS = {"auto": _lib.MV_COMM_AUTO, "rccl": _lib.MV_COMM_RCCL, "p2p": _lib.MV_COMM_P2P, "host": _lib.MV_COMM_HOST}

    def __init__(self, shards: Sequence["MvIndex"], transport: str = "auto"):
        self.shards = list(shards)
        devs = np.ascontiguousarray([s.device for s in self.shards], dtype=np.int32)
        h = C.c_void_p()
        check(lib().mv_comm_create(len(self.shards), devs.ctypes.data, self.TRANSPORTS[transport], C.byref(h)))
        self._h = h
        for i, s in enumerate(self.shards):
            check(lib().mv_comm_attach(self._h, i, s._h))

    @property
    def transport(self) -> str:
        code = lib().mv_comm_transport(self._h)
        return {v: k for k, v in self.TRANSPORTS.items()}[code]

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().mv_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query(self, q: Any, k: int, mode: str = "float", allow: Optional[np.ndarray] = None, want_stats: bool = False, q_fde: Any = None):
        qa, code = as_rows(q)
        k = int(k)
        scores = np.empty(max(k, 1), np.float32)
        ids = np.empty(max(k, 1), np.int64)
        n = C.c_int32()
        ab = None if allow is None else np.ascontiguousarray(allow, dtype=np.uint32)
        st = (QueryStatsC * len(self.shards))()
        if q_fde is not None:  # the caller's own FDE of the query: every shard's coarse stage uses it (MvIndex.query)
            qf = _fde_block(q_fde, 1, self.shards[0].fde_config.output_dim)
            check(lib().mv_comm_query_topk_fde(self._h, qa.ctypes.data, code, qa.shape[0], qf.ctypes.data, k, MODES[mode], None if ab is None else ab.ctypes.data,
                                               0 if ab is None else ab.size, scores.ctypes.data, ids.ctypes.data, C.byref(n),
                                               C.cast(st, C.c_void_p) if want_stats else None))
        else:
            check(lib().mv_comm_query_topk(self._h, qa.ctypes.data, code, qa.shape[0], k, MODES[mode], None if ab is None else ab.ctypes.data,
                                           0 if ab is None else ab.size, scores.ctypes.data, ids.ctypes.data, C.byref(n),
                                           C.cast(st, C.c_void_p) if want_stats else None))
        res = (scores[: n.value].copy(), ids[: n.value].copy())
        return res + ([QueryStats.from_c(x) for x in st],) if want_stats else res

    def query_batch(self, queries: Sequence[Any], k: int, mode: str = "fde_then_float", allow: Optional[np.ndarray] = None,
                    want_stats: bool = False, allows: Optional[Sequence[Optional[np.ndarray]]] = None, n_docs: int = 0, q_fdes: Any = None):
        """mv_comm_query_topk_batch: a batch of requests against the sharded corpus (arguments as MvIndex.query_batch).
        "fde_then_float": one FDE-slab pass per shard and 32 requests, one exchange of all their candidate records, every
        request's share of its GLOBAL candidate list reranked in one launch per shard.  -> [(scores, ids)] per request
        [, per-shard QueryStats]."""
        blk, code, nmax = _stack_queries(queries)
        k = int(k)
        nq = blk.shape[0]
        scores = np.empty((nq, max(k, 1)), np.float32)
        ids = np.empty((nq, max(k, 1)), np.int64)
        n = np.zeros(nq, np.int32)
        ab, n_words, per_query = _allow_block(nq, allow, allows, n_docs)
        st = (QueryStatsC * len(self.shards))()
        if q_fdes is not None:
            qf = _fde_block(q_fdes, nq, self.shards[0].fde_config.output_dim)
            check(lib().mv_comm_query_topk_batch_fde(self._h, blk.ctypes.data, code, nq, nmax, qf.ctypes.data, k, MODES[mode], None if ab is None else ab.ctypes.data,
                                                     n_words, per_query, scores.ctypes.data, ids.ctypes.data, n.ctypes.data,
                                                     C.cast(st, C.c_void_p) if want_stats else None))
        else:
            check(lib().mv_comm_query_topk_batch(self._h, blk.ctypes.data, code, nq, nmax, k, MODES[mode], None if ab is None else ab.ctypes.data, n_words,
                                                 per_query, scores.ctypes.data, ids.ctypes.data, n.ctypes.data, C.cast(st, C.c_void_p) if want_stats else None))
        res = [(scores[i, : n[i]].copy(), ids[i, : n[i]].copy()) for i in range(nq)]
        return (res, [QueryStats.from_c(x) for x in st]) if want_stats else res
