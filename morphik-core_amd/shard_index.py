"""ShardedIndex -- R MvIndex shards + one mv_comm behind the interface the stores use for ONE index.

The reference builds one store object in one process (core/services_init.py:141-165); a node with 8 MI355X therefore
needs the shard fan-out and the top-k exchange BELOW the store, not in a launcher.  Shard r owns the fixed global id
range [r * capacity_per_shard, (r + 1) * capacity_per_shard) -- ascending with r, which is what makes the merged order
(score desc, id asc) the single-index order -- on device `devices[r]` (several shards may name the same device: logical
shards, used by the tests on a 1-GPU box).  Queries go through libmvmaxsim's communicator (RCCL all-gather of k pairs
over xGMI, peer copies, or the host reference path); this file only routes writes and remaps ids.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np


def _checkpoint_has_exact_split(shard_path: str) -> bool:
    """MV_WITH_EXACT_SPLIT in the flags of a shard file's header (8-byte magic + mv_index_config)."""
    import ctypes as C

    from ._lib import MV_WITH_EXACT_SPLIT, ConfigC

    try:
        with open(shard_path, "rb") as f:
            hdr = f.read(8 + C.sizeof(ConfigC))
        return bool(ConfigC.from_buffer_copy(hdr[8:]).flags & MV_WITH_EXACT_SPLIT)
    except (OSError, ValueError):
        return False


class ShardedIndex:
    def __init__(self, capacity_pages: int, stride_rows: int, devices: Sequence[int], with_float: bool = True, with_binary: bool = False,
                 with_fde: bool = False, with_fp8: bool = False, fde=None, transport: str = "auto", id_base: int = 0,
                 index_cls=None, comm_cls=None, with_host_exact: bool = False, with_exact_split: bool = False, with_float_lo: bool = False, packed: bool = False, capacity_rows: int = 0, with_fde_e4m3: bool = False, with_fde_fp4: bool = False):
        from .index import MvIndex, ShardComm

        index_cls = index_cls or MvIndex
        comm_cls = comm_cls or ShardComm
        self.devices = [int(d) for d in devices]
        self.n_shards = len(self.devices)
        if self.n_shards < 1:
            raise ValueError("ShardedIndex needs at least one device")
        self.per = -(-int(capacity_pages) // self.n_shards)  # ceil: slots per shard
        self.stride_rows = int(stride_rows)
        self.id_base = int(id_base)
        self.device = self.devices[0]
        extra = {"with_host_exact": True} if with_host_exact else {}  # every shard pins the exact rows of ITS pages (per / n of the corpus)
        if packed:
            extra["packed"] = True  # every shard packs ITS pages; the row budget is split like the page budget
            extra["capacity_rows"] = (-(-int(capacity_rows) // self.n_shards) + 15) // 16 * 16 if capacity_rows else 0
        if with_fde_e4m3:
            extra["with_fde_e4m3"] = True  # every shard's coarse stage reads the e4m3 copy of ITS FDE slab
        if with_fde_fp4:
            extra["with_fde_fp4"] = True  # ... or the fp4 copy (single requests)
        if with_float_lo:
            extra["with_float_lo"] = True  # every shard keeps the lo half of its bf16 slab (fp32-faithful rerank / scan)
        if with_exact_split:
            extra["with_exact_split"] = True  # ... after filling the HBM its other slabs leave free
            import os

            if len(set(self.devices)) < self.n_shards and "MV_EXACT_HBM_MAX_PAGES" not in os.environ:
                # the split is "whatever this device has free": the first shard created on a device would leave none for the slabs of the next
                raise ValueError("exact_tier=\"split\" takes the device's free HBM per shard: give every shard its own device "
                                 "(or bound the HBM part with MV_EXACT_HBM_MAX_PAGES)")
        self.shards = [
            index_cls(capacity_pages=self.per, stride_rows=stride_rows, device=d, with_float=with_float, with_binary=with_binary,
                      with_fde=with_fde, with_fp8=with_fp8, fde=fde, id_base=self.id_base + r * self.per, **extra)
            for r, d in enumerate(self.devices)
        ]
        self.comm = comm_cls(self.shards, transport=transport)

    # -- lifecycle
    def close(self) -> None:
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = None
        for s in getattr(self, "shards", []):
            s.close()
        self.shards = []

    def __len__(self) -> int:
        return sum(len(s) for s in self.shards)

    @property
    def capacity(self) -> int:
        return self.per * self.n_shards

    @property
    def transport(self) -> str:
        return self.comm.transport

    def set_option(self, option: int, value: int) -> None:
        for s in self.shards:
            s.set_option(option, value)

    def _route(self, page: int) -> Tuple[int, int]:
        r, local = divmod(int(page) - self.id_base, self.per)
        if r < 0 or r >= self.n_shards:
            raise ValueError(f"page {page} is outside every shard")
        return r, local

    def _pick(self, n_pages: int, device: Optional[int] = None) -> int:
        """Least-full shard with room for the batch (on `device` when the rows already sit there)."""
        best, best_free = -1, -1
        for r, s in enumerate(self.shards):
            if device is not None and self.devices[r] != device:
                continue
            free = self.per - len(s)
            if free >= n_pages and free > best_free:
                best, best_free = r, free
        return best

    # -- build: a batch lands on ONE shard so that its pages get consecutive global ids
    def add(self, pages: Sequence[Any], doc_ordinals: Optional[Sequence[int]] = None) -> int:
        if len(pages) == 0:
            return self.id_base
        r = self._pick(len(pages))
        if r < 0:
            from ._lib import MvError

            raise MvError(-4, f"slab full: no shard has {len(pages)} free slots (capacity {self.per} per shard)")
        return self.shards[r].add(pages, doc_ordinals) + self.shards[r].id_base

    def add_device(self, d_ptr: int, dtype_code: int, n_rows: Sequence[int], doc_ordinals: Optional[Sequence[int]] = None,
                   device: Optional[int] = None) -> int:
        r = self._pick(len(n_rows), device if device is not None else self.devices[0])
        if r < 0:
            from ._lib import MvError

            raise MvError(-4, f"slab full: no shard on device {device} has {len(n_rows)} free slots")
        return self.shards[r].add_device(d_ptr, dtype_code, n_rows, doc_ordinals) + self.shards[r].id_base

    def import_fde(self, page0: int, fde: Any) -> None:
        """Caller-supplied document FDE vectors for pages [page0, page0 + n) (global ids of ONE add() call: they lie in one shard)."""
        r, local = self._route(page0)
        self.shards[r].import_fde(local, fde)

    def remove_page(self, page: int) -> None:
        r, local = self._route(page)
        self.shards[r].remove_page(local)

    def remove_doc(self, doc_ordinal: int) -> int:
        return sum(s.remove_doc(doc_ordinal) for s in self.shards)

    def page_rows(self, pages: Sequence[int]) -> np.ndarray:
        out = np.empty(len(pages), np.int32)
        for j, p in enumerate(pages):
            r, local = self._route(p)
            out[j] = self.shards[r].page_rows([local])[0]
        return out

    def compact(self) -> Dict[int, int]:
        """Every shard compacts on its own (ids stay inside the shard's range).  -> {old global id: new global id}."""
        remap: Dict[int, int] = {}
        for s in self.shards:
            o2n = s.compact()
            base = s.id_base
            for o, n in enumerate(o2n.tolist()):
                if n >= 0:
                    remap[base + o] = base + int(n)
        return remap

    def rebalance_exact_tier(self, max_moves: int = 0) -> int:
        """Every shard re-places the hot pages of ITS split exact tier (MvIndex.rebalance_exact_tier).  -> pages moved in all."""
        return sum(int(s.rebalance_exact_tier(max_moves)) for s in self.shards if hasattr(s, "rebalance_exact_tier"))

    def fde_placement_trial(self, trials: int = 3):
        """Every shard tries other allocations for ITS FDE slab (MvIndex.fde_placement_trial).  -> [(ms before, ms after, moves)] per shard."""
        return [tuple(s.fde_placement_trial(trials)) for s in self.shards if hasattr(s, "fde_placement_trial")]

    # -- query
    def query(self, q: Any, k: int, mode: str = "float", allow: Optional[np.ndarray] = None, want_stats: bool = False, q_fde: Any = None):
        if q_fde is not None:
            return self.comm.query(q, k, mode=mode, allow=allow, want_stats=want_stats, q_fde=q_fde)
        return self.comm.query(q, k, mode=mode, allow=allow, want_stats=want_stats)

    _SINGLE_STAGE = ("float", "float_fp8", "binary", "fde")  # one scan + top-k: the merge of per-shard top-k lists is exact

    def query_batch(self, queries: Sequence[Any], k: int, mode: str = "float", allow: Optional[np.ndarray] = None, want_stats: bool = False,
                    allows: Optional[Sequence[Optional[np.ndarray]]] = None, n_docs: int = 0, q_fdes: Any = None):
        """Coalesced requests on a sharded store.
        Single-stage modes: every shard runs the WHOLE batch through its own mv_query_topk_batch (one slab pass per shard for
        all requests; the shards run side by side, one host thread each -- the library releases the GIL), and the per-shard
        top-k lists of a request are merged with the communicator's rule (score desc, ties by ascending global id), which
        is the single-index order.  The two-stage FDE pipeline keeps its GLOBAL candidate rule (the coarse top-n is taken over
        all shards before the rerank), and so does "fp8_then_float" (the e4m3 scan's GLOBAL top-n is re-scored exactly): their
        requests go through the communicator's batched form (mv_comm_query_topk_batch: one FDE / e4m3 slab pass per shard and
        group of requests, one exchange of all their candidate records)."""
        fkw = {} if q_fdes is None else {"q_fdes": q_fdes}  # the callers' own query FDEs (MvIndex.query_batch)
        if mode in ("fde_then_float", "fp8_then_float") and len(queries) >= 2 and hasattr(self.comm, "query_batch"):
            return self.comm.query_batch(queries, k, mode=mode, allow=allow, allows=allows, n_docs=n_docs, want_stats=want_stats, **fkw)
        if mode not in self._SINGLE_STAGE or len(queries) < 2:
            out = []
            for j, q in enumerate(queries):
                a = allow if allows is None else allows[j]
                qk = {} if q_fdes is None else {"q_fde": np.asarray(q_fdes, np.float32).reshape(len(queries), -1)[j]}
                out.append(self.comm.query(q, k, mode=mode, allow=None if a is None else np.asarray(a, np.uint32), **qk))
            return out
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=self.n_shards, thread_name_prefix="mv-shard")

        def run(shard):
            return shard.query_batch(queries, k, mode=mode, allow=allow, allows=allows, n_docs=n_docs, **fkw)

        per_shard = list(self._pool.map(run, self.shards))
        out = []
        for j in range(len(queries)):
            sc = np.concatenate([np.asarray(per_shard[r][j][0], np.float32) for r in range(self.n_shards)])
            ids = np.concatenate([np.asarray(per_shard[r][j][1], np.int64) for r in range(self.n_shards)])
            order = np.lexsort((ids, -sc.astype(np.float64)))[: int(k)]
            out.append((sc[order], ids[order]))
        return out

    # -- persistence: one file per shard next to `path`
    def save(self, path: str) -> None:
        for r, s in enumerate(self.shards):
            s.save(f"{path}.shard{r}")

    @classmethod
    def load(cls, path: str, devices: Sequence[int], transport: str = "auto", index_cls=None, comm_cls=None) -> "ShardedIndex":
        from .index import MvIndex, ShardComm

        index_cls = index_cls or MvIndex
        comm_cls = comm_cls or ShardComm
        self = cls.__new__(cls)
        self.devices = [int(d) for d in devices]
        self.n_shards = len(self.devices)
        import os

        if os.path.exists(f"{path}.shard{self.n_shards}"):
            raise ValueError(f"{path}: more shard files than the {self.n_shards} devices given -- refusing to drop shards")
        if len(set(self.devices)) < self.n_shards and "MV_EXACT_HBM_MAX_PAGES" not in os.environ and _checkpoint_has_exact_split(f"{path}.shard0"):
            # the same rule as the constructor: mv_index_load sizes the HBM part of a split exact tier by what the device has free at
            # that moment -- the first shard loaded on a shared device would leave nothing for the slabs of the next
            raise ValueError("this checkpoint holds a split exact tier (exact_tier=\"split\"): give every shard its own device "
                             "(or bound the HBM part with MV_EXACT_HBM_MAX_PAGES)")
        self.shards = [index_cls.load(f"{path}.shard{r}", device=d) for r, d in enumerate(self.devices)]
        self.per = self.shards[0].capacity
        self.stride_rows = self.shards[0].stride_rows
        self.id_base = self.shards[0].id_base
        for r, sh in enumerate(self.shards):  # the files must be the consecutive id ranges of ONE checkpoint
            if sh.id_base != self.id_base + r * self.per or sh.capacity != self.per:
                for x in self.shards:
                    x.close()
                raise ValueError(f"{path}.shard{r}: id_base {sh.id_base} / capacity {sh.capacity} do not continue shard 0 "
                                 f"(id_base {self.id_base}, {self.per} slots per shard)")
        self.device = self.devices[0]
        self.comm = comm_cls(self.shards, transport=transport)
        return self
