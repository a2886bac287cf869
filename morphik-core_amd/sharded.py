"""Row-sharded corpus over N ranks (one process per GPU) -- SURVEY.md section 8(e).

Rank r owns the contiguous page range shard_range(n_total, r, R); its MvIndex is created with
id_base = lo so local results already carry GLOBAL page ids.  A query is broadcast by the host
(every rank receives the same (Q,128) array), each rank scans its own slab, and the only
exchange is one all-gather of k (score, id) pairs per rank -- RCCL over xGMI on the GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.  120 bytes per rank at k=10: pure
latency, no bandwidth term; no embedding ever crosses a link.

The reference has no distributed code at all; this is the one collective we add.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced: the first (n_total % world) ranks get one extra page."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_topk(scores, ids, k: int, compact: bool = True):
    """scores/ids: [R, kk] tensors of per-shard results, each row sorted (score desc, id asc) and padded
    with (-inf, -1); shards in rank order own ascending id ranges.  Returns the global top-k in the same
    order.  A stable descending sort keeps (rank, position) order among equal scores, which IS
    ascending id order, so ties resolve exactly as on a single index.
    compact=False returns exactly k entries still padded with (-inf, -1): no boolean indexing, hence no
    device->host synchronisation on the query path (the padded tail is dropped by whoever reads the result)."""
    import torch

    s = scores.reshape(-1)
    i = ids.reshape(-1)
    order = torch.sort(s, descending=True, stable=True).indices[:k]
    ms, mi = s[order], i[order]
    if not compact:
        return ms, mi
    keep = mi >= 0
    return ms[keep], mi[keep]


def allgather_topk(local_scores, local_ids, k: int, group=None, compact: bool = True):
    """local_*: [kk] tensors on this rank (cuda for nccl/RCCL, cpu for gloo). Every rank returns the merged top-k.
    The (score, id) pairs travel as ONE float64 buffer (fp32 scores and ids < 2^53 are exact in it): one collective
    of 16*kk bytes per rank instead of two."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():  # single process, no communicator: nothing to gather
        return merge_topk(local_scores[None], local_ids[None], k, compact)
    world = dist.get_world_size(group)  # a 1-rank communicator still runs the collective (RCCL smoke test)
    kk = local_scores.numel()
    mine = torch.cat([local_scores.reshape(-1).to(torch.float64), local_ids.reshape(-1).to(torch.float64)])
    allb = torch.empty(world * 2 * kk, dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(allb, mine, group=group)
    allb = allb.view(world, 2, kk)
    return merge_topk(allb[:, 0].to(local_scores.dtype), allb[:, 1].to(local_ids.dtype), k, compact)


def allgather_topk_batch(local_scores, local_ids, k: int, group=None):
    """Batched form: local_* are [B, kk] (row b = this shard's top-kk of query b, padded with -inf / -1).  ONE
    all-gather of B*kk pairs per rank; returns a list of (scores, ids) per query."""
    import torch
    import torch.distributed as dist

    B, kk = local_scores.shape
    if not dist.is_initialized():
        return [merge_topk(local_scores[b][None], local_ids[b][None], k) for b in range(B)]
    world = dist.get_world_size(group)
    gs = torch.empty(world * B * kk, dtype=local_scores.dtype, device=local_scores.device)
    gi = torch.empty(world * B * kk, dtype=local_ids.dtype, device=local_ids.device)
    dist.all_gather_into_tensor(gs, local_scores.contiguous().view(-1), group=group)
    dist.all_gather_into_tensor(gi, local_ids.contiguous().view(-1), group=group)
    gs, gi = gs.view(world, B, kk), gi.view(world, B, kk)
    return [merge_topk(gs[:, b], gi[:, b], k) for b in range(B)]


class ShardedSearcher:
    """Wraps a per-rank search function.  `local_topk(q, k)` must return two tensors of exactly k entries
    (padded with -inf / -1) on the collective's device; on the GPU it is MvIndex.query_device writing
    into torch-owned cuda buffers, in the CPU tests an oracle-backed stand-in."""

    def __init__(self, local_topk: Callable, group=None):
        self.local_topk = local_topk
        self.group = group

    def query(self, q, k: int, compact: bool = True):
        s, i = self.local_topk(q, k)
        return allgather_topk(s, i, k, self.group, compact)

    def query_batch(self, queries, k: int, local_topk_batch: Callable):
        """Several queries in one slab pass per shard (mv_query_topk_batch) and one all-gather for all of them.
        `local_topk_batch(queries, k)` -> two [B, k] tensors padded with (-inf, -1)."""
        s, i = local_topk_batch(queries, k)
        return allgather_topk_batch(s, i, k, self.group)


def make_gpu_local_topk(index, device=None, mode: str = "float", collect_stats: Optional[list] = None):
    """Adapter: MvIndex -> local_topk over torch cuda buffers (no host round trip of the results)."""
    import torch

    dev = torch.device("cuda", index.device) if device is None else device
    bufs = {}

    def local_topk(q, k):
        if k not in bufs:
            bufs[k] = (torch.empty(k, dtype=torch.float32, device=dev), torch.empty(k, dtype=torch.int64, device=dev))
        s, i = bufs[k]
        st = index.query_device(q, k, s.data_ptr(), i.data_ptr(), mode=mode, want_stats=collect_stats is not None)
        if collect_stats is not None:
            collect_stats.append(st)
        return s, i

    return local_topk


def make_gpu_local_topk_batch(index, device=None, mode: str = "float"):
    """Adapter: MvIndex.query_batch -> padded [B, k] tensors on the collective's device."""
    import numpy as np
    import torch

    dev = torch.device("cuda", index.device) if device is None else device

    def local_topk_batch(queries, k):
        res = index.query_batch(queries, k, mode=mode)
        s = np.full((len(res), k), -np.inf, np.float32)
        i = np.full((len(res), k), -1, np.int64)
        for b, (rs, ri) in enumerate(res):
            s[b, : len(rs)] = rs
            i[b, : len(ri)] = ri
        return torch.from_numpy(s).to(dev), torch.from_numpy(i).to(dev)

    return local_topk_batch


class GpuShardedSearcher:
    """The RCCL path with the fewest steps per query, none of which waits for the host:
      1. MvIndex.query_device_async -- scan + local top-k enqueued on the index's stream, torch's current stream ordered behind
         the result; the k ids and k scores land in the two halves of ONE preallocated block ({int64 ids[k], float scores[k]})
      2. ONE all_gather_into_tensor of that block (uint8, mv_topk_block_bytes(k) = 128 B per rank at k = 10)
      3. ONE library launch (mv_merge_topk_blocks, torch's current stream, behind the collective) -> merged top-k.
    The HIP-event timings of a scan are collected ONE QUERY LATER (the library keeps two sets of timing events): query i's record
    is finished right after query i + 1 was enqueued, so the host never holds the GPU up -- call flush() for the last one.  Results are
    padded with (-inf, -1) and become valid in stream order (read them after a synchronize or from the same stream)."""

    def __init__(self, index, device=None, mode: str = "float", group=None, collect_stats=None):
        import torch
        import torch.distributed as dist

        from ._lib import lib

        self.index, self.mode, self.group, self.stats = index, mode, group, collect_stats
        self.dev = torch.device("cuda", index.device) if device is None else device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._lib = lib()
        self._bufs = {}
        self._flip = 0
        self._pending = None  # the stats record of the last enqueued query (finished one query later, or by flush())

    # bits of mv_query_stats.reserved that only ROUTE a deferred record (csrc/mv_index_priv.h: done / deferred / parity); anything else
    # is stage accounting (the FDE modes, fp8_then_float, the MV_OPT_FLOAT_LO_SCAN 2 cascade), which reads the candidate list the NEXT
    # query overwrites: such a record must be finished before the next query is enqueued (mv_query_stats_finish: MV_ERR_STATE otherwise)
    _ROUTING_BITS = (1 << 24) | (1 << 25) | (1 << 26)

    @classmethod
    def _may_trail(cls, pending) -> bool:
        return (int(pending.reserved) & ~cls._ROUTING_BITS) == 0

    def flush(self) -> None:
        """Collect the timings of the last enqueued query (waits for its scan)."""
        if self._pending is not None and self.stats is not None:
            self.stats.append(self.index.finish_stats(self._pending))
        self._pending = None

    def _buffers(self, k):
        import torch

        # two buffer sets used alternately: the collective of query i may still be reading set A while the
        # library writes the local top-k of query i+1 (they run on different streams)
        self._flip ^= 1
        key = (k, self._flip)
        if key not in self._bufs:
            d, w = self.dev, self.world
            bb = int(self._lib.mv_topk_block_bytes(k))
            mine = torch.zeros(bb, dtype=torch.uint8, device=d)
            self._bufs[key] = (mine, mine.data_ptr() + 8 * k, mine.data_ptr(), torch.empty(w * bb, dtype=torch.uint8, device=d),
                               torch.empty(k, dtype=torch.float32, device=d), torch.empty(k, dtype=torch.int64, device=d))
        return self._bufs[key]

    def query(self, q, k: int):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from ._lib import check

        mine, p_scores, p_ids, gathered, os_, oi = self._buffers(k)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        pending = self.index.query_device_async(q, k, p_scores, p_ids, stream, mode=self.mode)
        if dist.is_initialized():
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
        else:
            gathered = mine
        check(self._lib.mv_merge_topk_blocks(self.index.device, C.c_void_p(gathered.data_ptr()), self.world, k, k,
                                             C.c_void_p(os_.data_ptr()), C.c_void_p(oi.data_ptr()), C.c_void_p(stream)))
        if self.stats is not None:
            prev, self._pending = self._pending, None
            if prev is not None:  # the query BEFORE this one: its scan ran while this one was being enqueued
                self.stats.append(self.index.finish_stats(prev))
            if self._may_trail(pending):
                self._pending = pending
            else:  # the record says so itself (not the mode's name): stage accounting -> collect now
                self.stats.append(self.index.finish_stats(pending))
        return os_, oi

    def exchange_only(self, k: int):
        """Steps 2 and 3 alone on the block of the last query(k): what the exchange adds to a step, measured by itself."""
        import ctypes as C

        import torch
        import torch.distributed as dist

        from ._lib import check

        if (k, self._flip) not in self._bufs:
            raise RuntimeError("exchange_only(k) repeats the exchange of the last query(q, k): run one first")
        mine, _ps, _pi, gathered, os_, oi = self._bufs[(k, self._flip)]
        if dist.is_initialized():
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
        else:
            gathered = mine
        check(self._lib.mv_merge_topk_blocks(self.index.device, C.c_void_p(gathered.data_ptr()), self.world, k, k, C.c_void_p(os_.data_ptr()),
                                             C.c_void_p(oi.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        return os_, oi


class HostShardedSearcher:
    """The same step when the collective runs on HOST memory (gloo: the CPU tests, and N ranks sharing one GPU): the local top-k
    comes back through the index's pinned result buffers (MvIndex.query: one stream synchronisation), the k (score, id) pairs
    travel as ONE 16-byte-per-pair float64 buffer (fp32 scores and ids < 2^53 are exact in it) in ONE all-gather into a
    preallocated tensor, and the merge is a numpy lexsort with the single index's tie rule (score desc, then rank / position =
    ascending id).  -> (scores, ids) numpy arrays of exactly k entries padded with (-inf, -1)."""

    def __init__(self, index, mode: str = "float", group=None, collect_stats=None):
        import torch.distributed as dist

        self.index, self.mode, self.group, self.stats = index, mode, group, collect_stats
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bufs = {}

    def query(self, q, k: int):
        import numpy as np
        import torch
        import torch.distributed as dist

        if k not in self._bufs:
            self._bufs[k] = (torch.empty(2 * k, dtype=torch.float64), torch.empty(self.world * 2 * k, dtype=torch.float64))
        mine, allb = self._bufs[k]
        res = self.index.query(q, k, mode=self.mode, want_stats=self.stats is not None)
        if self.stats is not None:
            self.stats.append(res[2])
        s, i = res[0], res[1]
        m = mine.numpy()
        m[:k] = -np.inf
        m[k:] = -1
        m[: len(s)] = s
        m[k : k + len(i)] = i
        return self.exchange_only(k)

    def exchange_only(self, k: int):
        """The all-gather of the packed buffer (as the last query(k) left it) and the merge -- by themselves."""
        import numpy as np
        import torch.distributed as dist

        if k not in self._bufs:
            raise RuntimeError("exchange_only(k) repeats the exchange of the last query(q, k): run one first")
        mine, allb = self._bufs[k]
        m = mine.numpy()
        if dist.is_initialized():
            dist.all_gather_into_tensor(allb, mine, group=self.group)
            g = allb.numpy().reshape(self.world, 2, k)
        else:
            g = m.reshape(1, 2, k)
        gs, gi = g[:, 0].reshape(-1), g[:, 1].reshape(-1)
        order = np.lexsort((np.arange(gs.size), -gs))[:k]  # stable: equal scores keep (rank, position) order
        return gs[order].astype(np.float32), gi[order].astype(np.int64)


class TwoStageShardedSearcher:
    """The staged pipelines on a row-sharded corpus with the SAME candidate set as one big
    index, whatever the rank count.  "fde_then_float" is the reference pipeline (FDE coarse search -> top-n candidates ->
    exact MaxSim rerank -> top-k, fast_multivector_store.py:521-556); "fp8_then_float" takes the e4m3 scan's top-n instead:

      1. every rank: coarse scan of its shard -> local top-n (score, global id)
      2. all-gather of n pairs per rank (16 B each) -> global coarse top-n, identical on every rank
      3. every rank keeps the candidates IT OWNS; the candidates' row counts travel in the same all-gather, so every
         rank knows the pad length of each candidate: the longest page of ITS batch of 128 in the GLOBAL list
         (score_multi_vector scores passages in batches of 128, each padded on its own by pad_sequence, :553-555)
      3b. (only with a pinned-host exact tier behind a list longer than MV_OPT_RERANK_N) pruning stage: every rank scores its
         owned candidates on its e4m3 slab, the n scores per rank are all-gathered, the n_mid best list positions stay
      4. every rank: exact MaxSim of its own candidates (no embedding crosses xGMI) -> local top-k
      5. all-gather of k pairs per rank -> merged top-k

    Small collectives, all latency-bound.  This class is the host-driven form whose callables keep it testable on
    CPU (gloo + oracle); GpuTwoStageSearcher below is the device-resident form of the same steps.
      local_coarse(q, n, allow) -> (scores[n], ids[n]) tensors on the collective's device, padded (-inf, -1), GLOBAL ids
      local_rows(global_ids)    -> int array of row counts of owned pages
      local_rerank(q, global_ids, pads) -> float32 array of exact MaxSim scores of owned pages (pads[i] = pad length)
      local_prune(q, global_ids, pads)  -> float32 array of e4m3 MaxSim scores of owned pages (optional; with n_mid)
    `id_range` = [lo, hi) of global ids this rank owns."""

    BATCH = 128  # score_multi_vector's passage batch size

    def __init__(self, local_coarse: Callable, local_rows: Callable, local_rerank: Callable, id_range: Tuple[int, int],
                 group=None, pad_semantics: bool = True, local_prune: Optional[Callable] = None, n_mid: int = 0):
        self.local_coarse, self.local_rows, self.local_rerank = local_coarse, local_rows, local_rerank
        self.lo, self.hi = int(id_range[0]), int(id_range[1])
        self.group, self.pad = group, pad_semantics
        self.local_prune, self.n_mid = local_prune, int(n_mid)

    # The local phases are public so that R logical shards on ONE device can be driven without a process group
    # (tests; SURVEY 8e "R logical shards on one device"); query() chains them with the collectives in between.
    def coarse(self, q, k: int, coarse_n: Optional[int] = None, allow=None):
        n = int(coarse_n) if coarse_n else min(10 * k, 75)  # reference: top_k = min(10 * k, 75) (:529)
        cs, ci = self.local_coarse(q, n, allow)
        return n, cs, ci

    def batch_pads(self, global_ids, global_rows):
        """Pad length of every entry of the GLOBAL candidate list (coarse rank order): longest page of its batch of 128."""
        import numpy as np

        gid = np.asarray(global_ids, np.int64)
        rows = np.where(gid >= 0, np.asarray(global_rows, np.int64), 0)
        pads = np.zeros(gid.size, np.int32)
        if self.pad:
            for j in range(0, gid.size, self.BATCH):
                pads[j : j + self.BATCH] = rows[j : j + self.BATCH].max() if rows[j : j + self.BATCH].size else 0
        return pads

    def prune_scores(self, q, gid, pads):
        """This rank's e4m3 scores of the GLOBAL list: -inf at the positions other ranks own (float32 [n])."""
        import numpy as np

        out = np.full(gid.size, -np.inf, np.float32)
        own = np.nonzero((gid >= self.lo) & (gid < self.hi))[0]
        if own.size:
            out[own] = np.asarray(self.local_prune(q, gid[own], pads[own]), np.float32)
        return out

    @staticmethod
    def prune_keep(all_mid, n_mid: int):
        """all_mid: [world, n] pruning scores (one owner per position) -> boolean mask of the n_mid best list positions
        (score desc, ties by list position -- the rule of the library's selection)."""
        import numpy as np

        comb = np.max(np.asarray(all_mid, np.float32), axis=0)
        order = np.lexsort((np.arange(comb.size), -comb.astype(np.float64)))[: int(n_mid)]
        keep = np.zeros(comb.size, bool)
        keep[order[np.isfinite(comb[order])]] = True
        return keep

    def rerank(self, q, gid, pads, k: int):
        """gid / pads: the GLOBAL candidate list and its pad lengths.  -> local top-k (scores[k], ids[k]) of the candidates
        this rank owns, padded with (-inf, -1), ordered (score desc, coarse rank asc) like a single index."""
        import numpy as np
        import torch

        ls = torch.full((k,), float("-inf"), dtype=torch.float32)
        li = torch.full((k,), -1, dtype=torch.int64)
        own = np.nonzero((gid >= self.lo) & (gid < self.hi))[0]
        if own.size:
            mine = gid[own]
            sc = np.asarray(self.local_rerank(q, mine, pads[own]), np.float32)
            order = np.lexsort((own, -sc.astype(np.float64)))[:k]
            ls[: order.size] = torch.from_numpy(sc[order])
            li[: order.size] = torch.from_numpy(mine[order])
        return ls, li

    def query(self, q, k: int, coarse_n: Optional[int] = None, allow=None):
        import numpy as np
        import torch
        import torch.distributed as dist

        n, cs, ci = self.coarse(q, k, coarse_n, allow)
        # the row counts of this rank's coarse candidates ride in the same all-gather as their (score, id) pairs, so the
        # pad lengths of the GLOBAL candidate list need no collective of their own
        ids_h = ci.detach().cpu().numpy().astype(np.int64)
        rows_h = np.zeros(n, np.float64)
        if self.pad and (ids_h >= 0).any():
            rows_h[ids_h >= 0] = self.local_rows(ids_h[ids_h >= 0])
        mine3 = torch.cat([cs.reshape(-1).to(torch.float64), ci.reshape(-1).to(torch.float64),
                           torch.from_numpy(rows_h).to(cs.device)])
        if dist.is_initialized():
            world = dist.get_world_size(self.group)
            allb = torch.empty(world * 3 * n, dtype=torch.float64, device=mine3.device)
            dist.all_gather_into_tensor(allb, mine3, group=self.group)
            allb = allb.view(world, 3, n)
        else:
            allb = mine3.view(1, 3, n)
        gs, gi, gr = allb[:, 0].reshape(-1), allb[:, 1].reshape(-1), allb[:, 2].reshape(-1)
        order = torch.sort(gs, descending=True, stable=True).indices[:n]  # (score desc, id asc): ranks own ascending ids
        sel = torch.stack([gi[order], gr[order]]).cpu().numpy()  # ONE device->host copy: global top-n ids + row counts
        gid, grows = sel[0].astype(np.int64), sel[1]
        pads = self.batch_pads(gid, grows)
        if self.local_prune is not None and 0 < self.n_mid < n:  # pruning stage: n e4m3 scores per rank, one small all-gather
            mid = torch.from_numpy(self.prune_scores(q, gid, pads)).to(cs.device)
            if dist.is_initialized():
                allm = torch.empty(dist.get_world_size(self.group) * n, dtype=torch.float32, device=mid.device)
                dist.all_gather_into_tensor(allm, mid, group=self.group)
                allm = allm.view(-1, n)
            else:
                allm = mid.view(1, n)
            gid = np.where(self.prune_keep(allm.cpu().numpy(), self.n_mid), gid, -1)  # the list keeps its order and pad lengths
        ls, li = self.rerank(q, gid, pads, k)
        return allgather_topk(ls.to(cs.device), li.to(cs.device), k, self.group)


def make_gpu_two_stage(index, device=None, group=None, mode: str = "fde_then_float", k: int = 10, coarse_n: int = 0, n_q_rows: int = 32) -> TwoStageShardedSearcher:
    """Host-driven TwoStageShardedSearcher over one MvIndex shard: coarse = MV_MODE_FDE_ONLY (or, for "fp8_then_float", the
    e4m3 scan's) top-n, rerank = mv_score_candidates_pads on the pages this rank owns (exact tier: bf16 slab, else the
    pinned-host tier, else the e4m3 slab), pruning stage per the index's own plan.  The cross-check of GpuTwoStageSearcher."""
    import numpy as np
    import torch

    from . import _lib

    dev = torch.device("cuda", index.device) if device is None else device
    base = int(index.id_base)
    fde = mode == "fde_then_float"

    def coarse(q, n, allow):
        s, i = index.query(q, n, mode="fde" if fde else "float_fp8", allow=allow)
        ps = np.full(n, -np.inf, np.float32)
        pi = np.full(n, -1, np.int64)
        ps[: len(s)], pi[: len(i)] = s, i
        return torch.from_numpy(ps).to(dev), torch.from_numpy(pi).to(dev)

    def rows(gids):
        return index.page_rows(np.asarray(gids, np.int64) - base)

    def rerank(q, gids, pads):
        return index.score_candidates(q, np.asarray(gids, np.int64) - base, pads=pads)

    def prune(q, gids, pads):  # the e4m3 scores of the named pages (MV_OPT_EXACT_TIER 2), then back to the index's own tier
        prev = index.get_option(_lib.MV_OPT_EXACT_TIER, 0)
        index.set_option(_lib.MV_OPT_EXACT_TIER, 2)
        try:
            return index.score_candidates(q, np.asarray(gids, np.int64) - base, pads=pads)
        finally:
            index.set_option(_lib.MV_OPT_EXACT_TIER, prev)

    n = int(coarse_n) if coarse_n else min(10 * k, 75)
    n_mid, _tier = index.rerank_plan(n, k, n_q_rows, mode=mode)
    return TwoStageShardedSearcher(coarse, rows, rerank, (base, base + len(index)), group, pad_semantics=fde,
                                   local_prune=prune if n_mid else None, n_mid=n_mid)


class GpuTwoStageSearcher:
    """Device-resident form of TwoStageShardedSearcher for one process per GPU over RCCL: the library leaves the coarse
    candidates (16-byte records: score, rows, global id) in a cuda buffer (mv_two_stage_coarse_device), ONE
    all_gather_into_tensor moves n records per rank, (pruning stage, when the index's rerank plan has one:
    mv_two_stage_mid_device + an all-gather of n floats per rank,) the library derives the global top-n, its owned
    candidates, their per-batch pad lengths, reranks on the exact tier and selects the local top-k
    (mv_two_stage_rerank_device), a last all-gather moves the k pairs and mv_merge_topk finishes.  Everything is ordered on
    torch's current stream: no host synchronisation and no host copy of any intermediate on the query path."""

    REC = 16  # sizeof(mv_cand_rec)

    def __init__(self, index, device=None, group=None, mode: str = "fde_then_float"):
        import torch
        import torch.distributed as dist

        self.index, self.group, self.mode = index, group, mode
        self.dev = torch.device("cuda", index.device) if device is None else device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bufs = {}
        self._flip = 0

    def _buffers(self, n, k):
        import torch

        self._flip ^= 1  # two sets: collectives of query i may still read set A while query i+1 fills set B
        key = (n, k, self._flip)
        if key not in self._bufs:
            d, w = self.dev, self.world
            self._bufs[key] = (torch.empty(n * self.REC, dtype=torch.uint8, device=d), torch.empty(w * n * self.REC, dtype=torch.uint8, device=d),
                               torch.empty(k, dtype=torch.float32, device=d), torch.empty(k, dtype=torch.int64, device=d),
                               torch.empty(w * k, dtype=torch.float32, device=d), torch.empty(w * k, dtype=torch.int64, device=d),
                               torch.empty(k, dtype=torch.float32, device=d), torch.empty(k, dtype=torch.int64, device=d),
                               torch.empty(n, dtype=torch.float32, device=d), torch.empty(w * n, dtype=torch.float32, device=d))
        return self._bufs[key]

    def query(self, q, k: int, coarse_n: Optional[int] = None, allow=None, q_fde=None):
        """q_fde: the caller's own FDE vector of the query (every rank passes the same one), MvIndex.query."""
        import ctypes as C

        import numpy as np
        import torch
        import torch.distributed as dist

        from ._lib import check, lib

        n = int(coarse_n) if coarse_n else min(10 * k, 75)
        recs, allrecs, ls, li, gs, gi, os_, oi, mid, allmid = self._buffers(n, k)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        n_q = int(np.shape(q)[0]) if np.ndim(q) == 2 else 1
        n_mid, _tier = self.index.rerank_plan(n, k, n_q, mode=self.mode)  # host-side rule, identical on every rank
        self.index.two_stage_coarse_device(q, n, recs.data_ptr(), allow=allow, stream=stream, mode=self.mode, **({} if q_fde is None else {"q_fde": q_fde}))
        if dist.is_initialized():
            dist.all_gather_into_tensor(allrecs, recs, group=self.group)
        else:
            allrecs = recs
        if n_mid:
            self.index.two_stage_mid_device(q, allrecs.data_ptr(), self.world, n, mid.data_ptr(), stream=stream, mode=self.mode)
            if dist.is_initialized():
                dist.all_gather_into_tensor(allmid, mid, group=self.group)
            else:
                allmid = mid
        self.index.two_stage_rerank_device(q, allrecs.data_ptr(), self.world, n, k, ls.data_ptr(), li.data_ptr(), stream=stream, mode=self.mode,
                                           d_all_mid_ptr=allmid.data_ptr() if n_mid else 0, n_mid=n_mid)
        if dist.is_initialized():
            dist.all_gather_into_tensor(gs, ls, group=self.group)
            dist.all_gather_into_tensor(gi, li, group=self.group)
        else:
            gs, gi = ls, li
        check(lib().mv_merge_topk(self.index.device, C.c_void_p(gs.data_ptr()), C.c_void_p(gi.data_ptr()), self.world, k, k,
                                  C.c_void_p(os_.data_ptr()), C.c_void_p(oi.data_ptr()), C.c_void_p(stream)))
        return os_, oi
