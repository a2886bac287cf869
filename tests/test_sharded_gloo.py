"""N>1 path on CPU: world_size-2 (and 3) gloo process groups; each rank owns a row shard served by an
oracle-backed index; the all-gather + merge must equal the single-index top-k including ties."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from morphik_core_amd import sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _corpus(n, rows=8):
    from oracle import oracle as orc

    base = [orc.synth_rows(5, i, 0, rows) for i in range(12)]
    # duplicates -> exact score ties across shards
    return [base[i % 12] if i % 3 else orc.synth_rows(6, i, 0, rows) for i in range(n)]


def _worker(rank, world, port, n_total, k, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    lo, hi = sharded.shard_range(n_total, rank, world)
    pages = _corpus(n_total)[lo:hi]
    ix = OracleIndex(capacity_pages=max(hi - lo, 1), stride_rows=8, id_base=lo)
    ix.add(pages)

    def local_topk(q, kk):
        s, i = ix.query(q, kk)
        ps = np.full(kk, -np.inf, np.float32)
        pi = np.full(kk, -1, np.int64)
        ps[: len(s)] = s
        pi[: len(i)] = i
        return torch.from_numpy(ps), torch.from_numpy(pi)

    searcher = sharded.ShardedSearcher(local_topk)
    q = orc.synth_rows(4321, 0, 0, 6)
    s, i = searcher.query(q, k)

    def local_topk_batch(qs, kk):  # the batched form: [B, kk] per shard, one all-gather for all queries
        rows = [local_topk(qq, kk) for qq in qs]
        return torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])

    qs = [orc.synth_rows(4321, j, 0, 6) for j in range(3)]
    batch = searcher.query_batch(qs, k, local_topk_batch)
    assert batch[0][1].tolist() == i.tolist() and batch[0][0].tolist() == s.tolist()
    for qq, (bs, bi) in zip(qs, batch):
        ss, si = searcher.query(qq, k)
        assert bi.tolist() == si.tolist() and bs.tolist() == ss.tolist()
    # the host-memory step bench.py runs under gloo (ONE packed all-gather into a preallocated buffer, numpy merge): the same answer,
    # padded to exactly k entries with (-inf, -1)
    host = sharded.HostShardedSearcher(ix, mode="float")
    for qq in [q] + qs:
        hs, hi_ = host.query(qq, k)
        ss, si = searcher.query(qq, k)
        assert hs.shape == (k,) and hi_.shape == (k,)
        keep = hi_ >= 0
        assert hi_[keep].tolist() == si.tolist() and hs[keep].tolist() == ss.tolist()
        assert np.isneginf(hs[~keep]).all() and (hi_[~keep] == -1).all()
    out_q.put((rank, s.numpy().tolist(), i.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,k", [(2, 41, 10), (3, 50, 7), (2, 3, 10)])
def test_sharded_topk_equals_single_index(world, n_total, k):
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, k, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = OracleIndex(capacity_pages=n_total, stride_rows=8)
    one.add(_corpus(n_total))
    ws, wi = one.query(orc.synth_rows(4321, 0, 0, 6), k)
    for _rank, s, i in results:  # every rank holds the same merged answer
        assert i == wi.tolist() and s == ws.tolist()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 1000, 1_000_000):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_merge_topk_tie_rule():
    s = torch.tensor([[3.0, 2.0, float("-inf")], [3.0, 3.0, 1.0]])
    i = torch.tensor([[4, 1, -1], [10, 12, 11]])
    ms, mi = sharded.merge_topk(s, i, 4)
    assert mi.tolist() == [4, 10, 12, 1] and ms.tolist() == [3.0, 3.0, 3.0, 2.0]
    ms, mi = sharded.merge_topk(s, i, 10)
    assert mi.tolist() == [4, 10, 12, 1, 11]


# --------------------------------------------------------------------------- config 4: FDE coarse -> exact rerank, sharded
def _ragged_corpus(n):
    from oracle import oracle as orc

    return [orc.synth_rows(7, i, 0, 3 + (i * 5) % 9) for i in range(n)]  # 3..11 rows: the pad-to-longest rule matters


def _fde_cfg():
    from oracle import oracle as orc

    return orc.FdeConfig(128, 4, 3, 8, 1)


def _two_stage(ix, lo):
    def coarse(q, n, allow):
        s, i = ix.query(q, n, mode="fde", allow=allow)
        ok = np.isfinite(s)
        ps = np.full(n, -np.inf, np.float32)
        pi = np.full(n, -1, np.int64)
        ps[: ok.sum()], pi[: ok.sum()] = s[ok], i[ok]
        return torch.from_numpy(ps), torch.from_numpy(pi)

    return sharded.TwoStageShardedSearcher(coarse, lambda g: ix.page_rows(np.asarray(g) - lo),
                                           lambda q, g, pads: ix.score_candidates(q, np.asarray(g) - lo, pads=pads),
                                           (lo, lo + len(ix)))


def _worker_two_stage(rank, world, port, n_total, k, coarse_n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    lo, hi = sharded.shard_range(n_total, rank, world)
    ix = OracleIndex(capacity_pages=max(hi - lo, 1), stride_rows=16, id_base=lo, fde=_fde_cfg())
    ix.add(_ragged_corpus(n_total)[lo:hi], doc_ordinals=[(lo + j) % 5 for j in range(hi - lo)])
    searcher = _two_stage(ix, lo)
    res = []
    for j, allow in enumerate([None, np.array([0b01101], np.uint32)]):
        s, i = searcher.query(orc.synth_rows(4321, j, 0, 6), k, coarse_n=coarse_n, allow=allow)
        res.append((s.numpy().tolist(), i.numpy().tolist()))
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,k,coarse_n", [(2, 37, 5, 12), (3, 40, 4, 9), (2, 2, 5, 12), (2, 330, 6, 300)])
def test_two_stage_fde_pipeline_is_rank_count_invariant(world, n_total, k, coarse_n):
    """The sharded FDE_THEN_FLOAT (global coarse top-n, owners rerank with the pad length of each candidate's batch of
    128 in the GLOBAL list) returns exactly what ONE index returns: same candidate set, same pad-to-longest clamp per
    batch, same top-k.  The 330-page / 300-candidate case spans three rerank batches."""
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_stage, args=(r, world, port, n_total, k, coarse_n, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = OracleIndex(capacity_pages=n_total, stride_rows=16, fde=_fde_cfg())
    one.add(_ragged_corpus(n_total), doc_ordinals=[j % 5 for j in range(n_total)])
    for j, allow in enumerate([None, np.array([0b01101], np.uint32)]):
        ws, wi = one.query(orc.synth_rows(4321, j, 0, 6), k, mode="fde_then_float", allow=allow, coarse_n=coarse_n)
        for _rank, res in results:
            assert res[j][1] == wi.tolist() and res[j][0] == ws.tolist()
    # the same class without a process group == the single index as well
    solo = _two_stage(one, 0)
    s, i = solo.query(orc.synth_rows(4321, 0, 0, 6), k, coarse_n=coarse_n)
    ws, wi = one.query(orc.synth_rows(4321, 0, 0, 6), k, mode="fde_then_float", coarse_n=coarse_n)
    assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()


# --------------------------------------------------------------------------- the pruning stage (exact tier in host RAM behind a long coarse list)
def _fp8_scores(ix, q, local_pages, pads):
    """e4m3 MaxSim (oracle quantiser + oracle scan) of the named local pages of an OracleIndex: the pruning stage's scores."""
    from oracle import oracle as orc

    q = np.asarray(q)
    qf = orc.bf16_to_f32(q) if q.dtype == np.uint16 else np.asarray(q, np.float32)
    out = []
    for p, pad in zip(local_pages, pads):
        rows = orc.f32_to_bf16(ix.pages[int(p)])
        codes, inv = orc.quantize_page_fp8(rows, 16)
        out.append(orc.maxsim_fp8(qf, codes, rows.shape[0], float(inv), int(pad)))
    return np.array(out, np.float32)


def _two_stage_pruned(ix, lo, n_mid):
    base = _two_stage(ix, lo)
    return sharded.TwoStageShardedSearcher(base.local_coarse, base.local_rows, base.local_rerank, (lo, lo + len(ix)),
                                           local_prune=lambda q, g, pads: _fp8_scores(ix, q, np.asarray(g) - lo, pads), n_mid=n_mid)


def _worker_pruned(rank, world, port, n_total, k, coarse_n, n_mid, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    lo, hi = sharded.shard_range(n_total, rank, world)
    ix = OracleIndex(capacity_pages=max(hi - lo, 1), stride_rows=16, id_base=lo, fde=_fde_cfg())
    ix.add(_ragged_corpus(n_total)[lo:hi])
    searcher = _two_stage_pruned(ix, lo, n_mid)
    res = []
    for j in range(3):
        s, i = searcher.query(orc.synth_rows(4321, j, 0, 6), k, coarse_n=coarse_n)
        res.append((s.numpy().tolist(), i.numpy().tolist()))
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,k,coarse_n,n_mid", [(2, 90, 5, 60, 16), (3, 200, 6, 150, 40)])
def test_two_stage_with_the_pruning_stage_is_rank_count_invariant(world, n_total, k, coarse_n, n_mid):
    """configs[3] on shards whose exact tier is host memory: between the coarse exchange and the exact rerank every rank scores
    the entries of the GLOBAL list it owns on its e4m3 slab, the scores are all-gathered, the n_mid best list POSITIONS stay
    (ties by position).  Over gloo, world 2 and 3, the answer is the one a single index composes: coarse top-n -> batch pad
    lengths over the WHOLE list -> pruning -> exact rerank -> top-k (the device form is in test_gpu_exact_tier.py)."""
    from oracle import oracle as orc
    from tests.fake_index import OracleIndex

    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pruned, args=(r, world, port, n_total, k, coarse_n, n_mid, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out_q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = OracleIndex(capacity_pages=n_total, stride_rows=16, fde=_fde_cfg())
    one.add(_ragged_corpus(n_total))
    differs = 0
    for j in range(3):
        q = orc.synth_rows(4321, j, 0, 6)
        cs, ci = orc.topk(one.score_all(q, "fde"), coarse_n)
        ci = ci[np.isfinite(cs)]
        rows = one.page_rows(ci)
        pads = np.concatenate([np.full(len(rows[b : b + 128]), rows[b : b + 128].max()) for b in range(0, len(rows), 128)])
        f8 = _fp8_scores(one, q, ci, pads)
        keep = np.lexsort((np.arange(ci.size), -f8.astype(np.float64)))[:n_mid]
        exact = np.full(ci.size, -np.inf, np.float32)
        exact[keep] = one.score_candidates(q, ci[keep], pads=pads[keep])
        order = np.lexsort((np.arange(ci.size), -exact.astype(np.float64)))[:k]
        ws, wi = exact[order], ci[order]
        for _rank, res in results:
            assert res[j][1] == wi.tolist() and res[j][0] == ws.tolist()
        us, ui = one.query(q, k, mode="fde_then_float", coarse_n=coarse_n)
        differs += int(ui.tolist() != wi.tolist())
    # without a process group the class composes the same answer
    solo = _two_stage_pruned(one, 0, n_mid)
    s, i = solo.query(orc.synth_rows(4321, 0, 0, 6), k, coarse_n=coarse_n)
    assert i.tolist() == results[0][1][0][1] and s.tolist() == results[0][1][0][0]
