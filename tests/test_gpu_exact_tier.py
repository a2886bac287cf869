"""GPU tests of the exact rerank at the shard shape of BASELINE configs[3] / [4]: an index WITHOUT a bf16
slab (FDE + e4m3 slabs in HBM) keeps its exact bf16 rows in pinned host memory, and MV_MODE_FDE_THEN_FLOAT /
MV_MODE_FP8_THEN_FLOAT rerank on THAT tier -- the reference reranks with exact fp32 MaxSim on fp32 pages
(core/vector_store/fast_multivector_store.py:553-556, upcast at load :736,774), not on a quantised copy.

Checked: against the float oracle (orc.maxsim_f32 on the bf16 rows, NOT the fp8 oracle) within north_star's 1e-3, against an
index that holds the bf16 slab in HBM (bit-identical scores: same kernel, same rows), single / batched / R logical shards
through mv_comm (peer-copy and host transports) / the device-resident stages of the one-process-per-GPU flow.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)

RTOL = 1e-3  # north_star tolerance for float MaxSim


def _idx(**kw):
    from morphik_core_amd.index import MvIndex

    return MvIndex(**kw)


def _batch_pads(rows, batch=128):
    rows = np.asarray(rows)
    pads = np.empty_like(rows)
    for j in range(0, len(rows), batch):
        pads[j : j + batch] = rows[j : j + batch].max()
    return pads


def _corpus(n, stride, seed=31):
    """Ragged pages; every seventh page nearly repeats page 3 (a cluster of near-ties the e4m3 noise reorders)."""
    pages = [orc.synth_rows(seed, i, 0, 6 + (i * 7) % (stride - 6)) for i in range(n)]
    base = orc.bf16_to_f32(orc.synth_rows(seed, 3, 0, stride))
    rng = np.random.default_rng(seed)
    for i in range(10, n, 7):
        x = base[: pages[i].shape[0]] + 0.002 * rng.standard_normal((pages[i].shape[0], 128)).astype(np.float32)
        pages[i] = orc.f32_to_bf16(x / np.linalg.norm(x, axis=1, keepdims=True))
    return pages


def _oracle_cascade(ix, q, pages, k, coarse_n, n_mid, allow=None):
    """The reference pipeline composed on the host from the library's own coarse scores: coarse top-n (coarse rank order) -> per
    batch-of-128 pad lengths -> [e4m3 pruning to n_mid list positions, the device's own e4m3 scores] -> EXACT float MaxSim
    (oracle, fp32 on the bf16 rows) -> top-k by (score desc, list position asc)."""
    from morphik_core_amd import _lib

    qf = orc.bf16_to_f32(q)
    coarse = ix.score_all(q, mode="fde", allow=allow)
    cs, ci = orc.topk(coarse, coarse_n)
    ci = ci[np.isfinite(cs)]
    rows = np.array([pages[c].shape[0] for c in ci])
    pads = _batch_pads(rows)
    keep = np.ones(ci.size, bool)
    if n_mid and ci.size > n_mid:
        prev = ix.get_option(_lib.MV_OPT_EXACT_TIER, 0)
        ix.set_option(_lib.MV_OPT_EXACT_TIER, 2)  # the e4m3 scores of the named pages (checked against the fp8 oracle elsewhere)
        f8 = ix.score_candidates(q, ci, pads=pads)
        ix.set_option(_lib.MV_OPT_EXACT_TIER, prev)
        sel = np.lexsort((np.arange(ci.size), -f8.astype(np.float64)))[:n_mid]
        keep[:] = False
        keep[sel] = True
    exact = np.array([orc.maxsim_f32(qf, orc.bf16_to_f32(pages[c]), int(p)) if kp else -np.inf for c, p, kp in zip(ci, pads, keep)], np.float32)
    order = np.lexsort((np.arange(ci.size), -exact.astype(np.float64)))[:k]
    order = order[np.isfinite(exact[order])]
    return exact[order], ci[order]


@pytest.mark.parametrize("coarse_n,rerank_n", [(75, 128), (300, 64), (300, 1024)])
def test_fde_then_float_reranks_exactly_from_the_pinned_host_tier(coarse_n, rerank_n):
    """configs[3] at its shard shape: no bf16 slab; coarse top-n -> (n > MV_OPT_RERANK_N: e4m3 pruning) -> exact bf16 MaxSim read
    from pinned host memory -> top-k.  Scores within 1e-3 of the FLOAT oracle, identical to an index with the bf16 slab in HBM."""
    from morphik_core_amd import MvError, _lib
    from morphik_core_amd.index import allow_bitmap

    N, stride, k = 900, 48, 10
    pages = _corpus(N, stride)
    ords = [i % 13 for i in range(N)]
    host = _idx(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    hbm = _idx(capacity_pages=N, stride_rows=stride, with_float=True, with_fde=True)
    f8 = _idx(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True)  # no exact tier at all
    for ix in (host, hbm, f8):
        ix.add(pages, doc_ordinals=ords)
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        ix.set_option(_lib.MV_OPT_RERANK_N, rerank_n)
    n_mid, tier = host.rerank_plan(coarse_n, k, 20)
    assert tier == "host" and n_mid == (max(rerank_n, k) if coarse_n > max(rerank_n, k) else 0)
    assert hbm.rerank_plan(coarse_n, k, 20) == (0, "hbm") and f8.rerank_plan(coarse_n, k, 20) == (0, "fp8")
    host.remove_page(24)
    hbm.remove_page(24)
    f8.remove_page(24)
    near = orc.synth_rows(31, 3, 0, 20)  # a query inside the near-tie cluster: the e4m3 rerank reorders it, the exact one must not
    qs = [near] + [orc.synth_rows(4321, j, 0, 14 + 3 * j) for j in range(4)]
    differs = 0
    for j, q in enumerate(qs):
        for al in (None, allow_bitmap([0, 1, 2, 3, 5, 7, 8, 11, 12])):
            ws, wi = _oracle_cascade(host, q, pages, k, coarse_n, n_mid, allow=al)
            s, i, st = host.query(q, k, mode="fde_then_float", allow=al, want_stats=True)
            assert i.tolist() == wi.tolist(), (j, i.tolist(), wi.tolist())
            np.testing.assert_allclose(s, ws, rtol=RTOL, atol=1e-6)
            assert st.rerank_ms > 0 and st.coarse_ms > 0
            hs, hi = hbm.query(q, k, mode="fde_then_float", allow=al)  # the bf16 slab in HBM: same rows, same kernel, no pruning stage
            if n_mid == 0 or j > 0:  # (inside the near-tie cluster a cut at 64 of ~130 near-identical e4m3 scores is arbitrary)
                assert hi.tolist() == i.tolist() and hs.tolist() == s.tolist()
            fs, fi = f8.query(q, k, mode="fde_then_float", allow=al)  # the e4m3 rerank: close, not exact
            np.testing.assert_allclose(fs, s, rtol=3e-2)
            differs += int(fi.tolist() != i.tolist() or np.max(np.abs(fs - s) / np.abs(s)) > RTOL)
    assert differs > 0  # the corpus does separate the two reranks
    # a batch of requests: the batched pipeline (one FDE pass, every list pruned / reranked in one launch) == the lone calls
    for (bs, bi), q in zip(host.query_batch(qs, k, mode="fde_then_float"), qs):
        s, i = host.query(q, k, mode="fde_then_float")
        assert bi.tolist() == i.tolist()
        np.testing.assert_allclose(bs, s, rtol=1e-5)
    # mv_score_candidates names pages explicitly: exact tier too
    cand = np.arange(0, 200, 3)
    got = host.score_candidates(qs[1], cand, pad_to=-1)
    assert got.tolist() == hbm.score_candidates(qs[1], cand, pad_to=-1).tolist()
    # an index with an FDE slab only has nothing to rerank on: loud
    bare = _idx(capacity_pages=8, stride_rows=stride, with_float=False, with_fde=True)
    bare.add(pages[:8])
    with pytest.raises(MvError):
        bare.query(qs[0], 3, mode="fde_then_float")
    for ix in (host, hbm, f8, bare):
        ix.close()


@pytest.mark.parametrize("transport", ["p2p", "host"])
@pytest.mark.parametrize("mode,coarse_n,rerank_n", [("fde_then_float", 75, 128), ("fde_then_float", 300, 64), ("fp8_then_float", 0, 96)])
def test_sharded_exact_tier_equals_single_index(transport, mode, coarse_n, rerank_n):
    """configs[3] / [4] on a sharded corpus: R = 1, 2, 4 logical shards, every shard with ITS pinned-host exact tier, through
    mv_comm (single requests and batches) == ONE index holding every page -- ids and scores, near-ties and duplicates
    included: the candidate list (FDE top-n / e4m3 top-n) and the pruning cut are GLOBAL, whatever the shard count."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import ShardComm, allow_bitmap

    N, stride, k = 720, 48, 8
    pages = _corpus(N, stride, seed=32)
    pages[N - 5] = pages[17].copy()  # exact duplicates across shard boundaries: ties resolve by id, as on one index
    pages[400] = pages[17].copy()
    ords = [i % 11 for i in range(N)]
    kw = dict(stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)

    def opts(ix):
        if coarse_n:
            ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        ix.set_option(_lib.MV_OPT_RERANK_N, rerank_n)

    one = _idx(capacity_pages=N, **kw)
    one.add(pages, doc_ordinals=ords)
    opts(one)
    one.remove_page(33)
    qs = [orc.synth_rows(32, 3, 0, 20), orc.bf16_to_f32(pages[17][:12])] + [orc.synth_rows(4321, 40 + j, 0, 10 + (j * 5) % 23) for j in range(34)]  # 36 > one group of 32
    al = allow_bitmap([0, 1, 3, 4, 6, 7, 9, 10])
    per_req = [None if j % 3 == 0 else allow_bitmap([(j + d) % 11 for d in range(6)]) for j in range(len(qs))]
    want = [one.query(q, k, mode=mode) for q in qs[:6]]
    want_al = [one.query(q, k, mode=mode, allow=al) for q in qs[:6]]
    if mode == "fp8_then_float":  # (a batch's first stage defaults to ONE e4m3 term per query row; with two it nominates the lone call's candidates)
        one.set_option(_lib.MV_OPT_BATCH_VARIANT, 0)
    for (bs, bi), (s, i) in zip(one.query_batch(qs[:6], k, mode=mode), want):  # the batched pipeline agrees with the lone calls on one index
        assert bi.tolist() == i.tolist()
    one.set_option(_lib.MV_OPT_BATCH_VARIANT, -1)
    want_b = one.query_batch(qs, k, mode=mode)
    want_bp = one.query_batch(qs, k, mode=mode, allows=per_req, n_docs=11)
    for R in (1, 2, 4):
        per = N // R
        shards = []
        for r in range(R):
            sh = _idx(capacity_pages=per, id_base=r * per, **kw)
            sh.add(pages[r * per : (r + 1) * per], doc_ordinals=ords[r * per : (r + 1) * per])
            opts(sh)
            shards.append(sh)
        shards[33 // per].remove_page(33 % per)
        comm = ShardComm(shards, transport=transport)
        for q, (ws, wi), (was, wai) in zip(qs[:6], want, want_al):
            s, i, st = comm.query(q, k, mode=mode, want_stats=True)
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (R, transport)
            assert len(st) == R and all(x.total_device_ms > 0 for x in st)
            s, i = comm.query(q, k, mode=mode, allow=al)
            assert i.tolist() == wai.tolist() and s.tolist() == was.tolist(), (R, transport)
        for wantb, kwq in ((want_b, {}), (want_bp, dict(allows=per_req, n_docs=11))):
            got = comm.query_batch(qs, k, mode=mode, **kwq)
            for j, ((s, i), (ws, wi)) in enumerate(zip(got, wantb)):
                assert i.tolist() == wi.tolist(), (R, transport, j)
                assert s.tolist() == ws.tolist(), (R, transport, j)
        comm.close()
        for sh in shards:
            sh.close()
    one.close()


@pytest.mark.parametrize("mode,coarse_n,rerank_n", [("fde_then_float", 300, 64), ("fp8_then_float", 96, 96)])
def test_device_resident_and_host_driven_stages_with_the_host_tier(mode, coarse_n, rerank_n):
    """The stages of the one-process-per-GPU flow (mv_two_stage_coarse / mid / rerank_device; sharded.GpuTwoStageSearcher and
    the host-driven TwoStageShardedSearcher) over R logical shards with pinned-host exact tiers == the single index."""
    import torch

    from morphik_core_amd import _lib, sharded

    N, stride, k = 480, 48, 6
    pages = _corpus(N, stride, seed=33)
    kw = dict(stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)

    def opts(ix):
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        ix.set_option(_lib.MV_OPT_RERANK_N, rerank_n)

    one = _idx(capacity_pages=N, **kw)
    one.add(pages)
    opts(one)
    qs = [orc.synth_rows(33, 3, 0, 20)] + [orc.synth_rows(4321, 70 + j, 0, 20) for j in range(2)]
    dev = torch.device("cuda", 0)
    n_mid, tier = one.rerank_plan(coarse_n, k, 20, mode=mode)
    assert tier == "host" and (n_mid == 64 if mode == "fde_then_float" else n_mid == 0)
    for R in (1, 2, 4):
        per = N // R
        shards, searchers = [], []
        for r in range(R):
            sh = _idx(capacity_pages=per, id_base=r * per, **kw)
            sh.add(pages[r * per : (r + 1) * per])
            opts(sh)
            shards.append(sh)
            searchers.append(sharded.make_gpu_two_stage(sh, mode=mode, k=k, coarse_n=coarse_n, n_q_rows=20))
        for q in qs:
            ws, wi = one.query(q, k, mode=mode)
            # --- host-driven stages (the gloo-testable form), the collectives replaced by concatenation in shard order
            co = [se.coarse(q, k, coarse_n, None) for se in searchers]
            gs = torch.stack([c[1] for c in co]).reshape(-1)
            gi = torch.stack([c[2] for c in co]).reshape(-1)
            order = torch.sort(gs, descending=True, stable=True).indices[:coarse_n]
            gid = gi[order].cpu().numpy().astype(np.int64)
            grows = np.array([pages[g].shape[0] if g >= 0 else 0 for g in gid])
            pads = searchers[0].batch_pads(gid, grows)
            if n_mid:
                allm = np.stack([se.prune_scores(q, gid, pads) for se in searchers])
                gid = np.where(sharded.TwoStageShardedSearcher.prune_keep(allm, n_mid), gid, -1)
            loc = [se.rerank(q, gid, pads, k) for se in searchers]
            ms, mi = sharded.merge_topk(torch.stack([l[0] for l in loc]), torch.stack([l[1] for l in loc]), k)
            assert mi.tolist() == wi.tolist() and ms.tolist() == ws.tolist(), (R, "host-driven")
            # --- device-resident stages
            recs = [torch.empty(coarse_n * 16, dtype=torch.uint8, device=dev) for _ in range(R)]
            for sh, rb in zip(shards, recs):
                sh.two_stage_coarse_device(q, coarse_n, rb.data_ptr(), mode=mode)
            allrecs = torch.cat(recs)
            allmid = None
            if n_mid:
                mids = [torch.empty(coarse_n, dtype=torch.float32, device=dev) for _ in range(R)]
                for sh, mb in zip(shards, mids):
                    sh.two_stage_mid_device(q, allrecs.data_ptr(), R, coarse_n, mb.data_ptr(), mode=mode)
                allmid = torch.cat(mids)
                owners = (torch.isfinite(allmid.view(R, coarse_n)).sum(0)).cpu().numpy()
                assert (owners <= 1).all() and owners.sum() > n_mid  # one owner per list position
            ls = torch.empty((R, k), dtype=torch.float32, device=dev)
            li = torch.empty((R, k), dtype=torch.int64, device=dev)
            for r, sh in enumerate(shards):
                sh.two_stage_rerank_device(q, allrecs.data_ptr(), R, coarse_n, k, ls[r].data_ptr(), li[r].data_ptr(), mode=mode,
                                           d_all_mid_ptr=allmid.data_ptr() if n_mid else 0, n_mid=n_mid)
            ds, di = sharded.merge_topk(ls.cpu(), li.cpu(), k)
            assert di.tolist() == wi.tolist() and ds.tolist() == ws.tolist(), (R, "device-resident")
        if R == 1:  # GpuTwoStageSearcher without a process group: the stream-ordered pipeline, pruning stage included
            se = sharded.GpuTwoStageSearcher(shards[0], mode=mode)
            for q in qs:
                ws, wi = one.query(q, k, mode=mode)
                s, i = se.query(q, k, coarse_n=coarse_n)
                torch.cuda.synchronize()
                assert i.cpu().tolist() == wi.tolist() and s.cpu().tolist() == ws.tolist()
        for sh in shards:
            sh.close()
    one.close()


def test_sharded_stores_with_host_exact_tiers_behind_the_plugin_surface():
    """create_store("mi355x_sharded_fast_host_exact") / ("mi355x_sharded_fp8_exact"): the BaseVectorStore surface over R logical
    shards whose exact rows live in pinned host memory; query_similar returns the exact scores (float oracle, 1e-3)."""
    import asyncio

    from morphik_core_amd.models import DocumentChunk
    from morphik_core_amd.store import create_store

    stride = 48
    pages = _corpus(120, stride, seed=34)
    chunks = [DocumentChunk(document_id=f"d{i // 4}", content=f"p{i}", embedding=orc.bf16_to_f32(p), chunk_number=i % 4, metadata={}) for i, p in enumerate(pages)]
    q = orc.bf16_to_f32(orc.synth_rows(34, 3, 0, 20))
    exact = np.array([orc.maxsim_f32(q, orc.bf16_to_f32(p), 0) for p in pages], np.float32)
    for provider in ("mi355x_sharded_fast_host_exact", "mi355x_sharded_fp8_exact", "mi355x_fast_host_exact"):
        kw = dict(devices=[0, 0, 0], transport="p2p") if "sharded" in provider else {}
        st = create_store(provider, capacity_pages=300, stride_rows=stride, **kw)
        assert st.initialize()

        async def run():
            ok, ids, _m = await st.store_embeddings(chunks[:60], app_id=None)
            assert ok and len(ids) == 60
            ok, ids, _m = await st.store_embeddings(chunks[60:], app_id=None)
            assert ok
            return await st.query_similar(q, k=5)

        hits = asyncio.run(run())
        assert len(hits) == 5
        got = {(h.document_id, h.chunk_number): h.score for h in hits}
        for (doc, cn), sc in got.items():
            p = int(doc[1:]) * 4 + cn
            if "fast" in provider:  # FDE pipeline: the reference pads every rerank batch to its longest page (clamp at 0)
                assert sc >= exact[p] - 1e-3 * abs(exact[p])
            else:
                assert abs(sc - exact[p]) <= RTOL * abs(exact[p])
        if "fast" not in provider:  # exhaustive e4m3 scan + exact re-score: the exact top-5
            top = np.lexsort((np.arange(len(pages)), -exact.astype(np.float64)))[:5]
            assert [int(h.document_id[1:]) * 4 + h.chunk_number for h in hits] == top.tolist()
        st.close()


def test_host_exact_tier_beyond_the_memory_budget_is_refused_before_pinning(monkeypatch):
    """A container over its memory cgroup limit is KILLED in the middle of hipHostMalloc (measured on the MI355X pool: memory.max
    300 GiB on a 3 TiB host; the 328 GB tier of a 1.25 M-page shard took the box down twice): mv_index_create reads the limits
    and refuses up front, loudly."""
    from morphik_core_amd import MvError, _lib

    assert _lib.lib().mv_host_pin_budget_bytes() > 0
    monkeypatch.setenv("MV_HOST_EXACT_MAX_BYTES", str(64 << 20))
    assert _lib.lib().mv_host_pin_budget_bytes() <= 64 << 20
    with pytest.raises(MvError) as e:
        _idx(capacity_pages=1000, stride_rows=1024, with_float=False, with_fp8=True, with_host_exact=True)  # 262 MB > 64 MiB
    assert "may pin only" in str(e.value)
    ix = _idx(capacity_pages=100, stride_rows=1024, with_float=False, with_fp8=True, with_host_exact=True)  # 26 MB: fine
    ix.close()


@pytest.mark.parametrize("split", [False, True])
def test_exact_pipelines_at_shard_scale_return_float_oracle_scores(split):
    """configs[3] / [4] at (as much as this box allows of) their per-GPU shard shape: FDE + e4m3 slabs in HBM, the exact bf16 rows
    of EVERY page in pinned host memory -- or (split) in the free HBM first and pinned host memory for the rest, which is what
    lets the FULL 1.25 M-page shard run inside the pool's 300 GiB containers.  A 1.25 M-page shard needs 328 GB pinned; the page count is cut to HALF of what the
    process may still pin (mv_host_pin_budget_bytes: the container's memory cgroup limit, 300 GiB on the MI355X pool) and to the
    free HBM.  The oracle cannot scan a corpus of this size, so the test checks size-independent properties: the planted top-10
    comes back in rank order from every exact pipeline, single and batched, with scores within 1e-3 of orc.maxsim_bf16 (the
    FLOAT oracle) on the rows read back from the tier -- which is also where the kernels read them."""
    import torch

    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import synth_rows

    patches, qt, k = 1024, 32, 10
    budget = int(_lib.lib().mv_host_pin_budget_bytes())
    free_b, _tot = torch.cuda.mem_get_info(0)
    page_b, slab_b = patches * 256, patches * 128 + 20480 + 64 + 33 * 4
    if not split:  # every exact row pinned: 300 k pages = 79 GB (the FULL 1.25 M-page shard is the split case below)
        n = int(min(300_000, 0.5 * budget // page_b, (free_b - (16 << 30)) // slab_b))
    else:
        # BASELINE configs[3]'s FULL shard: 1.25 M pages.  The FDE + e4m3 slabs take 190 GB of HBM, the exact rows of the leading
        # pages fill what is left (minus the library's 12 GiB reserve), the rest is pinned -- at most 0.78 of the pin budget here
        n = int(min(1_250_000, (free_b - (16 << 30)) // slab_b))
        while n > 20_000 and (n - max(0, (free_b - n * slab_b - (13 << 30)) // page_b)) * page_b > 0.78 * budget:
            n -= 10_000
    assert n >= 20_000, f"only {n} pages fit (pin budget {budget / 1e9:.0f} GB, free HBM {free_b / 1e9:.0f} GB)"
    ix = _idx(capacity_pages=n, stride_rows=patches, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True, with_exact_split=split)
    if split:
        assert 0 < ix.exact_hbm_pages < n and (n - ix.exact_hbm_pages) * page_b <= 0.8 * budget, (n, ix.exact_hbm_pages)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n, n_rows=patches)
    qs = [synth_rows(synth.SEED_QUERIES, qi, qt) for qi in range(6)]
    spec = synth.planted_spec(qs, n, patches, n_ranks=k)
    synth.plant_neighbours_any(ix, spec, synth.SEED_CORPUS, patches, 0, n)
    planted = [[p for (qq, _r, p, _a, _b) in spec if qq == qi] for qi in range(len(qs))]
    print(f"shard-scale exact tier: {n} pages, {ix.exact_hbm_pages * page_b / 1e9:.0f} GB of exact rows in HBM, {(n - ix.exact_hbm_pages) * page_b / 1e9:.0f} GB pinned "
          f"(budget {budget / 1e9:.0f} GB)")

    def check(s, i, qi):
        assert i.tolist() == planted[qi], (qi, i.tolist(), planted[qi])
        want = np.array([orc.maxsim_bf16(qs[qi], ix.read_pages(p, 1)[0]) for p in planted[qi]], np.float32)
        np.testing.assert_allclose(s, want, rtol=RTOL)

    for coarse_n in (75, 1000):  # 75: the reference's min(10 k, 75), straight to the exact tier; 1000: through the e4m3 pruning stage
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        assert ix.rerank_plan(coarse_n, k, qt) == ((128 if coarse_n > 128 else 0), "host")
        for qi, q in enumerate(qs):
            s, i, st = ix.query(q, k, mode="fde_then_float", want_stats=True)
            check(s, i, qi)
            assert st.rerank_ms > 0
        for qi, (s, i) in enumerate(ix.query_batch(qs, k, mode="fde_then_float")):
            check(s, i, qi)
    for qi, q in enumerate(qs):
        s, i = ix.query(q, k, mode="fp8_then_float")
        check(s, i, qi)
    for qi, (s, i) in enumerate(ix.query_batch(qs, k, mode="fp8_then_float")):
        check(s, i, qi)
    ix.close()


# ------------------------------------------------------------------ the SPLIT exact tier (MV_WITH_EXACT_SPLIT)
class _SplitAt:
    """MV_EXACT_HBM_MAX_PAGES for the creation of a small index: forces the split where a real shard's free HBM would put it."""

    def __init__(self, pages):
        self.pages = pages

    def __enter__(self):
        import os

        self.prev = os.environ.get("MV_EXACT_HBM_MAX_PAGES")
        os.environ["MV_EXACT_HBM_MAX_PAGES"] = str(self.pages)

    def __exit__(self, *a):
        import os

        if self.prev is None:
            os.environ.pop("MV_EXACT_HBM_MAX_PAGES", None)
        else:
            os.environ["MV_EXACT_HBM_MAX_PAGES"] = self.prev


@pytest.mark.parametrize("split_at", [0, 137, 10_000])
def test_split_exact_tier_answers_like_the_unsplit_tier(split_at):
    """The exact tier split between HBM (pages [0, split)) and pinned host memory (the rest): every rerank entry point returns the
    SAME scores and ids, bit for bit, as the unsplit host tier -- single queries (with and without the e4m3 pruning stage, with an
    allow-list), batches (one-launch rerank and the per-query form of long queries), named candidates, the staged device entry
    points.  split 0 = all host, 10 000 > capacity = all HBM."""
    import torch

    from morphik_core_amd import MvError, _lib
    from morphik_core_amd.index import allow_bitmap

    N, stride, k = 420, 48, 7
    pages = _corpus(N, stride, seed=35)
    kw = dict(capacity_pages=N + 4, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    host = _idx(**kw)
    with _SplitAt(split_at):
        split = _idx(with_exact_split=True, **kw)
    assert split.exact_hbm_pages == min(split_at, N + 4) and host.exact_hbm_pages == 0
    with pytest.raises(MvError, match="MV_WITH_EXACT_SPLIT"):
        _idx(capacity_pages=8, stride_rows=stride, with_float=True, with_fp8=True, with_host_exact=True, with_exact_split=True)
    for ix in (host, split):
        ix.add(pages[:200])
        ix.add(pages[200:])  # a second ingest that starts on the HBM side or the host side, depending on the split
    qs = [orc.synth_rows(35, 3, 0, 20)] + [orc.synth_rows(977, j, 0, 20) for j in range(4)]
    long_q = orc.synth_rows(978, 0, 0, 80)  # > 64 rows: the batched rerank runs per query
    allow = allow_bitmap(np.arange(0, N, 3), N)
    for coarse_n, rerank_n in ((75, 128), (300, 64)):
        for ix in (host, split):
            ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
            ix.set_option(_lib.MV_OPT_RERANK_N, rerank_n)
        assert split.rerank_plan(coarse_n, k, 20) == host.rerank_plan(coarse_n, k, 20)
        for q in qs + [long_q]:
            for mode in ("fde_then_float", "fp8_then_float"):
                for al in (None, allow):
                    ws, wi = host.query(q, k, mode=mode, allow=al)
                    s, i = split.query(q, k, mode=mode, allow=al)
                    assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (coarse_n, mode, al is not None)
        for mode in ("fde_then_float", "fp8_then_float"):
            for batch in (qs, [long_q, long_q[:80]]):
                want = host.query_batch(batch, k, mode=mode)
                got = split.query_batch(batch, k, mode=mode)
                for (ws, wi), (s, i) in zip(want, got):
                    assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (coarse_n, mode, "batch")
    cand = np.array([0, 136, 137, 138, 419, 3, 200, 199, 10, 17], np.int32)
    for q in qs[:2]:
        a, b = host.score_candidates(q, cand), split.score_candidates(q, cand)
        assert a.tolist() == b.tolist()
        want = np.array([orc.maxsim_bf16(q, pages[c]) for c in cand], np.float32)
        np.testing.assert_allclose(b, want, rtol=RTOL)
    # the staged entry points of the one-process-per-GPU flow
    dev = torch.device("cuda", 0)
    coarse_n = 300
    for mode in ("fde_then_float", "fp8_then_float"):
        n_mid, tier = split.rerank_plan(coarse_n, k, 20, mode=mode)
        assert tier == "host"
        outs = []
        for ix in (host, split):
            recs = torch.empty(coarse_n * 16, dtype=torch.uint8, device=dev)
            ix.two_stage_coarse_device(qs[0], coarse_n, recs.data_ptr(), mode=mode)
            mid = None
            if n_mid:
                mid = torch.empty(coarse_n, dtype=torch.float32, device=dev)
                ix.two_stage_mid_device(qs[0], recs.data_ptr(), 1, coarse_n, mid.data_ptr(), mode=mode)
            ls = torch.empty(k, dtype=torch.float32, device=dev)
            li = torch.empty(k, dtype=torch.int64, device=dev)
            ix.two_stage_rerank_device(qs[0], recs.data_ptr(), 1, coarse_n, k, ls.data_ptr(), li.data_ptr(), mode=mode,
                                       d_all_mid_ptr=mid.data_ptr() if n_mid else 0, n_mid=n_mid)
            torch.cuda.synchronize()
            outs.append((ls.cpu().tolist(), li.cpu().tolist()))
        assert outs[0] == outs[1], mode
    host.close()
    split.close()


def test_split_exact_tier_writers_compaction_and_checkpoints(tmp_path):
    """replace_page / write_rows / read_pages on both sides of the split, compaction that moves pages ACROSS it, and a checkpoint
    that is reloaded with a different split: the tier's rows stay those of an unsplit index put through the same calls."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex

    N, stride, k = 300, 32, 5
    pages = _corpus(N, stride, seed=36)
    kw = dict(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    host = _idx(**kw)
    with _SplitAt(101):
        split = _idx(with_exact_split=True, **kw)
    qs = [orc.synth_rows(36, 3, 0, 16), orc.synth_rows(979, 1, 0, 16)]

    def same(a, b, tag):
        assert len(a) == len(b), tag
        np.testing.assert_array_equal(a.read_pages(0, len(a)), b.read_pages(0, len(b)), err_msg=tag)
        for q in qs:
            for mode in ("fde_then_float", "fp8_then_float"):
                ws, wi = a.query(q, k, mode=mode)
                s, i = b.query(q, k, mode=mode)
                assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (tag, mode)

    for ix in (host, split):
        ix.add(pages, doc_ordinals=np.arange(N, dtype=np.int32) // 3)
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 60)
    same(host, split, "ingest")
    new_rows = orc.synth_rows(555, 0, 0, 24)
    for ix in (host, split):
        for page in (0, 100, 101, 299):
            ix.replace_page(page, new_rows[: 10 + page % 7])
        ix.write_rows(100, 2, new_rows[:4])
        ix.write_rows(101, 0, new_rows[4:9])
    same(host, split, "replace / write")
    np.testing.assert_array_equal(split.read_pages(95, 12), host.read_pages(95, 12))
    for ix in (host, split):
        for d in (2, 20, 33, 34, 70):  # pages 6-8, 60-62, 99-104 (straddling the split), 210-212
            ix.remove_doc(d)
        ix.compact()
    assert len(split) == N - 15
    same(host, split, "compact")
    path = str(tmp_path / "split.idx")
    split.save(path)
    with _SplitAt(40):
        re = MvIndex.load(path, device=0)
    assert re.exact_hbm_pages == 40
    re.set_option(_lib.MV_OPT_FDE_COARSE_N, 60)  # options are the caller's, not the checkpoint's
    same(host, re, "reload with another split")
    re.close()
    re = MvIndex.load(path, device=0)  # no cap: everything the free HBM holds -- the whole (small) tier
    assert re.exact_hbm_pages == N
    re.set_option(_lib.MV_OPT_FDE_COARSE_N, 60)
    same(host, re, "reload all in HBM")
    re.close()
    host.close()
    split.close()


def test_split_exact_tier_rebalance_moves_hot_pages_into_hbm_and_changes_no_answer(tmp_path):
    """the HBM part of a split exact tier should hold the HOT pages, not the leading ones.  Every rerank counts the
    exact reads per page; mv_index_exact_tier_rebalance swaps the most-read host-resident pages with the least-read HBM-resident ones.
    Checked: the reads of a repeated workload move from host memory to HBM; ids and scores stay those of the unsplit host tier bit for
    bit -- single, batched, named candidates, staged, after a second rebalance (composed placements), through writers, a checkpoint
    (the file keeps page order: reloaded with any split) and a compaction (which restores slot == page first)."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex

    N, stride, k = 400, 32, 6
    pages = _corpus(N, stride, seed=37)
    kw = dict(capacity_pages=N + 8, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    host = _idx(**kw)
    with _SplitAt(60):
        split = _idx(with_exact_split=True, **kw)
    for ix in (host, split):
        ix.add(pages, doc_ordinals=np.arange(N, dtype=np.int32) // 2)
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
    # a workload whose candidates sit almost entirely behind the split (pages >= 60): "topic" queries built from late pages
    hot_q = [pages[300 + 7 * j][:16] for j in range(8)]
    other_q = [orc.synth_rows(980, j, 0, 16) for j in range(4)]

    def same(tag, qs=None):
        for q in (qs or hot_q[:3] + other_q[:2]):
            for mode in ("fde_then_float", "fp8_then_float"):
                ws, wi = host.query(q, k, mode=mode)
                s, i = split.query(q, k, mode=mode)
                assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (tag, mode)
        want = host.query_batch(hot_q[:5], k, mode="fde_then_float")
        got = split.query_batch(hot_q[:5], k, mode="fde_then_float")
        for (ws, wi), (s, i) in zip(want, got):
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (tag, "batch")

    def workload_reads():  # (reads served from HBM, from host memory) of one pass of the hot workload
        a = split.exact_tier_hits()
        for q in hot_q:
            split.query(q, k, mode="fde_then_float")
        b = split.exact_tier_hits()
        return b[0] - a[0], b[1] - a[1]

    assert host.rebalance_exact_tier() == 0 and host.exact_tier_hits() == (0, 0)  # no split: a no-op
    assert split.rebalance_exact_tier() == 0  # nothing read yet
    hb0, ho0 = workload_reads()
    assert hb0 + ho0 == 8 * 40 and ho0 > hb0  # 40 candidates per request went to the exact tier, most of them behind the split
    moved = split.rebalance_exact_tier()
    assert 0 < moved <= 60 and split.exact_tier_hits() == (0, 0)  # counters cleared
    same("after the first rebalance")
    hb1, ho1 = workload_reads()
    assert hb1 + ho1 == 8 * 40 and hb1 > hb0 and ho1 < ho0  # the same workload now reads (mostly) HBM
    # a different workload, a second rebalance: placements compose
    for q in other_q * 3:
        split.query(q, k, mode="fp8_then_float")
    assert split.rebalance_exact_tier(max_moves=10) <= 10
    same("after the second rebalance")
    cand = np.array([0, 59, 60, 61, 300, 307, 399, 3], np.int32)
    assert split.score_candidates(hot_q[0], cand).tolist() == host.score_candidates(hot_q[0], cand).tolist()
    np.testing.assert_array_equal(split.read_pages(0, N), host.read_pages(0, N))
    # writers address pages, not slots
    new_rows = orc.synth_rows(556, 0, 0, 20)
    for ix in (host, split):
        ix.replace_page(300, new_rows[:14])
        ix.replace_page(5, new_rows[:9])
        ix.write_rows(307, 1, new_rows[:3])
        ix.add([new_rows[:11]], doc_ordinals=[999])
    same("writers on a rebalanced tier")
    np.testing.assert_array_equal(split.read_pages(295, 20), host.read_pages(295, 20))
    # checkpoint: page order in the file, any split on reload
    path = str(tmp_path / "rebalanced.idx")
    split.save(path)
    with _SplitAt(25):
        re = MvIndex.load(path, device=0)
    re.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
    np.testing.assert_array_equal(re.read_pages(0, len(re)), host.read_pages(0, len(host)))
    ws, wi = host.query(hot_q[1], k, mode="fde_then_float")
    s, i = re.query(hot_q[1], k, mode="fde_then_float")
    assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    re.close()
    # compaction on a rebalanced tier
    for ix in (host, split):
        for d in (1, 20, 150, 155):
            ix.remove_doc(d)
        ix.compact()
    assert len(split) == len(host) == N + 1 - 8
    np.testing.assert_array_equal(split.read_pages(0, len(split)), host.read_pages(0, len(host)))
    same("after compaction", qs=other_q[:2])
    host.close()
    split.close()


def test_split_exact_tier_rebalance_on_uniform_pages():
    """Regression (round 6, found by bench.py): the kernels that take a page -> row table read the row count unconditionally; on an index whose
    pages all fill stride_rows (not "ragged": the scans get no n_rows array) the re-placed tier must still hand it over.  Uniform synthetic pages,
    single and batched reranks before and after a rebalance, against the unsplit host tier."""
    from morphik_core_amd import _lib

    N, stride, k = 300, 64, 5
    kw = dict(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    host = _idx(**kw)
    with _SplitAt(40):
        split = _idx(with_exact_split=True, **kw)
    for ix in (host, split):
        ix.fill_synthetic(1234, 0, N)  # every page has exactly stride_rows rows
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 30)
    qs = [orc.synth_rows(4321, j, 0, 16) for j in range(6)]
    for rnd in range(2):
        for q in qs:
            for mode in ("fde_then_float", "fp8_then_float"):
                ws, wi = host.query(q, k, mode=mode)
                s, i = split.query(q, k, mode=mode)
                assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (rnd, mode)
        for (ws, wi), (s, i) in zip(host.query_batch(qs, k, mode="fde_then_float"), split.query_batch(qs, k, mode="fde_then_float")):
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (rnd, "batch")
        if rnd == 0:
            assert split.rebalance_exact_tier() > 0
    host.close()
    split.close()


def test_split_exact_tier_behind_the_store():
    """create_store("mi355x_fast_split_exact" / "mi355x_sharded_fast_split_exact"): the plugin surface over split tiers returns the
    hits of the unsplit host-tier providers, score for score."""
    import asyncio

    from morphik_core_amd.models import DocumentChunk
    from morphik_core_amd.store import create_store

    stride = 48
    pages = _corpus(120, stride, seed=37)
    chunks = [DocumentChunk(document_id=f"d{i // 4}", content=f"p{i}", embedding=orc.bf16_to_f32(p), chunk_number=i % 4, metadata={}) for i, p in enumerate(pages)]
    q = orc.bf16_to_f32(orc.synth_rows(37, 3, 0, 20))
    res = {}
    for provider in ("mi355x_fast_host_exact", "mi355x_fast_split_exact", "mi355x_sharded_fast_host_exact", "mi355x_sharded_fast_split_exact"):
        kw = dict(devices=[0, 0, 0], transport="p2p") if "sharded" in provider else {}
        with _SplitAt(23):  # of 300 (one index) or 100 (each of three shards) slots
            st = create_store(provider, capacity_pages=300, stride_rows=stride, **kw)
            assert st.initialize()

        async def run():
            ok, ids, _m = await st.store_embeddings(chunks[:50], app_id=None)
            assert ok and len(ids) == 50
            ok, ids, _m = await st.store_embeddings(chunks[50:], app_id=None)
            assert ok
            return await st.query_similar(q, k=6)

        hits = asyncio.run(run())
        res[provider] = [(h.document_id, h.chunk_number, h.score) for h in hits]
        if "split" in provider:
            shards = st._index.shards if "sharded" in provider else [st._index]
            assert [sh.exact_hbm_pages for sh in shards] == [23] * len(shards)
        st.close()
    assert res["mi355x_fast_split_exact"] == res["mi355x_fast_host_exact"] and len(res["mi355x_fast_host_exact"]) == 6
    assert res["mi355x_sharded_fast_split_exact"] == res["mi355x_sharded_fast_host_exact"]


def test_lean_shard_fde_slab_plus_split_exact_tier_without_the_e4m3_slab():
    """An FDE shard WITHOUT the e4m3 slab (store option prune_slab=False): there is no pruning stage -- every coarse candidate goes to the
    (split) exact tier, whatever the list length -- and the answers are those of an index that keeps the slab but never prunes
    (MV_OPT_RERANK_N at its maximum).  The modes that need the e4m3 slab are refused."""
    from morphik_core_amd import MvError, _lib

    N, stride, k = 420, 48, 7
    pages = _corpus(N, stride, seed=38)
    full = _idx(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_fp8=True, with_host_exact=True)
    with _SplitAt(150):
        lean = _idx(capacity_pages=N, stride_rows=stride, with_float=False, with_fde=True, with_host_exact=True, with_exact_split=True)
    assert lean.exact_hbm_pages == 150
    for ix in (full, lean):
        ix.add(pages)
    full.set_option(_lib.MV_OPT_RERANK_N, 1024)
    qs = [orc.synth_rows(38, 3, 0, 20)] + [orc.synth_rows(981, j, 0, 20) for j in range(3)]
    for coarse_n in (75, 300):
        for ix in (full, lean):
            ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        assert lean.rerank_plan(coarse_n, k, 20) == (0, "host")
        for q in qs:
            ws, wi = full.query(q, k, mode="fde_then_float")
            s, i = lean.query(q, k, mode="fde_then_float")
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), coarse_n
            want = np.array([orc.maxsim_bf16(q, pages[p], int(_cascade_pad(lean, q, pages, coarse_n, p))) for p in i], np.float32)
            np.testing.assert_allclose(s, want, rtol=RTOL)
        for (ws, wi), (s, i) in zip(full.query_batch(qs, k, mode="fde_then_float"), lean.query_batch(qs, k, mode="fde_then_float")):
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (coarse_n, "batch")
    for mode in ("fp8_then_float", "float_fp8"):
        with pytest.raises(MvError):
            lean.query(qs[0], k, mode=mode)
    full.close()
    lean.close()


def _cascade_pad(ix, q, pages, coarse_n, page):
    """Pad length the reference's rerank gives `page`: the longest page of its batch of 128 in the coarse list (coarse rank order)."""
    coarse = ix.score_all(q, mode="fde")
    cs, ci = orc.topk(coarse, coarse_n)
    ci = ci[np.isfinite(cs)]
    rows = np.array([pages[c].shape[0] for c in ci])
    pads = _batch_pads(rows)
    return pads[list(ci).index(page)]


# ------------------------------------------------------------------ bring-your-own FDE (mv_index_import_fde, mv_query_topk_fde)
def test_caller_supplied_fde_vectors_drive_the_coarse_stage():
    """Document FDE vectors imported into the slab and query FDE vectors handed to the scan -- the path a deployment takes that keeps the
    reference's own `fde` extension (or vectors exported from a TurboPuffer namespace) for the encodings:
    (a) arbitrary vectors: the coarse scores (mode "fde", k = all) are cosine(q_fde, bf16(d_fde)) computed on the host, 1e-4;
    (b) the pipeline reranks exactly the candidates those scores nominate (float oracle on the rows), single, batched, 2 shards;
    (c) the library's OWN encodings fed back through the same doors reproduce the internal path (coarse scores 1e-3, same answers);
    (d) NaN / Inf and a wrong width are refused, so are modes without an FDE stage."""
    from morphik_core_amd import MvError, _lib
    from morphik_core_amd.index import ShardComm, fde_encode

    N, stride, k = 300, 32, 6
    pages = _corpus(N, stride, seed=39)
    kw = dict(stride_rows=stride, with_float=True, with_fde=True)
    ix = _idx(capacity_pages=N, **kw)
    ix.add(pages)
    D = ix.fde_config.output_dim
    rng = np.random.default_rng(39)
    docs = rng.standard_normal((N, D)).astype(np.float32)
    qs = [orc.synth_rows(39, 3, 0, 20)] + [orc.synth_rows(983, j, 0, 20) for j in range(4)]
    qf = rng.standard_normal((len(qs), D)).astype(np.float32)
    docs[7] = 3.0 * qf[0] + 0.1 * docs[7]  # a planted coarse neighbour of query 0
    ix.import_fde(0, docs[:120])
    ix.import_fde(120, docs[120:])
    db = orc.bf16_to_f32(orc.f32_to_bf16(docs))
    np.testing.assert_array_equal(ix.read_fde(100, 40), db[100:140])  # the slab holds the rounded vectors, read back through mv_index_read_fde
    want_all = (db @ qf.T) / np.linalg.norm(db, axis=1, keepdims=True)  # [page][query]: the scan's cosine rule (1/|d| of the rounded vector)
    ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
    for j, q in enumerate(qs):
        s, i = ix.query(q, N, mode="fde", q_fde=qf[j])
        assert sorted(i.tolist()) == list(range(N))
        np.testing.assert_allclose(s, want_all[i, j], rtol=2e-4, atol=2e-5)
        # (b) the candidates of the pipeline are the coarse top-40 of THOSE scores; the rerank is the float oracle on the rows
        cand = np.lexsort((np.arange(N), -want_all[:, j].astype(np.float64)))[:40]
        rows = np.array([pages[c].shape[0] for c in cand])
        pads = _batch_pads(rows)
        exact = np.array([orc.maxsim_bf16(q, pages[c], int(p)) for c, p in zip(cand, pads)], np.float32)
        order = np.lexsort((np.arange(40), -exact.astype(np.float64)))[:k]
        s, i = ix.query(q, k, mode="fde_then_float", q_fde=qf[j])
        assert set(i.tolist()) <= set(cand.tolist())
        np.testing.assert_allclose(s, exact[order], rtol=RTOL)
        assert i.tolist() == cand[order].tolist() or len(set(i.tolist()) ^ set(cand[order].tolist())) <= 2  # near-ties of the coarse cut
    assert ix.query(qs[0], 1, mode="fde", q_fde=qf[0])[1].tolist() == [7]
    single = [ix.query(q, k, mode="fde_then_float", q_fde=qf[j]) for j, q in enumerate(qs)]
    for (ws, wi), (s, i) in zip(single, ix.query_batch(qs, k, mode="fde_then_float", q_fdes=qf)):
        assert i.tolist() == wi.tolist()
        np.testing.assert_allclose(s, ws, rtol=1e-6)
    # two shards behind the communicator: the same answers (global candidate rule), single and batched
    shards = []
    for r in range(2):
        sh = _idx(capacity_pages=N // 2, id_base=r * (N // 2), **kw)
        sh.add(pages[r * (N // 2) : (r + 1) * (N // 2)])
        sh.import_fde(0, docs[r * (N // 2) : (r + 1) * (N // 2)])
        sh.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
        shards.append(sh)
    comm = ShardComm(shards, transport="p2p")
    for j, q in enumerate(qs):
        s, i = comm.query(q, k, mode="fde_then_float", q_fde=qf[j])
        assert i.tolist() == single[j][1].tolist() and s.tolist() == single[j][0].tolist()
    for (ws, wi), (s, i) in zip(single, comm.query_batch(qs, k, mode="fde_then_float", q_fdes=qf)):
        assert i.tolist() == wi.tolist()
        np.testing.assert_allclose(s, ws, rtol=1e-6)
    comm.close()
    # the stream-ordered staged pipeline of the one-process-per-GPU flow (no process group: one rank) with the caller's query FDE
    from morphik_core_amd import sharded
    import torch

    se = sharded.GpuTwoStageSearcher(ix, mode="fde_then_float")
    for j, q in enumerate(qs[:2]):
        s, i = se.query(q, k, coarse_n=40, q_fde=qf[j])
        torch.cuda.synchronize()
        assert i.cpu().tolist() == single[j][1].tolist() and s.cpu().tolist() == single[j][0].tolist()
    for sh in shards:
        sh.close()
    # (c) the library's own encodings through the same doors == the internal path
    own = _idx(capacity_pages=N, **kw)
    own.add(pages)
    own.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
    mine = np.stack([fde_encode(orc.bf16_to_f32(p), ix.fde_config, is_query=False) for p in pages])
    ix.import_fde(0, mine)
    for q in qs:
        ws, wi = own.query(q, k, mode="fde_then_float")
        s, i = ix.query(q, k, mode="fde_then_float", q_fde=fde_encode(orc.bf16_to_f32(q), ix.fde_config, is_query=True))
        # (the ingest kernels and mv_fde_encode may round an FDE element differently in the last bit: a near-tie at the coarse cut may swap)
        common = set(i.tolist()) & set(wi.tolist())
        assert len(common) >= k - 1
        assert {int(a): float(b) for a, b in zip(i, s) if int(a) in common} == {int(a): float(b) for a, b in zip(wi, ws) if int(a) in common}
        ca, cb = own.query(q, N, mode="fde"), ix.query(q, N, mode="fde", q_fde=fde_encode(orc.bf16_to_f32(q), ix.fde_config, is_query=True))
        np.testing.assert_allclose(cb[0][np.argsort(cb[1])], ca[0][np.argsort(ca[1])], rtol=1e-3, atol=1e-5)
    # (d) refusals
    bad = qf[0].copy()
    bad[5] = np.nan
    with pytest.raises(MvError, match="NaN"):
        ix.query(qs[0], k, mode="fde_then_float", q_fde=bad)
    with pytest.raises(MvError, match="NaN"):
        ix.import_fde(3, bad[None, :])
    with pytest.raises(ValueError, match="floats"):
        ix.query(qs[0], k, mode="fde_then_float", q_fde=qf[0][:100])
    with pytest.raises(MvError, match="no FDE stage"):
        ix.query(qs[0], k, mode="float", q_fde=qf[0])
    own.close()
    ix.close()
