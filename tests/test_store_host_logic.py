"""CPU tests of the store's HOST logic (ids, payloads, filters, metrics, async surface) with an
oracle-backed index injected; the same scenarios run on the real GPU index in test_gpu_store.py."""
import numpy as np
import pytest

from morphik_core_amd.store import MI355XFastMultiVectorStore, MI355XMultiVectorStore, create_store
from tests import store_scenarios as sc
from tests.fake_index import OracleIndex


def _store(cls=MI355XMultiVectorStore, **kw):
    s = cls(capacity_pages=64, stride_rows=32, index_factory=OracleIndex, **kw)
    assert s.initialize() is True
    return s


@pytest.mark.parametrize("scenario", sc.ALL, ids=lambda f: f.__name__)
@pytest.mark.parametrize("mode", ["binary", "float"])
def test_reference_store_scenarios(scenario, mode):
    sc.run(scenario(_store(mode=mode)))


def test_known_ranking_binary_scores_exact():
    sc.run(sc.scenario_known_ranking(_store(mode="binary"), exact_binary=True))
    sc.run(sc.scenario_known_ranking(_store(mode="float"), exact_binary=False))


def test_initialize_never_raises_and_is_loud_without_gpu():
    s = MI355XMultiVectorStore(capacity_pages=4, stride_rows=16)  # real index: no GPU in the CPU container
    import torch

    if not torch.cuda.is_available():
        assert s.initialize() is False  # returns False, never raises (multi_vector_store.py:325-327)
        with pytest.raises(Exception):
            sc.run(s.query_similar(np.ones((2, 128), np.float32), k=1))  # no CPU fallback: query raises


def test_fast_store_filters_by_app_namespace():
    s = _store(MI355XFastMultiVectorStore, mode="float")
    rng = np.random.default_rng(0)
    a = sc.make_chunks(rng, n_docs=1, chunks_per_doc=2)
    b = [c.model_copy(update={"document_id": "other"}) for c in sc.make_chunks(rng, n_docs=1, chunks_per_doc=2)]
    sc.run(s.store_embeddings(a, app_id="app-a"))
    sc.run(s.store_embeddings(b, app_id="app-b"))
    res = sc.run(s.query_similar(a[0].embedding, k=10, app_id="app-b"))
    assert {r.document_id for r in res} == {"other"}
    # FastMultiVectorStore always queries ONE namespace, self.ns(app_id) (fast_multivector_store.py:526): without an app_id
    # that is the default namespace -- never every tenant's pages
    assert sc.run(s.query_similar(a[0].embedding, k=10)) == []
    c = [x.model_copy(update={"document_id": "mine"}) for x in sc.make_chunks(rng, n_docs=1, chunks_per_doc=2)]
    sc.run(s.store_embeddings(c))  # no app_id on the way in -> the default namespace too
    assert {r.document_id for r in sc.run(s.query_similar(c[0].embedding, k=10))} == {"mine"}
    assert {r.document_id for r in sc.run(s.query_similar(c[0].embedding, k=10, app_id="default"))} == {"mine"}
    # a doc_ids filter cannot reach into another app's namespace
    assert sc.run(s.query_similar(a[0].embedding, k=10, doc_ids=["other"], app_id="app-a")) == []
    # store_embeddings without app_id resolves the document's app through the callback (the reference's documents-table lookup)
    s2 = _store(MI355XFastMultiVectorStore, mode="float", app_id_resolver=lambda doc: {"doc0": "app-z"}.get(doc))
    sc.run(s2.store_embeddings(a))
    assert len(sc.run(s2.query_similar(a[0].embedding, k=10, app_id="app-z"))) == 2
    assert sc.run(s2.query_similar(a[0].embedding, k=10)) == []


def test_same_document_id_under_two_apps_is_two_documents():
    """FastMultiVectorStore writes to self.ns(app_id) and leaves every other namespace alone (fast_multivector_store.py:440-502,
    :643): a re-ingest of a document_id under app B must not expose, move or delete app A's chunks."""
    s = _store(MI355XFastMultiVectorStore, mode="float")
    rng = np.random.default_rng(5)
    a = sc.make_chunks(rng, n_docs=1, chunks_per_doc=3)  # doc0, chunks 0..2, app-a
    sc.run(s.store_embeddings(a, app_id="app-a"))
    b0 = a[0].model_copy(update={"content": "b-version"})
    sc.run(s.store_embeddings([b0], app_id="app-b"))  # the same document_id, ONE chunk, another tenant
    seen_a = sc.run(s.query_similar(a[1].embedding, k=10, app_id="app-a"))
    assert sorted((c.chunk_number, c.content) for c in seen_a) == sorted((c.chunk_number, c.content) for c in a)  # A keeps all three, its own versions
    seen_b = sc.run(s.query_similar(a[1].embedding, k=10, app_id="app-b"))
    assert [(c.chunk_number, c.content) for c in seen_b] == [(0, "b-version")]  # B sees its one chunk, none of A's
    assert [c.content for c in sc.run(s.get_chunks_by_id([("doc0", 0), ("doc0", 2)], app_id="app-b"))] == ["b-version"]
    assert len(sc.run(s.get_chunks_by_id([("doc0", 0), ("doc0", 2)], app_id="app-a"))) == 2
    assert sc.run(s.query_similar(a[1].embedding, k=10, doc_ids=["doc0"], app_id="app-b"))[0].content == "b-version"
    assert sc.run(s.delete_chunks_by_document_id("doc0", app_id="app-b")) is True  # B deletes ITS copy only
    assert len(sc.run(s.query_similar(a[1].embedding, k=10, app_id="app-a"))) == 3
    assert sc.run(s.query_similar(a[1].embedding, k=10, app_id="app-b")) == []
    # a failed add hands its fresh ordinals back and leaves the other namespace untouched
    full = MI355XFastMultiVectorStore(capacity_pages=3, stride_rows=32, mode="float", index_factory=OracleIndex)
    sc.run(full.store_embeddings(a, app_id="app-a"))
    ords = dict(full._doc_ord)
    with pytest.raises(Exception):
        sc.run(full.store_embeddings([b0], app_id="app-b"))  # slab full
    assert full._doc_ord == ords and len(sc.run(full.query_similar(a[1].embedding, k=10, app_id="app-a"))) == 3


def test_too_many_vectors_is_an_error_and_factory():
    s = _store(mode="float")
    from morphik_core_amd.models import DocumentChunk

    with pytest.raises(ValueError):
        sc.run(s.store_embeddings([DocumentChunk(document_id="d", chunk_number=0, content="", embedding=np.ones((33, 128), np.float32))]))
    assert isinstance(create_store("mi355x_fast", capacity_pages=4), MI355XFastMultiVectorStore)
    assert create_store("mi355x_float", capacity_pages=4).mode == "float"
    with pytest.raises(ValueError):
        create_store("postgres")


def test_exact_tier_choices_map_to_the_slab_flags_of_the_index():
    """exact_tier "hbm" / "host" / "split" (and the providers that name them) -> the slabs and exact-tier flags MvIndex is created with:
    the bf16 slab in HBM; FDE + e4m3 slabs with the exact rows pinned; the same with the leading pages' exact rows in the free HBM."""
    want = {"hbm": dict(with_float=True, with_fp8=False), "host": dict(with_float=False, with_fp8=True, with_host_exact=True),
            "split": dict(with_float=False, with_fp8=True, with_host_exact=True, with_exact_split=True)}
    for tier, provider in (("hbm", "mi355x_fast"), ("host", "mi355x_fast_host_exact"), ("split", "mi355x_fast_split_exact"),
                           ("host", "mi355x_sharded_fast_host_exact"), ("split", "mi355x_sharded_fast_split_exact")):
        kw = dict(devices=[0, 1]) if "sharded" in provider else {}
        st = create_store(provider, capacity_pages=8, **kw)
        assert st.exact_tier == tier and st.mode == "fde_then_float"
        fl = st._slab_flags()
        assert fl["with_fde"] and not fl["with_binary"]
        for k, v in want[tier].items():
            assert fl.get(k, False) == v, (provider, k)
        assert ("with_exact_split" in fl) == (tier == "split") and ("with_host_exact" in fl) == (tier != "hbm")
    for provider, kw in (("mi355x_fast_split_exact_lean", {}), ("mi355x_sharded_fast_split_exact_lean", dict(devices=[0, 1]))):
        fl = create_store(provider, capacity_pages=8, **kw)._slab_flags()  # no e4m3 pruning slab: FDE slab + the split exact tier only
        assert fl["with_fde"] and fl["with_host_exact"] and fl["with_exact_split"] and not fl["with_fp8"] and not fl["with_float"]
    fl = create_store("mi355x_fp8_exact", capacity_pages=8)._slab_flags()  # configs[4]: e4m3 scan + exact re-score, no FDE
    assert fl["with_fp8"] and fl["with_host_exact"] and not fl["with_fde"] and not fl["with_float"]
    for provider, kw in (("mi355x_fp8_split_exact", {}), ("mi355x_sharded_fp8_split_exact", dict(devices=[0, 1]))):
        fl = create_store(provider, capacity_pages=8, **kw)._slab_flags()
        assert fl["with_fp8"] and fl["with_host_exact"] and fl["with_exact_split"] and not fl["with_fde"] and not fl["with_float"]
    with pytest.raises(ValueError, match="exact_tier"):
        MI355XFastMultiVectorStore(capacity_pages=8, exact_tier="nvme")
    # a split takes "what the device has free": two shards of one store on ONE device are refused (unless the HBM part is bounded)
    from morphik_core_amd.shard_index import ShardedIndex
    from tests.fake_index import OracleComm

    with pytest.raises(ValueError, match="its own device"):
        ShardedIndex(capacity_pages=8, stride_rows=32, devices=[0, 0], with_float=False, with_fde=True, with_fp8=True, with_host_exact=True,
                     with_exact_split=True, index_cls=OracleIndex, comm_cls=OracleComm)


def test_fast_hit_chunks_are_indistinguishable_from_validated_ones(monkeypatch):
    """models.hit_chunk_builder: search hits are built without re-validating values that were validated at ingest (1.5 -> 0.8 us per chunk,
    a third of the store's per-request Python).  The objects must be DocumentChunks in every observable way, must not share their metadata
    dict with the store's cache (a caller may edit its hits), and MV_FAST_HIT_CHUNKS=0 must give the validating constructor back."""
    from morphik_core_amd import models
    from morphik_core_amd.models import DocumentChunk

    fast = models.hit_chunk_builder()
    assert fast.__name__ == "fast"  # this pydantic version passes the builder's own equivalence check
    meta = {"is_image": True, "page": 3}
    a = fast("doc", 2, "body", meta, 0.75)
    b = DocumentChunk(document_id="doc", chunk_number=2, content="body", embedding=[], metadata=meta, score=0.75)
    assert isinstance(a, DocumentChunk) and a == b and a.model_dump() == b.model_dump() and a.model_dump_json() == b.model_dump_json()
    assert a.model_fields_set == b.model_fields_set and repr(a) == repr(b)
    a.metadata["page"] = 99
    assert meta["page"] == 3 and a.embedding == [] and a.embedding is not fast("doc", 2, "body", meta, 0.75).embedding
    assert a.model_copy(update={"score": 0.1}).score == 0.1 and DocumentChunk.model_validate(a.model_dump()) == DocumentChunk.model_validate(b.model_dump() | {"metadata": {"is_image": True, "page": 99}})
    monkeypatch.setenv("MV_FAST_HIT_CHUNKS", "0")
    assert models.hit_chunk_builder().__name__ == "slow"
    # through a store: the hits of both builders are the same chunks
    rng = np.random.default_rng(8)
    chunks = sc.make_chunks(rng, n_docs=3, chunks_per_doc=3)
    slow_store = _store(mode="float")  # built while the variable is set
    monkeypatch.delenv("MV_FAST_HIT_CHUNKS")
    fast_store = _store(mode="float")
    assert slow_store._hit_chunk.__name__ == "slow" and fast_store._hit_chunk.__name__ == "fast"
    sc.run(slow_store.store_embeddings(chunks))
    sc.run(fast_store.store_embeddings(chunks))
    for c in chunks[:4]:
        assert sc.run(fast_store.query_similar(c.embedding, k=5)) == sc.run(slow_store.query_similar(c.embedding, k=5))


def test_concurrent_requests_are_coalesced_into_one_batched_scan():
    """batch_window_ms > 0: concurrent query_similar calls (different k, different doc_ids filters) ride one
    mv_query_topk_batch pass and each gets exactly what a lone call would have returned."""
    import asyncio

    rng = np.random.default_rng(2)
    chunks = sc.make_chunks(rng, n_docs=4, chunks_per_doc=3)
    plain = _store(mode="float")
    fused = _store(mode="float", batch_window_ms=20.0, max_batch=8)
    sc.run(plain.store_embeddings(chunks))
    sc.run(fused.store_embeddings(chunks))
    reqs = [(chunks[i].embedding, 1 + i % 4, None if i % 3 else [chunks[i].document_id, chunks[0].document_id]) for i in range(11)]

    async def fire(store):
        return await asyncio.gather(*(store.query_similar(q, k=k, doc_ids=d) for q, k, d in reqs))

    want = sc.run(fire(plain))
    got = sc.run(fire(fused))
    assert fused.coalesced_batches == [8, 3]  # 11 concurrent requests: one full batch, one flushed by the window
    for w, g, (_q, k, d) in zip(want, got, reqs):
        assert [(c.document_id, c.chunk_number) for c in g] == [(c.document_id, c.chunk_number) for c in w]
        assert [c.score for c in g] == [c.score for c in w] and len(g) <= k
        if d:
            assert {c.document_id for c in g} <= set(d)


def test_adaptive_coalescing_dispatches_a_lone_request_at_once_and_batches_what_arrives_meanwhile():
    """batch_window_ms < 0 (group commit): no timer -- a request that finds the index idle goes out immediately (with the
    requests that became ready in the same loop tick); requests arriving while a pass is in flight ride the next pass."""
    import asyncio
    import time

    rng = np.random.default_rng(4)
    chunks = sc.make_chunks(rng, n_docs=4, chunks_per_doc=3)
    plain = _store(mode="float")
    fused = _store(mode="float", batch_window_ms=-1, max_batch=8)
    sc.run(plain.store_embeddings(chunks))
    sc.run(fused.store_embeddings(chunks))
    t0 = time.perf_counter()
    lone = sc.run(fused.query_similar(chunks[0].embedding, k=2))
    assert fused.coalesced_batches == [1] and time.perf_counter() - t0 < 5.0
    assert [(c.document_id, c.chunk_number) for c in lone] == [(c.document_id, c.chunk_number) for c in sc.run(plain.query_similar(chunks[0].embedding, k=2))]
    reqs = [(chunks[i].embedding, 1 + i % 4, None if i % 3 else [chunks[i].document_id, chunks[0].document_id]) for i in range(11)]

    async def fire(store):
        return await asyncio.gather(*(store.query_similar(q, k=k, doc_ids=d) for q, k, d in reqs))

    fused.coalesced_batches.clear()
    want, got = sc.run(fire(plain)), sc.run(fire(fused))
    assert fused.coalesced_batches == [8, 3]  # one tick: 8 go out at once, the rest ride the pass that follows
    for w, g in zip(want, got):
        assert [(c.document_id, c.chunk_number, c.score) for c in g] == [(c.document_id, c.chunk_number, c.score) for c in w]

    async def staggered():  # a second wave arrives while the first pass is in flight
        first = [asyncio.ensure_future(fused.query_similar(q, k=k, doc_ids=d)) for q, k, d in reqs[:3]]
        await asyncio.sleep(0)  # the first wave is dispatched
        await asyncio.sleep(0)
        second = [asyncio.ensure_future(fused.query_similar(q, k=k, doc_ids=d)) for q, k, d in reqs[3:9]]
        return await asyncio.gather(*first, *second)

    fused.coalesced_batches.clear()
    got2 = sc.run(staggered())
    assert sum(fused.coalesced_batches) == 9 and fused.coalesced_batches[0] == 3 and len(fused.coalesced_batches) <= 3
    for w, g in zip(want[:9], got2):
        assert [(c.document_id, c.chunk_number, c.score) for c in g] == [(c.document_id, c.chunk_number, c.score) for c in w]


def test_adaptive_coalescing_keeps_a_second_full_batch_in_flight():
    """pipeline_depth = 2 (default): behind a pass in flight, a FULL batch is dispatched at once (the library serialises the two
    passes; the event loop builds the first pass's hits while the second runs); a partial batch still waits for an idle index.
    pipeline_depth = 1 is the strict one-pass-at-a-time form.  Answers are those of a lone call either way."""
    import asyncio
    import threading
    import time

    rng = np.random.default_rng(6)
    chunks = sc.make_chunks(rng, n_docs=4, chunks_per_doc=3)

    class SlowIndex(OracleIndex):
        live = 0
        peak = 0
        mu = threading.Lock()

        def query_batch(self, *a, **kw):
            with SlowIndex.mu:
                SlowIndex.live += 1
                SlowIndex.peak = max(SlowIndex.peak, SlowIndex.live)
            time.sleep(0.05)
            try:
                return super().query_batch(*a, **kw)
            finally:
                with SlowIndex.mu:
                    SlowIndex.live -= 1

    plain = _store(mode="float")
    sc.run(plain.store_embeddings(chunks))
    reqs = [(chunks[i % 12].embedding, 1 + i % 4) for i in range(20)]
    want = [sc.run(plain.query_similar(q, k=k)) for q, k in reqs]
    for depth, peak, sizes in ((2, 2, [8, 8, 4]), (1, 1, [8, 8, 4])):
        fused = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=SlowIndex, mode="float", batch_window_ms=-1, max_batch=8,
                                       pipeline_depth=depth)
        assert fused.initialize()
        sc.run(fused.store_embeddings(chunks))
        SlowIndex.peak = 0

        async def fire():
            return await asyncio.gather(*(fused.query_similar(q, k=k) for q, k in reqs))

        got = sc.run(fire())
        assert fused.coalesced_batches == sizes and SlowIndex.peak == peak
        for w, g in zip(want, got):
            assert [(c.document_id, c.chunk_number, c.score) for c in g] == [(c.document_id, c.chunk_number, c.score) for c in w]


def test_concurrent_requests_on_the_fast_store_are_coalesced_by_k():
    """mode fde_then_float: coalesced requests share a batched call only with requests of the same k (the candidate rule
    min(10k, 75) depends on k), so every request gets exactly what a lone call returns."""
    import asyncio

    rng = np.random.default_rng(3)
    chunks = sc.make_chunks(rng, n_docs=4, chunks_per_doc=3)
    from oracle import oracle as orc

    def fde_index(**kw):
        return OracleIndex(fde=orc.FdeConfig.reference_default(), **kw)

    plain = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=fde_index, mode="fde_then_float")
    fused = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=fde_index, mode="fde_then_float", batch_window_ms=20.0, max_batch=8)
    assert plain.initialize() and fused.initialize()
    sc.run(plain.store_embeddings(chunks))
    sc.run(fused.store_embeddings(chunks))
    reqs = [(chunks[i].embedding, 1 + i % 3, None if i % 3 else [chunks[i].document_id, chunks[0].document_id]) for i in range(10)]

    async def fire(store):
        return await asyncio.gather(*(store.query_similar(q, k=k, doc_ids=d) for q, k, d in reqs))

    want, got = sc.run(fire(plain)), sc.run(fire(fused))
    assert fused.coalesced_batches == [8, 2]
    for w, g, (_q, k, _d) in zip(want, got, reqs):
        assert [(c.document_id, c.chunk_number, c.score) for c in g] == [(c.document_id, c.chunk_number, c.score) for c in w] and len(g) <= k


class _KeyedFde:
    """The API of the reference's `fde` extension (fast_multivector_store.py:325-331, :447-449, :521) with a transparent encoder behind
    it: a page / query whose first element is x encodes to the unit vector e_j, j = round(10 x) -- so the coarse score of a page is 1
    when its key equals the query's and 0 otherwise, whatever the rows say."""

    class FixedDimensionalEncodingConfig:
        def __init__(self, **kw):
            self.kw = kw

    def __init__(self):
        self.docs = self.queries = 0

    @staticmethod
    def _enc(rows):
        v = np.zeros(10240, np.float32)
        v[int(round(10 * float(np.asarray(rows, np.float32)[0, 0]))) % 10240] = 1.0
        return v

    def generate_document_encoding(self, emb, cfg):
        assert cfg.kw == dict(dimension=128, num_repetitions=20, num_simhash_projections=5, projection_dimension=16, projection_type="AMS_SKETCH")
        self.docs += 1
        return self._enc(emb)

    def generate_query_encoding(self, q, cfg):
        self.queries += 1
        return self._enc(q)


def test_bring_your_own_fde_module_drives_the_candidate_stage(tmp_path):
    """fde_module=: the store calls the deployment's own FDE encoder (the reference's extension where it is installed) for every chunk and
    every query, imports the document vectors into the FDE slab and hands the query vectors to the scan.  With the keyed stand-in above
    the candidate list of a k = 1 request (min(10 k, 75) = 10 pages) is exactly the pages of the query's key -- the exact best page, under
    another key, must NOT come back, although the store without the module finds it."""
    from morphik_core_amd.models import DocumentChunk
    from oracle import oracle as orc

    class fde_index(OracleIndex):  # an index class (the store's checkpoint path calls index_factory.load)
        def __init__(self, *a, **kw):
            kw["fde"] = orc.FdeConfig.reference_default()
            super().__init__(*a, **kw)

    rng = np.random.default_rng(5)
    q = rng.standard_normal((6, 128)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0, 0] = 0.3  # key 3
    chunks = []
    for i in range(40):
        e = rng.standard_normal((8, 128)).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        e[0, 0] = 0.3 if i % 4 == 0 else 0.7  # ten pages of key 3, thirty of key 7
        if i == 17:  # the exact best page of the query -- under the OTHER key
            e[1:7] = q
        if i == 8:   # the best page among the query's own key
            e[1:4] = q[:3]
        chunks.append(DocumentChunk(document_id=f"d{i}", content=f"c{i}", embedding=e, chunk_number=0, metadata={}))
    mod = _KeyedFde()
    own = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=fde_index, mode="fde_then_float")
    byo = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=fde_index, mode="fde_then_float", fde_module=mod)
    assert own.initialize() and byo.initialize()
    for st in (own, byo):
        ok, ids, _m = sc.run(st.store_embeddings(chunks))
        assert ok and len(ids) == 40
    assert mod.docs == 40
    exact_best = sc.run(own.query_similar(q, k=3))[0]  # 30 candidates of 40: the exact best is among them
    assert exact_best.document_id == "d17"
    hit = sc.run(byo.query_similar(q, k=1))
    assert [h.document_id for h in hit] == ["d8"] and mod.queries == 1  # ten candidates: the pages of key 3, reranked exactly
    # coalesced requests carry their own query encodings
    fused = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, index_factory=fde_index, mode="fde_then_float", fde_module=mod, batch_window_ms=20.0, max_batch=8)
    assert fused.initialize()
    sc.run(fused.store_embeddings(chunks))
    q7 = q.copy()
    q7[0, 0] = 0.7

    async def both():
        import asyncio

        return await asyncio.gather(fused.query_similar(q, k=1), fused.query_similar(q7, k=1))

    a, b = sc.run(both())
    assert [h.document_id for h in a] == ["d8"] and b[0].document_id != "d8"
    assert fused.last_query_timing.get("batched_queries") == 2
    # a checkpoint remembers whose vectors its FDE slab holds
    byo.save(str(tmp_path / "ck"))
    with pytest.raises(RuntimeError, match="fde_module"):
        MI355XMultiVectorStore.load(str(tmp_path / "ck"), index_factory=fde_index)
    back = MI355XMultiVectorStore.load(str(tmp_path / "ck"), index_factory=fde_index, fde_module=mod)
    assert [h.document_id for h in sc.run(back.query_similar(q, k=1))] == ["d8"]
    own.save(str(tmp_path / "ck2"))
    with pytest.raises(RuntimeError, match="without fde_module"):
        MI355XMultiVectorStore.load(str(tmp_path / "ck2"), index_factory=fde_index, fde_module=mod)
    with pytest.raises(ValueError, match="fde_then_float"):
        MI355XMultiVectorStore(capacity_pages=8, mode="float", fde_module=mod)


def test_owner_server_takes_the_deployments_fde_module_by_name(monkeypatch):
    """store_server --fde-module <importable name>: the owner process imports the deployment's encoder and builds the store over it;
    a module that is not installed fails loudly."""
    import argparse

    from morphik_core_amd import store as store_mod
    from morphik_core_amd import store_server

    seen = {}
    real = store_mod.create_store

    def spy(provider, **kw):
        seen.update(kw)
        kw["index_factory"] = OracleIndex  # no GPU here: the oracle-backed index stands in
        return real(provider, **kw)

    monkeypatch.setattr(store_mod, "create_store", spy)
    a = argparse.Namespace(provider="mi355x_fast", capacity_pages=16, stride_rows=32, devices="", load="", batch_window_ms=-1.0, max_batch=8,
                           payload_dir="", fde_module="tests.fake_fde_module")
    st = store_server.build_store(a)
    import tests.fake_fde_module as fm

    assert seen["fde_module"] is fm and st.fde_module is fm and st._fde_ext_cfg.kw["num_repetitions"] == 20
    a.fde_module = "no_such_fde_module_anywhere"
    with pytest.raises(ModuleNotFoundError):
        store_server.build_store(a)


def test_compaction_reclaims_slots_and_keeps_answers():
    rng = np.random.default_rng(7)
    chunks = sc.make_chunks(rng, n_docs=5, chunks_per_doc=3)
    s = _store(mode="float")
    sc.run(s.store_embeddings(chunks))
    sc.run(s.delete_chunks_by_document_id("doc1"))
    sc.run(s.store_embeddings([chunks[0]]))  # upsert of an existing (doc, chunk): the old slot is tombstoned
    keep = [c for c in chunks if c.document_id != "doc1"]
    before = [sc.run(s.query_similar(c.embedding, k=5)) for c in keep]
    assert s.compact() == 4 and s.compact() == 0
    assert len(s._require_index()) == len(keep)
    after = [sc.run(s.query_similar(c.embedding, k=5)) for c in keep]
    for b, a in zip(before, after):
        assert [(c.document_id, c.chunk_number, c.content) for c in a] == [(c.document_id, c.chunk_number, c.content) for c in b]
        assert [c.score for c in a] == [c.score for c in b]
    got = sc.run(s.get_chunks_by_id([("doc4", 2), ("doc1", 0)]))
    assert [(c.document_id, c.chunk_number) for c in got] == [("doc4", 2)]
    # the reclaimed capacity is usable again and filters still resolve
    more = [c.model_copy(update={"document_id": "new"}) for c in sc.make_chunks(rng, n_docs=1, chunks_per_doc=2)]
    sc.run(s.store_embeddings(more))
    assert {c.document_id for c in sc.run(s.query_similar(more[0].embedding, k=2, doc_ids=["new"]))} == {"new"}


def test_import_of_a_reference_npy_tree_into_the_store(tmp_path):
    """multivector/{document_id}/{chunk_number}.npy trees written by FastMultiVectorStore import page by page."""
    from morphik_core_amd import formats

    rng = np.random.default_rng(5)
    truth = {}
    for d in ("alpha", "beta"):
        for c in (0, 1, 3):
            e = rng.standard_normal((rng.integers(3, 20), 128)).astype(np.float32)
            e /= np.linalg.norm(e, axis=1, keepdims=True)
            p = tmp_path / "multivector" / d
            p.mkdir(parents=True, exist_ok=True)
            (p / f"{c}.npy").write_bytes(formats.save_npy_page(e))
            truth[(d, c)] = e
    s = _store(mode="float")
    assert formats.import_npy_tree_into_store(s, str(tmp_path), batch=4) == 6
    for (d, c), e in truth.items():
        hit = sc.run(s.query_similar(e, k=1))[0]
        assert (hit.document_id, hit.chunk_number) == (d, c)  # self-retrieval through the imported pages
    assert {c.document_id for c in sc.run(s.query_similar(truth[("beta", 1)], k=6, doc_ids=["beta"]))} == {"beta"}


def test_coalesced_requests_all_see_a_scan_failure():
    import asyncio

    s = _store(mode="float", batch_window_ms=5.0, max_batch=4)
    sc.run(s.store_embeddings(sc.make_chunks(np.random.default_rng(0), n_docs=1, chunks_per_doc=2)))

    def boom(*a, **k):
        raise RuntimeError("scan failed")

    s._index.query_batch = boom  # type: ignore[attr-defined]

    async def fire():
        return await asyncio.gather(*(s.query_similar(np.ones((3, 128), np.float32), k=1) for _ in range(3)), return_exceptions=True)

    res = sc.run(fire())
    assert len(res) == 3 and all(isinstance(r, RuntimeError) for r in res)  # query errors propagate to every waiter


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", ["float", "binary"])
def test_random_op_sequences_match_a_brute_force_model(seed, mode):
    sc.run(sc.scenario_random_ops_against_model(_store(mode=mode), seed=seed, n_ops=50, mode=mode, capacity=64))


def test_optional_min_score_cut():
    """The reference accepts min_score but never applies it to multivector hits (default None = same); a number cuts."""
    rng = np.random.default_rng(5)
    chunks = sc.make_chunks(rng, n_docs=2, chunks_per_doc=3)
    plain, cut = _store(mode="float"), _store(mode="float", min_score=0.0)
    for st in (plain, cut):
        sc.run(st.store_embeddings(chunks))
    res = sc.run(plain.query_similar(chunks[0].embedding, k=6))
    assert len(res) == 6
    cut.min_score = (res[1].score + res[2].score) / 2
    got = sc.run(cut.query_similar(chunks[0].embedding, k=6))
    assert [(r.document_id, r.chunk_number) for r in got] == [(r.document_id, r.chunk_number) for r in res[:2]]


# --------------------------------------------------------------------------- payloads through .storage
class MemStorage:
    """BaseStorage stand-in (core/storage/base_storage.py): the three coroutines the stores use."""

    def __init__(self):
        self.objects, self.uploads, self.downloads, self.fail_upload = {}, 0, 0, False

    async def upload_from_base64(self, content, key, content_type=None, bucket=""):
        import base64

        if self.fail_upload:
            raise RuntimeError("storage down")
        self.objects[(bucket, key)] = (base64.b64decode(content), content_type)
        self.uploads += 1
        return bucket, key

    async def download_file(self, bucket, key):
        self.downloads += 1
        return self.objects[(bucket, key)][0]

    async def delete_file(self, bucket, key):
        return self.objects.pop((bucket, key), None) is not None


def _png_data_uri(n=200):
    import base64

    return "data:image/png;base64," + base64.b64encode(b"\x89PNG\r\n\x1a\n" + bytes(range(256)) * (n // 256 + 1)).decode()


def test_payloads_live_in_storage_and_skip_image_content_returns_the_key():
    """multi_vector_store.py:650-699 (upload, keep the key), :778-817 / :866-919 (download on hit unless the caller skips image
    payloads), :921-946 (delete the objects with the document)."""
    from morphik_core_amd.models import DocumentChunk
    from morphik_core_amd.payloads import MULTIVECTOR_CHUNKS_BUCKET

    st = MemStorage()
    s = _store(mode="float", storage=st)
    rng = np.random.default_rng(9)
    img = _png_data_uri()
    chunks = [DocumentChunk(document_id="d", chunk_number=0, content=img, embedding=sc.rand_emb(rng, 9), metadata={"is_image": True}),
              DocumentChunk(document_id="d", chunk_number=1, content="plain text of page 2", embedding=sc.rand_emb(rng, 9), metadata={"is_image": False})]
    ok, ids, m = sc.run(s.store_embeddings(chunks, app_id="app1"))
    assert ok and m["chunk_payload_objects"] == 2 and m["chunk_payload_bytes"] > 200 and m["chunk_payload_backend"] == "memstorage"
    assert set(st.objects) == {(MULTIVECTOR_CHUNKS_BUCKET, "app1/d/0.png"), (MULTIVECTOR_CHUNKS_BUCKET, "app1/d/1.txt")}
    assert st.objects[(MULTIVECTOR_CHUNKS_BUCKET, "app1/d/1.txt")] == (b"plain text of page 2", "text/plain")
    assert all(len(r[2]) < 64 for r in s._rows.values())  # only keys in RAM
    hit = sc.run(s.query_similar(chunks[0].embedding, k=2))
    assert [h.chunk_number for h in hit] == [0, 1] and hit[0].content == img and hit[1].content == "plain text of page 2"
    n_dl = st.downloads
    skipped = sc.run(s.query_similar(chunks[0].embedding, k=2, skip_image_content=True))
    assert skipped[0].content == "app1/d/0.png" and skipped[1].content == "plain text of page 2" and st.downloads == n_dl + 1
    got = sc.run(s.get_chunks_by_id([("d", 0)], skip_image_content=True))
    assert got[0].content == "app1/d/0.png"
    assert sc.run(s.get_chunks_by_id([("d", 0)]))[0].content == img
    assert sc.run(s.delete_chunks_by_document_id("d")) is True and st.objects == {}
    # more uploads than the concurrency limit, from a SECOND event loop (asyncio.run per call): the limiter is per loop
    many = [DocumentChunk(document_id="bulk", chunk_number=i, content=f"text {i}", embedding=sc.rand_emb(rng, 4), metadata={}) for i in range(40)]
    ok, ids, m = sc.run(s.store_embeddings(many, app_id="app1"))
    assert ok and m["chunk_payload_objects"] == 40 and len(st.objects) == 40
    assert sc.run(s.delete_chunks_by_document_id("bulk")) is True and st.objects == {}
    # a failing upload keeps the content inline (the reference's database-column fallback) and the store keeps working
    st.fail_upload = True
    ok, ids, m = sc.run(s.store_embeddings([chunks[1]], app_id="app1"))
    assert ok and m["chunk_payload_objects"] == 0
    assert sc.run(s.query_similar(chunks[1].embedding, k=1))[0].content == "plain text of page 2"


def test_content_that_looks_like_another_tenants_key_is_just_text():
    """store_embeddings must not take a chunk's text for a storage key because it LOOKS like one.  Keys are
    predictable (`<app>/<doc>/<n>.txt`): app-b ingests a chunk whose text is app-a's key.  The reference always uploads on
    store and never trusts content as a key (multi_vector_store.py:650-676): app-b gets its own text back, app-a's payload is
    neither returned to app-b nor deleted with app-b's document."""
    from morphik_core_amd.models import DocumentChunk
    from morphik_core_amd.payloads import MULTIVECTOR_CHUNKS_BUCKET
    from morphik_core_amd.store import ROW_OWN_KEY, row_origin

    st = MemStorage()
    s = _store(mode="float", storage=st)
    rng = np.random.default_rng(21)
    victim = DocumentChunk(document_id="doc", chunk_number=0, content="SECRET of app-a", embedding=sc.rand_emb(rng, 6), metadata={})
    sc.run(s.store_embeddings([victim], app_id="app-a"))
    victim_key = (MULTIVECTOR_CHUNKS_BUCKET, "app-a/doc/0.txt")
    assert st.objects[victim_key][0] == b"SECRET of app-a"
    evil = DocumentChunk(document_id="mine", chunk_number=0, content="app-a/doc/0.txt", embedding=sc.rand_emb(rng, 6), metadata={})
    ok, _ids, m = sc.run(s.store_embeddings([evil], app_id="app-b"))
    assert ok and m["chunk_payload_objects"] == 1  # uploaded like any other content
    assert all(row_origin(r) == ROW_OWN_KEY for r in s._rows.values())
    got = sc.run(s.get_chunks_by_id([("mine", 0)], app_id="app-b"))
    assert got[0].content == "app-a/doc/0.txt"  # its own text, not the victim's payload
    hit = sc.run(s.query_similar(evil.embedding, k=1, app_id="app-b"))
    assert hit[0].content == "app-a/doc/0.txt"
    assert sc.run(s.delete_chunks_by_document_id("mine", app_id="app-b")) is True
    assert st.objects[victim_key][0] == b"SECRET of app-a"  # still there
    assert sc.run(s.get_chunks_by_id([("doc", 0)], app_id="app-a"))[0].content == "SECRET of app-a"
    # without a storage object the same text is inline content: nothing is dereferenced, nothing is reported as a key
    s2 = _store(mode="float")
    sc.run(s2.store_embeddings([evil], app_id="app-b"))
    assert sc.run(s2.get_chunks_by_id([("mine", 0)], app_id="app-b"))[0].content == "app-a/doc/0.txt"
    ok, left = sc.run(s2.delete_chunks_returning_keys("mine", app_id="app-b"))
    assert ok and left == []
    # a checkpoint written before the origin column existed keeps its old reading (key-shaped content = a key: this store's when it
    # has storage, a remote client's otherwise -- test_checkpoint_without_the_origin_column_keeps_key_shaped_rows_as_keys)
    from morphik_core_amd.store import ROW_LEGACY_KEY

    assert row_origin(("d", 0, "app/d/0.txt", "{}", "app")) == ROW_LEGACY_KEY and row_origin(("d", 0, "some text", "{}", "app")) == 0


def test_owner_with_its_own_storage_hands_a_clients_keys_back_and_never_dereferences_them():
    """ADVICE r4 (high, second half): the remote client flags the keys IT uploaded (`content_is_key`); the owner -- even one with
    a storage object of its own -- stores them as the client's, never downloads or deletes them itself, flags them on the way
    back, and returns them on delete so the client removes its objects.  An unflagged key-shaped text from a client is content."""
    from morphik_core_amd.store_server import MI355XRemoteMultiVectorStore, create_app
    from tests.test_encoder_and_formats import _serve

    owner = MI355XFastMultiVectorStore(capacity_pages=16, stride_rows=32, mode="float", index_factory=OracleIndex, storage=MemStorage())
    assert owner.initialize()
    url, stop = _serve(create_app(owner))
    try:
        st = MemStorage()
        remote = MI355XRemoteMultiVectorStore(url, storage=st)
        ch = sc.make_chunks(np.random.default_rng(31), n_docs=1, chunks_per_doc=2)
        for c in ch:
            c.metadata = {"is_image": False}
        sc.run(remote.store_embeddings(ch, app_id="t"))
        assert st.uploads == 2 and owner.storage.uploads == 0 and owner.storage.downloads == 0
        assert sc.run(remote.query_similar(ch[1].embedding, k=1, app_id="t"))[0].content == ch[1].content
        assert owner.storage.downloads == 0 and st.downloads == 1
        assert sc.run(remote.delete_chunks_by_document_id(ch[0].document_id, app_id="t")) is True
        assert st.objects == {}  # the client removed ITS objects: the owner handed the keys back
        # a client WITHOUT storage sends text that looks like a key: the owner uploads it as content and returns it as content
        plain = MI355XRemoteMultiVectorStore(url)
        odd = sc.make_chunks(np.random.default_rng(32), n_docs=1, chunks_per_doc=1)
        odd[0].content, odd[0].metadata = "t/other-doc/0.txt", {"is_image": False}
        sc.run(plain.store_embeddings(odd, app_id="t"))
        assert owner.storage.uploads == 1
        assert sc.run(plain.query_similar(odd[0].embedding, k=1, app_id="t"))[0].content == "t/other-doc/0.txt"
    finally:
        stop()


def test_fde_vectors_of_the_wrong_width_stop_at_the_boundary():
    """ADVICE r4 (medium / low): an fde_module whose output width differs from the index's FDE must raise, not over-read a host
    buffer in the *_fde C entry points or regroup document vectors into another number of pages; a failed import leaves no
    live page behind."""
    from morphik_core_amd.index import _fde_block

    with pytest.raises(ValueError):
        _fde_block(np.zeros(100, np.float32), 1, 10240)
    with pytest.raises(ValueError):
        _fde_block(np.zeros((2, 10240), np.float32), 3, 10240)
    assert _fde_block(np.zeros(20480, np.float32), 2, 10240).shape == (2, 10240)

    class Boom(OracleIndex):
        def import_fde(self, page0, fde):
            raise RuntimeError("import failed")

    import tests.fake_fde_module as fm

    s = MI355XFastMultiVectorStore(capacity_pages=16, stride_rows=32, mode="fde_then_float", index_factory=Boom, fde_module=fm)
    assert s.initialize()
    ch = sc.make_chunks(np.random.default_rng(41), n_docs=1, chunks_per_doc=2)
    with pytest.raises(RuntimeError):
        sc.run(s.store_embeddings(ch, app_id="t"))
    assert len(s) == 0 and not any(s._index.alive)  # the appended pages were retired, no bookkeeping kept


# --------------------------------------------------------------------------- compaction vs queries, upsert order
def test_query_that_overlaps_a_compaction_is_rerun_against_the_new_numbering():
    rng = np.random.default_rng(11)
    chunks = sc.make_chunks(rng, n_docs=4, chunks_per_doc=3)
    s = _store(mode="float")
    sc.run(s.store_embeddings(chunks))
    sc.run(s.delete_chunks_by_document_id("doc0"))  # pages 0..2 tombstoned: compaction will renumber everything else
    want = sc.run(s.query_similar(chunks[7].embedding, k=3))
    real = s._index.query
    calls = []

    def racing_query(q, k, **kw):
        out = real(q, k, **kw)
        if not calls:  # a compaction lands between this query's scan and its id lookup
            calls.append(1)
            assert s.compact() == 3
        else:
            calls.append(2)
        return out

    s._index.query = racing_query
    got = sc.run(s.query_similar(chunks[7].embedding, k=3))
    assert calls == [1, 2]  # the stale scan was thrown away and repeated
    assert [(c.document_id, c.chunk_number, c.content, c.score) for c in got] == [(c.document_id, c.chunk_number, c.content, c.score) for c in want]


def test_failed_upsert_keeps_the_previous_version_and_ordinals_are_reclaimed():
    rng = np.random.default_rng(12)
    s = MI355XMultiVectorStore(capacity_pages=4, stride_rows=32, mode="float", index_factory=OracleIndex)
    chunks = sc.make_chunks(rng, n_docs=1, chunks_per_doc=4)
    sc.run(s.store_embeddings(chunks))
    with pytest.raises(Exception):  # slab full: the add fails BEFORE anything is tombstoned
        sc.run(s.store_embeddings([chunks[1].model_copy(update={"content": "v2"})]))
    assert sc.run(s.query_similar(chunks[1].embedding, k=1))[0].content == chunks[1].content
    assert len(s) == 4
    sc.run(s.delete_chunks_by_document_id("doc0"))
    assert s.compact() == 4 and s._doc_ord == {} and s._doc_app == {}  # no page left: the ordinal is forgotten
    sc.run(s.store_embeddings(chunks[:2]))
    assert len(sc.run(s.query_similar(chunks[0].embedding, k=5, doc_ids=["doc0"]))) == 2


def test_checkpoint_generations_survive_an_interrupted_save_and_do_not_hold_the_store_lock(tmp_path, monkeypatch):
    """a crash / kill during a (periodic) save must leave the PREVIOUS checkpoint loadable -- the owner process holds
    the only copy of the corpus -- and the slab dump must not run under the store lock the event loop's query paths take."""
    import os
    import threading
    import time

    rng = np.random.default_rng(3)
    s = MI355XFastMultiVectorStore(capacity_pages=32, stride_rows=32, mode="float", index_factory=OracleIndex)
    chunks = sc.make_chunks(rng, n_docs=3, chunks_per_doc=2)
    sc.run(s.store_embeddings(chunks[:4], app_id="t"))
    d = str(tmp_path / "ckpt")
    s.save(d)
    gen1 = open(os.path.join(d, "CURRENT")).read()
    assert sorted(os.listdir(d)) == sorted(["CURRENT", gen1]) and MI355XFastMultiVectorStore.checkpoint_path(d) == os.path.join(d, gen1)
    sc.run(s.store_embeddings(chunks[4:], app_id="t"))
    # (1) the index dump dies half way (kill -9, OOM, disk full): CURRENT still names generation 1, which still loads
    real_save = OracleIndex.save

    def dying_save(self, path):
        with open(path + ".tmp", "wb") as f:
            f.write(b"half a slab")
        raise OSError("killed mid-dump")

    monkeypatch.setattr(OracleIndex, "save", dying_save)
    with pytest.raises(OSError):
        s.save(d)
    assert open(os.path.join(d, "CURRENT")).read() == gen1 and sorted(os.listdir(d)) == sorted(["CURRENT", gen1])
    back = MI355XFastMultiVectorStore.load(d, index_factory=OracleIndex)
    assert len(back) == 4 and sc.run(back.query_similar(chunks[1].embedding, k=1, app_id="t"))[0].content == chunks[1].content
    monkeypatch.setattr(OracleIndex, "save", real_save)
    # (2) everything written, the process dies before CURRENT is switched: still generation 1
    real_replace = os.replace

    def no_switch(a, b):
        if os.path.basename(b) == "CURRENT":
            raise OSError("killed before the switch")
        return real_replace(a, b)

    monkeypatch.setattr(os, "replace", no_switch)
    with pytest.raises(OSError):
        s.save(d)
    monkeypatch.setattr(os, "replace", real_replace)
    assert open(os.path.join(d, "CURRENT")).read() == gen1
    assert len(MI355XFastMultiVectorStore.load(d, index_factory=OracleIndex)) == 4
    # (3) a save that completes replaces the generation; the store lock is free while the slabs are being written
    free_during_dump = []

    def slow_save(self, path):
        time.sleep(0.3)
        return real_save(self, path)

    monkeypatch.setattr(OracleIndex, "save", slow_save)
    t = threading.Thread(target=s.save, args=(d,))
    t.start()
    time.sleep(0.1)
    got = s._lock.acquire(timeout=0.05)  # what query_similar / get_chunks_by_id take on the event loop
    free_during_dump.append(got)
    if got:
        s._lock.release()
    res = sc.run(s.query_similar(chunks[5].embedding, k=1, app_id="t"))  # served while the dump is in flight
    assert t.is_alive() and res[0].content == chunks[5].content
    t.join()
    assert free_during_dump == [True]
    gen2 = open(os.path.join(d, "CURRENT")).read()
    assert gen2 != gen1 and sorted(os.listdir(d)) == sorted(["CURRENT", gen2])
    back = MI355XFastMultiVectorStore.load(d, index_factory=OracleIndex)
    assert len(back) == 6 and sc.run(back.query_similar(chunks[5].embedding, k=1, app_id="t"))[0].content == chunks[5].content
    # a store_embeddings that arrives during a save waits for it (writers are gated) and lands in the NEXT checkpoint
    monkeypatch.setattr(OracleIndex, "save", real_save)


# --------------------------------------------------------------------------- sharded store (host logic; GPU: test_gpu_store.py)
def _sharded(R, **kw):
    from morphik_core_amd.store import MI355XShardedMultiVectorStore
    from tests.fake_index import OracleComm

    s = MI355XShardedMultiVectorStore(devices=[0] * R, capacity_pages=64, stride_rows=32, index_factory=OracleIndex, comm_factory=OracleComm, **kw)
    assert s.initialize() is True
    return s


@pytest.mark.parametrize("R", [1, 2, 4])
@pytest.mark.parametrize("scenario", sc.ALL, ids=lambda f: f.__name__)
def test_sharded_store_passes_the_reference_scenarios(scenario, R):
    sc.run(scenario(_sharded(R, mode="float")))


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("mode", ["float", "binary"])
def test_sharded_store_random_ops_match_the_model(R, mode):
    sc.run(sc.scenario_random_ops_against_model(_sharded(R, mode=mode), seed=R, n_ops=50, mode=mode, capacity=40))


def test_sharded_index_batch_merges_per_shard_topk_like_the_communicator():
    """ShardedIndex.query_batch (single-stage modes): every shard serves the whole batch, the per-shard lists are merged with
    the communicator's rule -> exactly the communicator's (= single index) answers, ties and per-request filters included."""
    from morphik_core_amd.index import allow_bitmap
    from morphik_core_amd.shard_index import ShardedIndex
    from tests.fake_index import OracleComm

    rng = np.random.default_rng(11)
    ix = ShardedIndex(capacity_pages=48, stride_rows=16, devices=[0, 0, 0], index_cls=OracleIndex, comm_cls=OracleComm)
    base = [sc.rand_emb(rng, 6) for _ in range(8)]
    for b in range(6):  # six batches of five pages; duplicates give exact ties across shards
        ix.add([base[(b + i) % 8] for i in range(5)], doc_ordinals=[(b * 5 + i) % 7 for i in range(5)])
    queries = [sc.rand_emb(rng, 4) for _ in range(5)]
    allows = [None, allow_bitmap([0, 2, 4]), allow_bitmap([1, 3]), None, allow_bitmap([6])]
    for mode in ("float", "binary"):
        got = ix.query_batch(queries, 7, mode=mode, allows=allows, n_docs=7)
        for (s, i), q, a in zip(got, queries, allows):
            ws, wi = ix.query(q, 7, mode=mode, allow=a)
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), mode
    ix.close()


def test_sharded_index_routes_imported_fde_vectors_and_query_fdes_to_the_shards():
    """ShardedIndex.import_fde (global page ids of one add() -> the owning shard) and q_fde / q_fdes through query and query_batch:
    with caller-supplied vectors on every page the coarse ("fde") scores are dot products the test computes itself."""
    from morphik_core_amd.shard_index import ShardedIndex
    from oracle import oracle as orc
    from tests.fake_index import OracleComm

    class fde_index(OracleIndex):
        def __init__(self, *a, **kw):
            kw["fde"] = orc.FdeConfig.reference_default()
            super().__init__(*a, **kw)

    rng = np.random.default_rng(13)
    ix = ShardedIndex(capacity_pages=30, stride_rows=16, devices=[0, 0, 0], index_cls=fde_index, comm_cls=OracleComm, with_float=True, with_fde=True)
    D = 10240
    docs = {}
    for b in range(5):
        first = ix.add([sc.rand_emb(rng, 5) for _ in range(4)], doc_ordinals=[b] * 4)
        vec = rng.standard_normal((4, D)).astype(np.float32)
        ix.import_fde(first, vec)
        for i in range(4):
            docs[first + i] = vec[i]
    queries = [sc.rand_emb(rng, 4) for _ in range(3)]
    qf = rng.standard_normal((3, D)).astype(np.float32)
    ids = sorted(docs)
    db = orc.bf16_to_f32(orc.f32_to_bf16(np.stack([docs[i] for i in ids])))
    want = (db @ qf.T) / np.linalg.norm(db, axis=1, keepdims=True)
    for j, q in enumerate(queries):
        s, i = ix.query(q, 20, mode="fde", q_fde=qf[j])
        assert sorted(i.tolist()) == ids
        np.testing.assert_allclose(s, want[[ids.index(int(x)) for x in i], j], rtol=2e-3, atol=1e-4)
    for j, (s, i) in enumerate(ix.query_batch(queries, 5, mode="fde", q_fdes=qf)):
        ws, wi = ix.query(queries[j], 5, mode="fde", q_fde=qf[j])
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    ix.close()


def test_sharded_store_equals_single_store_and_spreads_the_pages():
    rng = np.random.default_rng(21)
    chunks = sc.make_chunks(rng, n_docs=6, chunks_per_doc=3)
    one, four = _store(mode="float"), _sharded(4, mode="float")
    for st in (one, four):
        for d in range(6):  # one store_embeddings call per document, like the ingestion worker
            sc.run(st.store_embeddings(chunks[d * 3 : d * 3 + 3]))
    assert [len(sh) for sh in four._index.shards] == [6, 6, 3, 3]  # least-full routing
    for c in chunks[::4]:
        for filt in (None, ["doc1", "doc4"]):
            a = sc.run(one.query_similar(c.embedding, k=7, doc_ids=filt))
            b = sc.run(four.query_similar(c.embedding, k=7, doc_ids=filt))
            assert [(x.document_id, x.chunk_number, x.content) for x in a] == [(x.document_id, x.chunk_number, x.content) for x in b]
            assert [x.score for x in a] == [x.score for x in b]


# --------------------------------------------------------------------------- one slab, many processes: owner server + remote store
def test_remote_store_passes_the_reference_scenarios_through_the_owner_server():
    """the ingestion worker builds its own store object (ingestion_worker.py:102-142); with the store being
    HBM, every process but one uses MI355XRemoteMultiVectorStore, which forwards the four coroutines to the owner over HTTP.
    The reference's store scenarios must pass through that hop unchanged."""
    from morphik_core_amd.store_server import MI355XRemoteMultiVectorStore, create_app
    from tests.test_encoder_and_formats import _serve

    for scenario in sc.ALL:
        owner = _store(mode="float")
        url, stop = _serve(create_app(owner, api_key="s3"))
        try:
            remote = MI355XRemoteMultiVectorStore(url, api_key="s3")
            assert remote.initialize() is True
            sc.run(scenario(remote))
        finally:
            stop()
    assert MI355XRemoteMultiVectorStore("http://127.0.0.1:9", api_key="x").initialize() is False  # nobody there: False, no raise


def test_remote_store_two_clients_see_each_others_pages_and_errors_propagate():
    from morphik_core_amd.store_server import MI355XRemoteMultiVectorStore, create_app
    from tests.test_encoder_and_formats import _serve

    owner = MI355XFastMultiVectorStore(capacity_pages=8, stride_rows=32, mode="float", index_factory=OracleIndex)
    assert owner.initialize()
    url, stop = _serve(create_app(owner))
    try:
        worker, api = MI355XRemoteMultiVectorStore(url), MI355XRemoteMultiVectorStore(url)
        rng = np.random.default_rng(3)
        chunks = sc.make_chunks(rng, n_docs=2, chunks_per_doc=3)
        ok, ids, metrics = sc.run(worker.store_embeddings(chunks, app_id="tenant"))  # the ingestion worker's process
        assert ok and len(ids) == 6 and metrics["vector_store_rows"] == 6
        hit = sc.run(api.query_similar(chunks[4].embedding, k=2, app_id="tenant"))  # the API server's process
        assert (hit[0].document_id, hit[0].chunk_number, hit[0].content, hit[0].metadata) == ("doc1", 1, chunks[4].content, chunks[4].metadata)
        assert sc.run(api.query_similar(chunks[4].embedding, k=2, app_id="other")) == []
        with pytest.raises(RuntimeError, match="slab full"):  # the owner's exception text reaches the caller
            sc.run(worker.store_embeddings(sc.make_chunks(rng, n_docs=1, chunks_per_doc=4), app_id="tenant"))
        with pytest.raises(RuntimeError, match="401"):
            stop()
            url2, stop = _serve(create_app(owner, api_key="k"))
            sc.run(MI355XRemoteMultiVectorStore(url2, api_key="wrong").query_similar(chunks[0].embedding, k=1))
    finally:
        stop()


def test_remote_client_with_its_own_storage_keeps_payloads_off_the_wire():
    """MI355XRemoteMultiVectorStore(storage=...) uploads chunk payloads through ITS storage object (as the local
    stores do, multi_vector_store.py:650-676); the owner only ever sees storage keys, hits are resolved on the client, and
    image payloads stay keys when the caller asks for skip_image_content."""
    import base64

    from morphik_core_amd.store_server import MI355XRemoteMultiVectorStore, create_app
    from tests.test_encoder_and_formats import _serve

    owner = MI355XFastMultiVectorStore(capacity_pages=16, stride_rows=32, mode="float", index_factory=OracleIndex)
    assert owner.initialize()
    url, stop = _serve(create_app(owner))
    try:
        st = MemStorage()
        remote = MI355XRemoteMultiVectorStore(url, storage=st)
        rng = np.random.default_rng(12)
        chunks = sc.make_chunks(rng, n_docs=2, chunks_per_doc=2)
        img = "data:image/png;base64," + base64.b64encode(b"\x89PNG\r\n\x1a\n" + b"x" * 5000).decode()
        chunks[1].content, chunks[1].metadata = img, {"is_image": True}
        chunks[2].metadata = {"is_image": False}
        ok, ids, _m = sc.run(remote.store_embeddings(chunks, app_id="t"))
        assert ok and len(ids) == 4 and st.uploads == 4
        assert all(len(r[2]) < 100 and "/" in r[2] for r in owner._rows.values())  # the owner holds keys only
        hit = sc.run(remote.query_similar(chunks[1].embedding, k=1, app_id="t"))
        assert hit[0].content == img  # resolved through the client's storage
        hit = sc.run(remote.query_similar(chunks[1].embedding, k=1, app_id="t", skip_image_content=True))
        assert hit[0].content.endswith(".png") and "/" in hit[0].content  # the key itself
        got = sc.run(remote.get_chunks_by_id([(chunks[2].document_id, chunks[2].chunk_number)], app_id="t"))
        assert got[0].content == chunks[2].content
        # deleting a document removes ITS payload objects from the client's storage too (the owner has no storage
        # object and hands the keys back: multi_vector_store.py:921-951 deletes storage objects on delete) ...
        assert len(st.objects) == 4
        assert sc.run(remote.delete_chunks_by_document_id(chunks[0].document_id, app_id="t")) is True
        assert len(st.objects) == 2 and sc.run(remote.query_similar(chunks[0].embedding, k=4, app_id="t"))[0].document_id == chunks[2].document_id
        # ... a store_embeddings the owner refuses leaves no orphaned uploads behind ...
        too_long = sc.make_chunks(rng, n_docs=1, chunks_per_doc=1)
        too_long[0].embedding = np.ones((40, 128), np.float32)  # stride_rows is 32
        with pytest.raises(RuntimeError):
            sc.run(remote.store_embeddings(too_long, app_id="t"))
        assert len(st.objects) == 2
    finally:
        stop()
    # ... and an owner that HAS a payload store keeps a client's key as the key (no second upload of the key string as "content")
    owner2 = MI355XFastMultiVectorStore(capacity_pages=16, stride_rows=32, mode="float", index_factory=OracleIndex, storage=MemStorage())
    assert owner2.initialize()
    url2, stop2 = _serve(create_app(owner2))
    try:
        st2 = MemStorage()
        remote2 = MI355XRemoteMultiVectorStore(url2, storage=st2)
        ch = sc.make_chunks(np.random.default_rng(13), n_docs=1, chunks_per_doc=2)
        ch[0].content, ch[0].metadata = _png_data_uri(), {"is_image": True}
        sc.run(remote2.store_embeddings(ch, app_id="t"))
        assert owner2.storage.uploads == 0 and st2.uploads == 2
        hit = sc.run(remote2.query_similar(ch[0].embedding, k=1, app_id="t"))
        assert hit[0].content == ch[0].content
    finally:
        stop2()


def test_checkpoint_without_the_origin_column_keeps_key_shaped_rows_as_keys(tmp_path):
    """rows of checkpoints written before round 5 carry no origin.  Those builds treated key-shaped content as a key --
    this store's own when it has storage, a remote client's otherwise (flagged on the wire, handed back on delete).  After loading
    such a checkpoint the same must hold: an owner WITHOUT storage flags the rows as client keys and returns them on delete; an owner
    WITH storage dereferences and deletes them itself."""
    import json
    import os

    from morphik_core_amd.store import ROW_LEGACY_KEY, row_origin

    rng = np.random.default_rng(21)
    chunks = sc.make_chunks(rng, n_docs=2, chunks_per_doc=2)
    st = MemStorage()
    s = MI355XFastMultiVectorStore(capacity_pages=16, stride_rows=32, mode="float", index_factory=OracleIndex, storage=st)
    assert s.initialize()
    sc.run(s.store_embeddings(chunks, app_id="t"))
    assert st.uploads == 4
    d = str(tmp_path / "old")
    s.save(d)
    gdir = MI355XFastMultiVectorStore.checkpoint_path(d)
    with open(os.path.join(gdir, "store.json")) as f:
        book = json.load(f)
    book["rows"] = [r[:6] for r in book["rows"]]  # the pre-round-5 layout: [page, doc, chunk, content, meta, app]
    assert all(len(r) == 6 for r in book["rows"])
    with open(os.path.join(gdir, "store.json"), "w") as f:
        json.dump(book, f)
    # (a) owner WITHOUT a storage object (the remote client holds the payloads): rows are flagged as keys, keys come back on delete
    bare = MI355XFastMultiVectorStore.load(d, index_factory=OracleIndex)
    assert all(row_origin(r) == ROW_LEGACY_KEY for r in bare._rows.values())
    ids = [(c.document_id, c.chunk_number) for c in chunks]
    assert bare.content_key_flags(ids, app_id="t") == [True] * 4
    ok, keys = sc.run(bare.delete_chunks_returning_keys(chunks[0].document_id, app_id="t"))
    assert ok and len(keys) == 2 and all(k in {kk for _b, kk in st.objects} for k in keys)
    # (b) owner WITH the storage object: contents are dereferenced, delete removes the objects, nothing is flagged for a client
    own = MI355XFastMultiVectorStore.load(d, index_factory=OracleIndex, storage=st)
    assert own.content_key_flags(ids, app_id="t") == [False] * 4
    hit = sc.run(own.query_similar(chunks[1].embedding, k=1, app_id="t"))
    want = sc.run(s.query_similar(chunks[1].embedding, k=1, app_id="t"))  # the store that wrote the checkpoint (rows WITH origins)
    assert hit[0].content == want[0].content and hit[0].content != own._rows[own._page_of[(own._nk(chunks[1].document_id, "t"), chunks[1].chunk_number)]][2]
    n_before = len(st.objects)
    assert sc.run(own.delete_chunks_by_document_id(chunks[2].document_id, app_id="t")) is True
    assert len(st.objects) == n_before - 2
    # a fresh save writes the origin column (legacy rows keep their marker)
    own.save(str(tmp_path / "new"))
    with open(os.path.join(MI355XFastMultiVectorStore.checkpoint_path(str(tmp_path / "new")), "store.json")) as f:
        assert all(len(r) == 7 and r[6] == ROW_LEGACY_KEY for r in json.load(f)["rows"])
