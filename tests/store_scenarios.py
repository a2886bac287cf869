"""The reference's store tests (core/tests/unit/test_multivector.py) re-expressed against the MI355X
stores; run on the CPU with an oracle-backed index (host logic) and on the GPU with the real one."""
import asyncio

import numpy as np

from morphik_core_amd.models import DocumentChunk


def run(coro):
    return asyncio.run(coro)


def rand_emb(rng, n, d=128):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def make_chunks(rng, n_docs=3, chunks_per_doc=4, rows=24):
    out = []
    for d in range(n_docs):
        for c in range(chunks_per_doc):
            out.append(DocumentChunk(document_id=f"doc{d}", chunk_number=c, content=f"content {d}/{c}",
                                     embedding=rand_emb(rng, rows), metadata={"page": c, "doc": d, "is_image": bool(c % 2)}))
    return out


async def scenario_store_query_roundtrip(store):
    """test_multivector.py:152-181: store, self-query ranks first, scores non-increasing, embedding=[]"""
    rng = np.random.default_rng(1)
    chunks = make_chunks(rng)
    ok, ids, metrics = await store.store_embeddings(chunks)
    assert ok is True and ids == [f"{c.document_id}-{c.chunk_number}" for c in chunks]
    for key in ("chunk_payload_upload_s", "chunk_payload_objects", "chunk_payload_bytes", "chunk_payload_backend",
                "multivector_upload_s", "multivector_objects", "multivector_bytes", "multivector_backend",
                "vector_store_write_s", "vector_store_backend", "vector_store_rows", "cache_write_s", "cache_write_objects"):
        assert key in metrics
    assert metrics["vector_store_rows"] == len(chunks)
    target = chunks[5]
    res = await store.query_similar(target.embedding, k=4)
    assert len(res) == 4
    assert (res[0].document_id, res[0].chunk_number) == (target.document_id, target.chunk_number)
    assert all(res[i].score >= res[i + 1].score for i in range(len(res) - 1))
    assert all(r.embedding == [] and isinstance(r.score, float) for r in res)
    assert res[0].content == target.content and res[0].metadata == target.metadata  # :259-294 metadata roundtrip
    big = await store.query_similar(target.embedding, k=100)  # k may exceed N
    assert len(big) == len(chunks)


async def scenario_doc_filter(store):
    """test_multivector.py:184-202: results are a subset of doc_ids; falsy doc_ids = no filter"""
    rng = np.random.default_rng(2)
    chunks = make_chunks(rng)
    await store.store_embeddings(chunks)
    q = rand_emb(rng, 8)
    res = await store.query_similar(q, k=50, doc_ids=["doc1"])
    assert len(res) == 4 and {r.document_id for r in res} == {"doc1"}
    assert len(await store.query_similar(q, k=50, doc_ids=[])) == len(chunks)
    assert await store.query_similar(q, k=50, doc_ids=["nope"]) == []
    res = await store.query_similar(q, k=50, doc_ids=["doc0", "doc2", "nope"])
    assert {r.document_id for r in res} == {"doc0", "doc2"}


async def scenario_empty_and_missing(store):
    """test_multivector.py:205-211 + chunks without embeddings are skipped (:631-635)"""
    ok, ids, metrics = await store.store_embeddings([])
    assert ok is True and ids == [] and metrics["vector_store_rows"] == 0
    rng = np.random.default_rng(3)
    assert await store.query_similar(rand_emb(rng, 4), k=3) == []
    c = DocumentChunk(document_id="d", chunk_number=0, content="x", embedding=None, metadata={})
    ok, ids, _ = await store.store_embeddings([c])
    assert ok is True and ids == []
    assert await store.get_chunks_by_id([]) == []


async def scenario_known_ranking(store, exact_binary):
    """test_multivector.py:214-256: doc1 = 3 x (+1*64,-1*64), doc2 = negation, query = the half pattern"""
    half = np.concatenate([np.ones(64), -np.ones(64)]).astype(np.float32)
    await store.store_embeddings([
        DocumentChunk(document_id="doc1", chunk_number=0, content="a", embedding=np.stack([half] * 3), metadata={}),
        DocumentChunk(document_id="doc2", chunk_number=0, content="b", embedding=np.stack([-half] * 3), metadata={}),
    ])
    res = await store.query_similar(half[None, :], k=2)
    assert [r.document_id for r in res] == ["doc1", "doc2"]
    if exact_binary:
        assert [r.score for r in res] == [1.0, 0.0]  # SQL max_sim values


async def scenario_get_delete_upsert(store):
    rng = np.random.default_rng(4)
    chunks = make_chunks(rng, n_docs=2, chunks_per_doc=3)
    await store.store_embeddings(chunks)
    got = await store.get_chunks_by_id([("doc1", 2), ("doc0", 0), ("doc1", 2), ("zzz", 9)])
    assert [(g.document_id, g.chunk_number) for g in got] == [("doc1", 2), ("doc0", 0)]
    assert all(g.score == 0.0 and g.embedding == [] for g in got)
    assert await store.delete_chunks_by_document_id("doc0") is True
    assert await store.delete_chunks_by_document_id("never-stored") is True
    res = await store.query_similar(chunks[0].embedding, k=10)
    assert len(res) == 3 and all(r.document_id == "doc1" for r in res)
    assert await store.get_chunks_by_id([("doc0", 0)]) == []
    # upsert replaces the page
    new = DocumentChunk(document_id="doc1", chunk_number=1, content="replaced", embedding=rand_emb(rng, 7), metadata={"v": 2})
    await store.store_embeddings([new])
    res = await store.query_similar(new.embedding, k=10)
    assert len(res) == 3 and res[0].content == "replaced" and res[0].metadata == {"v": 2}


async def scenario_input_tolerance(store):
    """torch tensors / lists of tensors / nested lists are accepted (multi_vector_store.py:334-337)"""
    import torch

    rng = np.random.default_rng(5)
    e = rand_emb(rng, 10)
    await store.store_embeddings([
        DocumentChunk(document_id="t", chunk_number=0, content="t0", embedding=torch.from_numpy(e), metadata={}),
        DocumentChunk(document_id="t", chunk_number=1, content="t1", embedding=rand_emb(rng, 10).tolist(), metadata={}),
    ])
    for q in (torch.from_numpy(e), [torch.from_numpy(r) for r in e], e.tolist(), e):
        res = await store.query_similar(q, k=1)
        assert res[0].content == "t0"


ALL = [scenario_store_query_roundtrip, scenario_doc_filter, scenario_empty_and_missing, scenario_get_delete_upsert,
       scenario_input_tolerance]


async def scenario_random_ops_against_model(store, seed=0, n_ops=60, mode="float", capacity=64, fp32_pages=False):
    """Model-based check of the store's bookkeeping: a random sequence of store (incl. upserts), delete, compact and
    filtered queries; after every query the answer must be the brute-force top-k of the LIVE chunks (a plain dict is the
    model), scored by the oracle.  Catches id remapping / tombstone / ordinal-reuse mistakes that fixed scenarios miss."""
    from oracle import oracle as orc

    rng = np.random.default_rng(seed)
    model = {}  # (doc, chunk) -> embedding fp32 (bf16-rounded like the slab)
    docs = [f"d{j}" for j in range(7)]
    used_slots = 0

    def bf16r(x):
        if fp32_pages:  # a store with fp32_pages keeps hi + lo halves: the page scores as the fp32 input does (to 2^-18)
            return np.asarray(x, np.float32)
        return orc.bf16_to_f32(orc.f32_to_bf16(x))

    for step in range(n_ops):
        op = rng.choice(["store", "store", "delete", "compact", "query", "query"])
        if op == "store":
            n = int(rng.integers(1, 5))
            if used_slots + n > capacity:
                used_slots -= store.compact()
                if used_slots + n > capacity:
                    continue
            chunks, seen = [], set()
            for _ in range(n):
                key = (str(rng.choice(docs)), int(rng.integers(0, 4)))
                if key in seen:
                    continue
                seen.add(key)
                e = rand_emb(rng, int(rng.integers(1, 25)))
                chunks.append(DocumentChunk(document_id=key[0], chunk_number=key[1], content=f"{key[0]}/{key[1]}@{step}", embedding=e,
                                            metadata={"step": step}))
            ok, ids, _m = await store.store_embeddings(chunks)
            assert ok and len(ids) == len(chunks)
            used_slots += len(chunks)
            for c in chunks:
                model[(c.document_id, c.chunk_number)] = (bf16r(c.embedding), c.content)
        elif op == "delete":
            d = str(rng.choice(docs))
            assert await store.delete_chunks_by_document_id(d) is True
            for key in [k for k in model if k[0] == d]:
                del model[key]
        elif op == "compact":
            used_slots -= store.compact()
            assert used_slots == len(model) == len(store)
        else:
            q = rand_emb(rng, int(rng.integers(1, 9)))
            filt = None
            if rng.random() < 0.5:
                filt = [str(d) for d in rng.choice(docs + ["nope"], size=int(rng.integers(1, 4)), replace=False)]
            k = int(rng.integers(1, 8))
            res = await store.query_similar(q, k=k, doc_ids=filt)
            live = [(key, v) for key, v in model.items() if filt is None or key[0] in filt]
            qb = np.asarray(q, np.float32)  # an fp32 query is scored exactly (split hi + lo); only the slab's pages are bf16
            if mode == "binary":
                want = sorted(((float(orc.maxsim_binary(orc.sign_pack(v[0]), orc.sign_pack(q))), key) for key, v in live), key=lambda t: -t[0])
            else:
                want = sorted(((float(orc.maxsim_f32(qb, v[0])), key) for key, v in live), key=lambda t: -t[0])
            assert len(res) == min(k, len(live))
            got_scores = [r.score for r in res]
            assert all(got_scores[i] >= got_scores[i + 1] for i in range(len(res) - 1))
            np.testing.assert_allclose(got_scores, [w[0] for w in want[: len(res)]], rtol=1e-4, atol=1e-5)
            cutoff = want[len(res) - 1][0] if res else 0.0
            for r in res:  # every hit is a live chunk with its CURRENT payload; ties at the cut may pick either chunk
                assert (r.document_id, r.chunk_number) in model and r.content == model[(r.document_id, r.chunk_number)][1]
                assert filt is None or r.document_id in filt
                assert r.score >= cutoff - 1e-4 * abs(cutoff) - 1e-5
            got = await store.get_chunks_by_id([(key[0], key[1]) for key in list(model)[:3]] + [("nope", 0)])
            assert {(g.document_id, g.chunk_number) for g in got} == set(list(model)[:3])
