"""GPU tests of the PACKED page layout (MV_LAYOUT_PACKED): ragged pages back to back in whole 16-row tiles, a row-offset table
instead of `page * stride_rows` in every kernel.  The reference's real encoder (ColQwen2.5, core/embedding/colpali_embedding_model.py:47-52)
emits a different token count per page; a fixed slot per page then wastes the HBM between a page's rows and stride_rows.

Bar: the packed index returns the bits of the fixed-stride index (same kernels, same arithmetic order, another base address) in every
mode and through every writer, and both agree with the oracle.

Run on the MI355X box:  python -m pytest tests/test_gpu_packed_layout.py -m gpu -x -q
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)

LENS = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 5, 64, 20, 200, 208, 1, 96, 113, 7, 160, 208, 33]


def _pages(seed=77, lens=LENS):
    return [orc.synth_rows(seed, i, 0, n) if n else np.zeros((0, 128), np.uint16) for i, n in enumerate(lens)]


def _pair(stride=208, cap=64, **kw):
    from morphik_core_amd.index import MvIndex

    fixed = MvIndex(capacity_pages=cap, stride_rows=stride, **kw)
    packed = MvIndex(capacity_pages=cap, stride_rows=stride, packed=True, capacity_rows=kw.pop("capacity_rows", 0), **kw)
    return fixed, packed


def _same(a, b):
    assert np.array_equal(np.asarray(a).view(np.uint32) if np.asarray(a).dtype == np.float32 else np.asarray(a),
                          np.asarray(b).view(np.uint32) if np.asarray(b).dtype == np.float32 else np.asarray(b))


@pytest.mark.parametrize("variant", [0, 6, 7])
def test_packed_float_scan_and_candidates_equal_the_fixed_layout_bit_for_bit(variant):
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    pages = _pages()
    fx, pk = _pair()
    for ix in (fx, pk):
        ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
        ix.add(pages, doc_ordinals=[i // 3 for i in range(len(pages))])
    assert pk.rows_used == sum((n + 15) // 16 * 16 for n in LENS) and fx.rows_used == len(LENS) * 208
    for nq in (21, 40, 70, 130):
        q = orc.synth_rows(4321, 100 + nq, 0, nq)
        a, b = fx.score_all(q), pk.score_all(q)
        _same(a, b)
        want = np.array([orc.maxsim_bf16(q, p) for p in pages], np.float32)
        alive = np.array([not (nq != 21 and i == 7) for i in range(len(pages))])  # page 7 is removed at the end of the first round
        np.testing.assert_allclose(b[alive], want[alive], rtol=1e-4, atol=1e-6)
        assert np.all(np.isneginf(b[~alive]))
        cand = [1, 4, 11, 12, 15, 16, 23, 0]
        _same(fx.score_candidates(q, cand, 208), pk.score_candidates(q, cand, 208))
        _same(fx.score_candidates(q, np.arange(len(pages)), pad_to=-1), pk.score_candidates(q, np.arange(len(pages)), pad_to=-1))
        for ix in (fx, pk):
            if nq == 21:
                ix.remove_page(7)
        allow = allow_bitmap([d for d in range(8) if d % 4 != 1], 8)
        _same(fx.score_all(q, allow=allow), pk.score_all(q, allow=allow))
        (s1, i1), (s2, i2) = fx.query(q, 9, allow=allow), pk.query(q, 9, allow=allow)
        assert i1.tolist() == i2.tolist()
        _same(s1, s2)
    # the images the readers hand out are the fixed-stride ones
    _same(fx.read_pages(0, len(pages)), pk.read_pages(0, len(pages)))
    fx.close()
    pk.close()


def test_packed_sign_bit_fp8_fde_slabs_and_pipelines_equal_the_fixed_layout():
    from morphik_core_amd import _lib

    pages = _pages()
    fx, pk = _pair(with_binary=True, with_fde=True, with_fp8=True)
    for ix in (fx, pk):
        ix.add(pages[:10])
        ix.add([orc.bf16_to_f32(p) for p in pages[10:]])  # fp32 ingest: sign bits from the fp32 rows, FDE from the fp32 rows
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 12)
    q = orc.synth_rows(4321, 5, 0, 32)
    for mode in ("binary", "float_fp8", "fde"):
        _same(fx.score_all(q, mode=mode), pk.score_all(q, mode=mode))
    for bv in (0, 4):
        for ix in (fx, pk):
            ix.set_option(_lib.MV_OPT_BINARY_VARIANT, bv)
        _same(fx.score_all(q, mode="binary"), pk.score_all(q, mode="binary"))
    # the sign-bit scan is exact against the oracle
    bits_q = orc.sign_pack(orc.bf16_to_f32(q))
    want = [orc.maxsim_binary(orc.sign_pack(orc.bf16_to_f32(p)) if len(p) else np.zeros((0, 16), np.uint8), bits_q) for p in pages]
    assert pk.score_all(q, mode="binary").astype(np.float64).tolist() == want
    c1, s1 = fx.read_fp8(0, len(pages))
    c2, s2 = pk.read_fp8(0, len(pages))
    _same(c1, c2)
    _same(s1, s2)
    _same(fx.read_fde(0, len(pages)), pk.read_fde(0, len(pages)))
    for mode in ("fde_then_float", "fp8_then_float", "float_fp8", "binary", "float"):
        (a, ia), (b, ib) = fx.query(q, 6, mode=mode), pk.query(q, 6, mode=mode)
        assert ia.tolist() == ib.tolist(), mode
        _same(a, b)
    # batches: the page-split / row-split bf16 kernels, the e4m3 batch kernel, the batched FDE pipeline and its one-launch rerank
    qs = [orc.synth_rows(4321, 50 + j, 0, 32) for j in range(16)]
    for mode, nb in (("float", 4), ("float", 16), ("float_fp8", 4), ("float_fp8", 16), ("fde_then_float", 5), ("fp8_then_float", 5)):
        ra, rb = fx.query_batch(qs[:nb], 6, mode=mode), pk.query_batch(qs[:nb], 6, mode=mode)
        for (a, ia), (b, ib) in zip(ra, rb):
            assert ia.tolist() == ib.tolist(), (mode, nb)
            _same(a, b)
    fx.close()
    pk.close()


def test_packed_split_bf16_tier():
    """MV_WITH_FLOAT_LO + MV_LAYOUT_PACKED: the lo slab shares the row table; fp32 pages / queries against the fp32 oracle."""
    rng = np.random.default_rng(3)
    lens = [40, 1, 17, 208, 64, 0, 33, 100]
    pages = [(lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True))(rng.standard_normal((n, 128)).astype(np.float32)) if n else np.zeros((0, 128), np.float32)
             for n in lens]
    q = rng.standard_normal((24, 128)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    fx, pk = _pair(with_float_lo=True)
    for ix in (fx, pk):
        ix.add(pages)
    _same(fx.score_all(q), pk.score_all(q))
    want = np.array([orc.maxsim_f32(q, p) for p in pages], np.float32)
    got = pk.score_all(q)
    assert np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max())
    _same(fx.read_pages_f32(0, len(pages)), pk.read_pages_f32(0, len(pages)))
    fx.close()
    pk.close()


def test_packed_writers_replace_write_compact_save_load(tmp_path):
    from morphik_core_amd._lib import MvError
    from morphik_core_amd.index import MvIndex

    pages = _pages()
    fx, pk = _pair(with_binary=True, with_fp8=True, with_fde=True)
    docs = [i // 2 for i in range(len(pages))]
    for ix in (fx, pk):
        ix.add(pages, doc_ordinals=docs)
    q = orc.synth_rows(4321, 9, 0, 32)
    # a replacement that fits the page's tiles (page 19: 113 rows -> a 112-row slot ... 128 rows of tiles)
    new = orc.synth_rows(99, 1, 0, 120)
    for ix in (fx, pk):
        ix.replace_page(19, new)
    with pytest.raises(MvError):
        pk.replace_page(2, orc.synth_rows(99, 2, 0, 17))  # page 2 owns ONE tile (15 rows): 17 rows do not fit
    rows = orc.synth_rows(99, 3, 0, 6)
    for ix in (fx, pk):
        ix.write_rows(15, 100, rows)
    with pytest.raises(MvError):
        pk.write_rows(1, 16, rows)  # page 1 (1 row) owns rows 0..15 only
    for mode in ("float", "binary", "float_fp8", "fde"):
        _same(fx.score_all(q, mode=mode), pk.score_all(q, mode=mode))
    # compaction: removed documents' tiles are reclaimed, the live pages keep their order and their answers
    used_before = pk.rows_used
    for d in (0, 3, 7, 11):
        assert fx.remove_doc(d) == pk.remove_doc(d)
    o1, o2 = fx.compact(), pk.compact()
    assert o1.tolist() == o2.tolist() and len(fx) == len(pk) == len(pages) - 8
    live = [i for i in range(len(pages)) if docs[i] not in (0, 3, 7, 11)]
    lens_now = [120 if i == 19 else LENS[i] for i in live]
    assert pk.rows_used == sum((LENS[i] + 15) // 16 * 16 for i in live) < used_before
    for mode in ("float", "binary", "float_fp8", "fde"):
        _same(fx.score_all(q, mode=mode), pk.score_all(q, mode=mode))
    _same(fx.read_pages(0, len(live)), pk.read_pages(0, len(live)))
    assert pk.page_rows(np.arange(len(live))).tolist() == lens_now
    # append after compaction lands behind the live pages
    more = [orc.synth_rows(55, j, 0, n) for j, n in enumerate((30, 208, 2))]
    for ix in (fx, pk):
        ix.add(more, doc_ordinals=[40, 40, 41])
    for mode in ("float", "binary", "float_fp8", "fde", ):
        _same(fx.score_all(q, mode=mode), pk.score_all(q, mode=mode))
    # checkpoint round trip
    path = os.path.join(tmp_path, "packed.mvidx")
    pk.save(path)
    before = {m: pk.score_all(q, mode=m) for m in ("float", "binary", "float_fp8", "fde")}
    used = pk.rows_used
    pk.close()
    back = MvIndex.load(path)
    assert back.rows_used == used and len(back) == len(fx)
    for m, v in before.items():
        _same(back.score_all(q, mode=m), v)
    (s1, i1), (s2, i2) = fx.query(q, 5, mode="fde_then_float"), back.query(q, 5, mode="fde_then_float")
    assert i1.tolist() == i2.tolist()
    _same(s1, s2)
    back.add([orc.synth_rows(56, 0, 0, 77)], doc_ordinals=[50])  # a loaded packed index keeps appending
    assert back.rows_used == used + 80
    back.close()
    fx.close()


def test_packed_capacity_is_counted_in_rows_and_holds_more_ragged_pages():
    """1000 pages of 550..1024 rows (a ColQwen-like spread): the fixed layout needs 1000 x 1024 rows; packed, the same pages fit a slab of
    their valid tiles -- and a slab sized for 790 fixed slots takes all 1000."""
    from morphik_core_amd._lib import MvError
    from morphik_core_amd.index import MvIndex, synth_ragged_rows

    n = 1000
    rows = [synth_ragged_rows(1234, u, 550, 1024) for u in range(n)]
    assert min(rows) >= 550 and max(rows) <= 1024 and len(set(rows)) > 300
    need = sum((r + 15) // 16 * 16 for r in rows)
    assert need < 0.80 * n * 1024
    pk = MvIndex(capacity_pages=n, stride_rows=1024, packed=True, capacity_rows=need)
    pk.fill_synthetic_ragged(1234, 0, n, 550, 1024)
    assert len(pk) == n and pk.rows_used == need == pk.capacity_rows
    assert pk.page_rows(np.arange(n)).tolist() == rows
    with pytest.raises(MvError):
        pk.add([orc.synth_rows(1, 0, 0, 16)])  # full, in rows and in pages
    fx = MvIndex(capacity_pages=n, stride_rows=1024)
    fx.fill_synthetic_ragged(1234, 0, n, 550, 1024)
    q = orc.synth_rows(4321, 0, 0, 32)
    _same(fx.score_all(q), pk.score_all(q))
    # the generator's pages are the oracle's rows of that unit, cut at n(u)
    img = pk.read_pages(0, 40)
    for u in (0, 7, 39):
        want = orc.synth_rows(1234, u, 0, rows[u])
        assert np.array_equal(img[u, : rows[u]], want) and not img[u, rows[u]:].any()
    want = np.array([orc.maxsim_bf16(q, img[u, : rows[u]]) for u in range(40)], np.float32)
    np.testing.assert_allclose(pk.score_all(q)[:40], want, rtol=1e-4)
    s1, i1 = fx.query(q, 10)
    s2, i2 = pk.query(q, 10)
    assert i1.tolist() == i2.tolist()
    _same(s1, s2)
    fx.close()
    pk.close()


def test_packed_sign_bit_import_and_sharded_communicator():
    from morphik_core_amd.index import MvIndex, ShardComm

    pages = _pages()
    bits = [orc.sign_pack(orc.bf16_to_f32(p)) if len(p) else np.zeros((0, 16), np.uint8) for p in pages]
    fx = MvIndex(capacity_pages=32, stride_rows=208, with_float=False, with_binary=True)
    pk = MvIndex(capacity_pages=32, stride_rows=208, with_float=False, with_binary=True, packed=True)
    for ix in (fx, pk):
        ix.add_bits(bits[:10])
        ix.add_bits(bits[10:])
    q = orc.synth_rows(4321, 2, 0, 20)
    _same(fx.score_all(q, mode="binary"), pk.score_all(q, mode="binary"))
    fx.close()
    pk.close()
    # two packed shards behind the C-ABI communicator == one packed index == one fixed index
    one = MvIndex(capacity_pages=32, stride_rows=208, with_fde=True, with_fp8=True, packed=True)
    one.add(pages)
    half = len(pages) // 2
    shards = []
    for r in range(2):
        sh = MvIndex(capacity_pages=16, stride_rows=208, with_fde=True, with_fp8=True, packed=True, id_base=r * half)
        sh.add(pages[r * half : (r + 1) * half])
        shards.append(sh)
    comm = ShardComm(shards)
    for mode in ("float", "float_fp8", "fde_then_float", "fp8_then_float"):
        (a, ia), (b, ib) = one.query(q, 7, mode=mode), comm.query(q, 7, mode=mode)
        assert ia.tolist() == ib.tolist(), mode
        _same(a, b)
    comm.close()
    for sh in shards:
        sh.close()
    one.close()


def test_packed_flag_is_refused_with_the_host_exact_tier():
    from morphik_core_amd._lib import MvError
    from morphik_core_amd.index import MvIndex

    with pytest.raises(MvError):
        MvIndex(capacity_pages=4, stride_rows=16, with_float=False, with_fp8=True, with_host_exact=True, packed=True)
    with pytest.raises(MvError):
        MvIndex(capacity_pages=4, stride_rows=16, packed=True, capacity_rows=24)  # not a multiple of 16


@pytest.mark.parametrize("mode", ["float", "fde_then_float", "binary"])
def test_store_with_the_packed_layout_answers_like_the_fixed_one(mode, tmp_path):
    """Behind the plugin surface: `MI355X*MultiVectorStore(packed_layout=True, capacity_rows=...)` -- ragged chunk embeddings, the same hits and
    scores as the fixed-stride store, a row budget instead of a slot budget, checkpoint round trip."""
    import morphik_core_amd.store as st
    from tests import store_scenarios as sc

    cls = st.MI355XFastMultiVectorStore if mode == "fde_then_float" else st.MI355XMultiVectorStore
    rng = np.random.default_rng(23)
    chunks = sc.make_chunks(rng, n_docs=5, chunks_per_doc=4, rows=24)
    for j, c in enumerate(chunks):  # 3 .. 60 rows
        c.embedding = sc.rand_emb(rng, 3 + (j * 7) % 58)
    need = sum((np.asarray(c.embedding).shape[0] + 15) // 16 * 16 for c in chunks)
    fixed = cls(capacity_pages=32, stride_rows=64, mode=mode)
    packed = cls(capacity_pages=32, stride_rows=64, mode=mode, packed_layout=True, capacity_rows=need + 64)
    assert fixed.initialize() and packed.initialize()
    try:
        for s in (fixed, packed):
            ok, ids, _m = sc.run(s.store_embeddings(chunks, app_id="t"))
            assert ok and len(ids) == len(chunks)
        assert packed._index.rows_used == need and packed._index.capacity_rows == need + 64 < 32 * 64
        for c in chunks[::3]:
            a = sc.run(fixed.query_similar(c.embedding, k=6, app_id="t"))
            b = sc.run(packed.query_similar(c.embedding, k=6, app_id="t"))
            assert [(h.document_id, h.chunk_number, h.score) for h in a] == [(h.document_id, h.chunk_number, h.score) for h in b]
        # the row budget is what fills up: one more 64-row chunk fits, the next does not
        from morphik_core_amd.models import DocumentChunk

        extra = DocumentChunk(document_id="x", chunk_number=0, content="x", embedding=sc.rand_emb(rng, 64), metadata={})
        assert sc.run(packed.store_embeddings([extra], app_id="t"))[0] is True
        extra2 = DocumentChunk(document_id="x", chunk_number=1, content="y", embedding=sc.rand_emb(rng, 40), metadata={})
        with pytest.raises(Exception):
            sc.run(packed.store_embeddings([extra2], app_id="t"))
        d = str(tmp_path / "ck")
        packed.save(d)
        back = cls.load(d)
        try:
            assert back.packed_layout and back._index.rows_used == need + 64
            b = sc.run(back.query_similar(chunks[3].embedding, k=4, app_id="t"))
            a = sc.run(packed.query_similar(chunks[3].embedding, k=4, app_id="t"))
            assert [(h.document_id, h.chunk_number, h.score) for h in a] == [(h.document_id, h.chunk_number, h.score) for h in b]
        finally:
            back.close()
    finally:
        fixed.close()
        packed.close()
