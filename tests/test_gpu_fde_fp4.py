"""MV_WITH_FDE_FP4 (round 6, closing session): the coarse stage of a single FDE request on an FP4 (e2m1) copy of the FDE slab.

The reference's coarse stage is a TurboPuffer ANN query over the documents' FDE vectors (core/vector_store/fast_multivector_store.py:526-532):
approximate by contract.  What is held here: the copy is the oracle's quantisation of the bf16 rows bit for bit (orc_quantize_fde_fp4: one
power-of-two scale per row -- half the covering one: the largest elements saturate --, round to nearest with ties to the even code, element 2i in the low nibble); the scan's scores are the fp32 dot
products of those codes (all 16 of them pass through v_cvt_scalef32_pk_f32_fp4, in both nibbles of all four bytes of a dword); every writer of
the FDE slab keeps the copy in step; the batched pass (both MFMA operands FP4, the queries as two e2m1 terms) scores the codes against that two-term
query; the pipeline's answers are
those of the bf16 coarse stage wherever the candidates decide nothing (planted neighbours)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


def _idx(**kw):
    from morphik_core_amd.index import MvIndex

    return MvIndex(**kw)


def _assert_copy_in_step(ix, n):
    want_c, want_s = orc.quantize_fde_fp4(orc.f32_to_bf16(ix.read_fde(0, n)))
    got_c, got_s = ix.read_fde_fp4(0, n)
    assert got_s.tolist() == want_s.tolist()
    assert np.array_equal(got_c, want_c)


def _pages(n, stride, seed=5):
    return [orc.synth_rows(seed, i, 0, 3 + (i * 7) % (stride - 2)) for i in range(n)]


def test_copy_is_the_oracles_quantisation_and_every_writer_keeps_it_in_step(tmp_path):
    from morphik_core_amd.index import MvIndex

    N, stride = 90, 32
    ix = _idx(capacity_pages=N + 10, stride_rows=stride, with_float=True, with_fde=True, with_fde_fp4=True)
    ix.add(_pages(N, stride), doc_ordinals=np.arange(N, dtype=np.int32) // 3)
    _assert_copy_in_step(ix, N)
    # all-zero row: scale 1, codes 0
    ix.add([np.zeros((4, 128), np.float32)], doc_ordinals=[500])
    c, s = ix.read_fde_fp4(N, 1)
    assert s.tolist() == [1.0] and not c.any()
    ix.replace_page(7, orc.synth_rows(77, 0, 0, 19))
    rng = np.random.default_rng(3)
    ix.import_fde(11, rng.standard_normal((3, ix.fde_config.output_dim)).astype(np.float32) * 37.0)
    _assert_copy_in_step(ix, N + 1)
    ix.remove_doc(2)
    ix.remove_page(40)
    ix.compact()
    n = len(ix)
    assert n == N + 1 - 4
    _assert_copy_in_step(ix, n)
    path = str(tmp_path / "fp4.idx")
    ix.save(path)
    re = MvIndex.load(path, device=0)
    assert np.array_equal(re.read_fde_fp4(0, n)[0], ix.read_fde_fp4(0, n)[0]) and re.read_fde_fp4(0, n)[1].tolist() == ix.read_fde_fp4(0, n)[1].tolist()
    q = orc.synth_rows(4321, 1, 0, 20)
    assert re.query(q, 5, mode="fde_then_float")[1].tolist() == ix.query(q, 5, mode="fde_then_float")[1].tolist()
    re.close()
    ix.close()


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 65, 700, 5003])
def test_scan_scores_are_the_fp32_dot_products_of_the_codes(n):
    """Rows that hold EVERY e2m1 code in every nibble position (imported, so the codes are known), a caller-supplied query FDE: the scan's score
    of a page is (sum_i q_i decode(code_i)) * scale / |d| to fp32 summation accuracy -- i.e. the hardware conversion decodes what the oracle's
    table decodes, low nibble first -- for page counts around the 16-row group and the 64-row workgroup, with tombstones and a doc filter,
    cosine on and off."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    stride = 16
    ix = _idx(capacity_pages=n, stride_rows=stride, with_float=False, with_fde=True, with_fde_fp4=True)
    od = ix.fde_config.output_dim
    ix.add([orc.synth_rows(9, i, 0, 5) for i in range(n)], doc_ordinals=np.arange(n, dtype=np.int32) // 2)
    lut = np.array([orc.lib().orc_fp4_decode(c) for c in range(16)], np.float32)
    rng = np.random.default_rng(n)
    m = min(n, 40)
    nib = rng.integers(0, 16, size=(m, od)).astype(np.uint8)
    nib[:, 0] = 7
    vals = lut[nib]
    vals[:, 0] = 12.0  # the row's amax = 12 -> scale 1 (12 * scale >= amax): element 0 saturates to code 7, every other value IS its decoded code
    rows = vals * np.float32(0.5) ** rng.integers(0, 3, size=(m, 1)).astype(np.float32)  # other scales too
    ix.import_fde(0, rows)
    got_c, got_s = ix.read_fde_fp4(0, m)
    got_nib = np.empty((m, od), np.uint8)
    got_nib[:, 0::2] = got_c & 15
    got_nib[:, 1::2] = got_c >> 4
    # -0.0 (code 8) is imported as a bf16 -0.0 and comes back as code 8; every other code is its own fixed point
    assert np.array_equal(got_nib, nib)
    if n > 3:
        ix.remove_page(2)
    qf = rng.standard_normal(od).astype(np.float32)
    full_c, full_s = ix.read_fde_fp4(0, n)
    deq = orc.fp4_decode(full_c).astype(np.float64)
    norms = np.linalg.norm(ix.read_fde(0, n).astype(np.float64), axis=1)
    allow = allow_bitmap([d for d in range((n + 1) // 2) if d % 3 != 1], (n + 1) // 2)
    for cosine in (1, 0):
        ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
        want = (deq @ qf.astype(np.float64)) * full_s
        if cosine:
            want = want / np.where(norms > 0, norms, 1.0)
        for al in (None, allow):
            s, i = ix.query(orc.synth_rows(1, 0, 0, 4), min(n, 1000), mode="fde", q_fde=qf, allow=al)
            live = [p for p in range(n) if not (n > 3 and p == 2) and (al is None or (p // 2) % 3 != 1)]
            assert set(i.tolist()) <= set(live) and len(i) == min(len(live), 1000)
            if n <= 1000:
                assert sorted(i.tolist()) == live
            else:  # the best 1000 of the live pages
                assert float(s.min()) >= float(np.sort(want[live])[-1000]) - 1e-4 * float(np.abs(want).max())
            np.testing.assert_allclose(s, want[i], rtol=2e-5, atol=1e-4 * float(np.abs(want).max()))
    ix.close()


def test_pipeline_on_the_fp4_coarse_stage_finds_the_planted_pages_single_and_batched():
    """Planted neighbours: both coarse stages put them among the candidates, the exact rerank returns the same ids and scores;
    MV_OPT_FDE_COARSE_SLAB switches the slab per query; coarse scores of the two slabs agree to the quantisation's few per cent of the
    largest score; a BATCH of requests on the fp4 copy returns the planted pages with the single request's exact scores."""
    from morphik_core_amd import _lib, synth

    N, stride = 3000, 64
    q = orc.synth_rows(4321, 0, 0, 32)
    plain = _idx(capacity_pages=N, stride_rows=stride, with_float=True, with_fde=True)
    ix = _idx(capacity_pages=N, stride_rows=stride, with_float=True, with_fde=True, with_fde_fp4=True)
    spec = synth.planted_spec([q], N, stride, n_ranks=10)
    for x in (plain, ix):
        x.fill_synthetic(1234, 0, N)
        for (_, _, p, row0, rows) in spec:
            page = x.read_pages(p, 1)[0]
            page[row0 : row0 + rows.shape[0]] = rows
            x.replace_page(p, page)
    planted = [p for (_, _, p, _, _) in spec]
    s4, i4 = ix.query(q, 10, mode="fde_then_float")
    c4 = ix.score_all(q, mode="fde")
    ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 0)
    s16, i16 = ix.query(q, 10, mode="fde_then_float")
    c16 = ix.score_all(q, mode="fde")
    ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 2)
    assert i4.tolist() == i16.tolist() == planted and s4.tolist() == s16.tolist()
    assert np.array_equal(c16, plain.score_all(q, mode="fde"))
    assert not np.array_equal(c4, c16)
    np.testing.assert_allclose(c4, c16, rtol=0, atol=6e-2 * float(np.abs(c16).max()))
    assert float(np.corrcoef(c4, c16)[0, 1]) > 0.98
    # a batch: the batched pass reads the fp4 copy too (both MFMA operands FP4); the planted query's answer is the single query's, and with
    # MV_OPT_FDE_COARSE_SLAB 0 the batch reads the bf16 slab and answers bit for bit like an index without the copy
    qs = [q] + [orc.synth_rows(4321, j, 0, 32) for j in range(1, 6)]
    sb, ib = ix.query_batch(qs, 10, mode="fde_then_float")[0]
    assert ib.tolist() == planted and sb.tolist() == s4.tolist()
    ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 0)
    got = ix.query_batch(qs, 10, mode="fde_then_float")
    want = plain.query_batch(qs, 10, mode="fde_then_float")
    for (gs, gi), (ws, wi) in zip(got, want):
        assert gi.tolist() == wi.tolist() and np.array_equal(gs, ws)
    ix.close()
    plain.close()


def test_recall_of_the_fp4_coarse_stage_on_hard_negatives_is_close_to_the_bf16_stages():
    """64 near-tied pages per query (exact scores 3e-4 apart per rank): of the exact top-10, how many does each coarse stage keep among its 75
    candidates?  Priced before the copy was built (tools/fde_4bit_recall_probe.py: 0.9875 against 0.9922 over 64 queries); here the fp4 stage may
    lose at most three pages in all of 120 against the bf16 stage."""
    from morphik_core_amd import _lib, synth

    N, stride, NQ = 20_000, 64, 12
    qs = [orc.synth_rows(4321, j, 0, 32) for j in range(NQ)]
    ix = _idx(capacity_pages=N, stride_rows=stride, with_float=True, with_fde=True, with_fde_fp4=True)
    ix.fill_synthetic(1234, 0, N)
    spec = synth.hard_spec(qs, N, stride)
    for (_qi, _j, p, row0, rows) in spec:
        page = ix.read_pages(p, 1)[0]
        page[row0 : row0 + rows.shape[0]] = rows
        ix.replace_page(p, page)
    ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 75)
    kept = {0: 0, 2: 0}
    for q in qs:
        truth = set(ix.query(q, 10, mode="float")[1].tolist())
        for slab in (0, 2):
            ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, slab)
            cand = set(ix.query(q, 75, mode="fde")[1].tolist())
            kept[slab] += len(truth & cand)
    assert kept[0] >= 0.95 * 10 * NQ
    assert kept[2] >= kept[0] - 3, kept
    ix.close()


def _two_term_fp4_query(qf):
    """numpy restatement of fde_batch_qscale4 / qprep4: one power-of-two scale s per query (the smallest with 6 s >= max|x|), hi = fp4(x / s),
    lo = fp4(4 (x / s - hi)) -> the fp64 vector the batched pass multiplies: (hi + lo / 4) s"""
    qf = np.asarray(qf, np.float32)
    amax = np.float32(np.abs(qf).max())
    if amax == 0:
        return qf.astype(np.float64)
    bits = int(np.array([amax], np.float32).view(np.uint32)[0])
    e0, mant = ((bits >> 23) & 0xFF) - 127, bits & 0x7FFFFF
    e = e0 - 2 if mant <= 0x400000 else e0 - 1
    x = (qf * np.float32(2.0) ** np.float32(-e)).astype(np.float32)
    enc = np.vectorize(lambda v: orc.lib().orc_fp4_decode(orc.lib().orc_fp4_encode(float(v))), otypes=[np.float32])
    hi = enc(x)
    lo = enc(((x - hi) * np.float32(4.0)).astype(np.float32))
    return (hi.astype(np.float64) + lo.astype(np.float64) / 4.0) * 2.0 ** e


@pytest.mark.parametrize("n,B", [(40, 5), (64, 16), (129, 20), (3000, 32), (70_001, 9)])
def test_batched_pass_on_the_fp4_copy_scores_the_codes_against_two_term_fp4_queries(n, B):
    """mv_query_topk_batch(mode "fde") on an index with the fp4 copy: ONE pass over the copy per 32 requests, both MFMA operands FP4
    (v_mfma_scale_f32_16x16x128_f8f6f4; queries as two e2m1 terms under one scale, the second at a block scale of 1/4; the chunk count padded from
    10 to 12 with zero query fragments).  Every returned score is the fp64 dot product of the page's codes with the query's two-term value times
    scale / |d| (to fp32 accumulation accuracy); the ranking is that model's; tombstones, a shared filter and per-request filters hold; corpus
    sizes around the 64-page tile, one and two query tiles."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    ix = _idx(capacity_pages=n, stride_rows=16, with_float=False, with_fde=True, with_fde_fp4=True)
    od = ix.fde_config.output_dim
    ix.fill_synthetic(1234, 0, n, pages_per_doc=3)
    ix.remove_doc(1)
    rng = np.random.default_rng(n + B)
    qfdes = (rng.standard_normal((B, od)) * rng.uniform(0.01, 30.0, size=(B, 1))).astype(np.float32)
    qfdes[B - 1, : od // 2] = 0.0
    queries = [orc.synth_rows(4321, b, 0, 8) for b in range(B)]
    m = min(n, 4000)  # the model is checked on the first m pages (a doc filter keeps the answers inside them)
    codes, scale = ix.read_fde_fp4(0, m)
    deq = orc.fp4_decode(codes).astype(np.float64)
    norms = np.linalg.norm(ix.read_fde(0, m).astype(np.float64), axis=1)
    n_docs = (n + 2) // 3
    docs_m = [d for d in range((m + 2) // 3) if d != 1 and (d + 1) * 3 <= m]
    shared = allow_bitmap(docs_m, n_docs)
    per_req = [allow_bitmap([d for d in docs_m if (d + b) % 4 != 0], n_docs) for b in range(B)]
    q2 = [_two_term_fp4_query(qfdes[b]) for b in range(B)]
    k = 25
    for cosine in (1, 0):
        ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
        fac = scale.astype(np.float64) / (norms if cosine else 1.0)
        for kind in ("shared", "per_request"):
            kw = dict(allow=shared) if kind == "shared" else dict(allows=per_req, n_docs=n_docs)
            got = ix.query_batch(queries, k, mode="fde", q_fdes=qfdes, **kw)
            for b in range(B):
                want = (deq @ q2[b]) * fac
                ok_docs = set(docs_m) if kind == "shared" else {d for d in docs_m if (d + b) % 4 != 0}
                live = np.array([p for p in range(m) if p // 3 in ok_docs], np.int64)
                s, i = got[b]
                assert len(i) == min(k, live.size) and set(i.tolist()) <= set(live.tolist())
                tol = 3e-5 * float(np.abs(want[live]).max()) + 1e-30
                np.testing.assert_allclose(s, want[i], rtol=0, atol=tol)
                assert float(s.min()) >= float(np.sort(want[live])[-len(i)]) - 2 * tol  # nothing better was left out
    # the same batch through the bf16 slab: other scores, (nearly) the same candidates
    ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 0)
    got16 = ix.query_batch(queries, k, mode="fde", q_fdes=qfdes, allow=shared)
    ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 2)
    got4 = ix.query_batch(queries, k, mode="fde", q_fdes=qfdes, allow=shared)
    overlap = np.mean([len(set(a[1].tolist()) & set(b[1].tolist())) / max(len(a[1]), 1) for a, b in zip(got16, got4)])
    assert overlap >= 0.6
    ix.close()


def test_batched_pipeline_on_the_fp4_copy_and_its_placement_trial():
    """fde_then_float batches on the fp4 coarse pass: planted neighbours come back with the exact scores of the single query; the placement
    trial moves the fp4 copy (the slab the pass reads) and changes nothing."""
    from morphik_core_amd import synth

    N, stride, B = 5000, 32, 12
    qs = [orc.synth_rows(4321, j, 0, 16) for j in range(B)]
    ix = _idx(capacity_pages=N, stride_rows=stride, with_fde=True, with_fde_fp4=True)
    ix.fill_synthetic(1234, 0, N)
    spec = synth.planted_spec(qs, N, stride, n_ranks=10)
    for (_, _, p, row0, rows) in spec:
        page = ix.read_pages(p, 1)[0]
        page[row0 : row0 + rows.shape[0]] = rows
        ix.replace_page(p, page)
    want = [ix.query(q, 10, mode="fde_then_float") for q in qs]
    got = ix.query_batch(qs, 10, mode="fde_then_float")
    for qi, ((ws, wi), (s, i)) in enumerate(zip(want, got)):
        assert i.tolist() == wi.tolist() == [p for (qq, _r, p, _a, _b) in spec if qq == qi] and s.tolist() == ws.tolist()
    codes0 = ix.read_fde_fp4(0, N)[0]
    before, after, moves = ix.fde_placement_trial(2)
    assert before > 0 and after > 0 and 0 <= moves <= 2
    assert np.array_equal(ix.read_fde_fp4(0, N)[0], codes0)
    for (ws, wi), (s, i) in zip(want, ix.query_batch(qs, 10, mode="fde_then_float")):
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    ix.close()


@pytest.mark.parametrize("R", [2, 3])
def test_shards_with_the_fp4_copy_answer_like_one_index(R):
    """mv_comm over R shards that each hold the fp4 copy of THEIR FDE rows: a row's codes and scale do not depend on its neighbours, so the
    staged pipeline (global candidate rule across shards) returns what one index with the copy returns -- single requests (fp32 query against
    the codes) and batches (two-term FP4 queries, both MFMA operands FP4), coarse-only and with the exact rerank, with a doc filter."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import ShardComm, allow_bitmap

    N, stride, k = 360, 32, 8
    per = N // R
    pages = _pages(N, stride, seed=21)
    ords = [i % 13 for i in range(N)]
    kw = dict(stride_rows=stride, with_float=True, with_fde=True, with_fde_fp4=True)
    one = _idx(capacity_pages=N, **kw)
    one.add(pages, doc_ordinals=ords)
    shards = []
    for r in range(R):
        sh = _idx(capacity_pages=per, id_base=r * per, **kw)
        sh.add(pages[r * per : (r + 1) * per], doc_ordinals=ords[r * per : (r + 1) * per])
        shards.append(sh)
    for ix in [one] + shards:
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 60)
    one.remove_page(100)
    shards[100 // per].remove_page(100 % per)
    comm = ShardComm(shards, transport="p2p")
    allow = allow_bitmap([0, 2, 3, 5, 7, 11, 12], 13)
    qs = [orc.synth_rows(4321, 40 + j, 0, 14) for j in range(7)]
    for al in (None, allow):
        for mode in ("fde_then_float", "fde"):
            for q in qs[:3]:
                ws, wi = one.query(q, k, mode=mode, allow=al)
                s, i = comm.query(q, k, mode=mode, allow=al)
                assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), (mode, R)
        want = one.query_batch(qs, k, mode="fde_then_float", allow=al)
        got = comm.query_batch(qs, k, mode="fde_then_float", allow=al)
        for (ws, wi), (s, i) in zip(want, got):
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist(), ("batch", R)
    comm.close()
    for sh in shards:
        sh.close()
    one.close()


def test_flag_and_option_errors():
    from morphik_core_amd import _lib
    from morphik_core_amd._lib import MvError

    with pytest.raises(MvError):
        _idx(capacity_pages=4, stride_rows=16, with_float=True, with_fde=False, with_fde_fp4=True)
    both = _idx(capacity_pages=4, stride_rows=16, with_float=True, with_fde=True, with_fde_fp4=True, with_fde_e4m3=True)  # the index may hold both copies
    both.add([orc.synth_rows(3, 0, 0, 9)], doc_ordinals=[0])
    q = orc.synth_rows(4321, 0, 0, 8)
    for slab in (2, 1, 0):  # default 2; every slab answers
        both.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, slab)
        assert both.query(q, 1, mode="fde_then_float")[1].tolist() == [0]
    both.close()
    ix = _idx(capacity_pages=4, stride_rows=16, with_float=True, with_fde=True)
    with pytest.raises(MvError):
        ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 2)
    with pytest.raises(MvError):
        ix.read_fde_fp4(0, 0)
    ix.close()
    ix = _idx(capacity_pages=4, stride_rows=16, with_float=True, with_fde=True, with_fde_fp4=True)
    with pytest.raises(MvError):
        ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 1)
    with pytest.raises(MvError):
        ix.set_option(_lib.MV_OPT_FDE_COARSE_SLAB, 3)
    ix.close()


@pytest.mark.parametrize("sharded", [False, True])
def test_store_with_fde_fp4_runs_the_reference_scenarios_and_survives_a_checkpoint(sharded, tmp_path):
    """store option fde_fp4 (single store and one store over three shards): the reference's store scenarios -- the rerank is exact, so the
    planted answers of the scenarios must come back -- and a checkpoint round trip keeps the option and the answers; fde_e4m3 and fde_fp4
    together are refused."""
    from morphik_core_amd.store import MI355XFastMultiVectorStore, MI355XShardedFastMultiVectorStore
    from tests import store_scenarios as sc

    def make(**kw):
        if sharded:
            s = MI355XShardedFastMultiVectorStore(devices=[0, 0, 0], transport="p2p", capacity_pages=96, stride_rows=32, mode="fde_then_float", fde_fp4=True, **kw)
        else:
            s = MI355XFastMultiVectorStore(capacity_pages=96, stride_rows=32, mode="fde_then_float", fde_fp4=True, **kw)
        assert s.initialize() is True
        return s

    with pytest.raises(ValueError):
        MI355XFastMultiVectorStore(capacity_pages=96, stride_rows=32, mode="fde_then_float", fde_fp4=True, fde_e4m3=True)
    for scenario in sc.ALL:
        s = make()
        try:
            sc.run(scenario(s))
        finally:
            s.close()
    s = make()
    rng = np.random.default_rng(11)
    chunks = sc.make_chunks(rng, n_docs=6, chunks_per_doc=4)
    sc.run(s.store_embeddings(chunks))
    want = [[(c.document_id, c.chunk_number, c.score) for c in sc.run(s.query_similar(ch.embedding, k=3))] for ch in chunks[:5]]
    assert all(w[0][:2] == (ch.document_id, ch.chunk_number) for w, ch in zip(want, chunks[:5]))
    path = str(tmp_path / "ckpt")
    s.save(path)
    s.close()
    cls = MI355XShardedFastMultiVectorStore if sharded else MI355XFastMultiVectorStore
    r = cls.load(path, **(dict(devices=[0, 0, 0], transport="p2p") if sharded else {}))
    assert r.fde_fp4 is True
    got = [[(c.document_id, c.chunk_number, c.score) for c in sc.run(r.query_similar(ch.embedding, k=3))] for ch in chunks[:5]]
    assert got == want
    r.close()
