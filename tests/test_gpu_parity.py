"""GPU parity tests: every HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs (and against the committed golden fixtures).  Bars: bit-exact for integer/byte/index
work; <= 1e-3 relative (north_star) for float scores -- observed ~1e-6.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)

RTOL = 1e-3  # north_star tolerance for float MaxSim


@pytest.fixture(scope="module")
def mv():
    import morphik_core_amd as m

    assert os.path.exists(m.library_path()), "libmvmaxsim.so missing: the HIP path must be the one that runs"
    from morphik_core_amd import _lib

    assert _lib.lib().mv_device_count() >= 1, "no MI355X visible"
    return m


def _idx(mv, **kw):
    from morphik_core_amd.index import MvIndex

    return MvIndex(**kw)


def _assert_topk_matches(got_s, got_i, want_s, want_i, rtol=RTOL):
    assert len(got_i) == len(want_i)
    np.testing.assert_allclose(got_s, want_s, rtol=rtol, atol=1e-6)
    # ids equal where oracle scores are separated by more than the tolerance; equal as sets otherwise
    assert set(got_i.tolist()) == set(want_i.tolist())
    sep = np.abs(np.diff(want_s)) > 2 * rtol * np.abs(want_s[:-1]).max() if len(want_s) > 1 else np.array([])
    if len(want_s) > 1 and sep.all():
        assert got_i.tolist() == want_i.tolist()


# ------------------------------------------------------------------ generator
def test_synth_generator_bit_identical_to_oracle(mv):
    from morphik_core_amd.index import synth_rows

    ix = _idx(mv, capacity_pages=16, stride_rows=64)
    ix.fill_synthetic(1234, 5, 9, n_rows=50)
    got = ix.read_pages(0, 9)
    want = np.zeros_like(got)
    want[:, :50] = orc.synth_pages(1234, 5, 9, 50)
    assert np.array_equal(got, want)
    assert np.array_equal(synth_rows(4321, 3, 32), orc.synth_rows(4321, 3, 0, 32))
    ix.close()


# ------------------------------------------------------------------ float MaxSim
@pytest.mark.parametrize("variant", [0, 6, 7])  # direct loads (cross-check), the two defaults (four waves / one wave per page on the nt LDS-DMA ring)
def test_float_maxsim_all_variants_small(mv, variant):
    from morphik_core_amd import _lib

    ix = _idx(mv, capacity_pages=512, stride_rows=64)
    ix.fill_synthetic(1234, 0, 301)
    ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
    pages = ix.read_pages(0, 301)
    for nq in (32, 7, 48):
        q = orc.synth_rows(4321, nq, 0, nq)
        want = orc.maxsim_bf16_slab(q, pages)
        got = ix.score_all(q)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-6)
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()  # expected ~1e-6
        gs, gi = ix.query(q, 10)
        ws, wi = orc.topk(want, 10)
        _assert_topk_matches(gs, gi, ws, wi)
    ix.close()


@pytest.mark.parametrize("nq", [1, 16, 17, 32, 64, 65, 100, 112, 128, 129, 200, 300, 384, 400, 512, 530, 900])
def test_float_maxsim_1024_patches_query_lengths(mv, nq):
    ix = _idx(mv, capacity_pages=64, stride_rows=1024)
    ix.fill_synthetic(1234, 100, 40)
    pages = ix.read_pages(0, 40)
    q = orc.synth_rows(4321, 1000 + nq, 0, nq)
    want = orc.maxsim_float_np(orc.bf16_to_f32(q), orc.bf16_to_f32(pages))
    got = ix.score_all(q)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    if nq in (100, 128, 200):  # long queries (5..8 row tiles per pass, several passes) on every page-split kernel family
        from morphik_core_amd import _lib

        ix.set_option(_lib.MV_OPT_LONG_QUERY_VARIANT, 0)  # default 1 = the row-split workgroup checked above
        for variant in (0, 6, 7):
            ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
            np.testing.assert_allclose(ix.score_all(q), want, rtol=1e-4, atol=1e-6)
    ix.close()


@pytest.mark.parametrize("nq", [70, 130, 390, 600])
def test_long_query_row_split_route_ragged_filter_tombstones(mv, nq):
    """Queries > 64 rows take the row-split (batched) workgroup as ONE query: ragged pages, doc filter, tombstones and
    top-k must agree with the oracle and with the page-split kernel route."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    lens = [(i * 37) % 200 + 1 for i in range(90)] + [0, 208, 16]
    pages = [orc.synth_rows(78, i, 0, n) if n else np.zeros((0, 128), np.uint16) for i, n in enumerate(lens)]
    ix = _idx(mv, capacity_pages=128, stride_rows=208)
    ix.add(pages, doc_ordinals=[i // 3 for i in range(len(pages))])
    ix.remove_page(7)
    q = orc.synth_rows(4321, 50 + nq, 0, nq)
    allow = allow_bitmap([d for d in range(31) if d % 4 != 1], 31)
    want = np.array([orc.maxsim_bf16(q, p) for p in pages], np.float32)
    ok = np.array([(i // 3) % 4 != 1 and i != 7 for i in range(len(pages))])
    res = {}
    for route in (1, 0):
        ix.set_option(_lib.MV_OPT_LONG_QUERY_VARIANT, route)
        got = ix.score_all(q, allow=allow)
        assert np.all(np.isneginf(got[~ok]))
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-4, atol=1e-6)
        res[route] = ix.query(q, 12, allow=allow)
        ws, wi = orc.topk(got, 12)
        assert res[route][1].tolist() == wi.tolist()
    assert set(res[0][1].tolist()) == set(res[1][1].tolist())
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    ix.close()


@pytest.mark.parametrize("variant", [0, 6, 7])
def test_float_maxsim_ragged_pages(mv, variant):
    """Ragged pages through mv_index_add: rows beyond n_rows never count (pad_to = 0)."""
    from morphik_core_amd import _lib

    lens = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 5, 64, 20]
    pages = [orc.synth_rows(77, i, 0, n) if n else np.zeros((0, 128), np.uint16) for i, n in enumerate(lens)]
    ix = _idx(mv, capacity_pages=32, stride_rows=64)
    ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
    ix.add(pages)
    q = orc.synth_rows(4321, 9, 0, 21)
    want = np.array([orc.maxsim_bf16(q, p) for p in pages], np.float32)
    got = ix.score_all(q)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-6)
    # zero-padding clamp of score_multi_vector: candidates scored as ONE batch pad to the longest
    cand = [1, 4, 11, 12]
    pad_to = max(lens[c] for c in cand)
    want_c = np.array([orc.maxsim_bf16(q, pages[c], pad_to) for c in cand], np.float32)
    np.testing.assert_allclose(ix.score_candidates(q, cand, pad_to), want_c, rtol=RTOL, atol=1e-6)
    ix.close()


def test_float_maxsim_golden_score_retrieval(mv, golden_dir):
    """Against transformers' score_retrieval outputs (tests/golden/maxsim_float.npz) at north_star's 1e-3 for ALL six cases:
    the four fp32 fixtures live in an index with the lo slab (MV_WITH_FLOAT_LO: x = hi + lo, three MFMA terms), the two
    bf16-representable ones in a plain index -- observed ~1e-6 either way."""
    g = np.load(os.path.join(golden_dir, "maxsim_float.npz"))
    for ci in range(int(g["n_cases"])):
        q, slab, n_rows, pad_to, want = (g[f"{k}{ci}"] for k in ("q", "slab", "n_rows", "pad_to", "scores"))
        stride = ((slab.shape[1] + 15) // 16) * 16
        exact = np.array_equal(orc.bf16_to_f32(orc.f32_to_bf16(slab)), slab)
        ix = _idx(mv, capacity_pages=slab.shape[0], stride_rows=stride, with_float_lo=not exact)
        ix.add([slab[i, : n_rows[i]] for i in range(slab.shape[0])])
        for j in range(0, slab.shape[0], 128):  # the reference pads per batch of 128
            cand = list(range(j, min(j + 128, slab.shape[0])))
            got = ix.score_candidates(q, cand, int(pad_to[j]))
            np.testing.assert_allclose(got, want[cand], rtol=RTOL, atol=1e-5)
            assert np.abs(got - want[cand]).max() <= 2e-5 * max(1.0, np.abs(want).max())  # expected ~1e-6 of the score scale
        # ONE call over the whole list with the reference rule (pad_to = -1): every batch of 128 pads on its own, on the
        # device -- the golden case with 133 pages crosses the batch boundary
        allc = list(range(slab.shape[0]))
        got_all = ix.score_candidates(q, allc, pad_to=-1)
        np.testing.assert_allclose(got_all, want, rtol=RTOL, atol=1e-5)
        per_batch = np.concatenate([ix.score_candidates(q, allc[j : j + 128], int(pad_to[j])) for j in range(0, len(allc), 128)])
        np.testing.assert_array_equal(got_all, per_batch)
        ix.close()


def test_doc_filter_and_tombstones(mv):
    from morphik_core_amd.index import allow_bitmap

    ix = _idx(mv, capacity_pages=256, stride_rows=32)
    ix.fill_synthetic(1234, 0, 200, pages_per_doc=4)  # 50 docs of 4 pages
    pages = ix.read_pages(0, 200)
    q = orc.synth_rows(4321, 0, 0, 32)
    full = orc.maxsim_bf16_slab(q, pages)
    allowed = [3, 7, 8, 40]
    bm = allow_bitmap(allowed, 50)
    got = ix.score_all(q, allow=bm)
    mask = np.isin(np.arange(200) // 4, allowed)
    assert np.all(np.isneginf(got[~mask]))
    np.testing.assert_allclose(got[mask], full[mask], rtol=RTOL)
    s, i = ix.query(q, 50, allow=bm)
    assert len(i) == 16 and set(i.tolist()) <= set(np.nonzero(mask)[0].tolist())  # results subset of the filter
    ws, wi = orc.topk(np.where(mask, full, -np.inf), 50)
    _assert_topk_matches(s, i, ws, wi)
    # delete document 7 -> its pages disappear (delete_chunks_by_document_id)
    assert ix.remove_doc(7) == 4
    s, i = ix.query(q, 50, allow=bm)
    assert len(i) == 12 and not (set(i.tolist()) & {28, 29, 30, 31})
    s, i = ix.query(q, 300)  # k > N, no filter
    assert len(i) == 196
    ix.close()


def test_selective_doc_filter_compaction_equals_in_scan_masking(mv):
    """A filter that allows < 25 % of the documents takes the compaction path (ordered candidate list + scan of the
    candidates only); the result must be identical to masking inside the full scan, ties included."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    n = 9000
    ix = _idx(mv, capacity_pages=n, stride_rows=64, with_fp8=True, with_binary=True)
    base = [orc.synth_rows(5, i, 0, 50) for i in range(30)]
    ix.add([base[i % 30] for i in range(n)], [i // 4 for i in range(n)])  # duplicates -> ties; 4 pages per doc; ragged (50 of 64 rows)
    ix.remove_doc(3)
    ix.remove_page(4001)
    q = orc.synth_rows(4321, 3, 0, 20)
    n_docs = n // 4
    rng = np.random.default_rng(1)
    for frac in (0.2, 0.02, 0.0005):
        docs = sorted(set(rng.choice(n_docs, size=max(1, int(n_docs * frac)), replace=False).tolist()) | {3, 1000})
        allow = allow_bitmap(docs, n_docs)
        for mode, bvar in (("float", -1), ("float_fp8", -1), ("binary", 0), ("binary", 4)):
            if mode == "binary":  # the popcount cross-check and the FP4 MFMA default
                ix.set_option(_lib.MV_OPT_BINARY_VARIANT, bvar)
            res = []
            for pct in (0, 25):
                ix.set_option(_lib.MV_OPT_FILTER_COMPACT_PCT, pct)
                res.append(ix.query(q, 40, mode=mode, allow=allow))
            assert res[0][1].tolist() == res[1][1].tolist() and res[0][0].tolist() == res[1][0].tolist()
            sc = ix.score_all(q, mode=mode, allow=allow)
            ws, wi = orc.topk(sc, 40)
            assert res[1][1].tolist() == wi.tolist()
            assert all((p // 4) in docs and (p // 4) != 3 and p != 4001 for p in res[1][1].tolist())
    # a filter that allows nothing live
    s, i = ix.query(q, 5, allow=allow_bitmap([3], n_docs))
    assert len(i) == 0
    ix.close()


def test_topk_ties_k_edge_cases_and_large_k(mv):
    ix = _idx(mv, capacity_pages=6000, stride_rows=16)
    base = [orc.synth_rows(5, i, 0, 16) for i in range(40)]
    pages = [base[i % 40] for i in range(5000)]  # every page duplicated 125 times -> massive ties
    ix.add(pages)
    q = orc.synth_rows(4321, 1, 0, 8)
    sc = ix.score_all(q)
    want = np.array([orc.maxsim_bf16(q, p) for p in base], np.float32)
    np.testing.assert_allclose(sc[:40], want, rtol=RTOL)
    assert np.array_equal(sc, np.tile(sc[:40], 125))  # identical pages -> bit-identical scores
    for k in (1, 10, 130, 1024, 1500, 6000):
        s, i = ix.query(q, k)
        ws, wi = orc.topk(sc, k)
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()  # (score desc, id asc) exactly
    s, i = ix.query(q, 0)
    assert len(i) == 0
    ix.close()
    empty = _idx(mv, capacity_pages=4, stride_rows=16)
    s, i = empty.query(q, 5)
    assert len(i) == 0  # empty store (test_multivector.py:205-211)
    empty.close()


def test_large_k_radix_selection_matches_oracle_order(mv):
    """k > 32 over many distinct scores takes the radix-threshold path (3 histogram passes + compaction + one sort);
    result and order must be exactly the oracle's (score desc, id asc), also through a filter, for k around the
    survivor capacity, and for k beyond the number of live pages."""
    from morphik_core_amd.index import allow_bitmap

    n = 30_000
    ix = _idx(mv, capacity_pages=n, stride_rows=16, with_binary=True)
    ix.fill_synthetic(1234, 0, n, pages_per_doc=3)
    ix.remove_doc(77)
    q = orc.synth_rows(4321, 2, 0, 16)
    allow = allow_bitmap([d for d in range(n // 3 + 1) if d % 5 != 0])
    for al in (None, allow):
        sc = ix.score_all(q, allow=al)
        for k in (33, 100, 1000, 1024):
            s, i = ix.query(q, k, allow=al)
            ws, wi = orc.topk(sc, k)
            assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    # sign-bit scores: a handful of distinct values -> the survivors overflow and the sort cascade takes over
    sb = ix.score_all(q, mode="binary")
    for k in (64, 1000):
        s, i = ix.query(q, k, mode="binary")
        ws, wi = orc.topk(sb, k)
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    # k larger than the live pages of a narrow filter
    narrow = allow_bitmap([1, 2, 3])
    s, i = ix.query(q, 500, allow=narrow)
    ws, wi = orc.topk(ix.score_all(q, allow=narrow), 500)
    assert len(i) == 9 and i.tolist() == wi.tolist()
    ix.close()


def test_self_retrieval_ranks_first(mv):
    """core/tests/unit/test_multivector.py:166-181: querying with a stored page's own embedding ranks
    it first, scores non-increasing."""
    ix = _idx(mv, capacity_pages=128, stride_rows=32)
    ix.fill_synthetic(1234, 0, 100)
    page = ix.read_pages(17, 1)[0]
    s, i = ix.query(page, 5)
    assert i[0] == 17 and np.all(np.diff(s) <= 0)
    assert s[0] == pytest.approx(32.0, rel=2e-2)  # 32 unit rows matching themselves
    ix.close()


def test_logical_shards_equal_single_index(mv):
    """Row-sharding invariance: R shards with id_base offsets + merge == one index."""
    N, stride = 960, 32
    one = _idx(mv, capacity_pages=N, stride_rows=stride)
    one.fill_synthetic(1234, 0, N)
    q = orc.synth_rows(4321, 2, 0, 32)
    s1, i1 = one.query(q, 10)
    for R in (2, 4, 8):
        per = N // R
        parts = []
        for r in range(R):
            sh = _idx(mv, capacity_pages=per, stride_rows=stride, id_base=r * per)
            sh.fill_synthetic(1234, r * per, per)
            parts.append(sh.query(q, 10))
            sh.close()
        s = np.concatenate([p[0] for p in parts])
        i = np.concatenate([p[1] for p in parts])
        ms, mi = orc.topk(s, 10, ids=i)
        assert mi.tolist() == i1.tolist() and ms.tolist() == s1.tolist()
    one.close()


def test_full_size_one_million_pages_properties(mv):
    """BASELINE configs[2] at FULL size (1 M pages x 1024 patches x 128-d bf16 = 262 GB): the oracle cannot score it in
    test time, so parity is carried by size-independent properties over ALL pages plus sampled oracle scores:
    planted recall@10 = 1.0, sorted + idempotent top-k, top-k == selection over the full score vector, additivity of
    MaxSim over query rows (score(q1 ++ q2) = score(q1) + score(q2) on every page), 64 sampled pages vs the oracle."""
    import torch

    from morphik_core_amd import synth

    free_b, _total = torch.cuda.mem_get_info(0)
    N, stride = 1_000_000, 1024
    if free_b < N * stride * 256 + (8 << 30):
        pytest.skip(f"needs {N * stride * 256 / 2**30:.0f} GiB of free HBM, have {free_b / 2**30:.0f}")
    ix = _idx(mv, capacity_pages=N, stride_rows=stride)
    try:
        ix.fill_synthetic(synth.SEED_CORPUS, 0, N)
        queries = [orc.synth_rows(synth.SEED_QUERIES, qi, 0, 32) for qi in range(2)]
        spec = synth.planted_spec(queries, N, stride)
        assert synth.plant_neighbours(ix, spec) == 20
        all_scores = []
        for qi, q in enumerate(queries):
            s, i = ix.query(q, 10)
            planted = [p for (qq, _r, p, _row0, _rows) in spec if qq == qi]
            assert i.tolist() == planted  # rank order of the planted neighbours, recall@10 = 1.0
            assert all(s[j] > s[j + 1] for j in range(9))
            s2, i2 = ix.query(q, 10)
            assert i2.tolist() == i.tolist() and s2.tolist() == s.tolist()  # idempotent, bit for bit
            sc = ix.score_all(q)
            assert sc.shape == (N,) and np.isfinite(sc).all()
            ws, wi = orc.topk(sc, 10)
            assert wi.tolist() == i.tolist() and ws.tolist() == s.tolist()  # device selection == selection over every score
            ws, wi = orc.topk(sc, 1000)
            s3, i3 = ix.query(q, 1000)  # radix-threshold selection path at full size
            assert i3.tolist() == wi.tolist() and s3.tolist() == ws.tolist()
            all_scores.append(sc)
        both = ix.score_all(np.concatenate(queries))  # 64 query rows in one pass
        np.testing.assert_allclose(both, all_scores[0] + all_scores[1], rtol=2e-6, atol=1e-5)
        rng = np.random.default_rng(3)
        sample = np.sort(rng.choice(N, 64, replace=False))
        qf = orc.bf16_to_f32(queries[0])
        for p in sample.tolist() + [N - 1, 0]:
            want = orc.maxsim_float_np(qf, orc.bf16_to_f32(ix.read_pages(p, 1)))[0]
            assert abs(all_scores[0][p] - want) <= 1e-4 * abs(want)
    finally:
        ix.close()


def test_full_shard_fp8_fde_slabs_properties(mv):
    """BASELINE configs[3] / [4] at their per-GPU shard shape (10 M pages / 8 GPUs = 1.25 M pages x 1024 patches in the e4m3,
    FDE and sign-bit slabs, 210 GB).  The oracle cannot score 1.25 M pages in test time, so parity is
    carried by size-independent properties over ALL pages plus sampled oracle scores ON THE DEVICE'S OWN CODES:
    fp8 scan -- 48 sampled pages (+ first, last, planted) vs the oracle's fp8 MaxSim of the codes read back, top-k == the
    selection over the full score vector (k = 10 and the radix path at k = 1000), idempotent bit for bit, additive over
    query rows, the batched scan == the single scan; FDE -- sampled coarse scores vs the oracle's FDE of the same page,
    coarse top-k == selection over the full coarse vector, the coarse -> fp8-rerank pipeline == the oracle's composition
    (fp8 MaxSim of the coarse top-n on the device's codes), batched pipeline == query by query; sign bits -- sampled pages
    vs the SQL restatement.  Recall: planted recall@10 = 1.0 on every path; hard negatives (truth = the oracle's exact bf16
    scores of the hard set, which sits far above the background) within the floors measured by bench.py at this size
    (fp8 0.86, FDE-75 -> fp8 0.86, coarse@75 0.99, coarse@1000 1.0 on 64 queries; 32 queries here, floors 0.05 below)."""
    import torch

    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import synth_rows

    N, stride = 1_250_000, 1024
    per_page = stride * 128 + stride * 16 + 20480 + 16 + 40 * 4
    free_b, _total = torch.cuda.mem_get_info(0)
    if free_b < N * per_page + (8 << 30):
        pytest.skip(f"needs {N * per_page / 2**30:.0f} GiB of free HBM, have {free_b / 2**30:.0f}")
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_float=False, with_fp8=True, with_fde=True, with_binary=True)
    try:
        ix.fill_synthetic(synth.SEED_CORPUS, 0, N)
        pq = [orc.synth_rows(synth.SEED_QUERIES, qi, 0, 32) for qi in range(2)]
        pspec = synth.planted_spec(pq, N, stride)
        hq = [synth_rows(synth.SEED_QUERIES, 1000 + j, 32) for j in range(32)]
        taken = {t[2] for t in pspec}
        hspec = [t for t in synth.hard_spec(hq, N, stride) if t[2] not in taken]
        assert synth.plant_neighbours_any(ix, pspec + hspec, synth.SEED_CORPUS, stride) == len(pspec) + len(hspec)
        planted = [[p for (qq, _r, p, _a, _b) in pspec if qq == qi] for qi in range(2)]

        overrides = {pp: (row0, rows) for (_q, _r, pp, row0, rows) in pspec + hspec}

        def host_page(p):  # the bf16 rows the slabs were built from: the generator's page with the planted rows on top
            pg = orc.synth_rows(synth.SEED_CORPUS, p, 0, stride)
            if p in overrides:
                row0, rows = overrides[p]
                pg[row0 : row0 + rows.shape[0]] = rows
            return pg

        rng = np.random.default_rng(3)
        sample = sorted(set(rng.choice(N, 48, replace=False).tolist() + [0, N - 1] + planted[0][:3] + [hspec[0][2]]))
        ocfg = orc.FdeConfig.reference_default()
        # ---- fp8 scan
        fp8_scores = []
        for qi, q in enumerate(pq):
            s, i = ix.query(q, 10, mode="float_fp8")
            assert i.tolist() == planted[qi]  # recall@10 = 1.0, in rank order
            s2, i2 = ix.query(q, 10, mode="float_fp8")
            assert i2.tolist() == i.tolist() and s2.tolist() == s.tolist()
            sc = ix.score_all(q, mode="float_fp8")
            assert sc.shape == (N,) and np.isfinite(sc).all()
            for k in (10, 1000):
                ws, wi = orc.topk(sc, k)
                s3, i3 = ix.query(q, k, mode="float_fp8")
                assert i3.tolist() == wi.tolist() and s3.tolist() == ws.tolist()
            fp8_scores.append(sc)
        both = ix.score_all(np.concatenate(pq), mode="float_fp8")
        np.testing.assert_allclose(both, fp8_scores[0] + fp8_scores[1], rtol=2e-6, atol=1e-5)
        qf = orc.bf16_to_f32(pq[0])
        for p in sample:
            codes, inv = ix.read_fp8(p, 1)
            wc, winv = orc.quantize_page_fp8(host_page(p), stride)
            assert np.array_equal(codes[0], wc) and inv[0] == np.float32(winv), p  # the slab holds the oracle's codes of that page
            want = orc.maxsim_fp8_np(qf, codes, inv, n_rows=[stride])[0]
            assert abs(fp8_scores[0][p] - want) <= 1e-4 * abs(want), p
        for (s, i), sc in zip(ix.query_batch(pq, 10, mode="float_fp8"), fp8_scores):  # one slab pass for both queries
            ws, wi = orc.topk(sc, 10)
            assert i.tolist() == wi.tolist()
            np.testing.assert_allclose(s, ws, rtol=1e-5)
        # ---- FDE coarse scan and the coarse -> fp8 rerank pipeline
        fq = orc.fde_encode(ocfg, qf, True)
        coarse = ix.score_all(pq[0], mode="fde")
        assert coarse.shape == (N,) and np.isfinite(coarse).all()
        for p in sample:
            fd = orc.f32_to_bf16(orc.fde_encode(ocfg, orc.bf16_to_f32(host_page(p)), False))[None]
            want = orc.fde_coarse_scores(fq, fd, use_cosine=True)[0]
            assert abs(coarse[p] - want) <= 2e-3 * abs(want) + 2e-4, p
        for n_coarse in (75, 1000):
            ix.set_option(_lib.MV_OPT_FDE_COARSE_N, n_coarse)
            cs, ci = ix.query(pq[0], n_coarse, mode="fde")
            ws, wi = orc.topk(coarse, n_coarse)
            assert ci.tolist() == wi.tolist()
            s, i = ix.query(pq[0], 10, mode="fde_then_float")
            assert i.tolist() == planted[0]
            rer = fp8_scores[0][ci]  # the exhaustive fp8 scan's scores of the coarse candidates: same kernel arithmetic, per page
            order = np.lexsort((np.arange(n_coarse), -rer.astype(np.float64)))[:10]
            assert i.tolist() == ci[order].tolist()
            np.testing.assert_allclose(s, rer[order], rtol=1e-5)
            bres = ix.query_batch(pq + hq[:14], 10, mode="fde_then_float")
            for q, (bs, bi) in zip(pq + hq[:14], bres):
                s1, i1 = ix.query(q, 10, mode="fde_then_float")
                np.testing.assert_allclose(bs, s1, rtol=1e-5)
                for j in np.nonzero(bi != i1)[0]:  # hard negatives: two pages inside one rounding step may swap between the two rerank kernels
                    nb = [x for x in (j - 1, j + 1) if 0 <= x < 10]
                    assert min(abs(s1[j] - s1[x]) for x in nb) <= 2e-5 * abs(s1[j]), (j, bi.tolist(), i1.tolist())
        # ---- sign bits
        bq = orc.sign_pack(qf)
        bsc = ix.score_all(pq[0], mode="binary")
        for p in sample[:12]:
            assert float(bsc[p]) == orc.maxsim_binary(orc.sign_pack(orc.bf16_to_f32(host_page(p))), bq), p
        assert ix.query(pq[0], 10, mode="binary")[1].tolist() == planted[0]
        # ---- hard negatives: truth = exact bf16 scores of the hard set by the oracle
        r8, r75, rc75, rc1000 = [], [], [], []
        for j, q in enumerate(hq):
            pages = synth.hard_pages_of(hspec, j)
            exact = np.array([orc.maxsim_bf16(q, host_page(p)) for p in pages], np.float32)
            top, _info = synth.exact_truth_from_scores(pages, exact)
            r8.append(synth.recall_at_k(ix.query(q, 10, mode="float_fp8")[1].tolist(), top))
            ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 75)
            r75.append(synth.recall_at_k(ix.query(q, 10, mode="fde_then_float")[1].tolist(), top))
            rc75.append(synth.recall_at_k(ix.query(q, 75, mode="fde")[1].tolist(), top))
            rc1000.append(synth.recall_at_k(ix.query(q, 1000, mode="fde")[1].tolist(), top))
        r8, r75, rc75, rc1000 = (float(np.mean(x)) for x in (r8, r75, rc75, rc1000))
        print(f"full shard, hard negatives (32 queries): fp8 {r8:.3f} fde75->fp8 {r75:.3f} coarse@75 {rc75:.3f} coarse@1000 {rc1000:.3f}")
        assert r8 >= 0.80 and r75 >= 0.80 and rc75 >= 0.95 and rc1000 >= 0.99
    finally:
        ix.close()


def test_planted_neighbours_recall_and_sampled_parity_midsize(mv):
    """20k pages x 1024 patches (5.2 GB): recall@10 == 1.0 on planted neighbours; sampled oracle parity."""
    from morphik_core_amd import synth

    N, stride = 20000, 1024
    ix = _idx(mv, capacity_pages=N, stride_rows=stride)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, N)
    queries = [orc.synth_rows(synth.SEED_QUERIES, qi, 0, 32) for qi in range(4)]
    spec = synth.planted_spec(queries, N, stride)
    assert synth.plant_neighbours(ix, spec) == 40
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(N, 64, replace=False))
    for qi, q in enumerate(queries):
        s, i = ix.query(q, 10)
        planted = [p for (qq, r, p, _, _) in spec if qq == qi]
        assert i.tolist() == planted  # rank order by construction (sigma rises with rank)
        assert synth.recall_at_k(i, planted) == 1.0
        # oracle on planted + sampled pages, regenerated on the CPU with the same overrides
        full = ix.score_all(q)
        check = sorted(set(sample.tolist()) | set(planted))
        for p in check:
            page = orc.synth_rows(synth.SEED_CORPUS, p, 0, stride)
            for (qq, r, pp, row0, rows) in spec:
                if pp == p:
                    page[row0 : row0 + rows.shape[0]] = rows
            want = orc.maxsim_float_np(orc.bf16_to_f32(q), orc.bf16_to_f32(page)[None])[0]
            assert abs(full[p] - want) <= RTOL * abs(want)
    ix.close()


# ------------------------------------------------------------------ binary path
def test_sign_pack_matches_reference_golden(mv, golden_dir):
    from morphik_core_amd.index import sign_pack

    g = np.load(os.path.join(golden_dir, "sign_pack.npz"))
    for ci in range(int(g["n_cases"])):
        assert np.array_equal(sign_pack(g[f"x{ci}"]), g[f"packed{ci}"]), f"case {ci}"
    # morphik_rust/src/binary_ops.rs:309-320
    assert sign_pack(np.array([1, -1, 1, -1, -1, 1, -1, 1], np.float32))[0, 0] == 0b10100101


def test_hamming_batch_matches_reference_golden(mv, golden_dir):
    from morphik_core_amd.index import hamming_batch

    g = np.load(os.path.join(golden_dir, "hamming.npz"))
    assert hamming_batch(bytes(g["q"]), [bytes(x) for x in g["b"]]) == g["batch"].tolist()
    assert hamming_batch(bytes([0xF0, 0xAA]), [bytes([0xF0, 0x55])]) == [8]  # binary_ops.rs:322-334
    with pytest.raises(ValueError):
        hamming_batch(b"ab", [b"abc"])


BINARY_VARIANTS = [0, 4]  # 0 = popcount on the VALU (the independent cross-check), 4 = FP4 MFMA (default); identical integers required


def _set_binary_variant(ix, variant):
    from morphik_core_amd import _lib

    ix.set_option(_lib.MV_OPT_BINARY_VARIANT, variant)


@pytest.mark.parametrize("variant", BINARY_VARIANTS)
@pytest.mark.parametrize("stride", [64, 80, 208])
def test_binary_maxsim_exact_vs_sql_restatement(mv, variant, stride):
    lens = [64, 1, 0, 33, 64, 17, 64, 64, 5, stride, stride - 1, stride - 15, 16, 48]
    rng = np.random.default_rng(4)
    pages = [rng.standard_normal((n, 128)).astype(np.float32) for n in lens]
    pages[4][3, :7] = 0.0  # zeros quantise to 0 bits
    pages[5][:, :] = -1.0  # all-zero bit rows
    pages[6][:, :] = 1.0   # all-one bit rows
    ix = _idx(mv, capacity_pages=len(lens), stride_rows=stride, with_float=False, with_binary=True)  # last page ends the slab
    _set_binary_variant(ix, variant)
    ix.add(pages)
    for nq in (1, 9, 16, 32, 40, 64, 70, 130):
        q = rng.standard_normal((nq, 128)).astype(np.float32)
        want = np.array([orc.maxsim_binary(orc.sign_pack(p) if len(p) else np.zeros((0, 16), np.uint8), orc.sign_pack(q)) for p in pages])
        got = ix.score_all(q, mode="binary")
        assert got.astype(np.float64).tolist() == want.tolist()  # integer arithmetic: exact
        s, i = ix.query(q, 4, mode="binary")
        ws, wi = orc.topk(want.astype(np.float32), 4)
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    ix.close()


@pytest.mark.parametrize("variant", BINARY_VARIANTS)
def test_binary_maxsim_reference_known_ranking(mv, variant):
    """core/tests/unit/test_multivector.py:214-256 -> doc1 first with 1.0, doc2 0.0."""
    half = np.concatenate([np.ones(64), -np.ones(64)]).astype(np.float32)
    ix = _idx(mv, capacity_pages=4, stride_rows=16, with_float=True, with_binary=True)
    _set_binary_variant(ix, variant)
    ix.add([np.stack([half] * 3), np.stack([-half] * 3)])
    s, i = ix.query(half[None], 2, mode="binary")
    assert i.tolist() == [0, 1] and s.tolist() == [1.0, 0.0]
    ix.close()


@pytest.mark.parametrize("variant", BINARY_VARIANTS)
def test_binary_maxsim_synthetic_slab_1024(mv, variant):
    ix = _idx(mv, capacity_pages=64, stride_rows=1024, with_binary=True)
    _set_binary_variant(ix, variant)
    ix.fill_synthetic(1234, 0, 48)
    pages = ix.read_pages(0, 48)
    bits = np.stack([orc.sign_pack(orc.bf16_to_f32(p)) for p in pages])
    for nq in (32, 20):
        q = orc.synth_rows(4321, 0, 0, nq)
        want = orc.maxsim_binary_np(bits, orc.sign_pack(orc.bf16_to_f32(q)))
        got = ix.score_all(q, mode="binary")
        assert got.astype(np.float64).tolist() == want.tolist()
    ix.close()
    # uniform corpora of whole 256-row pages whose page count is NOT a multiple of the burst form's four pages per workgroup,
    # queries of 1..64 rows (1..4 query tiles, the 64-row pass boundary) and beyond (two passes)
    for stride, n in ((256, 1001), (512, 403), (1024, 1030)):
        ix = _idx(mv, capacity_pages=n, stride_rows=stride, with_binary=True)
        ix.fill_synthetic(77, 0, n)
        for nq in (1, 17, 48, 64, 70):
            q = orc.synth_rows(4321, 30 + nq, 0, nq)
            _set_binary_variant(ix, 0)
            want = ix.score_all(q, mode="binary")  # the popcount kernel, itself checked against the oracle above
            _set_binary_variant(ix, variant)
            assert np.array_equal(ix.score_all(q, mode="binary"), want), (stride, n, nq)
        ix.close()


@pytest.mark.parametrize("n_rows", [1000, 1024])  # ragged / uniform pages
def test_binary_variants_agree_with_filter_and_tombstones_midsize(mv, n_rows):
    """The sign-bit kernels over 20 k pages with a doc filter and tombstones: identical vectors (20 003 pages: a partial last workgroup;
    a run of 140 tombstoned pages; queries of one and of several row tiles, one and two passes)."""
    n = 20_003
    ix = _idx(mv, capacity_pages=n, stride_rows=1024, with_binary=True)
    ix.fill_synthetic(1234, 0, n, n_rows=n_rows, pages_per_doc=7)
    for d in range(40, 60):  # 140 consecutive pages gone: whole workgroups have nothing to stream
        ix.remove_doc(d)
    for d in (3, 11, 500):
        ix.remove_doc(d)
    from morphik_core_amd.index import allow_bitmap

    allow = allow_bitmap([d for d in range(0, n // 7 + 1) if d % 3 != 1])
    for nq in (32, 9, 40, 100):
        q = orc.synth_rows(4321, nq, 0, nq)
        outs = []
        for v in BINARY_VARIANTS + [-1]:
            _set_binary_variant(ix, v)
            outs.append((ix.score_all(q, mode="binary", allow=allow), ix.query(q, 10, mode="binary", allow=allow), ix.score_all(q, mode="binary")))
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][2], o[2])
            assert outs[0][1][1].tolist() == o[1][1].tolist() and outs[0][1][0].tolist() == o[1][0].tolist()
        assert np.isinf(outs[0][0]).sum() > n // 4  # the filter really masked pages
    ix.close()


# ------------------------------------------------------------------ FDE
def test_fde_encode_matches_oracle(mv):
    from morphik_core_amd.index import FdeConfig, fde_encode

    cfg = FdeConfig()
    ocfg = orc.FdeConfig.reference_default()
    assert cfg.output_dim == 10240
    for scalar in ("0", "1"):  # f32-MFMA kernel (default) and the scalar kernel: the same fmaf chains
        os.environ["MV_FDE_SCALAR"] = scalar
        for n in (1, 15, 16, 17, 32, 63, 64, 65, 200, 1030):
            x = orc.bf16_to_f32(orc.synth_rows(9, n, 0, n))
            x[0, :5] *= 3.7  # not unit norm, not bf16-exact
            for is_q in (True, False):
                got = fde_encode(x, cfg, is_query=is_q)
                want = orc.fde_encode(ocfg, x, is_q)
                # partitions agree bit for bit (same fmaf chain) -> the support of the vectors is identical
                assert np.array_equal(got != 0, want != 0), (scalar, n, is_q)
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    os.environ.pop("MV_FDE_SCALAR", None)
    # other FDE shapes through the MFMA kernel (fewer repetitions / projections / narrower buckets)
    for (R, NS, PD) in ((3, 4, 8), (20, 6, 16), (7, 1, 16)):
        c2 = FdeConfig(num_repetitions=R, num_simhash_projections=NS, projection_dimension=PD, seed=5)
        o2 = orc.FdeConfig(128, R, NS, PD, 5)
        x = orc.bf16_to_f32(orc.synth_rows(11, 0, 0, 77))
        for is_q in (True, False):
            got, want = fde_encode(x, c2, is_query=is_q), orc.fde_encode(o2, x, is_q)
            assert np.array_equal(got != 0, want != 0)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_fde_slab_same_scores_from_both_encode_kernels(mv):
    from morphik_core_amd import _lib

    q = orc.synth_rows(4321, 0, 0, 32)
    outs = []
    for variant in (0, 1, 4):  # scalar kernel, f32-MFMA kernel, two-pass bf16-slab form (hash + one-hot MFMA bucket sums: the default)
        ix = _idx(mv, capacity_pages=300, stride_rows=208, with_fde=True)
        ix.set_option(_lib.MV_OPT_FDE_ENCODE_VARIANT, variant)
        ix.fill_synthetic(1234, 0, 300, n_rows=200)
        outs.append(ix.score_all(q, mode="fde"))
        ix.close()
    np.testing.assert_allclose(outs[0], outs[1], rtol=2e-3, atol=2e-4)  # bf16 slab, bucket sums in different orders
    np.testing.assert_allclose(outs[2], outs[1], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(outs[2], outs[0], rtol=2e-3, atol=2e-4)
    # the bf16-slab kernel on ragged pages (1 .. stride rows, an empty page) added from the host as bf16 and as fp32 (fp32 rows take the
    # f32-MFMA kernel: the slab kernel only sees pages that are already bf16), against the oracle's FDE of the same bf16 rows
    rng = np.random.default_rng(12)
    stride = 64
    lens = [1, 15, 16, 17, 33, 64, 0, 48] + [int(x) for x in rng.integers(1, stride + 1, 40)]
    pages = [orc.synth_rows(77, p, 0, stride)[:n] for p, n in enumerate(lens)]
    ocfg = orc.FdeConfig.reference_default()
    fq = orc.fde_encode(ocfg, orc.bf16_to_f32(q), True)
    want = np.array([orc.fde_coarse_scores(fq, orc.f32_to_bf16(orc.fde_encode(ocfg, orc.bf16_to_f32(pg), False))[None], use_cosine=True)[0]
                     if len(pg) else 0.0 for pg in pages], np.float32)
    for as_f32 in (False, True):
        for variant in (4, 1, 0):  # two-pass MFMA form (default), the f32-pipe form, the scalar kernel
            ix = _idx(mv, capacity_pages=len(pages), stride_rows=stride, with_fde=True)
            ix.set_option(_lib.MV_OPT_FDE_ENCODE_VARIANT, variant)
            ix.add([orc.bf16_to_f32(pg) if as_f32 else pg for pg in pages])
            got = ix.score_all(q, mode="fde")
            live = np.array([n > 0 for n in lens])
            np.testing.assert_allclose(got[live], want[live], rtol=2e-3, atol=2e-4)
            ix.close()


@pytest.mark.parametrize("R,NS,PD", [(10, 5, 16), (8, 5, 16), (7, 4, 16), (20, 4, 8), (3, 6, 16)])
def test_fde_index_with_other_fde_shapes(mv, R, NS, PD):
    """Indexes built with FDE shapes other than the reference's 20 x 32 x 16: 5 120 dims (the second unrolled single-query scan), 4 096
    (batched GEMM scan with 16 K chunks), 1 792 / 2 560 / 3 072 dims (the any-width scan; the batched entry point then runs query by query
    unless the width is a multiple of 1 024).  The document-encode kernel of the corpus build falls back where its 7-column-tile form does
    not apply.  Coarse scores against the oracle's FDE of the same rows; the batched pipeline against the single-query one."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import FdeConfig

    cfg = FdeConfig(num_repetitions=R, num_simhash_projections=NS, projection_dimension=PD, seed=9)
    ocfg = orc.FdeConfig(128, R, NS, PD, 9)
    N, stride = 333, 64
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_fde=True, fde=cfg)
    assert ix.fde_config.output_dim == R * (1 << NS) * PD
    ix.fill_synthetic(1234, 0, N, n_rows=50, pages_per_doc=3)
    ix.remove_doc(4)
    pages = ix.read_pages(0, N)[:, :50]
    queries = [orc.synth_rows(4321, 60 + b, 0, 20 + b) for b in range(5)]
    fds = np.stack([orc.f32_to_bf16(orc.fde_encode(ocfg, orc.bf16_to_f32(pg), False)) for pg in pages])
    for q in queries[:2]:
        fq = orc.fde_encode(ocfg, orc.bf16_to_f32(q), True)
        want = orc.fde_coarse_scores(fq, fds, use_cosine=True)
        got = ix.score_all(q, mode="fde")
        live = np.isfinite(got)
        assert (~live).sum() == 3  # the tombstoned document's pages
        np.testing.assert_allclose(got[live], want[live], rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(want).max())))
    for mode, k in (("fde", 40), ("fde_then_float", 5)):
        batch = ix.query_batch(queries, k, mode=mode)
        for q, (s, i) in zip(queries, batch):
            ws, wi = ix.query(q, k, mode=mode)
            assert len(i) == len(wi)
            np.testing.assert_allclose(s, ws, rtol=1e-4, atol=1e-6)
            assert len(set(i.tolist()) & set(wi.tolist())) >= len(wi) - 2  # near-ties may swap at the cut
    ix.close()


def test_fde_coarse_scan_and_pipeline(mv):
    from morphik_core_amd import _lib, synth

    N, stride = 600, 64
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_fde=True)
    ix.fill_synthetic(1234, 0, N)
    q = orc.synth_rows(4321, 0, 0, 32)
    spec = synth.planted_spec([q], N, stride, n_ranks=5)
    # plant BEFORE the FDE slab is built: rebuild by re-adding planted pages is not needed here,
    # so build a second index from host pages instead
    pages = ix.read_pages(0, N)
    for (_, _, p, row0, rows) in spec:
        pages[p, row0 : row0 + rows.shape[0]] = rows
    ix.close()
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_fde=True)
    ix.add(list(pages))
    ocfg = orc.FdeConfig.reference_default()
    qf = orc.bf16_to_f32(q)
    fq = orc.fde_encode(ocfg, qf, True)
    fds = np.stack([orc.fde_encode(ocfg, orc.bf16_to_f32(p), False) for p in pages])
    for cosine in (1, 0):
        ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
        want = orc.fde_coarse_scores(fq, orc.f32_to_bf16(fds), use_cosine=bool(cosine))
        got = ix.score_all(q, mode="fde")
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4)
    # coarse -> exact rerank: the planted pages come back on top with their exact scores
    ix.set_option(_lib.MV_OPT_FDE_COSINE, 1)
    s, i = ix.query(q, 5, mode="fde_then_float")
    exact = orc.maxsim_float_np(qf, orc.bf16_to_f32(pages))
    planted = [p for (_, _, p, _, _) in spec]
    assert i.tolist() == planted
    np.testing.assert_allclose(s, exact[planted], rtol=RTOL)
    ix.close()


@pytest.mark.parametrize("n", [1, 3, 17, 255, 2049, 70_001, 300_017])
def test_fde_scan_row_quarters_bit_identical_to_the_register_scan(mv, n):
    """The default coarse scan (round 5: row quarters through the nt LDS-DMA ring -- a fresh workgroup per 16 rows, wave w
    streams the w-th quarter of every row and keeps its slice of the query FDE, one barrier joins the four partial sums; filter
    and 1/norm evaluated per workgroup) and the register scan (variant 0: one wave per row on plain nt loads) share ONE
    arithmetic order: every score is BIT-identical -- unfiltered, with tombstoned pages, with a doc filter, cosine on and off --
    for page counts that give a single partial unit, whole units and a partial last unit; the top-k is identical too (variant 0
    accumulates the selection's first radix histogram itself, the default leaves it to the selection), and a sample agrees with
    the oracle's coarse scores."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    stride = 16
    ix = _idx(mv, capacity_pages=n, stride_rows=stride, with_fde=True, with_float=False, with_fp8=True)
    ix.fill_synthetic(1234, 0, n, n_rows=stride, pages_per_doc=3)
    rng = np.random.default_rng(n)
    for p in rng.choice(n, size=min(n // 3, 500), replace=False).tolist():
        ix.remove_page(int(p))
    n_docs = (n + 2) // 3
    allow = allow_bitmap([d for d in range(n_docs) if d % 5 != 1], n_docs)
    q = orc.synth_rows(4321, 7, 0, 32)
    got = {}
    for v in (0, 5, 6, -1):  # 6 (= -1, round 6): the row-quarter form with one 256 KiB-aligned block of the slab per workgroup (12 / 13 rows)
        ix.set_option(_lib.MV_OPT_FDE_SCAN_VARIANT, v)
        for cosine in (1, 0):
            ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
            got[v, cosine, "all"] = ix.score_all(q, mode="fde")
            got[v, cosine, "flt"] = ix.score_all(q, mode="fde", allow=allow)
        ix.set_option(_lib.MV_OPT_FDE_COSINE, 1)
        got[v, "top"] = ix.query(q, min(200, n), mode="fde", allow=allow)
        got[v, "top_all"] = ix.query(q, min(1000, n), mode="fde")
    for v in (5, 6, -1):
        for cosine in (1, 0):
            for key in ("all", "flt"):
                a, b = got[0, cosine, key], got[v, cosine, key]
                assert a.shape == (n,) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (v, cosine, key)
        for key in ("top", "top_all"):
            assert np.array_equal(got[0, key][1], got[v, key][1]) and np.array_equal(got[0, key][0], got[v, key][0])
    assert np.isneginf(got[5, 1, "flt"]).sum() >= np.isneginf(got[5, 1, "all"]).sum()
    with pytest.raises(Exception):  # the forms that lost are gone, not silently rerouted
        ix.set_option(_lib.MV_OPT_FDE_SCAN_VARIANT, 3)
        ix.score_all(q, mode="fde")
    ix.set_option(_lib.MV_OPT_FDE_SCAN_VARIANT, -1)
    # ... and the scores are the oracle's (sample of live pages; the slab's own FDE rows)
    live = np.flatnonzero(np.isfinite(got[5, 1, "all"]))[:: max(1, n // 40)][:40]
    if live.size:
        ocfg = orc.FdeConfig.reference_default()
        fq = orc.fde_encode(ocfg, orc.bf16_to_f32(q), True)
        rows = np.concatenate([ix.read_fde(int(p), 1) for p in live])
        want = orc.fde_coarse_scores(fq, orc.f32_to_bf16(rows), use_cosine=True)
        np.testing.assert_allclose(got[5, 1, "all"][live], want, rtol=2e-3, atol=2e-4)
    ix.close()


def test_gpu_fde_encoder_against_the_pinned_reference_fixture_or_its_stand_in(mv, tmp_path, monkeypatch):
    """mv_fde_encode measured against tests/golden/fde.npz -- the reference extension's own document / query
    encodings (oracle/gen_golden_fde.py) -- when that fixture exists; here, where the extension does not import, the SAME check runs
    on a fixture the recipe writes from the oracle-backed stand-in, so the code path is exercised on the device either way.  And the
    fixture's vectors drive the bring-your-own-FDE entry points (mv_index_import_fde / mv_query_topk_fde): planted page first."""
    import importlib.util

    from morphik_core_amd.index import fde_encode
    from tests.test_oracle_golden import GOLDEN, ROOT, fde_pin_report

    path = os.path.join(GOLDEN, "fde.npz")
    pinned = os.path.exists(path)
    if not pinned:
        spec = importlib.util.spec_from_file_location("gen_golden_fde", os.path.join(ROOT, "oracle", "gen_golden_fde.py"))
        g = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(g)
        import tests.fake_fde_module as fm

        monkeypatch.setattr(g, "find_extension", lambda: fm)
        path = str(tmp_path / "fde.npz")
        monkeypatch.setattr(g, "OUT", path)
        assert g.main() == 0
    rep = fde_pin_report(path, lambda p: fde_encode(p, is_query=False), lambda q: fde_encode(q, is_query=True))
    print("FDE pin report (mv_fde_encode vs %s):" % ("reference extension" if pinned else "oracle stand-in"), rep)
    if not pinned:  # against the oracle the device encoder is the parity-tested one: same partitions, fp32 summation order aside
        assert rep["median_cosine_doc_vectors"] > 0.99999 and rep["max_abs_diff_doc"] < 1e-3
    z = np.load(path)
    pages, queries = z["pages"], z["queries"]
    ix = _idx(mv, capacity_pages=pages.shape[0], stride_rows=pages.shape[1], with_fde=True)
    ix.add([orc.f32_to_bf16(p) for p in pages])
    ix.import_fde(0, z["doc_fde"], n_pages=pages.shape[0])
    for j, q in enumerate(queries):
        s, i = ix.query(orc.f32_to_bf16(q), 3, mode="fde_then_float", q_fde=z["q_fde"][j])
        assert i[0] == 3 * j + 1
    with pytest.raises(ValueError):
        ix.import_fde(0, z["doc_fde"][:, :-1])  # another width is refused, not regrouped
    ix.close()


@pytest.mark.parametrize("nq", [1, 15, 16, 32, 40, 70, 129])
def test_query_fde_encode_kernels_agree(mv, nq):
    """The query's FDE has three kernels (latency form: one block per repetition -- the default; bulk f32-MFMA; scalar).
    All are the same k-ordered fmaf chains: identical partitions, so the coarse scores agree to accumulation order, and
    they match the oracle."""
    from morphik_core_amd import _lib

    N = 300
    ix = _idx(mv, capacity_pages=N, stride_rows=64, with_fde=True)
    ix.fill_synthetic(1234, 0, N)
    q = orc.synth_rows(4321, 200 + nq, 0, nq)
    got = {}
    for v in (2, 1, 0):
        ix.set_option(_lib.MV_OPT_FDE_QUERY_ENCODE_VARIANT, v)
        got[v] = ix.score_all(q, mode="fde")
    np.testing.assert_allclose(got[1], got[2], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got[0], got[2], rtol=2e-5, atol=1e-6)
    ocfg = orc.FdeConfig.reference_default()
    pages = ix.read_pages(0, N)
    fq = orc.fde_encode(ocfg, orc.bf16_to_f32(q), True)
    fds = np.stack([orc.fde_encode(ocfg, orc.bf16_to_f32(p), False) for p in pages])
    want = orc.fde_coarse_scores(fq, orc.f32_to_bf16(fds), use_cosine=True)
    np.testing.assert_allclose(got[2], want, rtol=2e-3, atol=2e-4)
    ix.close()


# ------------------------------------------------------------------ batched queries (one slab pass, MFMA-bound form)
def test_fde_batched_coarse_scan_forms_agree_on_a_corpus_of_many_tiles_per_workgroup(mv):
    """The forms of the batched coarse pass (MV_OPT_FDE_BATCH_VARIANT 0 / 3 / 5) on a
    corpus large enough that every workgroup walks several tiles -- the deep-ring form's main phase takes groups of four 32-page
    tiles, which the small corpora of the test below never reach: 41 013 pages = 1 282 tiles of 32 (five per workgroup on 256 CUs,
    a partial last tile), tombstones, 32 + 5 requests: the same scores and ids bit for bit."""
    from morphik_core_amd import _lib

    n = 41_013
    ix = _idx(mv, capacity_pages=n, stride_rows=16, with_fde=True, with_float=False)
    ix.fill_synthetic(1234, 0, n, pages_per_doc=3)
    ix.remove_doc(7)
    queries = [orc.synth_rows(4321, b, 0, 24) for b in range(37)]
    for cosine in (1, 0):
        ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
        ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 0)
        want = ix.query_batch(queries, 50, mode="fde")
        ws, wi = ix.query(queries[0], 50, mode="fde")
        np.testing.assert_allclose(want[0][0], ws, rtol=1e-4, atol=1e-6)
        for form in (3, 5):
            ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, form)
            got = ix.query_batch(queries, 50, mode="fde")
            for (s0, i0), (sx, ix_) in zip(want, got):
                assert i0.tolist() == ix_.tolist() and s0.tolist() == sx.tolist(), (cosine, form)
    ix.close()


def test_fde_placement_trial_changes_where_the_slab_lives_and_nothing_else(mv):
    """mv_index_fde_placement_trial: candidates for the FDE slab are timed under the batched pass and the fastest kept.  Whatever it keeps,
    the batched and the single-query answers are the same bit for bit, later appends land in the slab it kept, an empty index and an
    index without an FDE slab behave as the header says."""
    n = 9_000
    ix = _idx(mv, capacity_pages=n + 500, stride_rows=16, with_fde=True, with_float=True)
    ix.fill_synthetic(1234, 0, n, pages_per_doc=3)
    ix.remove_doc(5)
    queries = [orc.synth_rows(4321, b, 0, 24) for b in range(20)]
    want = ix.query_batch(queries, 30, mode="fde")
    want1 = ix.query(queries[3], 30, mode="fde_then_float")
    moved_total = 0
    for trials in (0, 1, 4):
        before, after, moves = ix.fde_placement_trial(trials)
        assert before > 0 and after > 0 and 0 <= moves <= trials and (moves == 0 or after < before)
        moved_total += moves
        got = ix.query_batch(queries, 30, mode="fde")
        for (s0, i0), (s1, i1) in zip(want, got):
            assert i0.tolist() == i1.tolist() and s0.tolist() == s1.tolist()
        s1, i1 = ix.query(queries[3], 30, mode="fde_then_float")
        assert i1.tolist() == want1[1].tolist() and s1.tolist() == want1[0].tolist()
    ix.fill_synthetic(1234, n, 500, pages_per_doc=3)  # appends encode into the slab the trial kept
    ref = _idx(mv, capacity_pages=n + 500, stride_rows=16, with_fde=True, with_float=True)
    ref.fill_synthetic(1234, 0, n + 500, pages_per_doc=3)
    ref.remove_doc(5)
    for (s0, i0), (s1, i1) in zip(ref.query_batch(queries, 30, mode="fde"), ix.query_batch(queries, 30, mode="fde")):
        assert i0.tolist() == i1.tolist() and s0.tolist() == s1.tolist()
    ref.close()
    ix.close()
    empty = _idx(mv, capacity_pages=4096, stride_rows=16, with_fde=True, with_float=False)
    before, after, moves = empty.fde_placement_trial(2)  # timed over the whole (empty) slab
    assert before > 0 and after > 0
    empty.close()
    from morphik_core_amd._lib import MvError

    nofde = _idx(mv, capacity_pages=64, stride_rows=16)
    with pytest.raises(MvError):
        nofde.fde_placement_trial(1)
    nofde.close()


@pytest.mark.parametrize("n,stride", [(40, 16), (64, 16), (700, 32), (5000, 16)])
def test_fde_batched_coarse_scan_matches_the_single_query_scan(mv, n, stride):
    """mv_query_topk_batch in FDE mode: ONE pass over the FDE slab per 32 queries (bf16 MFMA, query FDE as bf16 hi + lo;
    37 queries = a pass with two query tiles + a pass with one).
    Coarse scores must agree with the single-query scan (fp32 query) to ~1e-5, through tombstones, a shared filter and
    per-query filters, for corpus sizes around the 64-page tile (partial last tile, fewer tiles than CUs) and with the
    radix selection (k > 32, n > 4096)."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    ix = _idx(mv, capacity_pages=n + 7, stride_rows=stride, with_fde=True, with_float=False)
    ix.fill_synthetic(1234, 0, n, pages_per_doc=3)
    ix.remove_doc(2)
    n_docs = (n + 2) // 3
    nq = 37
    queries = [orc.synth_rows(4321, b, 0, 20 + (b * 5) % 13) for b in range(nq)]  # ragged lengths, padded to 32 rows
    shared = allow_bitmap([d for d in range(n_docs) if d % 4 != 1])
    per_q = [None if b % 3 == 0 else allow_bitmap([d for d in range(n_docs) if (d + b) % 3 != 0]) for b in range(nq)]
    for cosine in (1, 0):
        ix.set_option(_lib.MV_OPT_FDE_COSINE, cosine)
        for k in (10, 64):
            kk = min(k, n)
            for kind in ("none", "shared", "per_query"):
                if kind == "none":
                    got = ix.query_batch(queries, kk, mode="fde")
                elif kind == "shared":
                    got = ix.query_batch(queries, kk, mode="fde", allow=shared)
                else:
                    got = ix.query_batch(queries, kk, mode="fde", allows=per_q, n_docs=n_docs)
                for b, (s, i) in enumerate(got):
                    al = None if kind == "none" else shared if kind == "shared" else per_q[b]
                    ws, wi = ix.query(queries[b], kk, mode="fde", allow=al)
                    assert len(i) == len(wi)
                    np.testing.assert_allclose(s, ws, rtol=1e-4, atol=1e-6)
                    assert len(set(i.tolist()) & set(wi.tolist())) >= len(wi) - 2  # near-ties may swap at the cut
                    full = ix.score_all(queries[b], mode="fde", allow=al)
                    np.testing.assert_allclose(s, full[i], rtol=1e-4, atol=1e-6)  # every returned id carries ITS score
    # MV_OPT_FDE_BATCH_VARIANT = 3: one page tile per query fragment (the default walks a workgroup's tiles in pairs): same
    # tile -> workgroup map, same order of the K chunks and of the partial sums -> the same scores bit for bit
    kk = min(64, n)
    ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 0)
    paired = ix.query_batch(queries, kk, mode="fde", allows=per_q, n_docs=n_docs)
    ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 3)
    single_tile = ix.query_batch(queries, kk, mode="fde", allows=per_q, n_docs=n_docs)
    for (s0, i0), (s3, i3) in zip(paired, single_tile):
        assert i0.tolist() == i3.tolist() and s0.tolist() == s3.tolist()
    # MV_OPT_FDE_BATCH_VARIANT = 5: the default scan with the round-2 selection -- the finish pass does not pre-bin the scores for the
    # radix selection's first pass (the default does when k > 32 and n > 4096; 3 does not either): the same answers
    ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 5)
    for kind_allows in (per_q, None):
        ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 5)
        three_pass = ix.query_batch(queries, kk, mode="fde", allows=kind_allows, n_docs=n_docs)
        ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 0)
        for rep in range(2):  # twice: the pre-binned histograms must be left clean for the next selection
            fused = ix.query_batch(queries, kk, mode="fde", allows=kind_allows, n_docs=n_docs)
            for (s0, i0), (s5, i5) in zip(fused, three_pass):
                assert i0.tolist() == i5.tolist() and s0.tolist() == s5.tolist()
    # the forms that lost by measurement (4, 6, 7, 8) are gone, not rerouted
    from morphik_core_amd._lib import MvError

    for form in (4, 6, 7, 8):
        with pytest.raises(MvError):
            ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, form)
    # MV_OPT_FDE_BATCH_VARIANT = 2: the query FDE rounded to bf16 (no lo term) -- the slab's own precision
    ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 2)
    kk = min(10, n)
    for (s, i), q in zip(ix.query_batch(queries, kk, mode="fde"), queries):
        full = ix.score_all(q, mode="fde")
        np.testing.assert_allclose(s, full[i], rtol=5e-3, atol=2e-3 * np.abs(full[np.isfinite(full)]).max())
        ws, _ = ix.query(q, kk, mode="fde")
        np.testing.assert_allclose(s, ws, rtol=5e-3, atol=2e-3 * np.abs(ws).max())
    # the query-by-query form of the same entry point (MV_OPT_FDE_BATCH_VARIANT = 1) is the single-query path itself
    ix.set_option(_lib.MV_OPT_FDE_BATCH_VARIANT, 1)
    for (s, i), q in zip(ix.query_batch(queries[:3], min(10, n), mode="fde"), queries[:3]):
        ws, wi = ix.query(q, min(10, n), mode="fde")
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    ix.close()


@pytest.mark.parametrize("slab", ["bf16", "fp8"])
def test_fde_batched_pipeline_equals_query_by_query(mv, slab):
    """FDE coarse -> exact rerank for a batch: same query FDE, same candidate rule (min(10k, 75) or MV_OPT_FDE_COARSE_N),
    same per-batch-of-128 pad rule and the same rerank kernel as the single-query pipeline -> the planted neighbours
    come back with bit-identical exact scores; ragged pages, a tombstone, per-query filters; stage accounting.
    slab = fp8: an index without the bf16 slab reranks on its e4m3 slab (BASELINE configs[3] shape)."""
    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import allow_bitmap

    N, stride, nb = 1500, 64, 35  # 35 queries = a group of 32 + a group of 3
    rng = np.random.default_rng(5)
    queries = [orc.synth_rows(4321, b, 0, 32 if b % 2 else 24) for b in range(nb)]
    base = orc.synth_pages(1234, 0, N, stride)
    spec = synth.planted_spec(queries, N, stride, n_ranks=5)
    for (_, _, p, row0, rows) in spec:
        base[p, row0 : row0 + rows.shape[0]] = rows
    planted_pages = {p for (_, _, p, _, _) in spec}
    lens = [stride if (p in planted_pages or p % 3 == 0) else int(rng.integers(40, stride + 1)) for p in range(N)]
    pages = [base[p, : lens[p]] for p in range(N)]
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_fde=True, with_float=slab == "bf16", with_fp8=slab == "fp8")
    row_bytes = 256 if slab == "bf16" else 128
    ix.add(pages, doc_ordinals=[p // 2 for p in range(N)])
    victim = next(p for p in range(N) if p not in planted_pages and (p ^ 1) not in planted_pages)
    ix.remove_doc(victim // 2)
    planted = {b: [p for (qi, _, p, _, _) in spec if qi == b] for b in range(nb)}
    for coarse_n, k in ((0, 5), (300, 5), (300, 10)):
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, coarse_n)
        got, st = ix.query_batch(queries, k, mode="fde_then_float", want_stats=True)
        for b, (s, i) in enumerate(got):
            ws, wi = ix.query(queries[b], k, mode="fde_then_float")
            assert i[:5].tolist() == planted[b] == wi[:5].tolist()
            assert s[:5].tolist() == ws[:5].tolist()  # same candidates' exact scores, same kernel: bit-identical
            common = set(i.tolist()) & set(wi.tolist())
            assert len(common) >= k - 1  # a candidate at the coarse cut may differ (bf16 hi+lo query FDE)
            ds, dw = dict(zip(i.tolist(), s.tolist())), dict(zip(wi.tolist(), ws.tolist()))
            assert all(ds[c] == dw[c] for c in common)
        nc = coarse_n if coarse_n else min(10 * k, 75)
        groups = 2  # 32 + 3 queries
        assert st.coarse_ms > 0 and st.rerank_ms > 0 and st.encode_ms > 0 and st.select_ms > 0
        live = N - 2
        assert st.pages_scored == live * nb
        cand_rows = st.bytes_scanned - groups * live * ix.fde_config.output_dim * 2
        assert cand_rows % row_bytes == 0 and 40 * nc * nb <= cand_rows // row_bytes <= stride * nc * nb
    # per-query doc filters: query b may not see its own best planted page's document
    allows = [allow_bitmap([d for d in range(N // 2) if d != planted[b][0] // 2]) for b in range(nb)]
    got = ix.query_batch(queries, 5, mode="fde_then_float", allows=allows, n_docs=N // 2)
    for b, (s, i) in enumerate(got):
        ws, wi = ix.query(queries[b], 5, mode="fde_then_float", allow=allows[b])
        assert planted[b][0] not in i.tolist()
        assert i[:4].tolist() == wi[:4].tolist() and s[:4].tolist() == ws[:4].tolist()
    ix.close()


def test_fde_batched_pipeline_fallback_branches(mv):
    """The branches of the batched FDE pipeline that do NOT take the one-launch rerank: queries longer than 128 rows (one
    rerank launch per query, in passes), a non-default float kernel variant, and an fp8-only index with queries longer than
    64 rows (served query by query) -- all must equal the single-query pipeline."""
    from morphik_core_amd import _lib

    N, stride = 900, 48
    for kind, nq_rows in (("bf16_long", 130), ("bf16_variant0", 20), ("fp8_long", 70)):
        ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_fde=True, with_float=kind != "fp8_long", with_fp8=kind == "fp8_long")
        ix.fill_synthetic(1234, 0, N, pages_per_doc=2)
        ix.remove_doc(3)
        if kind == "bf16_variant0":
            ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, 0)
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 200)
        queries = [orc.synth_rows(4321, b, 0, nq_rows - (b % 3)) for b in range(9)]
        got = ix.query_batch(queries, 6, mode="fde_then_float")
        for (s, i), q in zip(got, queries):
            ws, wi = ix.query(q, 6, mode="fde_then_float")
            common = set(i.tolist()) & set(wi.tolist())
            assert len(common) >= 5, kind
            ds, dw = dict(zip(i.tolist(), s.tolist())), dict(zip(wi.tolist(), ws.tolist()))
            assert all(ds[c] == dw[c] for c in common), kind
        ix.close()


@pytest.mark.parametrize("bvariant", [0, 3])  # auto (page-split form up to 128 query rows, row-split above), row-split always
@pytest.mark.parametrize("stride,nrows", [(1024, 1024), (1024, 1000), (208, 200), (64, 50), (16, 7)])
def test_batched_queries_equal_single_queries_and_oracle(mv, stride, nrows, bvariant):
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    n = 700 if stride < 1024 else 300
    ix = _idx(mv, capacity_pages=n, stride_rows=stride)  # the last page ends the slab (DMA pad)
    ix.set_option(_lib.MV_OPT_BATCH_VARIANT, bvariant)
    ix.fill_synthetic(1234, 0, n, n_rows=nrows, pages_per_doc=3)
    ix.remove_doc(5)
    allow = allow_bitmap([d for d in range(n // 3 + 1) if d % 4 != 2])
    pages = orc.bf16_to_f32(ix.read_pages(0, n)[:, :nrows])
    # [32, 32] / [16] * 4: <= 64 rows in all -- the single-row-tile instantiations of the row-split forms
    # [8, 8] / [16] * 3 / 5 / 7: 2, 3, 5, 7 row tiles in the page-split form; [32] * 6 / 10 / 12 / 14: 3, 5, 6, 7 row tiles per wave in the
    # row-split forms -- every instantiation a launcher can pick is run at least once
    for lens in ([32], [8, 8], [16] * 3, [32, 32], [16] * 4, [16] * 5, [32, 32, 32], [16] * 7, [32] * 6, [32] * 10, [32] * 12, [32] * 14, [32] * 16,
                 [32] * 20, [20, 32, 1, 17], [64] * 8, [16] * 32, [48] * 11, [100, 30]):
        qs = [orc.synth_rows(4321, 10 + j, 0, L) for j, L in enumerate(lens)]
        for al in (None, allow):
            got = ix.query_batch(qs, 7, allow=al)
            assert len(got) == len(qs)
            for q, (s, i) in zip(qs, got):
                ws, wi = ix.query(q, 7, allow=al)
                assert i.tolist() == wi.tolist()
                np.testing.assert_allclose(s, ws, rtol=1e-5)
        # against the oracle (no filter): exact float MaxSim per query
        got = ix.query_batch(qs, 5)
        for q, (s, i) in zip(qs[:3], got[:3]):
            want = orc.maxsim_float_np(orc.bf16_to_f32(q), pages)
            want[15:18] = -np.inf  # doc 5 was tombstoned (pages 15..17)
            ws, wi = orc.topk(want, 5)
            _assert_topk_matches(s, i, ws, wi)
    # per-query doc filters in one pass: every request keeps its own doc_ids bitmap (None = unfiltered)
    qs = [orc.synth_rows(4321, 40 + j, 0, 32) for j in range(6)]
    n_docs = n // 3 + 1
    allows = [None, allow_bitmap([1, 2, 3]), allow, allow_bitmap(range(0, n_docs, 2)), None, allow_bitmap([n_docs - 1])]
    got = ix.query_batch(qs, 6, allows=allows, n_docs=n_docs)
    for q, al, (s, i) in zip(qs, allows, got):
        ws, wi = ix.query(q, 6, allow=al)
        assert i.tolist() == wi.tolist()
        np.testing.assert_allclose(s, ws, rtol=1e-5)
    # other modes are served query by query through the same entry point
    ix.close()
    ix = _idx(mv, capacity_pages=64, stride_rows=32, with_binary=True)
    ix.fill_synthetic(1234, 0, 64)
    qs = [orc.synth_rows(4321, j, 0, 32) for j in range(3)]
    got = ix.query_batch(qs, 4, mode="binary")
    for q, (s, i) in zip(qs, got):
        ws, wi = ix.query(q, 4, mode="binary")
        assert i.tolist() == wi.tolist() and s.tolist() == ws.tolist()
    ix.close()


@pytest.mark.parametrize("stride", [64, 80, 208, 1024, 1040])
def test_batched_queries_on_a_uniform_corpus(mv, stride):
    """Uniform corpus (every page full, nothing masked; more pages than persistent workgroups), batches of 5..16 queries:
    strides whose tile count is not a multiple of the 4-tile chunk (80 -> 5 tiles, 1040 -> 65) end a page in a partial
    chunk.  Auto routing (page-split form <= 128 rows, row-split above) and the row-split form on its own (variant 3) must
    equal the single-query scan and the oracle."""
    from morphik_core_amd import _lib

    n = 1500 if stride <= 208 else 700  # more pages than persistent workgroups (512): several pages per workgroup
    ix = _idx(mv, capacity_pages=n, stride_rows=stride)
    ix.fill_synthetic(1234, 0, n)
    pages = orc.bf16_to_f32(ix.read_pages(0, n))
    for lens in ([32] * 5, [32] * 16, [48] * 10, [20, 32, 1, 17, 64, 33]):
        qs = [orc.synth_rows(4321, 70 + j, 0, L) for j, L in enumerate(lens)]
        res = {}
        for bv in (0, 3):
            ix.set_option(_lib.MV_OPT_BATCH_VARIANT, bv)
            res[bv] = ix.query_batch(qs, 9)
        for j, q in enumerate(qs):
            ws, wi = ix.query(q, 9)
            for bv in (0, 3):
                s, i = res[bv][j]
                assert i.tolist() == wi.tolist(), (stride, lens, bv, j)
                np.testing.assert_allclose(s, ws, rtol=1e-5)
        want = orc.maxsim_float_np(orc.bf16_to_f32(qs[0]), pages)
        ws, wi = orc.topk(want, 9)
        _assert_topk_matches(res[0][0][0], res[0][0][1], ws, wi)
    ix.close()


# ------------------------------------------------------------------ fp8 (e4m3) slab
def test_fp8_quantizer_bit_exact_vs_oracle(mv):
    ix = _idx(mv, capacity_pages=40, stride_rows=64, with_fp8=True)
    ix.fill_synthetic(1234, 0, 30, n_rows=50)
    rng = np.random.default_rng(5)
    extra = [rng.standard_normal((n, 128)).astype(np.float32) * s for n, s in ((64, 1.0), (1, 1e-3), (0, 1.0), (33, 40.0), (7, 1e-6))]
    ix.add(extra)
    pages = ix.read_pages(0, 35)
    codes, inv = ix.read_fp8(0, 35)
    lens = [50] * 30 + [64, 1, 0, 33, 7]
    for p in range(35):
        wc, wi = orc.quantize_page_fp8(pages[p, : lens[p]], 64)
        assert np.array_equal(codes[p], wc), p
        assert inv[p] == np.float32(wi), p
    ix.close()


@pytest.mark.parametrize("stride,nrows", [(1024, 1024), (1024, 1000), (64, 50), (48, 48), (16, 5)])
def test_fp8_maxsim_matches_oracle_on_same_codes(mv, stride, nrows):
    n = 24
    ix = _idx(mv, capacity_pages=n, stride_rows=stride, with_float=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, n, n_rows=nrows)
    codes, inv = ix.read_fp8(0, n)
    for nq in (1, 17, 32, 64, 80):
        q = orc.bf16_to_f32(orc.synth_rows(4321, nq, 0, nq))
        q[0, :9] *= 2.5  # not unit norm
        want = orc.maxsim_fp8_np(q, codes, inv, n_rows=[nrows] * n)
        got = ix.score_all(q, mode="float_fp8")
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-6)
        # quality: fp8 scores track the bf16 scores (3 mantissa bits on the page side; averaged over 128 dims)
        ref = ix.score_all(q, mode="float")
        assert np.all(np.abs(got - ref) <= 3e-2 * np.abs(ref) + 5e-3 * nq)
    ix.close()


def test_fp8_full_scan_page_pairs_equal_the_per_page_kernel_on_ragged_masked_odd_corpora(mv):
    """The full e4m3 scan streams two consecutive pages per workgroup (maxsim_fp8_pair_kernel); the candidate path scores
    one page per workgroup (maxsim_fp8_kernel).  Same arithmetic per page: bit-identical scores, whatever the two
    pages of a pair look like -- ragged lengths (incl. empty and one-row pages next to full ones), tombstones, filtered
    documents, an odd page count (the last workgroup has one page), more query rows than one pass holds."""
    from morphik_core_amd.index import allow_bitmap

    stride = 256
    rng = np.random.default_rng(77)
    lens = [256, 1, 0, 256, 33, 200, 97, 256, 256, 5, 128, 129, 31, 32, 64, 250, 3, 256, 160, 161, 0, 0, 255, 17, 224]
    assert len(lens) % 2 == 1
    ix = _idx(mv, capacity_pages=len(lens) + 4, stride_rows=stride, with_float=False, with_fp8=True)
    rows = [rng.standard_normal((n, 128)).astype(np.float32) * (0.5 + p % 3) for p, n in enumerate(lens)]
    ix.add(rows, doc_ordinals=list(range(len(lens))))
    longest_row = max(float(np.linalg.norm(r, axis=1).max()) for r in rows if len(r))
    codes, inv = ix.read_fp8(0, len(lens))
    every = list(range(len(lens)))
    for nq in (1, 32, 64, 80):
        q = rng.standard_normal((nq, 128)).astype(np.float32)
        want = orc.maxsim_fp8_np(q, codes, inv, n_rows=lens)
        got = ix.score_all(q, mode="float_fp8")
        per_page = ix.score_candidates(q, every)
        assert got.tobytes() == per_page.tobytes(), nq
        # against the fp64 oracle on the same codes: fp32 accumulation errs relative to the OPERANDS (sum_q |q| |p|), which a
        # one-row page of random signs does not cancel (its score is ~ 0 beside terms of ~ 200 per token)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=2e-6 * float(np.linalg.norm(q, axis=1).sum()) * longest_row)
    # masked pages inside pairs: a tombstone in the first half of one pair, a filtered document in the second half of another
    ix.remove_doc(7)
    allow = allow_bitmap([d for d in range(len(lens)) if d != 10])
    q = rng.standard_normal((32, 128)).astype(np.float32)
    got = ix.score_all(q, mode="float_fp8", allow=allow)
    ref = ix.score_candidates(q, every)
    for pg in every:
        if pg in (7, 10):
            assert got[pg] == -np.inf, pg
        else:
            assert got[pg].tobytes() == ref[pg].tobytes(), pg
    ix.close()


def test_fp8_only_index_ragged_filter_candidates_and_pipeline(mv):
    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import allow_bitmap

    N, stride, nrows = 3000, 64, 60
    ix = _idx(mv, capacity_pages=N, stride_rows=stride, with_float=False, with_fp8=True, with_fde=True, with_binary=True)
    ix.fill_synthetic(1234, 0, N, n_rows=nrows, pages_per_doc=4)  # staged in chunks: no float slab
    q_bf = orc.synth_rows(4321, 0, 0, 32)
    spec = synth.planted_spec([q_bf], N, nrows, n_ranks=5)
    assert synth.plant_neighbours_any(ix, spec, 1234, nrows) == 5
    planted = [p for (_, _, p, _, _) in spec]
    q = orc.bf16_to_f32(q_bf)
    codes, inv = ix.read_fp8(0, N)
    want = orc.maxsim_fp8_np(q, codes, inv, n_rows=[nrows] * N)
    got = ix.score_all(q_bf, mode="float_fp8")
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-6)
    s, i = ix.query(q_bf, 5, mode="float_fp8")
    assert i.tolist() == planted  # recall@5 = 1.0 against the exact (bf16) planted ranking
    # coarse -> rerank on the fp8 slab
    s2, i2 = ix.query(q_bf, 5, mode="fde_then_float")
    assert i2.tolist() == planted
    np.testing.assert_allclose(s2, s, rtol=1e-5)
    # explicit candidates on the fp8 slab
    cand = [planted[0], 17, planted[3], 2999]
    np.testing.assert_allclose(ix.score_candidates(q_bf, cand), want[cand], rtol=RTOL)
    # doc filter + tombstone
    ix.remove_doc(planted[0] // 4)
    allow = allow_bitmap([d for d in range(N // 4 + 1) if d != planted[1] // 4])
    s3, i3 = ix.query(q_bf, 3, mode="float_fp8", allow=allow)
    assert i3.tolist() == planted[2:5]
    # the binary slab was refreshed by replace_page too
    bits = np.stack([orc.sign_pack(orc.bf16_to_f32(synth_page)) for synth_page in [orc.synth_rows(1234, 5, 0, nrows)]])
    gb = ix.score_all(q_bf, mode="binary")
    assert float(gb[5]) == orc.maxsim_binary(bits[0], orc.sign_pack(q))
    ix.close()


def test_concurrent_threads_on_one_index_get_their_own_answers(mv):
    """The API server handles requests concurrently (to_thread): calls on one mv_index are serialised inside the
    library (one stream + one workspace per index); every caller must still see exactly its own result."""
    import threading

    ix = _idx(mv, capacity_pages=4096, stride_rows=64, with_binary=True)
    ix.fill_synthetic(1234, 0, 4096)
    qs = [orc.synth_rows(4321, j, 0, 32) for j in range(8)]
    want = [(ix.query(q, 5), ix.query(q, 5, mode="binary")) for q in qs]
    errs = []

    def worker(j):
        try:
            for _ in range(20):
                a, b = ix.query(qs[j], 5), ix.query(qs[j], 5, mode="binary")
                assert a[1].tolist() == want[j][0][1].tolist() and a[0].tolist() == want[j][0][0].tolist()
                assert b[1].tolist() == want[j][1][1].tolist() and b[0].tolist() == want[j][1][0].tolist()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(j,)) for j in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:2]
    ix.close()


def test_compaction_moves_every_slab_in_order(mv):
    n = 700
    ix = _idx(mv, capacity_pages=n, stride_rows=48, with_binary=True, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, n, n_rows=40, pages_per_doc=3)
    rng = np.random.default_rng(3)
    dead_docs = sorted(rng.choice(n // 3, size=60, replace=False).tolist())
    for d in dead_docs:
        ix.remove_doc(d)
    ix.remove_page(0)
    ix.remove_page(n - 1)
    q = orc.synth_rows(4321, 5, 0, 32)
    before = {m: ix.score_all(q, mode=m) for m in ("float", "binary", "float_fp8", "fde")}
    pages_before = ix.read_pages(0, n)
    o2n = ix.compact()
    live = np.nonzero(o2n >= 0)[0]
    assert len(ix) == live.size and np.array_equal(o2n[live], np.arange(live.size))
    assert np.all(np.isinf(before["float"][o2n < 0])) and not np.any(np.isinf(before["float"][live]))
    assert np.array_equal(ix.read_pages(0, live.size), pages_before[live])
    for m, sc in before.items():
        got = ix.score_all(q, mode=m)
        assert got.tolist() == sc[live].tolist(), m  # same bytes in a dense prefix -> bit-identical scores
    s, i = ix.query(q, 10)
    ws, wi = orc.topk(before["float"], 10)
    assert o2n[wi].tolist() == i.tolist() and s.tolist() == ws.tolist()
    first = ix.add([orc.bf16_to_f32(orc.synth_rows(9, 1, 0, 40))], [5])  # reclaimed capacity is usable
    assert first == live.size and ix.compact().size == live.size + 1  # nothing left to reclaim
    ix.close()


def test_ingest_runs_beside_queries_append_only_publish(mv):
    """SURVEY.md 8b / VERDICT r1 item 9: mv_index_add fills slab slots beyond the published size on the writer stream and
    publishes them with one atomic store -- it never takes the query lock.  Eight threads query while a ninth ingests
    30 batches: every answer equals the pre-ingest answer (the added pages cannot reach the planted top-10), queries
    never see a half-written page, and no query waits for an ingest call."""
    import threading
    import time

    from morphik_core_amd import synth

    base, extra, stride, nq = 20_000, 30 * 500, 64, 8
    ix = _idx(mv, capacity_pages=base + extra, stride_rows=stride, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, base)
    qs = [orc.synth_rows(4321, j, 0, 16) for j in range(nq)]
    spec = synth.planted_spec(qs, base, stride, n_ranks=10)
    synth.plant_neighbours_any(ix, spec, 1234, stride)
    want = [ix.query(q, 10) for q in qs]
    for qi, (s, i) in enumerate(want):
        assert sorted(i.tolist()) == sorted(p for (qq, _r, p, _a, _b) in spec if qq == qi)
    idle = []
    for _ in range(20):
        t0 = time.perf_counter()
        ix.query(qs[0], 10)
        idle.append(time.perf_counter() - t0)
    idle_med = float(np.median(idle))
    stop = threading.Event()
    errors, lat, seen_sizes = [], [], []

    def reader(j):
        try:
            while not stop.is_set():
                t0 = time.perf_counter()
                s, i = ix.query(qs[j], 10)
                lat.append(time.perf_counter() - t0)
                if i.tolist() != want[j][1].tolist() or s.tolist() != want[j][0].tolist():
                    errors.append((j, i.tolist()))
                seen_sizes.append(len(ix))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader, args=(j,)) for j in range(nq)]
    for t in threads:
        t.start()
    rng = np.random.default_rng(5)
    add_s = []
    for b in range(30):
        pages = [orc.synth_rows(777, base + b * 500 + p, 0, int(rng.integers(8, stride + 1))) for p in range(0, 500, 50)] * 50
        t0 = time.perf_counter()
        first = ix.add(pages, doc_ordinals=[base + b] * len(pages))
        add_s.append(time.perf_counter() - t0)
        assert first == base + b * 500
    stop.set()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert len(ix) == base + extra
    assert len(lat) > 50 and len(set(seen_sizes)) > 3  # queries really ran while the corpus grew
    # a query shares the GPU with the ingest kernels and the lock with 7 other readers, but never waits for an add call:
    # generous bound (8 readers take turns -> ~8x idle latency is expected)
    assert max(lat) < 40 * idle_med + 0.25, (max(lat), idle_med, max(add_s))
    # the appended pages are all there and scored like any other
    probe = pages[7]
    s, i = ix.query(probe, 3)
    assert int(i[0]) >= base and ix.page_rows([int(i[0])])[0] == probe.shape[0]
    ix.close()


# ------------------------------------------------------------------ persistence
def test_save_load_roundtrip(mv, tmp_path):
    from morphik_core_amd.index import MvIndex

    ix = _idx(mv, capacity_pages=64, stride_rows=32, with_binary=True, with_fde=True, with_fp8=True)
    ix.fill_synthetic(1234, 0, 50, pages_per_doc=5)
    ix.remove_doc(2)
    q = orc.synth_rows(4321, 0, 0, 32)
    before = {m: ix.query(q, 7, mode=m) for m in ("float", "binary", "fde_then_float", "float_fp8")}
    path = str(tmp_path / "ix.mv")
    ix.save(path)
    ix.close()
    ix2 = MvIndex.load(path)
    assert len(ix2) == 50
    for m, (s, i) in before.items():
        s2, i2 = ix2.query(q, 7, mode=m)
        assert i2.tolist() == i.tolist() and s2.tolist() == s.tolist()
    ix2.close()


def test_errors_are_loud(mv):
    from morphik_core_amd import MvError

    with pytest.raises(MvError):
        _idx(mv, capacity_pages=4, stride_rows=20)  # not a multiple of 16
    ix = _idx(mv, capacity_pages=2, stride_rows=16)
    with pytest.raises(MvError):
        ix.add([np.zeros((17, 128), np.float32)])  # longer than the stride
    ix.add([np.ones((3, 128), np.float32)] * 2)
    with pytest.raises(MvError):
        ix.add([np.ones((3, 128), np.float32)])  # slab full
    with pytest.raises(MvError):
        ix.query(np.ones((2, 128), np.float32), 1, mode="binary")  # slab not enabled
    ix.close()


@pytest.mark.parametrize("stride,nrows", [(1024, 1000), (208, 208), (64, 50), (16, 7)])  # ragged 1024-row pages, UNIFORM 208-row pages (no row masks), short ragged pages
def test_batched_fp8_scan_equals_single_query_scan_and_oracle(mv, stride, nrows):
    """mv_query_topk_batch, MV_MODE_FLOAT_FP8: one pass over the e4m3 slab for a group of queries (maxsim_batch_fp8_kernel,
    block-scaled MFMA with the page tile as the A operand) returns the single-query fp8 scan's answers, which are the
    oracle's on the device's own codes; per-request doc filters, tombstones, ragged pages, ragged query lengths.  Variant 7
    (one e4m3 term per query row) stays within e4m3 rounding of the two-term scores; variant 8 is the query-by-query form."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    n = 700 if stride < 1024 else 300
    ix = _idx(mv, capacity_pages=n, stride_rows=stride, with_float=False, with_fp8=True)
    ix.fill_synthetic(1234, 0, n, n_rows=nrows, pages_per_doc=3)
    ix.remove_doc(5)
    allow = allow_bitmap([d for d in range(n // 3 + 1) if d % 4 != 2])
    codes, inv = ix.read_fp8(0, n)
    # [32, 32] / [16] * 4 / [20, 9]: <= 64 query rows in all = ONE row tile per wave (its own instantiation of the kernel; two
    # requests over a full shard is how the store's coalescer meets it)
    # [32] * 6 / 8 / 10 / 12 / 14: 3 .. 7 row tiles per wave -- every instantiation of the kernel the launcher can pick is run
    for lens in ([32], [32, 32], [16] * 4, [20, 9], [32, 32, 32], [32] * 6, [32] * 8, [32] * 10, [32] * 12, [32] * 14, [32] * 16, [32] * 20,
                 [20, 32, 1, 17], [64] * 8, [16] * 32, [48] * 10):
        qs = [orc.synth_rows(4321, 10 + j, 0, L) for j, L in enumerate(lens)]
        for al in (None, allow):
            got = ix.query_batch(qs, 7, mode="float_fp8", allow=al)
            assert len(got) == len(qs)
            for q, (s, i) in zip(qs, got):
                ws, wi = ix.query(q, 7, mode="float_fp8", allow=al)
                assert i.tolist() == wi.tolist()
                np.testing.assert_allclose(s, ws, rtol=1e-5)
        got = ix.query_batch(qs, 5, mode="float_fp8")
        for q, (s, i) in zip(qs[:3], got[:3]):
            want = orc.maxsim_fp8_np(orc.bf16_to_f32(q), codes, inv, n_rows=[nrows] * n)
            want[15:18] = -np.inf  # doc 5 was tombstoned (pages 15..17)
            ws, wi = orc.topk(want, 5)
            _assert_topk_matches(s, i, ws, wi)
    qs = [orc.synth_rows(4321, 40 + j, 0, 32) for j in range(6)]
    n_docs = n // 3 + 1
    allows = [None, allow_bitmap([1, 2, 3]), allow, allow_bitmap(range(0, n_docs, 2)), None, allow_bitmap([n_docs - 1])]
    got = ix.query_batch(qs, 6, mode="float_fp8", allows=allows, n_docs=n_docs)
    for q, al, (s, i) in zip(qs, allows, got):
        ws, wi = ix.query(q, 6, mode="float_fp8", allow=al)
        assert i.tolist() == wi.tolist()
        np.testing.assert_allclose(s, ws, rtol=1e-5)
    two = ix.query_batch(qs, 6, mode="float_fp8")
    ix.set_option(_lib.MV_OPT_BATCH_VARIANT, 8)
    for (s, i), (ws, wi) in zip(ix.query_batch(qs, 6, mode="float_fp8"), two):
        assert i.tolist() == wi.tolist()
    ix.set_option(_lib.MV_OPT_BATCH_VARIANT, 7)  # single e4m3 term on the query side
    one = ix.query_batch(qs, 6, mode="float_fp8")
    for (s, i), (ws, wi) in zip(one, two):
        np.testing.assert_allclose(np.sort(s)[::-1], np.sort(ws)[::-1], rtol=3e-2)
    ix.close()


# ------------------------------------------------------------------ fp8 scan -> exact bf16 re-score from the exact tier
@pytest.mark.parametrize("tier", ["host", "hbm", "both"])
def test_fp8_then_float_rescoring_from_the_exact_tier(mv, tier, tmp_path):
    """MV_MODE_FP8_THEN_FLOAT: e4m3 scan -> top-n -> exact bf16 MaxSim of the n candidates read from the exact tier (pinned
    host memory mapped into the device, or the bf16 slab) -> top-k.  Checked against the oracle composed the same way on
    the device's own codes and rows: candidate set = fp8 top-n, final scores = the exact bf16 scores (1e-3 as north_star
    states; measured ~1e-7), final order = exact order.  Ragged pages, a doc filter, a tombstone, an upsert, compaction
    and a checkpoint round trip all keep the two tiers in step."""
    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import MvIndex, allow_bitmap

    N, stride, nrows, NR = 1200, 64, 60, 48
    ix = _idx(mv, capacity_pages=N + 8, stride_rows=stride, with_float=tier != "host", with_fp8=True, with_host_exact=tier != "hbm")
    if tier == "both":  # an index holding both exact tiers: the rerank is told to read the pinned-host one
        ix.set_option(_lib.MV_OPT_EXACT_TIER, 1)
    ix.fill_synthetic(1234, 0, N - 40, n_rows=nrows, pages_per_doc=4)
    rag = [orc.synth_rows(77, j, 0, 10 + j) for j in range(40)]  # ragged pages through the host add path
    ix.add(rag, doc_ordinals=[1000 + j for j in range(40)])
    ix.set_option(_lib.MV_OPT_RERANK_N, NR)
    q_bf = orc.synth_rows(4321, 0, 0, 32)
    q = orc.bf16_to_f32(q_bf)
    spec = synth.hard_spec([q_bf], N - 40, nrows, n_hard=24)
    synth.plant_neighbours_any(ix, spec, 1234, nrows)  # replace_page refreshes both tiers
    lens = [nrows] * (N - 40) + [10 + j for j in range(40)]

    def oracle_two_tier(k, allowed=None):
        pages = ix.read_pages(0, len(ix))  # exact tier rows (host tier when there is no bf16 slab)
        codes, inv = ix.read_fp8(0, len(ix))
        c8 = orc.maxsim_fp8_np(q, codes, inv, n_rows=lens[: len(ix)])
        if allowed is not None:
            c8 = np.where(allowed, c8, -np.inf)
        cs, ci = orc.topk(c8, NR)
        ex = np.array([orc.maxsim_bf16(q_bf, pages[c][: lens[c]]) for c in ci], np.float32)
        order = np.lexsort((ci, -ex.astype(np.float64)))[:k]
        return ex[order], ci[order], set(ci.tolist()), c8

    ws, wi, cand, c8 = oracle_two_tier(10)
    s, i, st = ix.query(q_bf, 10, mode="fp8_then_float", want_stats=True)
    _assert_topk_matches(s, i, ws, wi)
    assert st.rerank_ms > 0 and st.coarse_ms > 0
    # the hard set's exact top-10 comes back in exact order although the fp8 scan alone reorders it
    hard_pages = synth.hard_pages_of(spec, 0)
    pages_all = ix.read_pages(0, len(ix))
    exact = np.array([orc.maxsim_bf16(q_bf, pages_all[p][:nrows]) for p in hard_pages], np.float32)
    top, _info = synth.exact_truth_from_scores(hard_pages, exact)
    assert i.tolist() == top
    np.testing.assert_allclose(ix.score_all(q_bf, mode="fp8_then_float"), c8, rtol=RTOL, atol=1e-6)  # first-stage scores
    # a BATCH of requests: one pass of the batched fp8 scan, every request's top-n re-scored exactly in one launch -- the lone
    # calls' answers; with one e4m3 term per query row in the first stage (variant 7) the exact tier still restores them
    bqs = [q_bf] + [orc.synth_rows(4321, 50 + j, 0, 20 + j) for j in range(5)]
    lone = [ix.query(x, 10, mode="fp8_then_float") for x in bqs]
    for variant in (0, 7):
        ix.set_option(_lib.MV_OPT_BATCH_VARIANT, variant)
        for (bs, bi), (ls, li) in zip(ix.query_batch(bqs, 10, mode="fp8_then_float"), lone):
            if variant == 0:
                assert bi.tolist() == li.tolist()
                np.testing.assert_allclose(bs, ls, rtol=1e-5)
            else:
                assert bi.tolist()[:5] == li.tolist()[:5]  # the clear neighbours; the tail may differ where the fp8 cut differs
        assert ix.query_batch(bqs, 10, mode="fp8_then_float")[0][1].tolist() == top
    ix.set_option(_lib.MV_OPT_BATCH_VARIANT, -1)
    # doc filter (selective -> compacted candidate list) + tombstone
    ix.remove_doc(top[0] // 4)
    allowed_docs = [d for d in range((N - 40) // 4 + 1) if d % 3 != 1] + [1000 + j for j in range(0, 40, 2)]
    al = allow_bitmap(allowed_docs)
    docs = np.array([p // 4 for p in range(N - 40)] + [1000 + j for j in range(40)])
    mask = np.isin(docs, allowed_docs) & (docs != top[0] // 4)
    ws2, wi2, _c, _ = oracle_two_tier(7, mask)
    s2, i2 = ix.query(q_bf, 7, mode="fp8_then_float", allow=al)
    _assert_topk_matches(s2, i2, ws2, wi2)
    few = allow_bitmap([top[3] // 4, top[5] // 4, 1003])
    s3, i3 = ix.query(q_bf, 50, mode="fp8_then_float", allow=few)
    assert set(i3.tolist()) == {p for p in range(N) if docs[p] in (top[3] // 4, top[5] // 4, 1003) and docs[p] != top[0] // 4}
    # compaction moves both tiers; a checkpoint carries the exact tier
    o2n = ix.compact()
    lens = [lens[p] for p in range(N) if o2n[p] >= 0]
    ws4, wi4, _c, _ = oracle_two_tier(10)
    s4, i4 = ix.query(q_bf, 10, mode="fp8_then_float")
    _assert_topk_matches(s4, i4, ws4, wi4)
    path = str(tmp_path / "two_tier.mv")
    ix.save(path)
    ix2 = MvIndex.load(path)
    s5, i5 = ix2.query(q_bf, 10, mode="fp8_then_float")
    assert i5.tolist() == i4.tolist() and np.array_equal(s5, s4)
    assert np.array_equal(ix2.read_pages(0, len(ix2)), ix.read_pages(0, len(ix)))
    ix2.close()
    ix.close()
    if tier == "host":  # the mode refuses an index without an exact tier, loudly
        from morphik_core_amd import MvError

        ix3 = _idx(mv, capacity_pages=8, stride_rows=16, with_float=False, with_fp8=True)
        ix3.fill_synthetic(1, 0, 8)
        with pytest.raises(MvError):
            ix3.query(q_bf, 3, mode="fp8_then_float")
        ix3.close()


# ------------------------------------------------------------------ recall of the lossy paths on hard negatives (configs[3], [4])
# recall@10 floors of test_recall_of_lossy_paths...: measured on the box (profiles/r3/pytest_gpu_full_r3.log prints the values), minus <= 0.05
RECALL_FLOORS = {
    "hard": {"float_fp8": 0.80, "fp8_then_float": 0.995, "binary": 0.25, "fde75": 0.90, "fde1000": 0.98, "coarse75": 0.93, "coarse1000": 0.99},
    "clustered": {"float_fp8": 0.93, "fp8_then_float": 0.995, "binary": 0.55, "fde75": 0.93, "fde1000": 0.95, "coarse75": 0.95, "coarse1000": 0.97},
    "unstructured": {"float_fp8": 0.70, "fp8_then_float": 0.995},
}


def test_recall_of_lossy_paths_on_hard_negatives_and_unplanted_corpus(mv):
    """VERDICT r1 item 1 / r2 item 2.  A 40 k-page corpus with every slab and three query sets: 32 hard-negative queries (64
    pages each whose exact bf16 scores sit within ~2 % of each other; rank 10 and rank 11 differ by a few 1e-4 relative),
    32 clustered-topic queries (graded relevance, synth.clustered_spec) and 8 queries with no structure.  The exact float
    scan must return the exact top-10 (truth for the hard set = the ORACLE's scores on the device's own bytes); every lossy
    path is held to a floor within 0.05 of what it measures here: fp8 keeps most of the top-10 (its 0.3-1 % per-score
    deviation reorders near-ties) and ALL of it once its top 128 are re-scored exactly (fp8_then_float), the FDE coarse
    stage keeps the whole top-10 inside its top-1000 (and nearly all of it inside the reference's 75 candidates), sign
    bits do not resolve margins this small.  Without planted structure the FDE stage has nothing to find: only fp8 is bounded."""
    from morphik_core_amd import _lib, synth
    from morphik_core_amd.index import synth_rows

    n, rows, nq = 40_000, 256, 32
    ix = _idx(mv, capacity_pages=n, stride_rows=rows, with_float=True, with_binary=True, with_fde=True, with_fp8=True)
    ix.fill_synthetic(synth.SEED_CORPUS, 0, n)
    qs = [synth_rows(synth.SEED_QUERIES, 100 + j, 32) for j in range(nq)]
    spec = synth.hard_spec(qs, n, rows)
    cqs, cspec = synth.clustered_spec(nq, n, rows, exclude={t[2] for t in spec})
    synth.plant_neighbours_any(ix, spec + cspec, synth.SEED_CORPUS, rows)
    truths, gaps, near = [], [], []
    for j, q in enumerate(qs):
        pages = synth.hard_pages_of(spec, j)
        got = ix.score_candidates(q, pages, pad_to=0)
        if j < 6:
            want = np.array([orc.maxsim_bf16(q, ix.read_pages(p, 1)[0]) for p in pages], np.float32)  # oracle on the device's own bytes
            np.testing.assert_allclose(got, want, rtol=1e-5)
        top, info = synth.exact_truth_from_scores(pages, got)
        truths.append(top)
        gaps.append(info["gap_10_11"])
        near.append(info["within_2pct"])
        s, i = ix.query(q, 10, mode="float")
        assert sorted(i.tolist()) == sorted(top)  # the exact scan finds the exact top-10 (order may swap inside ~1e-6 ties)
    assert np.median(gaps) < 5e-3 and min(near) >= 50  # the corpus is as hard as VERDICT asked for
    ctruth, cgaps = [], []
    for q in cqs:  # graded relevance: truth = the exact scan (11 deep, for the margin)
        s, i = ix.query(q, 11, mode="float")
        ctruth.append(i[:10].tolist())
        cgaps.append(float((s[9] - s[10]) / abs(s[9])))
    rq = [synth_rows(synth.SEED_QUERIES, 300 + j, 32) for j in range(8)]
    rt = [ix.query(q, 10, mode="float")[1].tolist() for q in rq]

    def recall(mode, truth, queries, k=10):
        return float(np.mean([synth.recall_at_k(ix.query(q, k, mode=mode)[1].tolist(), t) for q, t in zip(queries, truth)]))

    got = {}
    for name, (queries, truth) in {"hard": (qs, truths), "clustered": (cqs, ctruth), "unstructured": (rq, rt)}.items():
        r = {"float_fp8": recall("float_fp8", truth, queries), "fp8_then_float": recall("fp8_then_float", truth, queries),
             "binary": recall("binary", truth, queries)}
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 75)
        r["fde75"] = recall("fde_then_float", truth, queries)
        r["coarse75"] = recall("fde", truth, queries, k=75)
        ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 1000)
        r["fde1000"] = recall("fde_then_float", truth, queries)
        r["coarse1000"] = recall("fde", truth, queries, k=1000)
        got[name] = r
        print(f"recall@10 {name} ({len(queries)} queries, median rank-10/11 gap "
              f"{np.median(gaps if name == 'hard' else cgaps if name == 'clustered' else [0]):.2e}): "
              + " ".join(f"{k}={v:.3f}" for k, v in r.items()))
    for name, floors in RECALL_FLOORS.items():
        for k, floor in floors.items():
            assert got[name][k] >= floor, (name, k, got[name][k], floor)
    ix.close()


# ------------------------------------------------------------------ advisor findings of round 1 (checkpoint validation, empty filters)
def test_checkpoint_is_replaced_atomically_and_a_corrupt_header_is_refused(mv, tmp_path):
    """mv_index_save writes <path>.tmp, fsyncs and renames (a crash mid-save keeps the previous checkpoint);
    mv_index_load treats the header as untrusted input: size beyond capacity, negative size, FDE width that does not match
    the config, row counts beyond the stride are refused loudly instead of overrunning the slabs."""
    import struct

    from morphik_core_amd import MvError
    from morphik_core_amd._lib import ConfigC
    from morphik_core_amd.index import MvIndex
    import ctypes as C

    ix = _idx(mv, capacity_pages=32, stride_rows=32, with_fde=True)
    ix.fill_synthetic(1234, 0, 20)
    path = str(tmp_path / "ix.mv")
    ix.save(path)
    first = open(path, "rb").read()
    ix.fill_synthetic(1234, 20, 5)
    ix.save(path)  # replaces the file in one rename; no temp file is left behind
    assert not os.path.exists(path + ".tmp") and len(open(path, "rb").read()) > len(first)
    q = orc.synth_rows(4321, 0, 0, 16)
    want = ix.query(q, 5)
    ix.close()
    back = MvIndex.load(path)
    got = back.query(q, 5)
    assert len(back) == 25 and got[1].tolist() == want[1].tolist() and got[0].tolist() == want[0].tolist()
    back.close()
    good = bytearray(open(path, "rb").read())
    off_size = 8 + C.sizeof(ConfigC)

    def corrupt(mutate, name):
        b = bytearray(good)
        mutate(b)
        p = str(tmp_path / name)
        open(p, "wb").write(b)
        with pytest.raises(MvError):
            MvIndex.load(p)

    corrupt(lambda b: b.__setitem__(slice(off_size, off_size + 8), struct.pack("<q", 33)), "size_over_capacity.mv")
    corrupt(lambda b: b.__setitem__(slice(off_size, off_size + 8), struct.pack("<q", -1)), "negative_size.mv")
    corrupt(lambda b: b.__setitem__(slice(off_size + 8, off_size + 16), struct.pack("<q", 12345)), "fde_width.mv")
    corrupt(lambda b: b.__setitem__(slice(off_size + 16, off_size + 20), struct.pack("<i", 33)), "row_count_over_stride.mv")
    corrupt(lambda b: b.__setitem__(slice(0, 8), b"NOTANIDX"), "magic.mv")
    with pytest.raises(MvError):
        MvIndex.load(str(tmp_path / "missing.mv"))


def test_filter_that_names_only_deleted_documents_and_long_queries(mv):
    """A compacted doc filter with ZERO live pages (the filter names only tombstoned documents) under a query of more than
    128 rows (several scan passes + the score accumulation launch): no launch with an empty grid, an empty answer."""
    from morphik_core_amd.index import allow_bitmap

    ix = _idx(mv, capacity_pages=400, stride_rows=32, with_fp8=True, with_binary=True)
    ix.fill_synthetic(1234, 0, 400, pages_per_doc=2)
    for d in (7, 8):
        ix.remove_doc(d)
    allow = allow_bitmap([7, 8], 200)
    for nq in (20, 150, 300):
        q = orc.synth_rows(4321, 900 + nq, 0, nq)
        for mode in ("float", "float_fp8", "binary"):
            s, i = ix.query(q, 10, mode=mode, allow=allow)
            assert len(i) == 0, (mode, nq)
        s, i = ix.query(q, 10, allow=allow_bitmap([7, 9], 200))
        assert sorted(i.tolist()) == [18, 19]
    ix.close()


def test_score_multi_vector_function_equals_the_reference_outputs(mv, golden_dir):
    """morphik_core_amd.scoring.score_multi_vector(qs, ps) -- the call FastMultiVectorStore makes (:553-555) -- against the
    committed outputs of the reference formula (tests/golden/maxsim_float.npz, transformers' score_retrieval), including
    the 133-passage case that crosses the batch-of-128 boundary, and its argument errors."""
    from morphik_core_amd.scoring import score_multi_vector

    g = np.load(os.path.join(golden_dir, "maxsim_float.npz"))
    for ci in range(int(g["n_cases"])):
        q, slab, n_rows, want = (g[f"{k}{ci}"] for k in ("q", "slab", "n_rows", "scores"))
        ps = [slab[i, : n_rows[i]] for i in range(slab.shape[0])]
        got = score_multi_vector([q, q[: max(1, q.shape[0] // 2)]], ps)
        assert got.shape == (2, len(ps)) and got.dtype == np.float32
        np.testing.assert_allclose(got[0], want, rtol=RTOL, atol=1e-5)  # fp32 passages: kept as split-bf16 pairs (scoring.py)
        assert np.abs(got[0] - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    with pytest.raises(ValueError):
        score_multi_vector([], [np.ones((2, 128), np.float32)])
    with pytest.raises(ValueError):
        score_multi_vector([np.ones((2, 128), np.float32)], [])


# ------------------------------------------------------------------ the binding INTEGRATION.md documents, executed
def test_integration_md_binding_create_add_query_matches_the_oracle(mv):
    """The ctypes stub of INTEGRATION.md section 2, extracted and run as documented: create / add / query through the raw C ABI
    against the oracle -- exact float MaxSim (1e-3), the sign-bit max_sim (exact), the FDE pipeline (same top pages)."""
    from tests.test_abi_exports import _integration_md_binding

    ns, _code = _integration_md_binding()
    rng = np.random.default_rng(7)
    pages = []
    for i in range(60):
        x = rng.standard_normal((20 + i % 13, 128)).astype(np.float32)
        pages.append(x / np.linalg.norm(x, axis=1, keepdims=True))
    q = pages[17][:12] + 0.05 * rng.standard_normal((12, 128)).astype(np.float32)
    h = ns["create"](64, stride_rows=48)
    assert ns["add"](h, pages, list(range(60))) == 0
    qb, pb = orc.f32_to_bf16(q), [orc.f32_to_bf16(p) for p in pages]
    want = np.array([orc.maxsim_bf16(qb, p) for p in pb], np.float32)
    ws, wi = orc.topk(want, 5)
    s, i = ns["query"](h, q, 5, 0)  # MV_MODE_FLOAT
    _assert_topk_matches(s, i, ws, wi)
    wantb = np.array([orc.maxsim_binary(orc.sign_pack(p), orc.sign_pack(q)) for p in pages], np.float32)
    bs, bi = ns["query"](h, q, 5, 1)  # MV_MODE_BINARY: SQL max_sim, exact
    wbs, wbi = orc.topk(wantb, 5)
    assert bs.tolist() == wbs.tolist() and bi.tolist() == wbi.tolist()
    fs, fi = ns["query"](h, q, 3, 2)  # MV_MODE_FDE_THEN_FLOAT: 30 coarse candidates of 60 pages, exact rerank
    assert fi[0] == 17 and abs(fs[0] - want[17]) <= RTOL * abs(want[17])
    allow = np.array([0xFFFDFFFF, 0xFFFFFFFF], np.uint32)  # document 17 filtered out
    s2, i2 = ns["query"](h, q, 5, 0, allow)
    assert 17 not in i2.tolist()
    ns["L"].mv_index_destroy.argtypes = [ns["C"].c_void_p]
    ns["L"].mv_index_destroy(h)


# ------------------------------------------------------------------ NaN / Inf / -0.0
def test_non_finite_inputs_are_refused_by_the_float_paths_and_defined_on_the_sign_bit_path(mv):
    """The reference's quantiser defines NaN / +-0 -> bit 0 (binary_ops.rs:81-136, fast_ops.py:191-227); its float path does not
    define them at all (einsum -> max -> topk ranks NaN first).  Here: float-derived slabs and float modes REFUSE non-finite rows,
    loudly and atomically; a sign-bit-only index and MV_MODE_BINARY take them by the quantiser's rule; -0.0 is an ordinary value."""
    from morphik_core_amd import MvError

    stride = 32
    rng = np.random.default_rng(11)

    def page(n=20):
        x = rng.standard_normal((n, 128)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True)

    good = [page() for _ in range(6)]
    ix = _idx(mv, capacity_pages=32, stride_rows=stride, with_float=True, with_binary=True, with_fde=True, with_fp8=True)
    ix.add(good)
    for bad_value in (np.nan, np.inf, -np.inf, 3.4e38):  # 3.4e38 is finite in fp32 and an Inf once rounded to bf16
        bad = page()
        bad[7, 33] = bad_value
        with pytest.raises(MvError, match="NaN / Inf"):
            ix.add([page(), bad, page()])  # all or nothing: the two good pages of the call are not added either
        assert len(ix) == 6
        with pytest.raises(MvError, match="NaN / Inf"):
            ix.add([orc.f32_to_bf16(bad)])  # the same rows handed over as bf16
        assert len(ix) == 6
        with pytest.raises(MvError, match="NaN / Inf"):
            ix.replace_page(2, orc.f32_to_bf16(bad))
    q = page(12)
    before = {m: ix.query(q, 4, mode=m) for m in ("float", "float_fp8", "binary", "fde_then_float")}
    neg0 = page()
    neg0[3] = -0.0  # a whole row of -0.0: an ordinary value
    pos0 = neg0.copy()
    pos0[3] = 0.0
    ix.add([neg0, pos0])
    assert len(ix) == 8
    for m in ("float", "float_fp8", "binary"):
        sc = ix.score_all(q, mode=m)
        assert sc[6] == sc[7] and np.isfinite(sc).all(), m
        s, i = ix.query(q, 4, mode=m)
        if 6 not in i.tolist() and 7 not in i.tolist():
            assert i.tolist() == before[m][1].tolist()  # the slots the refused adds touched were reused cleanly
    qn = q.copy()
    qn[5, 100] = np.nan
    qi = q.copy()
    qi[0, 0] = -np.inf
    for bad_q in (qn, qi):
        for m in ("float", "float_fp8", "fde_then_float", "fde"):
            with pytest.raises(MvError, match="NaN / Inf"):
                ix.query(bad_q, 3, mode=m)
            with pytest.raises(MvError, match="NaN / Inf"):
                ix.query_batch([q, bad_q], 3, mode=m)
        with pytest.raises(MvError):
            ix.score_candidates(bad_q, [0, 1])
        # the sign-bit mode defines them: bit = v > 0
        want = np.array([orc.maxsim_binary(orc.sign_pack(p), orc.sign_pack(bad_q)) for p in good + [neg0, pos0]], np.float32)
        assert ix.score_all(bad_q, mode="binary").tolist() == want.tolist()
    ix.close()
    # a sign-bit-only index takes non-finite PAGES too, by the same rule (golden fixtures pin the rule for mv_sign_pack)
    bx = _idx(mv, capacity_pages=8, stride_rows=stride, with_float=False, with_binary=True)
    weird = page()
    weird[0, :4] = [np.nan, np.inf, -np.inf, -0.0]
    bx.add([good[0], weird])
    want = np.array([orc.maxsim_binary(orc.sign_pack(p), orc.sign_pack(q)) for p in (good[0], weird)], np.float32)
    assert bx.score_all(q, mode="binary").tolist() == want.tolist()
    bits = orc.sign_pack(weird)
    assert (bits[0, 0] >> 4) == 0b0100  # NaN -> 0, +Inf -> 1, -Inf -> 0, -0.0 -> 0 (MSB first)
    bx.close()
