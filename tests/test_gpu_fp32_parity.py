"""GPU parity of the split-bf16 paths against the fp32 oracle (orc_maxsim_f32: the reference's fp32 einsum -> max -> sum,
fast_multivector_store.py:553-555 on fp32 pages, :736 / :774) on inputs that are NOT bf16-representable.

The reference keeps pages as fp32 `.npy` and scores in fp32.  This engine stores x = hi + lo (two bf16 slabs, the same 4 bytes
per element; MV_WITH_FLOAT_LO) and splits an fp32 query the same way; the kernels accumulate qhi.phi + qlo.phi + qhi.plo in fp32
before the max.  Bar: north_star's 1e-3 relative, asserted at 1e-3 and additionally at 5e-5 of the score scale (observed ~1e-6).
bf16-representable inputs must give the bits of the one-term kernels.

Run on the MI355X box:  python -m pytest tests/test_gpu_fp32_parity.py -m gpu -x -q
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402  (checker only)

RTOL = 1e-3


def _unit(rng, n, d=128):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def _idx(**kw):
    from morphik_core_amd.index import MvIndex

    return MvIndex(**kw)


def _want(q, pages, pad_to=0):
    return np.array([orc.maxsim_f32(q, p, pad_to) for p in pages], np.float32)


def _close(got, want, tight=5e-5):
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-5)
    assert np.abs(got - want).max() <= tight * max(1.0, float(np.abs(want).max())), float(np.abs(got - want).max())


@pytest.mark.parametrize("variant", [0, 6, 7])
def test_fp32_corpus_baseline_config0_shape(variant):
    """BASELINE configs[0]: 100 pages x 32 patches x 128-d random float32 (unit rows), fp32 query of 32 rows -- every entry
    point that scores floats, on the direct-load cross-check kernel and both ring kernels, against orc_maxsim_f32."""
    from morphik_core_amd import _lib

    rng = np.random.default_rng(20260930)
    pages = [_unit(rng, 32) for _ in range(100)]
    q = _unit(rng, 32)
    want = _want(q, pages)
    ix = _idx(capacity_pages=128, stride_rows=32, with_float_lo=True)
    ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
    ix.add(pages)
    _close(ix.score_all(q), want)  # full scan, hi + lo
    _close(ix.score_candidates(q, np.arange(100)), want)
    ws, wi = orc.topk(want, 10)
    for lo_scan in (1, 2):  # both halves of every page / hi-only scan + split-bf16 re-score of the best
        ix.set_option(_lib.MV_OPT_FLOAT_LO_SCAN, lo_scan)
        s, i = ix.query(q, 10)
        assert i.tolist() == wi.tolist()
        _close(s, ws)
    # the pages read back as fp32 are the input to 2^-17 (hi + lo keeps 16 significant bits)
    back = ix.read_pages_f32(0, 100)
    src = np.stack(pages)
    assert np.abs(back - src).max() <= 2.0 ** -17 * np.abs(src).max()
    # hi-only scan (MV_OPT_FLOAT_LO_SCAN 0): the scores of the bf16-rounded pages under the exact fp32 query
    ix.set_option(_lib.MV_OPT_FLOAT_LO_SCAN, 0)
    hi_pages = [orc.bf16_to_f32(orc.f32_to_bf16(p)) for p in pages]
    _close(ix.score_all(q), _want(q, hi_pages))
    ix.close()


def test_fp32_query_on_a_plain_bf16_index_is_not_rounded():
    """No lo slab: pages are bf16 (the north_star domain), but an fp32 QUERY is still scored exactly (hi + lo chains): the
    result is the fp32 product of the unrounded query with the stored bf16 pages."""
    rng = np.random.default_rng(7)
    pages_bf16 = [orc.f32_to_bf16(_unit(rng, n)) for n in (64, 17, 33, 1, 64, 48)]
    q = _unit(rng, 21)
    ix = _idx(capacity_pages=8, stride_rows=64)
    ix.add(pages_bf16)
    want = _want(q, [orc.bf16_to_f32(p) for p in pages_bf16])
    _close(ix.score_all(q), want, tight=2e-6)
    _close(ix.score_candidates(q, np.arange(6)), want, tight=2e-6)
    # ... and it differs measurably from the rounded query's score (the old behaviour), which the bf16 query reproduces bit for bit
    q_r = orc.f32_to_bf16(q)
    rounded = ix.score_all(q_r)
    assert np.array_equal(rounded, ix.score_all(orc.bf16_to_f32(q_r)))  # fp32 values that ARE bf16: the one-term kernel's bits
    assert np.abs(rounded - want).max() > 1e-5
    ix.close()


def test_bf16_representable_inputs_bit_identical_with_and_without_the_lo_slab():
    """lo = 0 everywhere: the three-term kernels return the bits of the one-term kernels (x + 0 = x), for the scan, the
    candidate scorer, ragged pages, the pad clamp and the top-k."""
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 5, 64, 20]
    pages = [orc.synth_rows(77, i, 0, n) if n else np.zeros((0, 128), np.uint16) for i, n in enumerate(lens)]
    q = orc.synth_rows(4321, 9, 0, 21)
    res = []
    for lo in (False, True):
        ix = _idx(capacity_pages=32, stride_rows=64, with_float_lo=lo)
        # added as float32 values that are bf16 (lo slab written by the ingest kernel: all zeros) and as bf16 bit patterns
        ix.add([orc.bf16_to_f32(p) for p in pages[:8]])
        ix.add(pages[8:])
        res.append((ix.score_all(q), ix.score_candidates(q, [1, 4, 11, 12], 64), ix.score_candidates(q, np.arange(15), pad_to=-1), ix.query(q, 5)))
        ix.close()
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][3][0], res[1][3][0]) and np.array_equal(res[0][3][1], res[1][3][1])


@pytest.mark.parametrize("variant", [0, 6, 7])
def test_fp32_ragged_pages_pad_clamp_filter_tombstones(variant):
    """Ragged fp32 pages (incl. empty and one-row pages, partial last tiles), the reference's zero-pad clamp, the doc filter and
    tombstones on the split-bf16 kernels."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    rng = np.random.default_rng(99)
    lens = [(i * 37) % 200 + 1 for i in range(60)] + [0, 208, 16, 1]
    pages = [_unit(rng, n) if n else np.zeros((0, 128), np.float32) for n in lens]
    q = _unit(rng, 40)
    ix = _idx(capacity_pages=80, stride_rows=208, with_float_lo=True)
    ix.set_option(_lib.MV_OPT_MAXSIM_VARIANT, variant)
    ix.add(pages, doc_ordinals=[i // 3 for i in range(len(pages))])
    want = _want(q, pages)
    _close(ix.score_all(q), want)
    cand = [1, 4, 11, 12, 60, 61, 63]
    pad = max(lens[c] for c in cand)
    _close(ix.score_candidates(q, cand, pad), _want(q, [pages[c] for c in cand], pad))
    # the reference rule over the whole list: one batch of 64 pages here, padded to its longest (208)
    _close(ix.score_candidates(q, np.arange(len(pages)), pad_to=-1), _want(q, pages, 208))
    ix.remove_page(7)
    allow = allow_bitmap([d for d in range(22) if d % 4 != 1], 22)
    ok = np.array([(i // 3) % 4 != 1 and i != 7 for i in range(len(pages))])
    got = ix.score_all(q, allow=allow)
    assert np.all(np.isneginf(got[~ok]))
    _close(got[ok], want[ok])
    s, i = ix.query(q, 12, allow=allow)
    ws, wi = orc.topk(np.where(ok, want, -np.inf), 12)
    assert i.tolist() == wi.tolist()
    _close(s, ws)
    ix.close()


@pytest.mark.parametrize("nq", [1, 16, 17, 64, 65, 100, 128, 129, 200])
def test_fp32_query_lengths_pass_splitting(nq):
    """The lo fragments take the registers of 64 query rows: longer queries run in passes of 64 rows (sum over query rows is
    linear) -- 1024-patch pages, the default kernel and the page-per-wave one."""
    rng = np.random.default_rng(1000 + nq)
    pages = [_unit(rng, 1024) for _ in range(12)]
    q = _unit(rng, nq)
    want = _want(q, pages)
    ix = _idx(capacity_pages=16, stride_rows=1024, with_float_lo=True)
    ix.add(pages)
    _close(ix.score_all(q), want)
    _close(ix.score_candidates(q, np.arange(12)), want)
    ix.close()
    # fp32 query against bf16 pages (two-term kernels), same pass rule
    ix = _idx(capacity_pages=16, stride_rows=1024)
    ix.add([orc.f32_to_bf16(p) for p in pages])
    _close(ix.score_all(q), _want(q, [orc.bf16_to_f32(orc.f32_to_bf16(p)) for p in pages]), tight=2e-6)
    ix.close()


def test_fp32_pages_survive_replace_write_compact_save_load(tmp_path):
    """Every writer keeps the two halves together: replace_page / write_rows (bf16 input: lo rows zeroed), compaction (the lo
    slab moves with the pages), save / load."""
    from morphik_core_amd.index import MvIndex

    rng = np.random.default_rng(5)
    pages = [_unit(rng, 48) for _ in range(20)]
    q = _unit(rng, 32)
    ix = _idx(capacity_pages=32, stride_rows=48, with_float_lo=True)
    ix.add(pages, doc_ordinals=list(range(20)))
    new3 = orc.f32_to_bf16(_unit(rng, 30))
    ix.replace_page(3, new3)
    pages[3] = orc.bf16_to_f32(new3)
    rows = orc.f32_to_bf16(_unit(rng, 5))
    ix.write_rows(6, 10, rows)
    pages[6] = pages[6].copy()
    pages[6][10:15] = orc.bf16_to_f32(rows)
    _close(ix.score_all(q), _want(q, pages))
    for d in (0, 5, 11):
        ix.remove_doc(d)
    o2n = ix.compact()
    live = [p for i, p in enumerate(pages) if i not in (0, 5, 11)]
    assert (o2n >= 0).sum() == len(live) == len(ix)
    want = _want(q, live)
    _close(ix.score_all(q), want)
    path = os.path.join(tmp_path, "lo.mvidx")
    ix.save(path)
    got_before = ix.score_all(q)
    ix.close()
    ix2 = MvIndex.load(path)
    assert np.array_equal(ix2.score_all(q), got_before)
    back = ix2.read_pages_f32(0, len(live))
    for i, p in enumerate(live):
        assert np.abs(back[i, : p.shape[0]] - p).max() <= 2.0 ** -17
        assert not back[i, p.shape[0]:].any()
    ix2.close()


def test_fp32_two_stage_pipelines_rerank_with_both_halves():
    """MV_MODE_FDE_THEN_FLOAT and MV_MODE_FP8_THEN_FLOAT on an index with the lo slab: the candidates' final scores are the fp32
    oracle's (the rerank reads hi and lo); single request == batched requests (fp32 queries in the batch block's lo half); two
    logical shards behind the communicator == one index."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import MvIndex, ShardComm

    rng = np.random.default_rng(11)
    n = 96
    pages = [_unit(rng, 64) for _ in range(n)]
    qs = [_unit(rng, 24), _unit(rng, 32), _unit(rng, 9)]
    ix = _idx(capacity_pages=n, stride_rows=64, with_float_lo=True, with_fde=True, with_fp8=True)
    ix.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
    ix.set_option(_lib.MV_OPT_PAD_SEMANTICS, 0)
    ix.add(pages)
    singles = {}
    for mode in ("fde_then_float", "fp8_then_float"):
        for j, q in enumerate(qs):
            s, i = ix.query(q, 8, mode=mode)
            want = _want(q, [pages[p] for p in i])
            _close(s, want)
            assert np.all(np.diff(s) <= 0)
            singles[mode, j] = (s, i)
        for j, (bs, bi) in enumerate(ix.query_batch(qs, 8, mode=mode)):
            assert bi.tolist() == singles[mode, j][1].tolist()
            np.testing.assert_allclose(bs, singles[mode, j][0], rtol=1e-6)
    # exact float batch (fp32 queries are served by the split-bf16 single-query kernels inside the one call)
    full = [ix.query(q, 8) for q in qs]
    for (bs, bi), (s, i) in zip(ix.query_batch(qs, 8), full):
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s)
    for q, (s, i) in zip(qs, full):
        ws, wi = orc.topk(_want(q, pages), 8)
        assert i.tolist() == wi.tolist()
        _close(s, ws)
    # two logical shards
    half = n // 2
    shards = []
    for r in range(2):
        sh = MvIndex(capacity_pages=half, stride_rows=64, device=0, id_base=r * half, with_float_lo=True, with_fde=True, with_fp8=True)
        sh.set_option(_lib.MV_OPT_FDE_COARSE_N, 40)
        sh.set_option(_lib.MV_OPT_PAD_SEMANTICS, 0)
        sh.add(pages[r * half : (r + 1) * half])
        shards.append(sh)
    comm = ShardComm(shards)
    for mode in ("float", "fde_then_float", "fp8_then_float"):
        for j, q in enumerate(qs):
            cs, ci = comm.query(q, 8, mode=mode)
            ws, wi = (full[j] if mode == "float" else singles[mode, j])
            assert ci.tolist() == wi.tolist(), mode
            np.testing.assert_allclose(cs, ws, rtol=1e-6)
    comm.close()
    for sh in shards:
        sh.close()
    ix.close()


def test_float_lo_flag_needs_the_float_slab():
    from morphik_core_amd._lib import MvError

    with pytest.raises(MvError):
        _idx(capacity_pages=4, stride_rows=16, with_float=False, with_fp8=True, with_float_lo=True)


@pytest.mark.parametrize("stride,n_pages", [(64, 300), (1024, 150)])
def test_fp32_batches_in_cascade_mode_take_one_slab_pass_per_group_and_equal_the_single_requests(stride, n_pages):
    """MV_OPT_FLOAT_LO_SCAN 2 (hi scan -> best max(MV_OPT_RERANK_N, k) -> split-bf16 re-score): a BATCH of fp32 requests is served by the
    batched bf16 MFMA scan + the one-launch split rerank of all lists (round 6; such batches ran query by query) and returns what
    the single requests return -- same ids, same fp32-faithful scores, the oracle's top-k -- with a shared filter, per-request
    filters, tombstones, ragged pages, more queries than one group holds, and queries longer than the one-launch rerank takes."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    rng = np.random.default_rng(stride + n_pages)
    lens = [int(x) for x in rng.integers(max(1, stride // 3), stride + 1, size=n_pages)]
    pages = [_unit(rng, n) for n in lens]
    ix = _idx(capacity_pages=n_pages, stride_rows=stride, with_float_lo=True)
    ix.add(pages, doc_ordinals=[i // 2 for i in range(n_pages)])
    ix.set_option(_lib.MV_OPT_FLOAT_LO_SCAN, 2)
    ix.set_option(_lib.MV_OPT_RERANK_N, 40)
    k = 9
    for nq, B in ((32, 5), (9, 20), (80, 3)):
        qs = [_unit(rng, nq) for _ in range(B)]
        singles = [ix.query(q, k) for q in qs]
        for q, (s, i) in zip(qs[:2], singles[:2]):
            ws, wi = orc.topk(_want(q, pages), k)
            assert i.tolist() == wi.tolist()
            _close(s, ws)
        got, st = ix.query_batch(qs, k, want_stats=True)
        groups = -(-B // min(512 // (-(-nq // 16) * 16), 32))
        if nq <= 64:
            assert st.score_launches == groups * 2  # ONE scan over the slab and ONE rerank launch per group of requests
        else:
            assert groups * 2 <= st.score_launches <= groups + 2 * B  # one scan per group; the lists of longer queries are re-scored per request, in passes of 64 rows
        for (bs, bi), (s, i) in zip(got, singles):
            assert bi.tolist() == i.tolist()
            assert np.array_equal(bs, s)
    # filters and tombstones
    ix.remove_doc(3)
    n_docs = (n_pages + 1) // 2
    shared = allow_bitmap([d for d in range(n_docs) if d % 4 != 1], n_docs)
    qs = [_unit(rng, 24) for _ in range(6)]
    per_req = [allow_bitmap([d for d in range(n_docs) if (d + b) % 3 != 0], n_docs) for b in range(6)]
    for (bs, bi), q in zip(ix.query_batch(qs, k, allow=shared), qs):
        s, i = ix.query(q, k, allow=shared)
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s)
    for (bs, bi), q, ab in zip(ix.query_batch(qs, k, allows=per_req, n_docs=n_docs), qs, per_req):
        s, i = ix.query(q, k, allow=ab)
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s)
        assert not (set(i.tolist()) & {6, 7})
    ix.close()


@pytest.mark.parametrize("packed", [False, True])
def test_fp32_cascade_batches_with_fewer_allowed_pages_than_the_rerank_list(packed):
    """The corners of the batched cascade (`float_cascade_batch_query`): filters that admit fewer pages than MV_OPT_RERANK_N and
    fewer than k (the hi scan's selection hands -1 entries to the rerank lists), a request whose filter admits nothing, a corpus
    smaller than k, the packed layout -- every request answers as it does alone and as the fp32 oracle does."""
    from morphik_core_amd import _lib
    from morphik_core_amd.index import allow_bitmap

    rng = np.random.default_rng(77 + packed)
    n_pages, stride, k = 60, 48, 10
    lens = [int(x) for x in rng.integers(5, stride + 1, size=n_pages)]
    pages = [_unit(rng, n) for n in lens]
    ix = _idx(capacity_pages=n_pages, stride_rows=stride, with_float_lo=True, packed=packed)
    ix.add(pages, doc_ordinals=list(range(n_pages)))
    ix.set_option(_lib.MV_OPT_FLOAT_LO_SCAN, 2)
    ix.set_option(_lib.MV_OPT_RERANK_N, 40)
    qs = [_unit(rng, 20) for _ in range(5)]
    allowed = [[3, 17, 41], list(range(0, 60, 9)), [], [59], list(range(60))]
    per_req = [allow_bitmap(a, n_pages) for a in allowed]
    got = ix.query_batch(qs, k, allows=per_req, n_docs=n_pages)
    for (bs, bi), q, a, ab in zip(got, qs, allowed, per_req):
        s, i = ix.query(q, k, allow=ab)
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s)
        assert len(bi) == min(k, len(a)) and set(bi.tolist()) <= set(a)
        if a:
            ws, wi = orc.topk(_want(q, [pages[p] for p in a]), k)
            assert bi.tolist() == [a[j] for j in wi.tolist()]
            _close(bs, ws)
    # a shared filter that leaves 4 pages, then tombstones that leave fewer pages than k in the whole index
    shared = allow_bitmap([1, 2, 30, 31], n_pages)
    for (bs, bi), q in zip(ix.query_batch(qs, k, allow=shared), qs):
        s, i = ix.query(q, k, allow=shared)
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s) and sorted(bi.tolist()) == [1, 2, 30, 31]
    for d in range(7, n_pages):
        ix.remove_doc(d)
    for (bs, bi), q in zip(ix.query_batch(qs, k), qs):
        s, i = ix.query(q, k)
        assert bi.tolist() == i.tolist() and np.array_equal(bs, s) and sorted(bi.tolist()) == list(range(7))
    ix.close()
