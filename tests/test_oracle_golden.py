"""Pin the CPU oracle against everything the reference offers for this path.

Golden fixtures under tests/golden/ were produced by oracle/gen_golden.py from the reference's
own importable code (core/utils/fast_ops.py) and from transformers' score_retrieval (the
in-container twin of colpali_engine.score_multi_vector).  Known answers are quoted from the
reference's tests with file:line.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---------------------------------------------------------------- A4 sign pack
def test_sign_pack_matches_reference_fast_ops(golden_dir):
    g = _load(golden_dir, "sign_pack.npz")
    for ci in range(int(g["n_cases"])):
        x, packed, bools = g[f"x{ci}"], g[f"packed{ci}"], g[f"bools{ci}"]
        got = orc.sign_pack(x)
        assert got.dtype == np.uint8 and got.shape == packed.shape
        assert np.array_equal(got, packed), f"case {ci}"
        assert np.array_equal(orc.sign_pack_np(x), packed), f"case {ci} (numpy)"
        d = x.shape[1]
        assert np.array_equal(np.unpackbits(got, axis=-1)[:, :d].astype(bool), bools)


def test_sign_pack_reference_unit_test_case(golden_dir):
    # core/tests/unit/test_multivector.py:94-109: [[0.1,-0.2,0.3],[-0.1,0.2,-0.3]] -> "101", "010"
    got = orc.sign_pack(np.array([[0.1, -0.2, 0.3], [-0.1, 0.2, -0.3]], np.float32))
    assert [format(b >> 5, "03b") for b in got[:, 0]] == ["101", "010"]
    g = _load(golden_dir, "sign_pack.npz")
    assert g["bools_ref_test"].tolist() == [[True, False, True], [False, True, False]]


def test_sign_pack_rust_known_answers():
    # morphik_rust/src/binary_ops.rs:309-320 -> 0b10100101 (MSB first)
    assert orc.sign_pack(np.array([1, -1, 1, -1, -1, 1, -1, 1], np.float32))[0, 0] == 0b10100101
    # binary_ops.rs:298-306: zero maps to 0
    bits = np.unpackbits(orc.sign_pack(np.array([1.0, -0.5, 0.1, -2.0, 0.0, 3.0, -1.0, 0.5], np.float32)))[:8]
    assert bits.astype(bool).tolist() == [True, False, True, False, False, True, False, True]


def test_hamming_matches_reference(golden_dir):
    g = _load(golden_dir, "hamming.npz")
    for a, b, hd in zip(g["a"], g["b"], g["hd"]):
        assert orc.hamming(a, b) == hd
    assert [orc.hamming(g["q"], y) for y in g["b"]] == g["batch"].tolist()
    # binary_ops.rs:322-334
    assert orc.hamming(bytes([0b11110000, 0b10101010]), bytes([0b11110000, 0b01010101])) == 8


# ---------------------------------------------------------------- A5 binary MaxSim
def test_binary_maxsim_reference_known_ranking():
    # core/tests/unit/test_multivector.py:214-256: doc1 = 3 x (+1*64, -1*64), doc2 = negation,
    # query = 1 x (+1*64, -1*64) -> ranking [doc1, doc2]; SQL scores are exactly 1.0 and 0.0.
    half = np.concatenate([np.ones(64), -np.ones(64)]).astype(np.float32)
    doc1 = orc.sign_pack(np.stack([half] * 3))
    doc2 = orc.sign_pack(np.stack([-half] * 3))
    q = orc.sign_pack(half[None])
    assert orc.maxsim_binary(doc1, q) == 1.0
    assert orc.maxsim_binary(doc2, q) == 0.0


def test_binary_maxsim_identity_with_pm1_gemm():
    # SURVEY 8(a) A5 identity: max_sim = 0.5*Q + (1/256) * sum_q max_d (s_q . s_d), s = 2b-1
    rng = np.random.default_rng(3)
    docs = rng.standard_normal((5, 37, 128)).astype(np.float32)
    q = rng.standard_normal((9, 128)).astype(np.float32)
    qb = orc.sign_pack(q)
    sq = np.where(q > 0, 1.0, -1.0)
    for d in docs:
        sd = np.where(d > 0, 1.0, -1.0)
        ident = 0.5 * q.shape[0] + (sq @ sd.T).max(1).sum() / 256.0
        assert orc.maxsim_binary(orc.sign_pack(d), qb) == ident
    vec = orc.maxsim_binary_np(np.stack([orc.sign_pack(d) for d in docs]), qb)
    assert vec.tolist() == [orc.maxsim_binary(orc.sign_pack(d), qb) for d in docs]


def test_binary_maxsim_empty():
    q = orc.sign_pack(np.ones((2, 128), np.float32))
    assert orc.maxsim_binary(np.zeros((0, 16), np.uint8), q) == 0.0  # COALESCE(..., 0.0)


# ---------------------------------------------------------------- A7 float MaxSim
def test_float_maxsim_matches_score_retrieval(golden_dir):
    g = _load(golden_dir, "maxsim_float.npz")
    for ci in range(int(g["n_cases"])):
        q, slab, n_rows, pad_to, want = (g[f"{k}{ci}"] for k in ("q", "slab", "n_rows", "pad_to", "scores"))
        got = np.array([orc.maxsim_f32(q, slab[i, : n_rows[i]], int(pad_to[i])) for i in range(slab.shape[0])], np.float32)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6, err_msg=f"case {ci}")
        # numpy restatement (the bench's CPU baseline) agrees too
        got_np = np.concatenate(
            [
                orc.maxsim_float_np(q, slab[j : j + 128], n_rows[j : j + 128], pad_to=int(pad_to[j]))
                for j in range(0, slab.shape[0], 128)
            ]
        )
        np.testing.assert_allclose(got_np, want, rtol=2e-6, atol=2e-6, err_msg=f"case {ci} numpy")


def test_float_maxsim_bf16_path_and_torch_formulation():
    q = orc.synth_rows(4321, 0, 0, 32)
    pages = orc.synth_pages(1234, 0, 6, 48)
    a = orc.maxsim_bf16_slab(q, pages)
    b = orc.maxsim_float_np(orc.bf16_to_f32(q), orc.bf16_to_f32(pages))
    c = orc.maxsim_float_torch(orc.bf16_to_f32(q), orc.bf16_to_f32(pages))
    np.testing.assert_allclose(a, b, rtol=1e-6)
    np.testing.assert_allclose(a, c, rtol=1e-6)


def test_float_maxsim_zero_query_row_contributes_zero():
    rng = np.random.default_rng(0)
    p = rng.standard_normal((10, 128)).astype(np.float32)
    q = rng.standard_normal((4, 128)).astype(np.float32)
    q0 = np.concatenate([q, np.zeros((3, 128), np.float32)])
    assert orc.maxsim_f32(q0, p) == pytest.approx(orc.maxsim_f32(q, p), rel=1e-7)


# ---------------------------------------------------------------- top-k tie rule
def test_topk_order_and_ties():
    s = np.array([1.0, 3.0, 3.0, -np.inf, 2.0, 3.0, np.nan], np.float32)
    sc, ids = orc.topk(s, 4)
    assert ids.tolist() == [1, 2, 5, 4] and sc.tolist() == [3.0, 3.0, 3.0, 2.0]
    sc, ids = orc.topk(s, 100)  # k may exceed N; masked (-inf) rows never come back
    assert ids.tolist() == [1, 2, 5, 4, 0]
    import torch

    r = np.random.default_rng(5).standard_normal(1000).astype(np.float32)
    tv, ti = torch.topk(torch.from_numpy(r), 17)  # reference: torch.topk (fast_multivector_store.py:556)
    sc, ids = orc.topk(r, 17)
    assert ids.tolist() == ti.tolist() and sc.tolist() == tv.tolist()


# ---------------------------------------------------------------- bf16 + generator
def test_bf16_rounding_matches_torch():
    import torch

    rng = np.random.default_rng(1)
    x = np.concatenate(
        [rng.standard_normal(5000).astype(np.float32) * 10.0 ** rng.integers(-20, 20, 5000), np.array([0.0, -0.0, np.inf, -np.inf, 1.0039062, 1.00390625, 3.3895314e38], np.float32)]
    ).astype(np.float32)
    want = torch.from_numpy(x).bfloat16().view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(orc.f32_to_bf16(x), want)
    assert np.array_equal(orc.f32_to_bf16_np(x), want)
    assert np.array_equal(orc.bf16_to_f32(want), torch.from_numpy(x).bfloat16().float().numpy())


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert orc.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2).tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == [
        0xD16CFE09,
        0x94FDCCEB,
        0x5001E420,
        0x24126EA1,
    ]


def test_synth_rows_are_unit_norm_and_addressable():
    a = orc.synth_rows(1234, 7, 0, 64)
    b = orc.synth_rows(1234, 7, 10, 5)
    assert np.array_equal(a[10:15], b)  # any (page,row) regenerates independently
    n = np.linalg.norm(orc.bf16_to_f32(a), axis=1)
    assert np.all(np.abs(n - 1.0) < 5e-3)
    assert not np.array_equal(a, orc.synth_rows(1235, 7, 0, 64))
    m = orc.bf16_to_f32(orc.synth_rows(1, 0, 0, 2048)).mean()
    assert abs(m) < 2e-3


# ---------------------------------------------------------------- FDE (parity unpinned upstream)
def test_fde_structural_invariants():
    cfg = orc.FdeConfig.reference_default()
    assert cfg.output_dim == 10240  # 20 x 32 x 16 (fast_multivector_store.py:325-331)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((50, 128)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    fq = orc.fde_encode(cfg, x, True)
    fd = orc.fde_encode(cfg, x, False)
    assert fq.shape == (10240,) and fd.shape == (10240,)
    # permutation invariance over rows (up to fp32 summation order)
    perm = rng.permutation(50)
    np.testing.assert_allclose(orc.fde_encode(cfg, x[perm], True), fq, rtol=0, atol=1e-5)
    # query encoding is additive over disjoint row sets
    np.testing.assert_allclose(orc.fde_encode(cfg, x[:20], True) + orc.fde_encode(cfg, x[20:], True), fq, atol=1e-5)
    # doc = per-partition mean: sum-encoding divided by partition counts
    parts = orc.fde_partitions(cfg, x)
    assert parts.min() >= 0 and parts.max() < 32
    fq3, fd3 = fq.reshape(20, 32, 16), fd.reshape(20, 32, 16)
    for r in range(20):
        cnt = np.bincount(parts[r], minlength=32)
        for b in range(32):
            if cnt[b] == 0:
                assert not fd3[r, b].any() and not fq3[r, b].any()
            else:
                np.testing.assert_allclose(fd3[r, b], fq3[r, b] / cnt[b], rtol=1e-6, atol=1e-7)
    G, H, S = orc.fde_matrices(cfg)
    assert H.min() >= 0 and H.max() < 16 and set(np.unique(S)) == {-1.0, 1.0}
    assert abs(G.mean()) < 0.05 and 0.9 < G.std() < 1.1


def test_fde_dot_tracks_maxsim():
    cfg = orc.FdeConfig.reference_default()
    q = orc.bf16_to_f32(orc.synth_rows(4321, 0, 0, 32))
    pages = orc.bf16_to_f32(orc.synth_pages(1234, 0, 40, 64))
    # plant a near-copy of the query in page 3
    pages[3, :32] = q
    fq = orc.fde_encode(cfg, q, True)
    fds = np.stack([orc.fde_encode(cfg, p, False) for p in pages])
    coarse = orc.fde_coarse_scores(fq, orc.f32_to_bf16(fds), use_cosine=False)
    exact = orc.maxsim_float_np(q, pages)
    assert int(np.argmax(coarse)) == 3 == int(np.argmax(exact))


# ------------------------------------------------------------------ fp8 (e4m3fn) oracle pieces
def test_e4m3_codec_matches_torch_float8():
    """The quantiser both sides use is OCP e4m3fn with RNE and saturation at 448 -- pinned against
    torch.float8_e4m3fn (an independent implementation) on 200k values incl. ties, subnormals and the clamp."""
    import torch

    rng = np.random.default_rng(11)
    x = (rng.standard_normal(200_000) * np.exp(rng.standard_normal(200_000) * 3)).astype(np.float32)
    x = np.concatenate([x, np.array([0.0, -0.0, 448.0, 464.0, 1e9, -1e9, 2.0**-9, 2.0**-10, 1.5 * 2.0**-9, 17.0, 18.0, 19.0], np.float32)])
    ref = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(orc.e4m3_encode(x), ref)
    codes = np.arange(256, dtype=np.uint8)
    finite = (codes & 0x7F) != 0x7F
    want = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(orc.e4m3_decode(codes)[finite], want[finite])


def test_fp8_quantised_scores_track_fp32_scores():
    q = orc.bf16_to_f32(orc.synth_rows(4321, 0, 0, 32))
    pages = orc.synth_pages(1234, 0, 6, 96)
    codes, inv = zip(*(orc.quantize_page_fp8(p, 96) for p in pages))
    for c, iv in zip(codes, inv):
        assert 224.0 < np.abs(orc.e4m3_decode(c)).max() <= 448.0  # power-of-two scale puts amax in (224, 448]
        assert np.log2(iv) == np.round(np.log2(iv))
    got = orc.maxsim_fp8_np(q, np.stack(codes), np.array(inv, np.float32))
    one = np.array([orc.maxsim_fp8(q, codes[i], 96, inv[i]) for i in range(6)], np.float32)
    np.testing.assert_allclose(got, one, rtol=1e-6)
    ref = orc.maxsim_float_np(q, orc.bf16_to_f32(pages))
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 3e-2
    # two-term query split: hi + lo/16 reproduces the scaled query to ~2^-7 of its row maximum
    hi, lo, fac = orc.fp8_query_prep(q)
    rec = (orc.e4m3_decode(hi) + orc.e4m3_decode(lo) / 16.0) * fac[:, None]
    assert np.max(np.abs(rec - q) / np.abs(q).max(axis=1, keepdims=True)) < 2.0**-7


def test_bf16_hi_lo_split_of_the_query_fde_bounds_the_batched_coarse_scores():
    """The batched FDE coarse scan (csrc/mv_fde.hip, fde_scan_batch2_kernel) feeds the fp32 query FDE to the bf16 MFMA as
    hi = bf16(x), lo = bf16(x - hi).  hi + lo keeps 16 mantissa bits (|x - hi - lo| <= 2^-17 |x|), so its coarse scores sit
    within ~1e-5 of the fp32-query scan's on the same bf16 slab -- the bound the GPU tests assert at 1e-4."""
    rng = np.random.default_rng(0)
    cfg = orc.FdeConfig.reference_default()
    q = rng.standard_normal((32, 128)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    fq = orc.fde_encode(cfg, q, True).astype(np.float32)
    hi = orc.bf16_to_f32(orc.f32_to_bf16(fq))
    lo = orc.bf16_to_f32(orc.f32_to_bf16(fq - hi))
    nz = fq != 0
    assert np.all(np.abs(fq - hi - lo)[nz] <= np.abs(fq[nz]) * 2.0 ** -16)
    pages = rng.standard_normal((64, 40, 128)).astype(np.float32)
    pages /= np.linalg.norm(pages, axis=2, keepdims=True)
    slab = orc.bf16_to_f32(orc.f32_to_bf16(np.stack([orc.fde_encode(cfg, p, False) for p in pages])))
    exact = slab.astype(np.float64) @ fq.astype(np.float64)
    split = slab.astype(np.float64) @ (hi.astype(np.float64) + lo.astype(np.float64))
    hi_only = slab.astype(np.float64) @ hi.astype(np.float64)
    scale = np.abs(exact).max()
    assert np.abs(split - exact).max() <= 2e-5 * scale          # MV_OPT_FDE_BATCH_VARIANT = 0 (default)
    assert np.abs(hi_only - exact).max() <= 5e-3 * scale        # = 2: the query at the slab's own precision
    assert np.abs(hi_only - exact).max() > np.abs(split - exact).max()


# ---------------------------------------------------------------- FDE pin: "pinned on first contact"
def fde_pin_report(fixture_path, encode_doc, encode_query):
    """What a fixture written by oracle/gen_golden_fde.py (the reference's extension, fast_multivector_store.py:325-331, :447-449, :521)
    says about an encoder pair `encode_doc(rows) / encode_query(rows) -> float32[10240]` of this repo.  The random tables of two
    implementations differ unless seed and generator are matched, so bit equality is REPORTED, not required; required is what the
    pipeline relies on: both rank the planted page of every query first among the fixture's pages, and both FDE dot products order
    the pages like exact MaxSim to a similar degree."""
    from scipy.stats import spearmanr

    z = np.load(fixture_path)
    pages, queries, ref_d, ref_q = z["pages"], z["queries"], z["doc_fde"], z["q_fde"]
    assert ref_d.shape == (pages.shape[0], 10240) and ref_q.shape == (queries.shape[0], 10240)
    our_d = np.stack([encode_doc(p) for p in pages])
    our_q = np.stack([encode_query(q) for q in queries])
    exact = np.stack([orc.maxsim_float_np(q, pages) for q in queries])
    rep = {"bit_equal_doc": bool(np.array_equal(our_d, ref_d)), "bit_equal_query": bool(np.array_equal(our_q, ref_q)),
           "max_abs_diff_doc": float(np.abs(our_d - ref_d).max()), "planted_top1_ref": 0, "planted_top1_ours": 0}
    cos = np.sum(our_d * ref_d, 1) / np.maximum(np.linalg.norm(our_d, axis=1) * np.linalg.norm(ref_d, axis=1), 1e-30)
    rep["median_cosine_doc_vectors"] = float(np.median(cos))
    rho_ref, rho_ours = [], []
    for j in range(queries.shape[0]):
        sr, so = ref_d @ ref_q[j], our_d @ our_q[j]
        rep["planted_top1_ref"] += int(np.argmax(sr) == 3 * j + 1)
        rep["planted_top1_ours"] += int(np.argmax(so) == 3 * j + 1)
        rho_ref.append(spearmanr(sr, exact[j]).correlation)
        rho_ours.append(spearmanr(so, exact[j]).correlation)
    rep["spearman_vs_maxsim_ref"], rep["spearman_vs_maxsim_ours"] = float(np.mean(rho_ref)), float(np.mean(rho_ours))
    nq = queries.shape[0]
    assert rep["planted_top1_ref"] == nq, rep  # the fixture itself is sane
    assert rep["planted_top1_ours"] == nq, rep
    assert rep["spearman_vs_maxsim_ours"] >= rep["spearman_vs_maxsim_ref"] - 0.1, rep
    return rep


def test_fde_against_the_reference_extension_when_pinned():
    path = os.path.join(GOLDEN, "fde.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fde.npz absent: the reference's fde extension is not importable here (run oracle/gen_golden_fde.py where it is); FDE parity unpinned")
    cfg = orc.FdeConfig.reference_default()
    rep = fde_pin_report(path, lambda p: orc.fde_encode(cfg, p, False), lambda q: orc.fde_encode(cfg, q, True))
    print("FDE pin report (oracle vs reference extension):", rep)


def test_fde_pin_recipe_runs_end_to_end_on_a_stand_in_extension(tmp_path, monkeypatch):
    """The recipe is executed, not just committed: oracle/gen_golden_fde.py with tests/fake_fde_module.py standing in for the
    extension writes a fixture of the documented layout, and the pin report accepts it (bit-equal here: the stand-in IS the oracle)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_golden_fde", os.path.join(ROOT, "oracle", "gen_golden_fde.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert g.CONFIG == dict(dimension=128, num_repetitions=20, num_simhash_projections=5, projection_dimension=16, projection_type="AMS_SKETCH")
    import tests.fake_fde_module as fm

    monkeypatch.setattr(g, "find_extension", lambda: fm)
    monkeypatch.setattr(g, "OUT", str(tmp_path / "fde.npz"))
    assert g.main() == 0
    cfg = orc.FdeConfig.reference_default()
    rep = fde_pin_report(str(tmp_path / "fde.npz"), lambda p: orc.fde_encode(cfg, p, False), lambda q: orc.fde_encode(cfg, q, True))
    assert rep["bit_equal_doc"] and rep["bit_equal_query"] and rep["median_cosine_doc_vectors"] > 0.999999
    monkeypatch.setattr(g, "find_extension", lambda: None)
    assert g.main() == 3  # absent extension: a message and a distinct exit code, no fixture


def test_fp4_quantiser_of_the_fde_copy_known_answers():
    """orc_quantize_fde_fp4 / orc_fp4_encode (the checker of MV_WITH_FDE_FP4; not a reference function -- the reference's coarse stage is an
    ANN index): the e2m1 grid {0, 0.5, 1, 1.5, 2, 3, 4, 6}, round to nearest with ties to the even code, sign in bit 3, element 2i in the
    low nibble, and the scale = the smallest power of two with 12 * scale >= max|x| (elements beyond 6 * scale saturate)."""
    L = orc.lib()
    for v, c in [(0.0, 0), (0.25, 0), (0.26, 1), (0.5, 1), (0.75, 2), (1.0, 2), (1.25, 2), (1.26, 3), (1.75, 4), (2.5, 4), (2.51, 5), (3.5, 6),
                 (5.0, 6), (5.01, 7), (6.0, 7), (100.0, 7), (-6.0, 15), (-0.5, 9), (-0.0, 8)]:
        assert L.orc_fp4_encode(v) == c, (v, c)
    assert [L.orc_fp4_decode(c) for c in range(16)] == [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]
    row = orc.f32_to_bf16(np.array([[6.0, -3.0, 0.5, 0.2, 1.5, -1.5, 0.0, 4.0]], np.float32))
    codes, sc = orc.quantize_fde_fp4(row)
    # scale = the smallest power of two with 12 * scale >= amax (HALF the covering scale): amax 6 -> 0.5; 6.0 / 0.5 = 12 saturates at code 7,
    # -3.0 -> -6 (15), 0.5 -> 1.0 (2), 0.2 -> 0.4 (1), 1.5 -> 3 (5), -1.5 -> -3 (13), 0 -> 0, 4.0 -> 8 saturates (7)
    assert sc.tolist() == [0.5] and codes.tolist() == [[0x07 | (0x0F << 4), 0x02 | (0x01 << 4), 0x05 | (0x0D << 4), 0x00 | (0x07 << 4)]]
    # amax 6.5: 12 * 0.5 < 6.5 <= 12 * 1; amax 3.0 = 12 * 0.25 exactly keeps 0.25; an all-zero row: scale 1
    assert orc.quantize_fde_fp4(orc.f32_to_bf16(np.array([[6.5, 1.0]], np.float32)))[1].tolist() == [1.0]
    assert orc.quantize_fde_fp4(orc.f32_to_bf16(np.array([[3.0, -0.25]], np.float32)))[1].tolist() == [0.25]
    assert orc.quantize_fde_fp4(orc.f32_to_bf16(np.zeros((1, 4), np.float32)))[1].tolist() == [1.0]
    # decode(codes) * scale reproduces a row to the grid's half-step everywhere
    rng = np.random.default_rng(4)
    x = orc.f32_to_bf16((rng.standard_normal((5, 2048)) * 0.07).astype(np.float32))
    codes, sc = orc.quantize_fde_fp4(x)
    back = orc.fp4_decode(codes) * sc[:, None]
    xf = orc.bf16_to_f32(x)
    inside = np.abs(xf) <= 6.0 * sc[:, None]
    assert np.all(np.abs(back - xf)[inside] <= sc[:, None].repeat(xf.shape[1], 1)[inside] * 1.0 + 1e-12)  # the widest gap of the grid is 2 (between 4 and 6): half of it
    assert np.all(np.abs(back)[~inside] == (6.0 * sc[:, None]).repeat(xf.shape[1], 1)[~inside])  # beyond 6 * scale: saturated at the top code
    assert float(np.median(np.linalg.norm(back - xf, axis=1) / np.linalg.norm(xf, axis=1))) < 0.15
