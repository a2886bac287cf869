"""The reference's store scenarios on the REAL MI355X index (through libmvmaxsim.so)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import store_scenarios as sc  # noqa: E402


def _store(mode, cls=None):
    from morphik_core_amd.store import MI355XFastMultiVectorStore, MI355XMultiVectorStore

    cls = cls or (MI355XFastMultiVectorStore if mode == "fde_then_float" else MI355XMultiVectorStore)
    s = cls(capacity_pages=64, stride_rows=32, mode=mode)
    assert s.initialize() is True
    return s


@pytest.mark.parametrize("scenario", sc.ALL, ids=lambda f: f.__name__)
@pytest.mark.parametrize("mode", ["binary", "float", "fde_then_float"])
def test_reference_store_scenarios_on_gpu(scenario, mode):
    s = _store(mode)
    try:
        sc.run(scenario(s))
    finally:
        s.close()


def test_known_ranking_on_gpu():
    for mode, exact in (("binary", True), ("float", False), ("fde_then_float", False)):
        s = _store(mode)
        sc.run(sc.scenario_known_ranking(s, exact_binary=exact))
        s.close()


def test_gpu_store_matches_oracle_backed_store():
    """Same chunks into the HIP-backed store and the oracle-backed one: identical rankings, scores within tolerance."""
    from morphik_core_amd.store import MI355XMultiVectorStore
    from tests.fake_index import OracleIndex

    rng = np.random.default_rng(11)
    chunks = sc.make_chunks(rng, n_docs=5, chunks_per_doc=6, rows=30)
    q = sc.rand_emb(rng, 19)
    for mode in ("binary", "float"):
        g = _store(mode)
        o = MI355XMultiVectorStore(capacity_pages=64, stride_rows=32, mode=mode, index_factory=OracleIndex)
        sc.run(g.store_embeddings(chunks))
        sc.run(o.store_embeddings(chunks))
        rg = sc.run(g.query_similar(q, k=12, doc_ids=["doc0", "doc3", "doc4"]))
        ro = sc.run(o.query_similar(q, k=12, doc_ids=["doc0", "doc3", "doc4"]))
        assert [(r.document_id, r.chunk_number) for r in rg] == [(r.document_id, r.chunk_number) for r in ro]
        if mode == "binary":
            assert [r.score for r in rg] == [r.score for r in ro]
        else:
            np.testing.assert_allclose([r.score for r in rg], [r.score for r in ro], rtol=1e-3)
        g.close()
