"""CPU tests of the host-side pieces around the scoring path: the encoder plugin (tiny random-init preset of the
ColPali architecture), the reference's disk / wire formats, and the /embeddings server protocol."""
import asyncio
import base64
import io
import os

import numpy as np
import pytest

from morphik_core_amd import formats
from morphik_core_amd.models import Chunk


@pytest.fixture(scope="module")
def tiny_embedder():
    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel

    return MI355XColpaliEmbeddingModel(preset="tiny", device="cpu", batch_size=2)


def _png(seed, size=20):
    from PIL import Image

    rng = np.random.default_rng(seed)
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 255, (size, size, 3), dtype=np.uint8)).save(buf, format="PNG")
    return buf.getvalue()


def test_encoder_output_contract_like_reference_tests(tiny_embedder):
    """core/tests/unit/test_colpali_embedding.py:56-59,72-77: ndarray, shape[1]==128, float32 -- queries and images."""
    q = asyncio.run(tiny_embedder.embed_for_query("what is the total revenue in 2023 ?"))
    assert isinstance(q, np.ndarray) and q.dtype == np.float32 and q.ndim == 2 and q.shape[1] == 128
    assert q.shape[0] == 1 + 8 + 10  # BOS + words + 10 augmentation tokens
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=2e-2)  # bf16 rows, L2-normalised by the head
    chunks = [
        Chunk(content="", metadata={"is_image": True, "_image_bytes": _png(1)}),
        Chunk(content="data:image/png;base64," + base64.b64encode(_png(2)).decode(), metadata={"is_image": True}),
        Chunk(content="a plain text chunk", metadata={}),
        Chunk(content="not an image at all", metadata={"is_image": True}),  # decode failure -> embedded as text (:83-100)
        Chunk(content="", metadata={"is_image": True, "_image_bytes": _png(3)}),
    ]
    embs = asyncio.run(tiny_embedder.embed_for_ingestion(chunks))
    assert len(embs) == 5
    n_img = tiny_embedder.n_image_tokens + 6
    assert [e.shape for e in embs] == [(n_img, 128), (n_img, 128), (1 + 4 + 10, 128), (1 + 5 + 10, 128), (n_img, 128)]
    assert all(e.dtype == np.float32 for e in embs)
    t = tiny_embedder.latest_ingest_timing()
    assert t["image_count"] == 3 and t["text_count"] == 2 and t["chunk_count"] == 5 and t["total"] > 0
    # deterministic, and batching does not change a page's rows
    again = asyncio.run(tiny_embedder.embed_for_ingestion([chunks[4]]))[0]
    np.testing.assert_allclose(again, embs[4], atol=2e-2)
    rows, n_rows = asyncio.run(tiny_embedder.embed_for_ingestion_device(chunks))
    assert n_rows == [e.shape[0] for e in embs] and tuple(rows.shape) == (sum(n_rows), 128)


def test_patch_embedding_gemm_equals_the_convolution(tiny_embedder):
    """The SigLIP patch embedding runs as unfold + GEMM (MIOpen's naive conv was 40 % of the encoder's GPU time); the
    rewired forward must be the same function as the Conv2d it replaces, parameters untouched."""
    import torch
    import torch.nn as nn

    convs = [m for m in tiny_embedder.model.modules() if isinstance(m, nn.Conv2d)]
    assert len(convs) == 1 and "forward" in convs[0].__dict__  # rewired instance, same module / state_dict keys
    c = convs[0]
    x = torch.randn(3, 3, tiny_embedder.image_size, tiny_embedder.image_size, dtype=c.weight.dtype)
    with torch.inference_mode():
        got, want = c(x), nn.Conv2d.forward(c, x)
    assert got.shape == want.shape
    torch.testing.assert_close(got.float(), want.float(), rtol=2e-2, atol=2e-2)
    assert "vlm.vision_tower.vision_model.embeddings.patch_embedding.weight" in "".join(tiny_embedder.model.state_dict().keys()) or any(
        k.endswith("patch_embedding.weight") for k in tiny_embedder.model.state_dict())


def test_npy_pages_roundtrip_and_tree_walk(tmp_path):
    rng = np.random.default_rng(0)
    pages = {("docA", 0): rng.standard_normal((5, 128)), ("docA", 2): rng.standard_normal((1, 128)), ("docB", 10): rng.standard_normal((7, 128))}
    for (d, c), e in pages.items():
        p = tmp_path / "multivector" / d
        p.mkdir(parents=True, exist_ok=True)
        (p / f"{c}.npy").write_bytes(formats.save_npy_page(e))  # == np.save(float32): fast_multivector_store.py:676-681
    got = list(formats.iter_npy_tree(str(tmp_path)))
    assert [(d, c) for d, c, _ in got] == [("docA", 0), ("docA", 2), ("docB", 10)]
    for d, c, e in got:
        assert e.dtype == np.float32 and np.array_equal(e, pages[(d, c)].astype(np.float32))
    assert formats.parse_npy_key("bucket/x/multivector/doc-1/42.npy") == ("doc-1", 42)
    with pytest.raises(ValueError):
        formats.load_npy_page(formats.save_npy_page(np.zeros((3, 64))))


def test_npz_wire_format_is_what_the_reference_client_decodes():
    rng = np.random.default_rng(1)
    embs = [rng.standard_normal((n, 128)).astype(np.float32) for n in (3, 1030, 17)]
    body = formats.encode_embeddings_npz(embs, "image")
    # the client's own steps (colpali_api_embedding_model.py:293-310)
    z = np.load(io.BytesIO(body))
    assert int(z["count"]) == 3 and str(z["input_type"]) == "image"
    for i, e in enumerate(embs):
        assert np.array_equal(z[f"emb_{i}"].astype(np.float32, copy=False), e)
    dec, it = formats.decode_embeddings_npz(body)
    assert it == "image" and all(np.array_equal(a, b) for a, b in zip(dec, embs))


def test_bit_rows_match_reference_quantiser_golden(golden_dir):
    """BIT(128) text rows <-> packed bytes: same image as fast_ops.binary_quantize_packed (golden fixture)."""
    g = np.load(os.path.join(golden_dir, "sign_pack.npz"))
    keys = [k for k in g.files if k.startswith("x") and g[k].ndim == 2 and g[k].shape[-1] == 128]
    assert keys
    for k in keys:
        x, packed = g[k], g["packed" + k[1:]]
        strings = ["".join("1" if v > 0 else "0" for v in row) for row in x]  # what _binary_quantize stores
        assert np.array_equal(formats.bit_rows_to_packed(strings), packed)
        assert formats.packed_to_bit_strings(packed) == strings
        assert np.array_equal(formats.bit_rows_to_packed([bytes(r) for r in packed]), packed)
    with pytest.raises(ValueError):
        formats.bit_rows_to_packed(["0101"])


def test_embed_server_speaks_the_reference_protocol(tiny_embedder):
    from fastapi.testclient import TestClient

    from morphik_core_amd.embed_server import create_app

    client = TestClient(create_app(tiny_embedder, api_key="k"))
    hdr = {"Authorization": "Bearer k"}
    r = client.post("/embeddings", json={"input_type": "text", "inputs": ["hello world", "second query here"]}, headers=hdr)
    assert r.status_code == 200
    embs, it = formats.decode_embeddings_npz(r.content)
    assert it == "text" and [e.shape for e in embs] == [(1 + 2 + 10, 128), (1 + 3 + 10, 128)]
    r = client.post("/embeddings", json={"input_type": "image", "inputs": [base64.b64encode(_png(5)).decode()]}, headers=hdr)
    assert r.status_code == 200
    embs, it = formats.decode_embeddings_npz(r.content)
    assert it == "image" and embs[0].shape == (tiny_embedder.n_image_tokens + 6, 128) and embs[0].dtype == np.float32
    assert client.post("/embeddings", json={"input_type": "text", "inputs": ["x"]}).status_code == 401
    assert client.post("/embeddings", json={"input_type": "text", "inputs": ["x"] * 300}, headers=hdr).status_code == 413  # client bisects
    assert client.post("/embeddings", json={"input_type": "audio", "inputs": ["x"]}, headers=hdr).status_code == 422
