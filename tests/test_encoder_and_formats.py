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


def test_fused_encoder_ops_leave_a_model_without_a_gpu_exactly_as_it_was():
    """encoder_ops.patch_encoder on a CPU model: every RMSNorm / gated MLP / tanh-gelu ViT MLP is found and wrapped, the gate|up weights become
    two views of one matrix holding the same numbers (state_dict unchanged), and -- no GPU tensor in sight -- every wrapped module falls
    back to its transformers forward: bit-identical outputs.  The kernels themselves are held to the modules on the GPU
    (tests/test_gpu_store.py::test_fused_encoder_ops_match_the_transformers_modules)."""
    import torch

    from morphik_core_amd import encoder_ops
    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel

    emb = MI355XColpaliEmbeddingModel(preset="tiny", device="cpu", batch_size=2, seed=4)
    assert emb.fused_ops == {"rmsnorm": 0, "gated_mlp": 0, "gelu_epilogue": 0}  # never patched on its own without a GPU
    model = emb.model
    before = {k: v.clone() for k, v in model.state_dict().items()}
    ids = torch.tensor([[1, 5, 9, 3, 2]])
    with torch.inference_mode():
        want = model(input_ids=ids, attention_mask=torch.ones_like(ids)).embeddings.clone()
    n = encoder_ops.patch_encoder(model)
    cfg = model.config.vlm_config
    assert n["rmsnorm"] == 2 * cfg.text_config.num_hidden_layers + 1 and n["gated_mlp"] == cfg.text_config.num_hidden_layers
    assert n["gelu_epilogue"] == cfg.vision_config.num_hidden_layers
    assert encoder_ops.patch_encoder(model) == {"rmsnorm": 0, "gated_mlp": 0, "gelu_epilogue": 0}  # idempotent
    after = model.state_dict()
    assert set(after) == set(before) and all(torch.equal(after[k], before[k]) for k in before)
    mlp = next(m for m in model.modules() if hasattr(m, "_mv_fused_w"))
    assert mlp.gate_proj.weight.data_ptr() == mlp._mv_fused_w.data_ptr()  # the halves ARE the fused matrix
    with torch.inference_mode():
        got = model(input_ids=ids, attention_mask=torch.ones_like(ids)).embeddings
    assert torch.equal(got, want)
    import copy

    twin = copy.deepcopy(model).float()  # re-materialises every parameter: the fused gate|up copy is stale there and must be let go, not used
    tmlp = next(m for m in twin.modules() if hasattr(m, "_mv_fused_w"))
    with torch.inference_mode():
        twin(input_ids=ids, attention_mask=torch.ones_like(ids))
    assert tmlp._mv_fused_w is None and tmlp.gate_proj.weight.dtype == torch.float32
    assert mlp._mv_fused_w is not None  # the original keeps its own
    encoder_ops.unpatch_encoder(model)
    assert not any(hasattr(m, "_mv_orig_forward") for m in model.modules())
    with torch.inference_mode():
        assert torch.equal(model(input_ids=ids, attention_mask=torch.ones_like(ids)).embeddings, want)


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


# --------------------------------------------------------------------------- checkpoint + processor branch, ColQwen2 family
def test_colpali_adapter_loads_a_checkpoint_directory_with_its_processor(tmp_path):
    """The `model_name_or_path` branch (from_pretrained + ColPaliProcessor: colpali_embedding_model.py:47-59): a tiny
    random-init checkpoint and an offline-built processor saved to disk, loaded back and driven end to end."""
    import asyncio

    from morphik_core_amd.embedding import MI355XColpaliEmbeddingModel
    from morphik_core_amd.models import Chunk
    from tests import offline_assets as oa

    d = oa.colpali_checkpoint(str(tmp_path / "ckpt"))
    m = MI355XColpaliEmbeddingModel(model_name_or_path=d, device="cpu", batch_size=2)
    assert m.processor is not None and m.random_init is False
    rng = np.random.default_rng(0)
    chunks = [Chunk(content="", metadata={"is_image": True, "_image_bytes": oa.png_bytes(oa.page_image(rng, 60, 80))}) for _ in range(3)]
    chunks.append(Chunk(content="total revenue by quarter", metadata={}))
    embs = asyncio.run(m.embed_for_ingestion(chunks))
    n_img = (56 // 14) ** 2
    assert [e.shape[0] for e in embs[:3]] == [n_img + 4] * 3  # 16 image tokens + the processor's prompt tokens
    assert all(e.dtype == np.float32 and e.shape[1] == 128 for e in embs)
    for e in embs:
        np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=2e-2)  # L2-normalised by the model head
    again = asyncio.run(m.embed_for_ingestion(chunks[:1]))
    np.testing.assert_allclose(again[0], embs[0], atol=2e-2)  # deterministic, batch-independent
    q = asyncio.run(m.embed_for_query("what is shown in this image"))
    assert q.ndim == 2 and q.shape[1] == 128 and q.dtype == np.float32
    t = m.latest_ingest_timing()
    assert t["image_count"] == 1 and t["model"] > 0


@pytest.mark.parametrize("preset", ["tiny", "tiny-2.5"])  # Qwen2-VL backbone / Qwen2.5-VL backbone (the reference's ColQwen2.5)
def test_colqwen2_adapter_ragged_pages_through_the_store(preset):
    """The reference's encoder family (ColQwen2.5: colpali_embedding_model.py:47-52) has DYNAMIC patch counts.  Tiny
    random-init ColQwen2ForRetrieval + offline processor: pages of different resolutions give different row counts, the
    store takes them as ragged pages, every page retrieves itself."""
    import asyncio

    import torch

    from morphik_core_amd.colqwen_embedding import MI355XColQwen2EmbeddingModel, build_random_colqwen2
    from morphik_core_amd.models import Chunk, DocumentChunk
    from morphik_core_amd.store import MI355XMultiVectorStore
    from tests import offline_assets as oa
    from tests.fake_index import OracleIndex

    proc, ids = oa.colqwen2_processor()
    emb = MI355XColQwen2EmbeddingModel(model=build_random_colqwen2(preset, ids, "cpu", torch.bfloat16), processor=proc, device="cpu", batch_size=2)
    rng = np.random.default_rng(1)
    sizes = [(60, 60), (56, 112), (112, 84), (30, 200), (84, 84)]
    chunks = [Chunk(content="", metadata={"is_image": True, "_image_bytes": oa.png_bytes(oa.page_image(rng, h, w))}) for h, w in sizes]
    chunks.append(Chunk(content="hello world", metadata={}))
    embs = asyncio.run(emb.embed_for_ingestion(chunks))
    n_rows = [e.shape[0] for e in embs]
    assert len(set(n_rows[:5])) >= 3, n_rows  # the row count follows the page's resolution
    for e in embs:
        assert e.dtype == np.float32 and e.shape[1] == 128
        np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=2e-2)
    rows, nr = asyncio.run(emb.embed_for_ingestion_device(chunks))
    assert nr == n_rows and rows.shape == (sum(n_rows), 128) and rows.dtype == torch.bfloat16
    store = MI355XMultiVectorStore(capacity_pages=16, stride_rows=32, mode="float", index_factory=OracleIndex)
    assert store.initialize()
    dcs = [DocumentChunk(document_id=f"d{i // 3}", chunk_number=i % 3, content=f"page {i}", embedding=e, metadata={"rows": e.shape[0]}) for i, e in enumerate(embs)]
    ok, ids_, m = asyncio.run(store.store_embeddings(dcs))
    assert ok and len(ids_) == 6 and m["multivector_bytes"] == sum(n_rows) * 256
    for i, e in enumerate(embs):
        hit = asyncio.run(store.query_similar(e, k=2))[0]
        assert hit.content == f"page {i}" and hit.metadata == {"rows": n_rows[i]}
    q = asyncio.run(emb.embed_for_query("total revenue by quarter"))
    assert q.shape[1] == 128 and len(asyncio.run(store.query_similar(q, k=3))) == 3


def _serve(app):
    """uvicorn on a free localhost port in a background thread -> (base url, stop())."""
    import socket
    import threading
    import time

    import uvicorn

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    server = uvicorn.Server(uvicorn.Config(app, host="127.0.0.1", port=port, log_level="warning"))
    th = threading.Thread(target=server.run, daemon=True)
    th.start()
    for _ in range(200):
        if server.started:
            break
        time.sleep(0.05)
    assert server.started

    def stop():
        server.should_exit = True
        th.join(timeout=10)

    return f"http://127.0.0.1:{port}", stop


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")
def test_the_references_own_api_client_drives_the_embed_server(tiny_embedder):
    """ColpaliApiEmbeddingModel (core/embedding/colpali_api_embedding_model.py) -- the reference's
    client class itself, imported from the reference checkout -- talks to the MI355X embed server over real HTTP:
    _call_api_endpoint (:273-310), embed_for_query (:312-319) and embed_for_ingestion with its batching (:109-271)."""
    import asyncio
    import sys
    import types

    from morphik_core_amd.embed_server import create_app
    from morphik_core_amd.models import Chunk

    url, stop = _serve(create_app(tiny_embedder, api_key="secret"))
    saved = {k: sys.modules.get(k) for k in ("core", "core.config", "core.embedding", "core.models")}
    try:
        # the client's package __init__ imports colpali_engine (absent here): register bare package modules that point at the
        # reference directories, and a settings stub carrying the two fields the client reads (:41-48)
        core = types.ModuleType("core")
        core.__path__ = [os.path.join(REF, "core")]
        emb_pkg = types.ModuleType("core.embedding")
        emb_pkg.__path__ = [os.path.join(REF, "core", "embedding")]
        models_pkg = types.ModuleType("core.models")
        models_pkg.__path__ = [os.path.join(REF, "core", "models")]
        cfg = types.ModuleType("core.config")
        cfg.get_settings = lambda: types.SimpleNamespace(MORPHIK_EMBEDDING_API_KEY="secret", MORPHIK_EMBEDDING_API_DOMAIN=[url])
        sys.modules.update({"core": core, "core.config": cfg, "core.embedding": emb_pkg, "core.models": models_pkg})
        from core.embedding.colpali_api_embedding_model import ColpaliApiEmbeddingModel  # the reference's class

        client = ColpaliApiEmbeddingModel()
        assert client.endpoints == [url + "/embeddings"]
        q = asyncio.run(client.embed_for_query("hello world"))
        want = asyncio.run(tiny_embedder.embed_for_query("hello world"))
        assert isinstance(q, np.ndarray) and q.dtype == np.float32 and q.shape == want.shape
        np.testing.assert_allclose(q, want, atol=2e-2)
        got = asyncio.run(client._call_api_endpoint(url + "/embeddings", [base64.b64encode(_png(3)).decode(), base64.b64encode(_png(4)).decode()], "image"))
        assert len(got) == 2 and all(g.shape == (tiny_embedder.n_image_tokens + 6, 128) and g.dtype == np.float32 for g in got)
        from core.models.chunk import Chunk as RefChunk  # the reference's own record type

        chunks = [RefChunk(content="data:image/png;base64," + base64.b64encode(_png(i)).decode(), metadata={"is_image": True}) for i in range(3)]
        chunks.insert(1, RefChunk(content="second query here", metadata={}))
        embs = asyncio.run(client.embed_for_ingestion(chunks))
        ours = asyncio.run(tiny_embedder.embed_for_ingestion([Chunk(content=c.content, metadata=dict(c.metadata)) for c in chunks]))
        assert len(embs) == 4
        for a, b in zip(embs, ours):  # order restored across the client's image / text partition
            assert np.asarray(a).shape == b.shape
            np.testing.assert_allclose(np.asarray(a, np.float32), b, atol=2e-2)
    finally:
        stop()
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in [k for k in sys.modules if k.startswith("core.")]:
            sys.modules.pop(k, None)
