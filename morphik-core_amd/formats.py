"""Importers / exporters for the reference's on-disk and wire formats -- so an existing
Morphik deployment can switch provider without re-embedding, and an MI355X embed server can answer the
reference's own client.

  .npy  per page      multivector/{document_id}/{chunk_number}.npy, float32 (P,128)
                      written by FastMultiVectorStore._save_multivector_to_storage_with_cache_time
                      (core/vector_store/fast_multivector_store.py:673-707), read back at :713-774
  .npz  embed API     {"count", "input_type", "emb_0", "emb_1", ...} float32 arrays
                      decoded by ColpaliApiEmbeddingModel._call_api_endpoint
                      (core/embedding/colpali_api_embedding_model.py:293-310)
  BIT(128)[] rows     MultiVectorStore's table column (core/vector_store/multi_vector_store.py:248); a row is a list
                      of 128-character '0'/'1' strings (psycopg text form) or pgvector Bit objects; sign-only --
                      floats are unrecoverable (cf. scripts/migrate_postgres_to_turbopuffer.py:233-300)

Pure host code (numpy); nothing here scores anything.
"""
from __future__ import annotations

import io
import os
import re
from typing import Any, Iterable, Iterator, List, Sequence, Tuple

import numpy as np

_NPY_KEY = re.compile(r"(?:^|/)multivector/(?P<doc>[^/]+)/(?P<chunk>\d+)\.npy$")


# --------------------------------------------------------------------------- .npy pages
def load_npy_page(src: Any) -> np.ndarray:
    """path / bytes / file object -> (P,128) float32 (the loader upcasts whatever was stored, like :736,774)."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        src = io.BytesIO(bytes(src))
    a = np.load(src, allow_pickle=False)
    a = np.asarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != 128:
        raise ValueError(f"multivector page must be (P,128); got {a.shape}")
    return np.ascontiguousarray(a)


def save_npy_page(emb: Any) -> bytes:
    """(P,128) -> the bytes FastMultiVectorStore uploads (np.save of float32)."""
    buf = io.BytesIO()
    np.save(buf, np.asarray(emb, dtype=np.float32))
    return buf.getvalue()


def parse_npy_key(key: str) -> Tuple[str, int]:
    """'.../multivector/{document_id}/{chunk_number}.npy' -> (document_id, chunk_number)."""
    m = _NPY_KEY.search(key.replace(os.sep, "/"))
    if not m:
        raise ValueError(f"not a multivector storage key: {key!r}")
    return m.group("doc"), int(m.group("chunk"))


def iter_npy_tree(root: str) -> Iterator[Tuple[str, int, np.ndarray]]:
    """Walk a LocalStorage tree: yields (document_id, chunk_number, (P,128) float32), documents and chunks in order."""
    base = os.path.join(root, "multivector") if os.path.isdir(os.path.join(root, "multivector")) else root
    for doc in sorted(os.listdir(base)):
        d = os.path.join(base, doc)
        if not os.path.isdir(d):
            continue
        chunks = sorted((int(f[:-4]), f) for f in os.listdir(d) if f.endswith(".npy") and f[:-4].isdigit())
        for n, f in chunks:
            yield doc, n, load_npy_page(os.path.join(d, f))


# --------------------------------------------------------------------------- .npz embed-API wire format
def encode_embeddings_npz(embeddings: Sequence[Any], input_type: str) -> bytes:
    """What an /embeddings endpoint answers: np.savez with count, input_type, emb_i (float32)."""
    buf = io.BytesIO()
    arrays = {f"emb_{i}": np.asarray(e, dtype=np.float32) for i, e in enumerate(embeddings)}
    np.savez(buf, count=np.array(len(embeddings)), input_type=np.array(input_type), **arrays)
    return buf.getvalue()


def decode_embeddings_npz(content: bytes) -> Tuple[List[np.ndarray], str]:
    """Inverse, exactly the client's steps (colpali_api_embedding_model.py:293-310)."""
    z = np.load(io.BytesIO(content), allow_pickle=False)
    count = int(z["count"])
    return [z[f"emb_{i}"].astype(np.float32, copy=False) for i in range(count)], str(z["input_type"])


# --------------------------------------------------------------------------- BIT(128)[] rows
def bit_rows_to_packed(rows: Iterable[Any]) -> np.ndarray:
    """One table row's `embeddings` column -> (P,16) uint8, MSB first (the byte image of BIT(128)).
    Accepts '0101...' strings, pgvector.Bit-like objects (to_text() / str()), 16-byte values, or 0/1 sequences."""
    out = []
    for r in rows:
        if isinstance(r, (bytes, bytearray, memoryview)):
            b = np.frombuffer(bytes(r), np.uint8)
        else:
            if hasattr(r, "to_text"):
                r = r.to_text()
            if isinstance(r, str):
                bits = np.frombuffer(r.strip().encode("ascii"), np.uint8) - ord("0")
            else:
                bits = np.asarray(r).astype(np.uint8)
            if bits.size != 128 or bits.max(initial=0) > 1:
                raise ValueError("BIT(128) value expected")
            b = np.packbits(bits, bitorder="big")
        if b.size != 16:
            raise ValueError("BIT(128) value must pack to 16 bytes")
        out.append(b)
    return np.ascontiguousarray(np.stack(out)) if out else np.zeros((0, 16), np.uint8)


def packed_to_bit_strings(packed: np.ndarray) -> List[str]:
    """(P,16) uint8 -> ['0101...'] as the reference inlines them into SQL (multi_vector_store.py:742-743)."""
    bits = np.unpackbits(np.ascontiguousarray(packed, dtype=np.uint8), axis=1, bitorder="big")
    return ["".join("1" if b else "0" for b in row) for row in bits]


def import_npy_tree_into_store(store: Any, root: str, batch: int = 64, app_id: Any = None) -> int:
    """Feed every multivector/{doc}/{chunk}.npy under `root` to an MI355X store (synchronously, in batches)."""
    import asyncio

    from .models import DocumentChunk

    pending, n = [], 0

    def flush():
        nonlocal pending, n
        if pending:
            asyncio.run(store.store_embeddings(pending, app_id))
            n += len(pending)
            pending = []

    for doc, chunk_no, emb in iter_npy_tree(root):
        pending.append(DocumentChunk(document_id=doc, content="", embedding=emb, chunk_number=chunk_no, metadata={}))
        if len(pending) >= batch:
            flush()
    flush()
    return n
