"""Boundary record types.  Inside a Morphik checkout the real classes are used
(core/models/chunk.py:9-38); elsewhere these field-for-field mirrors stand in."""
from __future__ import annotations

from typing import Any, Dict, List, Union

try:  # running inside morphik-core
    from core.models.chunk import Chunk, DocumentChunk  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001
    import numpy as np
    from pydantic import BaseModel, Field

    Embedding = Union[List[float], List[List[float]], np.ndarray]

    class DocumentChunk(BaseModel):
        """Represents a chunk stored in VectorStore (core/models/chunk.py:9-21)."""

        document_id: str
        content: str
        embedding: Any
        chunk_number: int
        metadata: Dict[str, Any] = Field(default_factory=dict)
        score: float = 0.0

        model_config = {"arbitrary_types_allowed": True}

    class Chunk(BaseModel):
        """core/models/chunk.py:24-38."""

        content: str
        metadata: Dict[str, Any] = Field(default_factory=dict)

        model_config = {"arbitrary_types_allowed": True}

        def to_document_chunk(self, document_id: str, chunk_number: int, embedding: Any) -> DocumentChunk:
            return DocumentChunk(document_id=document_id, content=self.content, embedding=embedding,
                                 chunk_number=chunk_number, metadata=self.metadata)


def hit_chunk_builder():
    """-> f(document_id, chunk_number, content, metadata, score): the DocumentChunk of one search hit (embedding = [], as the reference
    returns them: multi_vector_store.py:811).

    query_similar hands back values that were validated when they were stored; validating 10 x 6 fields again per request costs
    ~1.5 us per chunk -- 40 % of the store's per-request Python, which bounds the request rate at the plugin boundary once 32 requests
    share a 0.9 ms slab pass.  The fast builder fills the model's fields the way pydantic's own model_construct does.  It is used only if,
    on THIS pydantic version and THIS DocumentChunk class, its objects are indistinguishable from validated ones (equality, dump, field
    set, copy-with-update); otherwise -- or with MV_FAST_HIT_CHUNKS=0 -- the validating constructor is."""
    import os

    def slow(document_id, chunk_number, content, metadata, score):
        return DocumentChunk(document_id=document_id, chunk_number=chunk_number, content=content, embedding=[], metadata=metadata, score=score)

    if os.environ.get("MV_FAST_HIT_CHUNKS", "1") in ("0", "false", "no"):
        return slow
    try:
        cls = DocumentChunk
        fields = set(cls.model_fields)
        if fields != {"document_id", "content", "embedding", "chunk_number", "metadata", "score"}:
            return slow
        new, setattr_ = cls.__new__, object.__setattr__
        fset = frozenset(fields)

        def fast(document_id, chunk_number, content, metadata, score):
            o = new(cls)
            setattr_(o, "__dict__", {"document_id": document_id, "content": content, "embedding": [], "chunk_number": chunk_number,
                                     "metadata": dict(metadata), "score": score})  # own dict: the caller may edit its hits
            setattr_(o, "__pydantic_fields_set__", set(fset))
            setattr_(o, "__pydantic_extra__", None)
            setattr_(o, "__pydantic_private__", None)
            return o

        args = ("doc-\u00e9", 7, "content", {"is_image": True, "n": [1, 2]}, 0.25)
        a, b = fast(*args), slow(*args)
        ok = (a == b and b == a and a.model_dump() == b.model_dump() and a.model_fields_set == b.model_fields_set
              and a.model_copy(update={"score": 1.0}) == b.model_copy(update={"score": 1.0}) and a.model_dump_json() == b.model_dump_json()
              and type(a.score) is float and a.metadata is not args[3] and repr(a) == repr(b))
        a.score = 0.5
        b.score = 0.5
        return fast if ok and a == b else slow
    except Exception:  # noqa: BLE001 -- any surprise in the model class: validate as before
        return slow


try:
    from core.vector_store.base_vector_store import BaseVectorStore  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001
    from abc import ABC

    class BaseVectorStore(ABC):  # core/vector_store/base_vector_store.py:7-65 (same four coroutines)
        pass

try:
    from core.embedding.base_embedding_model import BaseEmbeddingModel  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001
    from abc import ABC as _ABC

    class BaseEmbeddingModel(_ABC):  # core/embedding/base_embedding_model.py:7-16
        pass


def build_store_metrics(*, chunk_payload_backend: str, multivector_backend: str, vector_store_backend: str,
                        chunk_payload_upload_s: float = 0.0, chunk_payload_objects: int = 0, multivector_upload_s: float = 0.0,
                        multivector_objects: int = 0, vector_store_write_s: float = 0.0, vector_store_rows: int = 0,
                        cache_write_s: float = 0.0, cache_write_objects: int = 0, chunk_payload_bytes: int = 0,
                        multivector_bytes: int = 0) -> Dict[str, Any]:
    """Same 13 keys as core/vector_store/utils.py:73-103 (the ingestion worker aggregates them)."""
    return {
        "chunk_payload_upload_s": chunk_payload_upload_s,
        "chunk_payload_objects": chunk_payload_objects,
        "chunk_payload_bytes": chunk_payload_bytes,
        "chunk_payload_backend": chunk_payload_backend,
        "multivector_upload_s": multivector_upload_s,
        "multivector_objects": multivector_objects,
        "multivector_bytes": multivector_bytes,
        "multivector_backend": multivector_backend,
        "vector_store_write_s": vector_store_write_s,
        "vector_store_backend": vector_store_backend,
        "vector_store_rows": vector_store_rows,
        "cache_write_s": cache_write_s,
        "cache_write_objects": cache_write_objects,
    }
