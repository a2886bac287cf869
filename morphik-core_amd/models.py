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
