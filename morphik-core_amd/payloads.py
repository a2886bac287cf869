"""Chunk payloads (page images / text) in the caller's blob storage -- the `.storage` half of the store boundary.

The reference never keeps chunk content next to the vectors: MultiVectorStore.store_embeddings uploads it to external
storage and keeps the storage KEY in its `content` column (core/vector_store/multi_vector_store.py:402-456, :666-699),
FastMultiVectorStore does the same per chunk (fast_multivector_store.py:455-462); hits download it again unless the
caller asked for `skip_image_content` and the chunk is an image, in which case the key itself is returned
(multi_vector_store.py:778-817, :866-919; fast_multivector_store.py:583-586).  In production `content` is a base64 page
image of hundreds of kilobytes, so a host dict of payloads would dwarf the bookkeeping.

`storage` is the caller's BaseStorage (core/storage/base_storage.py:5-90): upload_from_base64 / download_file /
delete_file coroutines.  No scoring here.
"""
from __future__ import annotations

import asyncio
import base64
import json
import logging
from typing import Any, Dict, Iterable, Optional, Tuple

logger = logging.getLogger(__name__)

MULTIVECTOR_CHUNKS_BUCKET = "multivector-chunks"  # core/vector_store/utils.py:11
DEFAULT_APP_ID = "default"  # fast_multivector_store.py:58, multi_vector_store.py (same constant)

_MIME_EXT = {
    "image/jpeg": ".jpg", "image/jpg": ".jpg", "image/png": ".png", "image/webp": ".webp", "image/gif": ".gif",
    "image/bmp": ".bmp", "image/tiff": ".tiff", "application/pdf": ".pdf", "text/plain": ".txt",
}
_MAGIC = ((b"\x89PNG\r\n\x1a\n", "image/png"), (b"\xff\xd8", "image/jpeg"), (b"GIF8", "image/gif"), (b"BM", "image/bmp"),
          (b"II*\x00", "image/tiff"), (b"MM\x00*", "image/tiff"), (b"%PDF", "application/pdf"))


def is_storage_key(value: Any) -> bool:
    """The reference's heuristic for "this content field holds a storage key, not content"
    (core/vector_store/utils.py:21-39): a short path with a slash, no whitespace, not a data URI / URL, none of the
    characters that appear in code or prose."""
    if not isinstance(value, str) or len(value) >= 500 or "/" not in value:
        return False
    if value.startswith("data:") or value.startswith("http"):
        return False
    return not any(ch.isspace() or ch in "();={}" for ch in value)


def sniff_mime(data: bytes) -> Optional[str]:
    for magic, mime in _MAGIC:
        if data.startswith(magic):
            return mime
    if data[:4] == b"RIFF" and b"WEBP" in data[:16]:
        return "image/webp"
    return None


def _decode_payload(content: str) -> Tuple[bytes, Optional[str]]:
    """content (data URI / bare base64 / text) -> (bytes to store, mime when known)."""
    if content.startswith("data:") and "," in content:
        header, b64 = content.split(",", 1)
        mime = header[5:].split(";", 1)[0] or None
        try:
            return base64.b64decode(b64), mime
        except Exception:  # noqa: BLE001
            return content.encode("utf-8"), mime
    try:
        raw = base64.b64decode(content, validate=True)
        return raw, sniff_mime(raw)
    except Exception:  # noqa: BLE001
        return content.encode("utf-8"), None


def extension_for(content: str, metadata: Dict[str, Any]) -> str:
    """Extension of the storage object (multi_vector_store.py:363-398): images by their MIME (data URI header, else magic
    bytes), everything else `.txt`."""
    if not metadata.get("is_image"):
        return ".txt"
    raw, mime = _decode_payload(content)
    mime = mime or sniff_mime(raw)
    return _MIME_EXT.get(mime or "", ".bin" if content.startswith("data:") else ".png")


def storage_key_for(app_id: Optional[str], document_id: str, chunk_number: int, extension: str) -> str:
    """f"{app_id}/{document_id}/{chunk_number}{ext}" (multi_vector_store.py:400-402)."""
    return f"{app_id or DEFAULT_APP_ID}/{document_id}/{chunk_number}{extension}"


def storage_backend_name(storage: Any) -> str:
    """core/vector_store/utils.py:63-70."""
    if storage is None:
        return "none"
    name = type(storage).__name__
    return {"S3Storage": "aws-s3", "LocalStorage": "local"}.get(name, name.lower())


class PayloadStore:
    """Uploads / fetches / deletes chunk content through the caller's storage object."""

    def __init__(self, storage: Any, bucket: str = MULTIVECTOR_CHUNKS_BUCKET, max_concurrency: int = 16):
        self.storage = storage
        self.bucket = bucket
        self._sems: Dict[int, asyncio.Semaphore] = {}  # one per event loop (a semaphore is bound to the loop it first waits in)
        self._max = int(max_concurrency)

    def _semaphore(self) -> asyncio.Semaphore:
        loop = asyncio.get_running_loop()
        sem = self._sems.get(id(loop))
        if sem is None:
            if len(self._sems) > 8:  # loops come and go (asyncio.run per call in scripts and tests)
                self._sems.clear()
            sem = self._sems[id(loop)] = asyncio.Semaphore(self._max)
        return sem

    async def put(self, content: str, document_id: str, chunk_number: int, metadata: Dict[str, Any],
                  app_id: Optional[str]) -> Tuple[Optional[str], int]:
        """-> (storage key, bytes stored), or (None, 0) when the upload failed (the caller keeps the content inline,
        as the reference falls back to the database column: multi_vector_store.py:672-677)."""
        try:
            ext = extension_for(content, metadata)
            key = storage_key_for(app_id, document_id, chunk_number, ext)
            if ext == ".txt":
                raw, ctype = content.encode("utf-8"), "text/plain"
                b64 = base64.b64encode(raw).decode("ascii")
            else:
                raw, mime = _decode_payload(content)
                ctype = mime or sniff_mime(raw)
                b64 = base64.b64encode(raw).decode("ascii")
            async with self._semaphore():
                await self.storage.upload_from_base64(content=b64, key=key, content_type=ctype, bucket=self.bucket)
            return key, len(raw)
        except Exception as e:  # noqa: BLE001
            logger.error(f"Failed to store content externally for {document_id}-{chunk_number}: {e}")
            return None, 0

    async def get(self, key: str, metadata: Dict[str, Any]) -> str:
        """Storage key -> content in the shape the reference returns (multi_vector_store.py:513-609): images as a data
        URI, text as str, undecodable bytes as base64.  On failure the key itself comes back (same fallback)."""
        try:
            async with self._semaphore():
                data = await self.storage.download_file(bucket=self.bucket, key=key)
            if not data:
                logger.error(f"No content downloaded for storage key: {key}")
                return key
            if metadata.get("is_image"):
                try:
                    text = data.decode("utf-8")
                    if text.strip().startswith("data:") and "," in text:
                        return text
                except Exception:  # noqa: BLE001
                    pass
                mime = metadata.get("mime_type") or sniff_mime(data) or "image/png"
                return f"data:{mime};base64," + base64.b64encode(data).decode("ascii")
            try:
                return data.decode("utf-8")
            except UnicodeDecodeError:
                return base64.b64encode(data).decode("ascii")
        except Exception as e:  # noqa: BLE001
            logger.error(f"Failed to retrieve content from storage key {key}: {e}")
            return key

    async def delete(self, keys: Iterable[str], document_id: str = "") -> None:
        """Best effort, like multi_vector_store.py:487-509."""
        keys = list(keys)
        if not keys or not hasattr(self.storage, "delete_file"):
            return
        results = await asyncio.gather(*[self.storage.delete_file(self.bucket, k) for k in keys], return_exceptions=True)
        for k, r in zip(keys, results):
            if isinstance(r, Exception):
                logger.warning("Failed to delete external storage key %s for document %s: %s", k, document_id, r)


_META_CACHE: Dict[str, Dict[str, Any]] = {}


def parse_metadata(meta_json: Optional[str]) -> Dict[str, Any]:
    """chunk_metadata column -> dict.  Every hit of every query passes through here: the decoded form of a JSON string is
    kept (bounded), and each caller gets its own top-level dict."""
    if not meta_json or meta_json == "{}":
        return {}
    m = _META_CACHE.get(meta_json)
    if m is None:
        try:
            m = json.loads(meta_json)
            if not isinstance(m, dict):
                m = {}
        except Exception:  # noqa: BLE001
            m = {}
        if len(_META_CACHE) >= 65536:
            _META_CACHE.clear()
        _META_CACHE[meta_json] = m
    return dict(m)


class LocalDirStorage:
    """The three coroutines of core/storage/base_storage.py the stores use, over a directory -- what the store-owner server
    (store_server.py --payload-dir) keeps chunk payloads in when no S3 / LocalStorage object of the host application is
    wired in: base64 page images of hundreds of kilobytes each belong on disk, not in the owner's RAM or in store.json."""

    def __init__(self, root: str):
        import os

        self.root = os.path.abspath(root)
        os.makedirs(self.root, exist_ok=True)

    def _path(self, bucket: str, key: str) -> str:
        import os

        p = os.path.abspath(os.path.join(self.root, bucket or "", key))
        if not p.startswith(self.root + os.sep):
            raise ValueError(f"storage key escapes the payload directory: {key!r}")
        return p

    async def upload_from_base64(self, content: str, key: str, content_type: Optional[str] = None, bucket: str = "") -> Tuple[str, str]:
        import os

        path = self._path(bucket, key)

        def write():
            os.makedirs(os.path.dirname(path), exist_ok=True)
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(base64.b64decode(content))
            os.replace(tmp, path)

        await asyncio.to_thread(write)
        return bucket, key

    async def download_file(self, bucket: str, key: str) -> bytes:
        path = self._path(bucket, key)

        def read():
            with open(path, "rb") as f:
                return f.read()

        return await asyncio.to_thread(read)

    async def delete_file(self, bucket: str, key: str) -> bool:
        import os

        try:
            os.remove(self._path(bucket, key))
            return True
        except FileNotFoundError:
            return False
