"""morphik_core_amd -- MI355X-native ColPali late-interaction retrieval path for Morphik.

Only what the hot path needs:
  csrc/          HIP kernels (gfx950) + the C ABI (include/mvmaxsim.h) -> libmvmaxsim.so
  _lib.py        ctypes binding of the C ABI (fails loudly when the library is missing)
  index.py       MvIndex: one GPU's shard of the page corpus; ShardComm: mv_comm over R shards in one process
  shard_index.py ShardedIndex: R shards + communicator behind the one-index interface the stores use
  store.py       MI355X{,Fast,Sharded,ShardedFast}MultiVectorStore: BaseVectorStore plugins (+ request coalescing)
  scoring.py     score_multi_vector(qs, ps): the reference's rerank call on loose multi-vectors
  payloads.py    chunk content in the caller's blob storage (keys in RAM, skip_image_content)
  store_server.py  one HBM slab, many processes: owner server + MI355XRemoteMultiVectorStore
  embedding.py   MI355XColpaliEmbeddingModel: BaseEmbeddingModel plugin (PyTorch-ROCm encoder, device-resident ingest)
  colqwen_embedding.py  MI355XColQwen2EmbeddingModel: the reference's encoder family (dynamic patch counts -> ragged pages)
  embed_server.py  the reference's /embeddings npz protocol served from an MI355X (one server per GPU)
  formats.py     importers for the reference's formats: .npy page tree, .npz wire format, BIT(128)[] rows
  sharded.py     row-sharded corpus over N ranks (one process per GPU), RCCL all-gather of per-shard top-k
  synth.py       synthetic corpus / planted-neighbour / hard-negative helpers shared by tests and bench
"""
from . import _lib  # noqa: F401
from ._lib import MvError, build_library, library_path  # noqa: F401
from .index import FdeConfig, MvIndex, QueryStats  # noqa: F401

__all__ = ["MvIndex", "FdeConfig", "QueryStats", "MvError", "build_library", "library_path"]
