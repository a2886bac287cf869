"""morphik_core_amd -- MI355X-native ColPali late-interaction retrieval path for Morphik.

Only what the hot path needs:
  csrc/          HIP kernels (gfx950) + the C ABI (include/mvmaxsim.h) -> libmvmaxsim.so
  _lib.py        ctypes binding of the C ABI (fails loudly when the library is missing)
  index.py       MvIndex: one GPU's shard of the page corpus
  store.py       MI355XMultiVectorStore / MI355XFastMultiVectorStore: BaseVectorStore plugins (+ request coalescing)
  embedding.py   MI355XColpaliEmbeddingModel: BaseEmbeddingModel plugin (PyTorch-ROCm encoder, device-resident ingest)
  embed_server.py  the reference's /embeddings npz protocol served from an MI355X (one server per GPU)
  formats.py     importers for the reference's formats: .npy page tree, .npz wire format, BIT(128)[] rows
  sharded.py     row-sharded corpus over N ranks, RCCL all-gather of per-shard top-k
  synth.py       synthetic corpus / planted-neighbour helpers shared by tests and bench
"""
from . import _lib  # noqa: F401
from ._lib import MvError, build_library, library_path  # noqa: F401
from .index import FdeConfig, MvIndex, QueryStats  # noqa: F401

__all__ = ["MvIndex", "FdeConfig", "QueryStats", "MvError", "build_library", "library_path"]
