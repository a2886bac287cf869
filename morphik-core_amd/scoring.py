"""score_multi_vector on the MI355X -- the function FastMultiVectorStore calls for its rerank.

    scores = self.processor.score_multi_vector([query], multivectors, device=self.device)[0]
                                                    (core/vector_store/fast_multivector_store.py:553-555)

colpali_engine's processor method (v0.3.13, not vendored; `transformers.ColPaliProcessor.score_retrieval` is the same
code): queries and passages are scored in batches of `batch_size` (128), each batch zero-padded to ITS longest member by
pad_sequence, `einsum("bnd,csd->bcns")`, max over passage tokens, sum over query tokens -> a (n_queries, n_passages)
float32 tensor.  Here the passages are appended to a scratch mv_index once (one H2D copy), every query is ONE call of
mv_score_candidates with the reference's per-batch pad rule computed on the device (pad_to = -1), and the scoring is the
fused HIP MaxSim kernel (no (Q x P) intermediate).  Inside the stores the passages already live in the slab; this
function exists for callers that hold loose multi-vectors -- and so that the parity tests read like the reference's call.

Precision.  The reference scores fp32 tensors in fp32 (its pages are fp32 `.npy` files: fast_multivector_store.py:676-681,
:736, :774).  float32 passages are therefore kept as split-bf16 pairs hi + lo (MV_WITH_FLOAT_LO: 16 significant bits, the same
4 bytes per element) and float32 queries are split the same way; the kernel accumulates qhi.phi + qlo.phi + qhi.plo in fp32
on the bf16 MFMA before the max.  The result equals the reference's fp32 score to ~1e-6 relative (summation order + the
dropped 2^-18 lo.lo term); bf16 inputs take the one-term kernel and are exact products summed in fp32, as before.
"""
from __future__ import annotations

from typing import Any, Optional, Sequence

import numpy as np


def _bf16_exact(a: np.ndarray) -> bool:
    """every float32 value of `a` is a bf16 value (its low 16 bits are zero)"""
    return not np.any(np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF))


def score_multi_vector(qs: Sequence[Any], ps: Sequence[Any], batch_size: int = 128, device: int = 0, index: Optional[Any] = None) -> np.ndarray:
    """qs: queries [(n_q_i, 128)], ps: passages [(n_p_j, 128)] (ndarray / torch tensor / list; fp32 or bf16).
    -> float32 ndarray [len(qs), len(ps)] (wrap it with torch.from_numpy for torch.topk, as the reference does).
    batch_size must be 128 (the reference's default, the only value its store uses): the per-batch pad rule is built
    into the library's candidate scorer."""
    from .index import MvIndex, as_rows

    if batch_size != 128:
        raise ValueError("score_multi_vector: batch_size is fixed at 128 (the reference store's value)")
    if len(qs) == 0:
        raise ValueError("No queries provided")  # colpali_engine raises the same
    if len(ps) == 0:
        raise ValueError("No passages provided")
    from ._lib import MV_F32

    rows = [as_rows(p) for p in ps]
    pages = [a for a, _ in rows]
    longest = max(p.shape[0] for p in pages)
    own = index is None
    # fp32 passages that are not bf16-representable keep their low bits in the lo slab (bf16 passages: no lo slab, same bits as ever)
    need_lo = own and any(c == MV_F32 and not _bf16_exact(a) for a, c in rows)
    ix = index or MvIndex(capacity_pages=len(pages), stride_rows=max(16, ((longest + 15) // 16) * 16), device=device, with_float_lo=need_lo)
    try:
        first = ix.add(pages) if own else 0
        cand = np.arange(first, first + len(pages), dtype=np.int32)
        out = np.empty((len(qs), len(pages)), np.float32)
        for i, q in enumerate(qs):
            out[i] = ix.score_candidates(q, cand, pad_to=-1)
        return out
    finally:
        if own:
            ix.close()
