"""A module with the API of the reference's `fde` extension (fast_multivector_store.py:27, :325-331, :447-449, :521), for tests of the
bring-your-own-FDE plumbing: the encoder behind it is the oracle's restatement.  Test infrastructure only."""
import numpy as np

from oracle import oracle as orc


class FixedDimensionalEncodingConfig:
    def __init__(self, **kw):
        self.kw = kw


def generate_document_encoding(emb, cfg):
    return orc.fde_encode(orc.FdeConfig.reference_default(), np.asarray(emb, np.float32), False)


def generate_query_encoding(q, cfg):
    return orc.fde_encode(orc.FdeConfig.reference_default(), np.asarray(q, np.float32), True)
