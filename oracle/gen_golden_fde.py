#!/usr/bin/env python3
"""Pin the FDE stage to the reference's own extension THE DAY it is importable.

The reference calls a C++ module that is absent from /root/reference (fde/pyproject.toml:70-72 builds it from sources that are not
vendored; `fixed-dimensional-encoding` 0.1.0, uv.lock:1085-1087):
    fde.FixedDimensionalEncodingConfig(dimension=128, num_repetitions=20, num_simhash_projections=5, projection_dimension=16,
                                       projection_type="AMS_SKETCH")                   fast_multivector_store.py:325-331
    fde.generate_document_encoding(np.array(chunk.embedding), self.fde_config)         :447-449
    fde.generate_query_encoding(query_embedding, self.fde_config)                      :521
Until then oracle/mv_oracle.c's orc_fde_encode restates the MUVERA construction with this repo's own Philox tables: "parity
unpinned" (the header of mv_oracle.c and DESIGN.md say so).

Where a module named `fde` or `fixed_dimensional_encoding` imports, this script writes tests/golden/fde.npz:
    pages   [n_pages, rows, 128] float32   seeded unit rows (the oracle generator: no reference needed to regenerate them)
    queries [n_q, q_rows, 128]   float32
    doc_fde [n_pages, 10240]     float32   generate_document_encoding of every page
    q_fde   [n_q, 10240]         float32   generate_query_encoding of every query
    config  json                            the five constructor arguments above + module name / version
tests/test_oracle_golden.py::test_fde_against_the_reference_extension_when_pinned loads it when present (skips otherwise) and reports
how this repo's encoder relates to it.  The two use different random tables unless the extension's seed / generator are matched, so the
test measures what survives that: the FDE dot products' agreement with exact MaxSim and with each other's ranking, not bit equality.

  python oracle/gen_golden_fde.py            (needs the extension; exits 3 with a message when it is absent)
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "fde.npz")
CONFIG = dict(dimension=128, num_repetitions=20, num_simhash_projections=5, projection_dimension=16, projection_type="AMS_SKETCH")
N_PAGES, ROWS, N_Q, Q_ROWS = 64, 96, 8, 32


def find_extension():
    for name in ("fde", "fixed_dimensional_encoding"):
        try:
            m = importlib.import_module(name)
        except Exception:  # noqa: BLE001
            continue
        if all(hasattr(m, a) for a in ("FixedDimensionalEncodingConfig", "generate_document_encoding", "generate_query_encoding")):
            return m
    return None


def inputs():
    from oracle import oracle as orc  # the seeded generator only

    pages = np.stack([orc.bf16_to_f32(orc.synth_rows(1234, p, 0, ROWS)) for p in range(N_PAGES)])
    queries = np.stack([orc.bf16_to_f32(orc.synth_rows(4321, q, 0, Q_ROWS)) for q in range(N_Q)])
    # planted near-duplicates so that the exact MaxSim ranking has structure: page 3q+1 carries query q's rows (plus noise)
    rng = np.random.default_rng(99)
    for q in range(N_Q):
        noisy = queries[q] + 0.1 * rng.standard_normal(queries[q].shape).astype(np.float32)
        pages[3 * q + 1, :Q_ROWS] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    return pages.astype(np.float32), queries.astype(np.float32)


def main():
    m = find_extension()
    if m is None:
        print("gen_golden_fde: neither `fde` nor `fixed_dimensional_encoding` imports here -- FDE parity stays unpinned", file=sys.stderr)
        return 3
    cfg = m.FixedDimensionalEncodingConfig(**CONFIG)
    pages, queries = inputs()
    doc_fde = np.stack([np.asarray(m.generate_document_encoding(np.array(p), cfg), np.float32).reshape(-1) for p in pages])
    q_fde = np.stack([np.asarray(m.generate_query_encoding(np.array(q), cfg), np.float32).reshape(-1) for q in queries])
    meta = dict(CONFIG, module=m.__name__, version=str(getattr(m, "__version__", "unknown")))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, pages=pages, queries=queries, doc_fde=doc_fde, q_fde=q_fde, config=np.array(json.dumps(meta)))
    print(f"wrote {OUT}: doc_fde {doc_fde.shape}, q_fde {q_fde.shape}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
