#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the pieces of the reference that are importable HERE.

Run in the build container only (needs /root/reference and transformers); the GPU box never
runs this -- it reads the committed fixtures.

  sign_pack.npz / hamming.npz : produced by the reference's own code,
      /root/reference/core/utils/fast_ops.py (binary_quantize, binary_quantize_packed,
      hamming_distance, hamming_distance_batch -- Python fallback branch, bit-identical by
      construction to the Rust branch morphik_rust/src/binary_ops.rs).
  maxsim_float.npz : produced by transformers' ColPaliProcessor.score_retrieval
      (processing_colpali.py:208), the in-container twin of colpali_engine's
      score_multi_vector that the reference calls at fast_multivector_store.py:553-555.
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def load_ref_fast_ops():
    spec = importlib.util.spec_from_file_location("ref_fast_ops", os.path.join(REF, "core/utils/fast_ops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert not mod.HAS_RUST, "expected the Python fallback branch in this container"
    return mod


def gen_sign_pack(fo):
    rng = np.random.default_rng(20240607)
    cases = {}
    specials = np.array(
        [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-45, 1.0, -1.0, 0.1, -0.2, 0.3, 3.0, -2.0, 0.5, -0.5], np.float32
    )
    dims = [3, 8, 13, 16, 128, 128, 128, 130]
    rows = [2, 4, 5, 1, 7, 32, 64, 3]
    for ci, (d, n) in enumerate(zip(dims, rows)):
        x = rng.standard_normal((n, d)).astype(np.float32)
        # sprinkle special values
        idx = rng.integers(0, x.size, size=min(x.size // 3 + 1, 40))
        x.reshape(-1)[idx] = specials[rng.integers(0, specials.size, size=idx.size)]
        packed = fo.binary_quantize_packed(x)
        bools = fo.binary_quantize(x)
        cases[f"x{ci}"] = x
        cases[f"packed{ci}"] = np.frombuffer(b"".join(packed), np.uint8).reshape(n, (d + 7) // 8)
        cases[f"bools{ci}"] = np.array(bools, dtype=np.bool_)
    # the reference unit test's literal case (core/tests/unit/test_multivector.py:94-109)
    x = np.array([[0.1, -0.2, 0.3], [-0.1, 0.2, -0.3]], np.float32)
    cases["x_ref_test"] = x
    cases["bools_ref_test"] = np.array(fo.binary_quantize(x), dtype=np.bool_)
    cases["n_cases"] = np.array(len(dims))
    np.savez_compressed(os.path.join(OUT, "sign_pack.npz"), **cases)


def gen_hamming(fo):
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, size=(64, 16), dtype=np.uint8)
    b = rng.integers(0, 256, size=(64, 16), dtype=np.uint8)
    hd = np.array([fo.hamming_distance(bytes(x), bytes(y)) for x, y in zip(a, b)], np.int64)
    q = a[0]
    batch = np.array(fo.hamming_distance_batch(bytes(q), [bytes(y) for y in b]), np.int64)
    np.savez_compressed(os.path.join(OUT, "hamming.npz"), a=a, b=b, hd=hd, q=q, batch=batch)


def gen_maxsim_float():
    import torch
    from transformers import ColPaliProcessor

    rng = np.random.default_rng(99)
    out = {}
    ci = 0

    def unit(n, d=128):
        x = rng.standard_normal((n, d)).astype(np.float32)
        return x / np.linalg.norm(x, axis=-1, keepdims=True)

    # (n_qrows, list of page lengths, bf16-representable?)
    specs = [
        (32, [32] * 6, False),  # BASELINE cfg-1 shape: fixed 32 patches, fp32
        (17, [5, 64, 33, 1, 64, 20, 7], False),  # ragged -> zero-padding clamp inside one batch
        (32, [48, 48, 48], True),  # bf16-representable values, fixed
        (21, [100, 3, 57, 100], True),  # bf16-representable, ragged
        (1, [16, 2], False),
        (40, [12] * 2 + [11] * 128 + [9, 7, 8], False),  # > 128 pages: crosses a batch boundary (batches pad independently)
    ]
    for nq, lens, as_bf16 in specs:
        q = unit(nq)
        pages = [unit(n) for n in lens]
        if as_bf16:
            q = torch.from_numpy(q).bfloat16().float().numpy()
            pages = [torch.from_numpy(p).bfloat16().float().numpy() for p in pages]
        scores = ColPaliProcessor.score_retrieval(
            None, [torch.from_numpy(q)], [torch.from_numpy(p) for p in pages], batch_size=128, output_dtype=torch.float32
        )[0].numpy()
        pmax = max(lens)
        slab = np.zeros((len(lens), pmax, 128), np.float32)
        for i, p in enumerate(pages):
            slab[i, : p.shape[0]] = p
        # pad_to of each page = longest page of ITS batch of 128 (pad_sequence is per batch)
        pad_to = np.empty(len(lens), np.int32)
        for j in range(0, len(lens), 128):
            pad_to[j : j + 128] = max(lens[j : j + 128])
        out[f"q{ci}"] = q
        out[f"slab{ci}"] = slab
        out[f"n_rows{ci}"] = np.array(lens, np.int32)
        out[f"pad_to{ci}"] = pad_to
        out[f"scores{ci}"] = scores.astype(np.float32)
        ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "maxsim_float.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs /root/reference (build container only)")
    fo = load_ref_fast_ops()
    gen_sign_pack(fo)
    gen_hamming(fo)
    gen_maxsim_float()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
