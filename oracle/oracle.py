"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see mv_oracle.c's header).

Two layers:
  * ctypes bindings of oracle/libmvoracle.so (the scalar C restatement; exact, slow);
  * numpy/torch restatements of the same formulas for sizes the scalar code cannot finish in
    seconds, and for the bench's cpu_baseline leg (`maxsim_float_np`, `maxsim_float_torch`).

Reference citations live next to each function in mv_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmvoracle.so")


def build(force: bool = False) -> str:
    """Compile oracle/mv_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "mv_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmvoracle.so"])
    return _LIB_PATH


class FdeConfig(C.Structure):
    """fast_multivector_store.py:325-331 (FixedDimensionalEncodingConfig)."""

    _fields_ = [
        ("dimension", C.c_int32),
        ("num_repetitions", C.c_int32),
        ("num_simhash_projections", C.c_int32),
        ("projection_dimension", C.c_int32),
        ("seed", C.c_uint64),
    ]

    @classmethod
    def reference_default(cls, seed: int = 1) -> "FdeConfig":
        return cls(128, 20, 5, 16, seed)

    @property
    def output_dim(self) -> int:
        return self.num_repetitions * (1 << self.num_simhash_projections) * self.projection_dimension


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
        L.orc_f32_to_bf16_n.argtypes = [vp, i64, vp]
        L.orc_bf16_to_f32_n.argtypes = [vp, i64, vp]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_synth_rows.argtypes = [u64, u64, i32, i32, i32, vp]
        L.orc_sign_pack.argtypes = [vp, i64, i32, vp]
        L.orc_hamming.argtypes = [vp, vp, i64]
        L.orc_hamming.restype = i64
        L.orc_maxsim_binary.argtypes = [vp, i32, vp, i32, i32]
        L.orc_maxsim_binary.restype = C.c_double
        L.orc_maxsim_f32.argtypes = [vp, i32, vp, i32, i32, i32]
        L.orc_maxsim_f32.restype = f32
        L.orc_maxsim_bf16.argtypes = [vp, i32, vp, i32, i32, i32]
        L.orc_maxsim_bf16.restype = f32
        L.orc_maxsim_bf16_slab.argtypes = [vp, i32, vp, vp, i64, i32, i32, i32, vp]
        L.orc_topk.argtypes = [vp, vp, i64, i64, vp, vp]
        L.orc_topk.restype = i64
        L.orc_fde_output_dim.argtypes = [C.POINTER(FdeConfig)]
        L.orc_fde_output_dim.restype = i64
        L.orc_fde_matrices.argtypes = [C.POINTER(FdeConfig), vp, vp, vp]
        L.orc_fde_encode.argtypes = [C.POINTER(FdeConfig), vp, i32, i32, vp]
        L.orc_fde_partitions.argtypes = [C.POINTER(FdeConfig), vp, i32, vp]
        L.orc_fde_coarse_scores.argtypes = [vp, vp, i64, i64, i32, vp]
        L.orc_e4m3_encode.argtypes = [f32]
        L.orc_e4m3_encode.restype = C.c_uint8
        L.orc_e4m3_decode.argtypes = [C.c_uint8]
        L.orc_e4m3_decode.restype = f32
        L.orc_quantize_page_fp8.argtypes = [vp, i32, i32, vp, vp]
        L.orc_fp8_query_prep.argtypes = [vp, i32, vp, vp, vp]
        L.orc_quantize_fde_fp4.argtypes = [vp, i32, vp, vp]
        L.orc_fp4_decode.argtypes = [C.c_uint32]
        L.orc_fp4_decode.restype = f32
        L.orc_fp4_encode.argtypes = [f32]
        L.orc_fp4_encode.restype = C.c_uint32
        L.orc_maxsim_fp8.argtypes = [vp, vp, vp, i32, vp, i32, f32, i32]
        L.orc_maxsim_fp8.restype = f32
        _lib = L
    return _lib


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------- bf16 helpers
def f32_to_bf16(x) -> np.ndarray:
    """RNE fp32 -> bf16 (as uint16), via the C oracle."""
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_f32_to_bf16_n(_p(x), x.size, _p(out))
    return out


def bf16_to_f32(u) -> np.ndarray:
    u = _c(u, np.uint16)
    return (u.astype(np.uint32) << 16).view(np.float32).reshape(u.shape)


def f32_to_bf16_np(x) -> np.ndarray:
    """Vectorised numpy RNE fp32 -> bf16 (checked against the C routine in the tests)."""
    u = _c(x, np.float32).view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


# --------------------------------------------------------------------------- generator
def philox(ctr: Sequence[int], key: Sequence[int]) -> np.ndarray:
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    out = np.empty(4, np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def synth_rows(seed: int, unit: int, row0: int, n_rows: int, dim: int = 128) -> np.ndarray:
    """bf16 (uint16) rows [n_rows, dim] of synthetic unit `unit` (a page id or a query id)."""
    out = np.empty((n_rows, dim), np.uint16)
    lib().orc_synth_rows(seed, unit, row0, n_rows, dim, _p(out))
    return out


def synth_pages(seed: int, page0: int, n_pages: int, n_rows: int, dim: int = 128) -> np.ndarray:
    out = np.empty((n_pages, n_rows, dim), np.uint16)
    for i in range(n_pages):
        lib().orc_synth_rows(seed, page0 + i, 0, n_rows, dim, C.c_void_p(out[i].ctypes.data))
    return out


# --------------------------------------------------------------------------- sign / hamming
def sign_pack(x) -> np.ndarray:
    x = _c(x, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    n, d = x.shape
    out = np.empty((n, (d + 7) // 8), np.uint8)
    lib().orc_sign_pack(_p(x), n, d, _p(out))
    return out


def sign_pack_np(x) -> np.ndarray:
    x = _c(x, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    return np.packbits(x > 0, axis=-1, bitorder="big")


def hamming(a, b) -> int:
    a = _c(np.frombuffer(bytes(a), np.uint8) if isinstance(a, (bytes, bytearray)) else a, np.uint8)
    b = _c(np.frombuffer(bytes(b), np.uint8) if isinstance(b, (bytes, bytearray)) else b, np.uint8)
    assert a.size == b.size
    return int(lib().orc_hamming(_p(a), _p(b), a.size))


def maxsim_binary(doc_bits, q_bits) -> float:
    """SQL max_sim on packed rows: doc_bits [P, nbytes] u8, q_bits [Q, nbytes] u8."""
    d = _c(doc_bits, np.uint8)
    q = _c(q_bits, np.uint8)
    nbytes = q.shape[-1] if q.ndim == 2 else (d.shape[-1] if d.ndim == 2 else 16)
    nd = d.shape[0] if d.ndim == 2 else 0
    nq = q.shape[0] if q.ndim == 2 else 0
    return float(lib().orc_maxsim_binary(_p(d), nd, _p(q), nq, nbytes))


def maxsim_binary_np(doc_bits, q_bits) -> np.ndarray:
    """Vectorised: doc_bits [N, P, nbytes], q_bits [Q, nbytes] -> float64 [N]."""
    d = _c(doc_bits, np.uint8)
    q = _c(q_bits, np.uint8)
    nbits = q.shape[-1] * 8
    x = d[:, None, :, :] ^ q[None, :, None, :]  # N,Q,P,B
    hd = np.unpackbits(x, axis=-1).sum(-1)  # N,Q,P
    return (1.0 - hd.min(-1).astype(np.float64) / max(nbits, 1)).sum(-1)


def maxsim_binary_popcount_np(doc_bits, q_bits, chunk: int = 64) -> np.ndarray:
    """Faster CPU formulation of the same SQL max_sim for timing: 64-bit popcounts (np.bitwise_count) in page chunks.
    doc_bits [N, P, 16] uint8, q_bits [Q, 16] uint8 -> float64 [N].  Equal to maxsim_binary_np bit for bit."""
    d = _c(doc_bits, np.uint8)
    q = _c(q_bits, np.uint8)
    assert d.shape[-1] == 16 and q.shape[-1] == 16
    d64 = d.view(np.uint64)  # [N, P, 2]
    q64 = q.view(np.uint64)  # [Q, 2]
    out = np.empty(d.shape[0], np.float64)
    for s0 in range(0, d.shape[0], chunk):
        x = d64[s0 : s0 + chunk, None, :, :] ^ q64[None, :, None, :]  # n, Q, P, 2
        hd = np.bitwise_count(x).sum(-1, dtype=np.uint16)              # n, Q, P
        out[s0 : s0 + chunk] = (1.0 - hd.min(-1).astype(np.float64) / 128.0).sum(-1)
    return out


# --------------------------------------------------------------------------- float MaxSim
def maxsim_f32(q, page, pad_to: int = 0) -> float:
    q = _c(q, np.float32)
    p = _c(page, np.float32)
    d = q.shape[-1]
    return float(lib().orc_maxsim_f32(_p(q), q.shape[0], _p(p), p.shape[0] if p.size else 0, d, pad_to))


def maxsim_bf16(q_u16, page_u16, pad_to: int = 0) -> float:
    q = _c(q_u16, np.uint16)
    p = _c(page_u16, np.uint16)
    d = q.shape[-1]
    return float(lib().orc_maxsim_bf16(_p(q), q.shape[0], _p(p), p.shape[0] if p.size else 0, d, pad_to))


def maxsim_bf16_slab(q_u16, slab_u16, n_rows: Optional[np.ndarray] = None, pad_to: int = 0) -> np.ndarray:
    q = _c(q_u16, np.uint16)
    s = _c(slab_u16, np.uint16)
    n, stride, d = s.shape
    out = np.empty(n, np.float32)
    nr = None if n_rows is None else _c(n_rows, np.int32)
    lib().orc_maxsim_bf16_slab(_p(q), q.shape[0], _p(s), None if nr is None else _p(nr), n, stride, d, pad_to, _p(out))
    return out


def maxsim_float_np(q_f32, pages_f32, n_rows: Optional[np.ndarray] = None, pad_to: int = 0, chunk: int = 256) -> np.ndarray:
    """numpy restatement for big inputs: (pages.reshape(-1,D) @ q.T) -> max over patches -> sum.

    pages_f32 [N, P, D] fp32 (fixed stride; rows >= n_rows[i] are ignored, or count as zero rows
    when pad_to > n_rows[i] -- the zero-padding semantics of score_multi_vector)."""
    q = _c(q_f32, np.float32)
    pages = np.asarray(pages_f32)
    n, p, d = pages.shape
    out = np.empty(n, np.float32)
    for s in range(0, n, chunk):
        blk = np.ascontiguousarray(pages[s : s + chunk], dtype=np.float32)
        sim = (blk.reshape(-1, d) @ q.T).reshape(blk.shape[0], p, q.shape[0])
        if n_rows is not None:
            nr = np.asarray(n_rows[s : s + chunk])
            mask = np.arange(p)[None, :] >= nr[:, None]
            sim = np.where(mask[:, :, None], -np.inf, sim)
            mx = sim.max(1)
            clamp = (pad_to > nr)[:, None]
            mx = np.where(clamp, np.maximum(mx, 0.0), mx)
            mx = np.where(np.isneginf(mx), 0.0, mx)
        else:
            mx = sim.max(1)
        out[s : s + chunk] = mx.astype(np.float32).sum(1, dtype=np.float32)
    return out


def maxsim_float_torch(q_f32, pages_f32, batch: int = 128):
    """The reference's own formulation (score_multi_vector / score_retrieval), one query,
    fixed-stride pages: einsum('bnd,csd->bcns').max(3).sum(2) over page batches of 128."""
    import torch

    q = torch.as_tensor(np.asarray(q_f32, np.float32))[None]
    pages = torch.as_tensor(np.asarray(pages_f32, np.float32))
    outs = []
    for j in range(0, pages.shape[0], batch):
        outs.append(torch.einsum("bnd,csd->bcns", q, pages[j : j + batch]).max(dim=3)[0].sum(dim=2))
    return torch.cat(outs, dim=1)[0].numpy()


# --------------------------------------------------------------------------- top-k
def topk(scores, k: int, ids=None) -> Tuple[np.ndarray, np.ndarray]:
    s = _c(scores, np.float32)
    i = None if ids is None else _c(ids, np.int64)
    k = int(max(0, k))
    os_ = np.empty(max(k, 1), np.float32)
    oi = np.empty(max(k, 1), np.int64)
    n = lib().orc_topk(_p(s), None if i is None else _p(i), s.size, k, _p(os_), _p(oi))
    return os_[:n].copy(), oi[:n].copy()


# --------------------------------------------------------------------------- FDE
def fde_matrices(cfg: FdeConfig):
    R, D, NS = cfg.num_repetitions, cfg.dimension, cfg.num_simhash_projections
    G = np.empty((R, D, NS), np.float32)
    H = np.empty((R, D), np.int32)
    S = np.empty((R, D), np.float32)
    lib().orc_fde_matrices(C.byref(cfg), _p(G), _p(H), _p(S))
    return G, H, S


def fde_encode(cfg: FdeConfig, x, is_query: bool) -> np.ndarray:
    x = _c(x, np.float32)
    out = np.empty(cfg.output_dim, np.float32)
    lib().orc_fde_encode(C.byref(cfg), _p(x), x.shape[0], 1 if is_query else 0, _p(out))
    return out


def fde_partitions(cfg: FdeConfig, x) -> np.ndarray:
    x = _c(x, np.float32)
    out = np.empty((cfg.num_repetitions, x.shape[0]), np.int32)
    lib().orc_fde_partitions(C.byref(cfg), _p(x), x.shape[0], _p(out))
    return out


def fde_coarse_scores(q_fde, d_slab_u16, use_cosine: bool = True) -> np.ndarray:
    q = _c(q_fde, np.float32)
    d = _c(d_slab_u16, np.uint16)
    out = np.empty(d.shape[0], np.float32)
    lib().orc_fde_coarse_scores(_p(q), _p(d), d.shape[0], d.shape[1], 1 if use_cosine else 0, _p(out))
    return out


# --------------------------------------------------------------------------- fp8 (e4m3fn) path
def e4m3_encode(x) -> np.ndarray:
    x = _c(x, np.float32)
    return np.array([lib().orc_e4m3_encode(float(v)) for v in x.ravel()], np.uint8).reshape(x.shape)


def e4m3_decode(c) -> np.ndarray:
    c = _c(c, np.uint8)
    lut = np.array([lib().orc_e4m3_decode(i) for i in range(256)], np.float32)
    return lut[c]


def quantize_page_fp8(rows_bf16, stride: int) -> Tuple[np.ndarray, float]:
    """bf16 rows [n,128] -> (codes [stride,128] uint8, inverse scale 2^-e)."""
    r = _c(rows_bf16, np.uint16).reshape(-1, 128)
    codes = np.empty((stride, 128), np.uint8)
    inv = C.c_float()
    lib().orc_quantize_page_fp8(_p(r), r.shape[0], stride, _p(codes), C.byref(inv))
    return codes, float(inv.value)


def quantize_fde_fp4(rows_bf16) -> Tuple[np.ndarray, np.ndarray]:
    """bf16 FDE rows [n, od] (uint16) -> (e2m1 codes [n, od / 2] uint8: element 2i in the low nibble, scales [n] float32)."""
    r = _c(rows_bf16, np.uint16)
    r = r.reshape(-1, r.shape[-1])
    codes = np.empty((r.shape[0], r.shape[1] // 2), np.uint8)
    sc = np.empty(r.shape[0], np.float32)
    one = C.c_float()
    for i in range(r.shape[0]):
        row = np.ascontiguousarray(r[i])
        out = np.empty(r.shape[1] // 2, np.uint8)
        lib().orc_quantize_fde_fp4(_p(row), r.shape[1], _p(out), C.byref(one))
        codes[i] = out
        sc[i] = one.value
    return codes, sc


def fp4_decode(codes) -> np.ndarray:
    """e2m1 code bytes [..., m] -> float32 [..., 2 m] (low nibble first)."""
    c = _c(codes, np.uint8)
    lut = np.array([lib().orc_fp4_decode(i) for i in range(16)], np.float32)
    out = np.empty(c.shape[:-1] + (c.shape[-1] * 2,), np.float32)
    out[..., 0::2] = lut[c & 15]
    out[..., 1::2] = lut[c >> 4]
    return out


def fp8_query_prep(q_f32):
    q = _c(q_f32, np.float32).reshape(-1, 128)
    hi = np.empty(q.shape, np.uint8)
    lo = np.empty(q.shape, np.uint8)
    fac = np.empty(q.shape[0], np.float32)
    lib().orc_fp8_query_prep(_p(q), q.shape[0], _p(hi), _p(lo), _p(fac))
    return hi, lo, fac


def maxsim_fp8(q_f32, codes, n_rows: int, inv_scale: float, pad_to: int = 0) -> float:
    hi, lo, fac = fp8_query_prep(q_f32)
    c = _c(codes, np.uint8)
    return float(lib().orc_maxsim_fp8(_p(hi), _p(lo), _p(fac), hi.shape[0], _p(c), int(n_rows), float(inv_scale), int(pad_to)))


def maxsim_fp8_np(q_f32, codes_pages, inv_scales, n_rows=None) -> np.ndarray:
    """Vectorised: codes_pages [N,P,128] uint8, inv_scales [N] -> scores [N] (fp64 accumulation, same operands)."""
    hi, lo, fac = fp8_query_prep(q_f32)
    a = e4m3_decode(hi).astype(np.float64) + e4m3_decode(lo).astype(np.float64) * 0.0625
    lut = np.array([lib().orc_e4m3_decode(i) for i in range(256)], np.float64)
    out = np.empty(len(codes_pages), np.float64)
    for i, pg in enumerate(codes_pages):
        nr = pg.shape[0] if n_rows is None else int(n_rows[i])
        if nr == 0:
            out[i] = 0.0
            continue
        s = lut[pg[:nr]] @ a.T  # [nr, Q]
        out[i] = float((s.max(axis=0) * fac.astype(np.float64)).sum() * float(inv_scales[i]))
    return out.astype(np.float32)
