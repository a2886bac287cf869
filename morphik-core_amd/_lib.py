"""ctypes binding of libmvmaxsim.so (include/mvmaxsim.h).

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVMAXSIM_LIB: load another build of the same library (the sanitizer builds of csrc/Makefile: libmvmaxsim_tsan.so / _asan.so)
_LIB = os.environ.get("MVMAXSIM_LIB") or os.path.join(_HERE, "libmvmaxsim.so")
_CSRC = os.path.join(_HERE, "csrc")


class MvError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libmvmaxsim error {status}: {message}")
        self.status = status


def library_path() -> str:
    return _LIB


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "-s", "-j8"])
    return _LIB


class FdeConfigC(C.Structure):
    _fields_ = [
        ("dimension", C.c_int32),
        ("num_repetitions", C.c_int32),
        ("num_simhash_projections", C.c_int32),
        ("projection_dimension", C.c_int32),
        ("seed", C.c_uint64),
    ]


class ConfigC(C.Structure):
    _fields_ = [
        ("dim", C.c_int32),
        ("stride_rows", C.c_int32),
        ("capacity_pages", C.c_int64),
        ("device", C.c_int32),
        ("flags", C.c_int32),
        ("id_base", C.c_int64),
        ("fde", FdeConfigC),
        ("capacity_rows", C.c_int64),  # MV_LAYOUT_PACKED: rows of the row-indexed slabs (0 = capacity_pages * stride_rows)
    ]


class QueryStatsC(C.Structure):
    _fields_ = [
        ("score_kernel_ms", C.c_float),
        ("topk_ms", C.c_float),
        ("total_device_ms", C.c_float),
        ("score_launches", C.c_int32),
        ("reserved", C.c_int32),
        ("pages_scored", C.c_int64),
        ("bytes_scanned", C.c_int64),
        ("encode_ms", C.c_float),
        ("coarse_ms", C.c_float),
        ("select_ms", C.c_float),
        ("rerank_ms", C.c_float),
    ]


MV_F32, MV_BF16 = 0, 1
MV_MODE_FLOAT, MV_MODE_BINARY, MV_MODE_FDE_THEN_FLOAT, MV_MODE_FDE_ONLY, MV_MODE_FLOAT_FP8, MV_MODE_FP8_THEN_FLOAT = 0, 1, 2, 3, 4, 5
MV_WITH_FLOAT, MV_WITH_BINARY, MV_WITH_FDE, MV_WITH_FP8, MV_WITH_HOST_EXACT, MV_WITH_EXACT_SPLIT = 1, 2, 4, 8, 16, 32
MV_WITH_FLOAT_LO = 64
MV_LAYOUT_PACKED = 128
MV_WITH_FDE_E4M3 = 256
MV_WITH_FDE_FP4 = 512
MV_OPT_MAXSIM_VARIANT, MV_OPT_FDE_COARSE_N, MV_OPT_FDE_COSINE, MV_OPT_PAD_SEMANTICS, MV_OPT_BINARY_VARIANT, MV_OPT_FDE_SCAN_VARIANT = 1, 2, 3, 4, 5, 6
MV_OPT_BATCH_VARIANT = 7
MV_OPT_FDE_ENCODE_VARIANT = 8
MV_OPT_FILTER_COMPACT_PCT = 9
MV_OPT_LONG_QUERY_VARIANT = 10
MV_OPT_FDE_QUERY_ENCODE_VARIANT = 11
MV_OPT_FDE_BATCH_VARIANT = 12
MV_OPT_RERANK_N = 13
MV_OPT_EXACT_TIER = 14
MV_OPT_FLOAT_LO_SCAN = 15
MV_OPT_FDE_COARSE_SLAB = 16
MV_CAL_READ_NT, MV_CAL_MFMA_BF16, MV_CAL_READ_LDSDMA, MV_CAL_MFMA_BF16_32X32 = 1, 2, 3, 4
MV_COMM_AUTO, MV_COMM_RCCL, MV_COMM_P2P, MV_COMM_HOST = 0, 1, 2, 3


class CandRecC(C.Structure):
    """mv_cand_rec: one coarse candidate of the sharded two-stage pipeline (16 bytes)."""

    _fields_ = [("score", C.c_float), ("rows", C.c_int32), ("id", C.c_int64)]


# every symbol include/mvmaxsim.h declares (tests check the .so exports all of them)
MV_ABI_VERSION = 7  # include/mvmaxsim.h: the header revision this binding's argument lists were written against

EXPORTS = [
    "mv_abi_version", "mv_last_error", "mv_version", "mv_device_count", "mv_host_pin_budget_bytes", "mv_index_exact_hbm_pages", "mv_index_exact_tier_rebalance", "mv_index_exact_tier_hits", "mv_index_fde_placement_trial", "mv_index_read_fde_e4m3", "mv_index_read_fde_fp4", "mv_index_create", "mv_index_destroy", "mv_index_set_option",
    "mv_index_size", "mv_index_capacity", "mv_index_rows_used", "mv_index_capacity_rows", "mv_index_add", "mv_index_add_device", "mv_index_add_bits", "mv_index_remove_doc",
    "mv_index_remove_page", "mv_index_compact", "mv_index_read_pages", "mv_index_read_pages_f32", "mv_index_write_rows", "mv_index_replace_page", "mv_index_read_fp8", "mv_index_fill_synthetic", "mv_index_fill_synthetic_ragged", "mv_synth_rows",
    "mv_query_topk", "mv_query_topk_device", "mv_query_topk_device_async", "mv_query_stats_finish", "mv_query_topk_batch", "mv_merge_topk", "mv_topk_block_bytes", "mv_merge_topk_blocks", "mv_score_all", "mv_score_candidates", "mv_score_candidates_pads", "mv_index_page_rows",
    "mv_two_stage_coarse_device", "mv_two_stage_mid_device", "mv_two_stage_rerank_device", "mv_index_rerank_plan", "mv_comm_create", "mv_comm_destroy", "mv_comm_attach", "mv_comm_transport",
    "mv_comm_query_topk", "mv_comm_query_topk_batch", "mv_sign_pack", "mv_hamming_batch",
    "mv_index_import_fde", "mv_index_read_fde", "mv_query_topk_fde", "mv_query_topk_batch_fde", "mv_comm_query_topk_fde", "mv_comm_query_topk_batch_fde", "mv_two_stage_coarse_device_fde",
    "mv_fde_output_dim", "mv_fde_encode", "mv_calibrate_read_bw", "mv_calibrate", "mv_index_save", "mv_index_load",
    "mv_enc_rmsnorm_bf16", "mv_enc_gated_act_bf16",
]

_lock = threading.Lock()
_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and its
    libraries ask for it by FILE name, while libmvmaxsim.so asks for the SONAME: if /opt/rocm's copy were loaded first,
    a later `import torch` would bring in a SECOND runtime and fail with "No HIP GPUs are available" (measured).
    Loading torch's copy first makes both resolve to the same, already-loaded runtime, in either import order.
    Without torch in the environment this is a no-op and the system ROCm runtime is used."""
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- best effort: the system runtime still works for torch-free processes
        pass


def lib() -> C.CDLL:
    """Load libmvmaxsim.so (built in-tree by csrc/Makefile). Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB):
            raise MvError(-2, f"{_LIB} not found: build it with morphik_core_amd.build_library() "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _preload_torch_hip_runtime()
        L = C.CDLL(_LIB)
        have = L.mv_abi_version() if hasattr(L, "mv_abi_version") else 0
        if have != MV_ABI_VERSION:
            # a library left over from another revision of the header: its entry points would be driven with the wrong
            # argument lists -- refuse it instead of corrupting memory (rebuild: morphik_core_amd.build_library(force=True))
            raise MvError(-5, f"{_LIB} implements ABI revision {have}, this binding needs {MV_ABI_VERSION}: rebuild it "
                              "(morphik_core_amd.build_library(force=True))")
        vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
        L.mv_last_error.restype = C.c_char_p
        L.mv_version.restype = C.c_char_p
        L.mv_device_count.restype = C.c_int
        L.mv_host_pin_budget_bytes.restype = C.c_int64
        L.mv_index_exact_hbm_pages.restype = C.c_int64
        L.mv_index_exact_hbm_pages.argtypes = [C.c_void_p]
        L.mv_index_exact_tier_rebalance.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64)]
        L.mv_index_exact_tier_hits.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.mv_index_read_fde_e4m3.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
        L.mv_index_read_fde_fp4.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
        L.mv_index_fde_placement_trial.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.mv_index_create.argtypes = [C.POINTER(ConfigC), C.POINTER(vp)]
        L.mv_index_destroy.argtypes = [vp]
        L.mv_index_destroy.restype = None
        L.mv_index_set_option.argtypes = [vp, C.c_int, i64]
        L.mv_index_size.argtypes = [vp]
        L.mv_index_size.restype = i64
        L.mv_index_capacity.argtypes = [vp]
        L.mv_index_capacity.restype = i64
        L.mv_index_rows_used.argtypes = [vp]
        L.mv_index_rows_used.restype = i64
        L.mv_index_capacity_rows.argtypes = [vp]
        L.mv_index_capacity_rows.restype = i64
        L.mv_index_fill_synthetic_ragged.argtypes = [vp, u64, u64, i64, i32, i32, i32]
        L.mv_index_add.argtypes = [vp, vp, C.c_int, vp, i64, vp, C.POINTER(i64)]
        L.mv_index_add_device.argtypes = [vp, vp, C.c_int, vp, i64, vp, C.POINTER(i64)]
        L.mv_index_add_bits.argtypes = [vp, vp, vp, i64, vp, C.POINTER(i64)]
        L.mv_index_remove_doc.argtypes = [vp, i32, C.POINTER(i64)]
        L.mv_index_remove_page.argtypes = [vp, i64]
        L.mv_index_compact.argtypes = [vp, vp, C.POINTER(i64)]
        L.mv_index_read_pages.argtypes = [vp, i64, i64, vp]
        L.mv_index_read_pages_f32.argtypes = [vp, i64, i64, vp]
        L.mv_index_write_rows.argtypes = [vp, i64, i32, i32, vp]
        L.mv_index_replace_page.argtypes = [vp, i64, vp, i32]
        L.mv_index_read_fp8.argtypes = [vp, i64, i64, vp, vp]
        L.mv_index_fill_synthetic.argtypes = [vp, u64, u64, i64, i32, i32]
        L.mv_synth_rows.argtypes = [C.c_int, u64, u64, i32, vp]
        L.mv_query_topk.argtypes = [vp, vp, C.c_int, i32, i32, C.c_int, vp, i64, vp, vp, C.POINTER(i32), C.POINTER(QueryStatsC)]
        L.mv_query_topk_device.argtypes = [vp, vp, C.c_int, i32, i32, C.c_int, vp, i64, vp, vp, vp, C.POINTER(QueryStatsC)]
        L.mv_query_topk_device_async.argtypes = [vp, vp, C.c_int, i32, i32, C.c_int, vp, i64, vp, vp, vp, C.POINTER(QueryStatsC)]
        L.mv_query_stats_finish.argtypes = [vp, C.POINTER(QueryStatsC)]
        L.mv_topk_block_bytes.argtypes = [i32]
        L.mv_topk_block_bytes.restype = C.c_int64
        L.mv_merge_topk_blocks.argtypes = [C.c_int, vp, i32, i32, i32, vp, vp, vp]
        L.mv_query_topk_batch.argtypes = [vp, vp, C.c_int, i32, i32, i32, C.c_int, vp, i64, i32, vp, vp, vp, C.POINTER(QueryStatsC)]
        L.mv_index_import_fde.argtypes = [vp, i64, i64, vp]
        L.mv_index_read_fde.argtypes = [vp, i64, i64, vp]
        L.mv_query_topk_fde.argtypes = [vp, vp, C.c_int, i32, vp, i32, C.c_int, vp, i64, vp, vp, C.POINTER(i32), C.POINTER(QueryStatsC)]
        L.mv_query_topk_batch_fde.argtypes = [vp, vp, C.c_int, i32, i32, vp, i32, C.c_int, vp, i64, i32, vp, vp, vp, C.POINTER(QueryStatsC)]
        L.mv_comm_query_topk_fde.argtypes = [vp, vp, C.c_int, i32, vp, i32, C.c_int, vp, i64, vp, vp, C.POINTER(i32), vp]
        L.mv_comm_query_topk_batch_fde.argtypes = [vp, vp, C.c_int, i32, i32, vp, i32, C.c_int, vp, i64, i32, vp, vp, vp, vp]
        L.mv_two_stage_coarse_device_fde.argtypes = [vp, vp, C.c_int, i32, vp, i32, C.c_int, vp, i64, vp, vp]
        L.mv_merge_topk.argtypes = [C.c_int, vp, vp, i32, i32, i32, vp, vp, vp]
        L.mv_score_all.argtypes = [vp, vp, C.c_int, i32, C.c_int, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(QueryStatsC)]
        L.mv_score_candidates.argtypes = [vp, vp, C.c_int, i32, vp, i32, i32, vp, C.POINTER(QueryStatsC)]
        L.mv_score_candidates_pads.argtypes = [vp, vp, C.c_int, i32, vp, i32, vp, vp, C.POINTER(QueryStatsC)]
        L.mv_two_stage_coarse_device.argtypes = [vp, vp, C.c_int, i32, i32, C.c_int, vp, i64, vp, vp]
        L.mv_two_stage_mid_device.argtypes = [vp, vp, C.c_int, i32, C.c_int, vp, i32, i32, vp, vp]
        L.mv_two_stage_rerank_device.argtypes = [vp, vp, C.c_int, i32, C.c_int, vp, i32, i32, vp, i32, i32, vp, vp, vp]
        L.mv_index_rerank_plan.argtypes = [vp, C.c_int, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.mv_comm_create.argtypes = [i32, vp, i32, C.POINTER(vp)]
        L.mv_comm_destroy.argtypes = [vp]
        L.mv_comm_destroy.restype = None
        L.mv_comm_attach.argtypes = [vp, i32, vp]
        L.mv_comm_transport.argtypes = [vp]
        L.mv_comm_query_topk.argtypes = [vp, vp, C.c_int, i32, i32, C.c_int, vp, i64, vp, vp, C.POINTER(i32), vp]
        L.mv_comm_query_topk_batch.argtypes = [vp, vp, C.c_int, i32, i32, i32, C.c_int, vp, i64, i32, vp, vp, vp, vp]
        L.mv_index_page_rows.argtypes = [vp, vp, i64, vp]
        L.mv_sign_pack.argtypes = [C.c_int, vp, i64, i32, vp]
        L.mv_hamming_batch.argtypes = [C.c_int, vp, vp, i64, i32, vp]
        L.mv_fde_output_dim.argtypes = [C.POINTER(FdeConfigC)]
        L.mv_fde_output_dim.restype = i64
        L.mv_fde_encode.argtypes = [C.c_int, C.POINTER(FdeConfigC), vp, i32, i32, vp]
        L.mv_calibrate_read_bw.argtypes = [C.c_int, i64, i32, C.POINTER(C.c_double)]
        L.mv_calibrate.argtypes = [C.c_int, C.c_int, i64, i32, C.POINTER(C.c_double)]
        L.mv_enc_rmsnorm_bf16.argtypes = [C.c_int, vp, vp, C.c_int, vp, i64, i32, C.c_float, C.c_float, C.c_int, vp]
        L.mv_enc_gated_act_bf16.argtypes = [C.c_int, vp, i64, vp, i64, vp, i64, i64, C.c_int, vp]
        L.mv_index_save.argtypes = [vp, C.c_char_p]
        L.mv_index_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
        _lib = L
        return L


def check(status: int) -> None:
    if status != 0:
        raise MvError(status, (lib().mv_last_error() or b"").decode("utf-8", "replace"))
