"""Fused elementwise chains for the encoder adapters (A1 / A2): RMSNorm and the gated-MLP activation of the PaliGemma
(ColPali) and Qwen2-VL / Qwen2.5-VL (ColQwen2) blocks as ONE HIP pass each (csrc/mv_encops.hip) instead of the strings of
framework kernels the transformers modules run -- about a fifth of the forward's device time on an MI355X
(profiles/r1/rocprofv3_kernel_stats_embed_b32.csv: GEMM 55 %, attention 17 %, elementwise strings ~20 %).

The reference formulation is `model(**processor(x))` under bf16 autocast
(core/embedding/colpali_embedding_model.py:251-262, 275-305); `patch_encoder(model)` keeps it: every patched module computes
what its transformers `forward` computes, in the same order and with the same roundings (the only difference is the summation
order of the mean inside the norm), and falls back to that `forward` for anything the kernels do not take (not bf16, not on a
GPU, not contiguous, autograd enabled).  Gate and up projections are also run as one GEMM (their weights become two views of
one [2I, H] matrix; checkpoints load and save as before).

There is no CPU path here and none is needed: without a GPU the modules simply keep their own forward.
"""
from __future__ import annotations

import ctypes as C
import os
import types
from typing import Any, Dict

from . import _lib

_ACT = {"gelu_pytorch_tanh": 0, "gelu_tanh": 0, "silu": 1, "swish": 1, "gelu": 2}


def _act_code(act_fn: Any, config: Any) -> int:
    name = getattr(config, "hidden_act", None) or getattr(config, "hidden_activation", None)
    if isinstance(name, str) and name in _ACT:
        return _ACT[name]
    cls = type(act_fn).__name__.lower()
    if "tanh" in cls:
        return 0
    if "silu" in cls or "swish" in cls:
        return 1
    if cls in ("geluactivation", "gelu"):
        return 2
    return -1


def _usable(torch, x) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and not torch.is_grad_enabled() and x.numel() > 0
            and x.data_ptr() % 16 == 0)


def _stream(torch, x) -> int:
    return torch.cuda.current_stream(x.device).cuda_stream


def patch_encoder(model: Any) -> Dict[str, int]:
    """Patch every RMSNorm / gated MLP / tanh-gelu ViT MLP of `model` in place.  -> modules patched per kind."""
    import torch

    lib = _lib.lib()
    counts = {"rmsnorm": 0, "gated_mlp": 0, "gelu_epilogue": 0}
    for mod in model.modules():
        cls = type(mod).__name__
        if cls.endswith("RMSNorm") and hasattr(mod, "weight") and not hasattr(mod, "_mv_orig_forward"):
            eps = float(getattr(mod, "eps", getattr(mod, "variance_epsilon", 1e-6)))
            gemma = cls.startswith("Gemma")  # (1 + w) in fp32, one rounding; everything else: Llama style (round, then * w)
            dim = int(mod.weight.shape[0])
            if dim % 8:
                continue
            mod._mv_orig_forward = mod.forward

            def norm_forward(self, x, _eps=eps, _gemma=gemma, _dim=dim):
                w = self.weight
                if not (_usable(torch, x) and x.shape[-1] == _dim and w.is_cuda and w.is_contiguous() and w.dtype in (torch.float32, torch.bfloat16)
                        and w.data_ptr() % 16 == 0 and w.device == x.device):
                    return self._mv_orig_forward(x)
                out = torch.empty_like(x)
                if out.data_ptr() % 16:
                    return self._mv_orig_forward(x)
                _lib.check(lib.mv_enc_rmsnorm_bf16(x.device.index or 0, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                                                   0 if w.dtype == torch.float32 else 1, C.c_void_p(out.data_ptr()), x.numel() // _dim, _dim,
                                                   _eps, 1.0 if _gemma else 0.0, 0 if _gemma else 1, C.c_void_p(_stream(torch, x))))
                return out

            mod.forward = types.MethodType(norm_forward, mod)
            counts["rmsnorm"] += 1
        elif all(hasattr(mod, a) for a in ("gate_proj", "up_proj", "down_proj", "act_fn")) and not hasattr(mod, "_mv_orig_forward"):
            g, u = mod.gate_proj, mod.up_proj
            if not (isinstance(g, torch.nn.Linear) and isinstance(u, torch.nn.Linear) and g.weight.shape == u.weight.shape):
                continue
            act = _act_code(mod.act_fn, getattr(mod, "config", None))
            inter, hidden = int(g.weight.shape[0]), int(g.weight.shape[1])
            if act < 0 or inter % 8 or (g.bias is None) != (u.bias is None):
                continue
            # one [2I, H] weight, gate_proj / up_proj become its two halves (views: no second copy, state_dict unchanged)
            with torch.no_grad():
                fused_w = torch.cat([g.weight.data, u.weight.data], 0).contiguous()
                g.weight.data = fused_w[:inter]
                u.weight.data = fused_w[inter:]
                fused_b = None
                if g.bias is not None:
                    fused_b = torch.cat([g.bias.data, u.bias.data], 0).contiguous()
                    g.bias.data = fused_b[:inter]
                    u.bias.data = fused_b[inter:]
            mod._mv_fused_w, mod._mv_fused_b = fused_w, fused_b
            mod._mv_orig_forward = mod.forward

            def mlp_forward(self, x, _act=act, _inter=inter, _hidden=hidden):
                fw = self._mv_fused_w
                if fw is not None and self.gate_proj.weight.data_ptr() != fw.data_ptr():
                    # .to() / .half() re-materialised the halves: the fused copy is stale -- release it (a 3.6 GB copy per model at
                    # Gemma-2B size) and leave this module to its own forward from now on
                    fw = self._mv_fused_w = self._mv_fused_b = None
                if fw is None or not (_usable(torch, x) and x.shape[-1] == _hidden and fw.is_cuda and fw.dtype == torch.bfloat16):
                    return self._mv_orig_forward(x)
                gu = torch.nn.functional.linear(x, fw, self._mv_fused_b)  # [..., 2I]: gate | up
                rows = gu.numel() // (2 * _inter)
                h = torch.empty(x.shape[:-1] + (_inter,), dtype=x.dtype, device=x.device)
                base = gu.data_ptr()
                _lib.check(lib.mv_enc_gated_act_bf16(x.device.index or 0, C.c_void_p(base), 2 * _inter, C.c_void_p(base + 2 * _inter), 2 * _inter,
                                                     C.c_void_p(h.data_ptr()), rows, _inter, _act, C.c_void_p(_stream(torch, x))))
                return self.down_proj(h)

            mod.forward = types.MethodType(mlp_forward, mod)
            counts["gated_mlp"] += 1
        elif all(hasattr(mod, a) for a in ("fc1", "fc2", "activation_fn")) and not hasattr(mod, "_mv_orig_forward"):
            # SigLIP's MLP: fc1 -> tanh-gelu -> fc2.  The activation rides fc1's GEMM as its epilogue (hipBLASLt through
            # torch._addmm_activation: bias + tanh-gelu on the fp32 accumulator, one rounding) instead of a pass of its own
            f1, f2 = mod.fc1, mod.fc2
            if not (isinstance(f1, torch.nn.Linear) and isinstance(f2, torch.nn.Linear) and f1.bias is not None):
                continue
            if _act_code(mod.activation_fn, getattr(mod, "config", None)) != 0 or not hasattr(torch, "_addmm_activation"):
                continue
            mod._mv_orig_forward = mod.forward

            def vit_mlp_forward(self, x):
                w = self.fc1.weight
                if not (_usable(torch, x) and w.is_cuda and w.dtype == torch.bfloat16):
                    return self._mv_orig_forward(x)
                h = torch._addmm_activation(self.fc1.bias, x.reshape(-1, x.shape[-1]), w.t(), use_gelu=True)
                return self.fc2(h).view(x.shape[:-1] + (self.fc2.out_features,))

            mod.forward = types.MethodType(vit_mlp_forward, mod)
            counts["gelu_epilogue"] += 1
    return counts


def unpatch_encoder(model: Any) -> None:
    """Give every patched module its transformers forward back (the fused gate|up weight views stay: they are the same numbers)."""
    for mod in model.modules():
        if hasattr(mod, "_mv_orig_forward"):
            mod.forward = mod._mv_orig_forward
            del mod._mv_orig_forward


def enabled_by_env() -> bool:
    return os.environ.get("MV_ENCODER_FUSED_OPS", "1") not in ("0", "false", "no")


_TUNED_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned")


def load_tuned_gemms(name: str = "tunableop_colpali_v1_2_gfx950.csv") -> bool:
    """hipBLASLt / rocBLAS solution selection for the encoder's GEMM shapes, tuned on an MI355X with PyTorch's TunableOp
    (tools/tune_encoder_gemms.py; + 9 % pages/s at the reference worker's 16 pages per forward) and shipped as a CSV: TunableOp is
    switched on WITHOUT tuning and reads the file.  PyTorch itself validates the file's ROCm / hipBLASLt / rocBLAS versions and GPU
    architecture against the running stack and ignores it on a mismatch; shapes that are not in it take the library default.
    MV_ENCODER_TUNED_GEMMS=0 leaves TunableOp alone.  -> True when the selections were loaded."""
    if os.environ.get("MV_ENCODER_TUNED_GEMMS", "1") in ("0", "false", "no"):
        return False
    path = os.path.join(_TUNED_DIR, name)
    if not os.path.exists(path):
        return False
    try:
        import torch.cuda.tunable as tun

        if tun.is_enabled():
            # the HOST application runs TunableOp itself (its own results file, tuning on or off): its table is not ours to merge entries
            # into, and its switch is not ours to flip -- the encoder takes whatever selections that setup gives it
            return False
        tuning_was = tun.tuning_is_enabled()
        tun.enable(True)
        tun.tuning_enable(False)
        try:
            ok = bool(tun.read_file(path))
        finally:
            tun.enable(False)  # back to the state found (off): switched on only around the page forwards (tuned_gemms() below) -- with TunableOp
            tun.tuning_enable(tuning_was)  # on, shapes that are NOT in the file (every query length) take a slower default path (15 -> 27 ms, measured)
        return ok
    except Exception:  # noqa: BLE001 -- an optimisation only: the library defaults are always correct
        return False


_tuned_lock = __import__("threading").Lock()
_tuned_users = 0
_tuned_prev = (False, True)


class tuned_gemms:
    """`with tuned_gemms(active):` -- TunableOp (loaded selections, no tuning) for the duration of a page-batch forward.  The switch is
    process-wide, so concurrent forwards are counted: it goes off when the last of them leaves."""

    def __init__(self, active: bool):
        self.active = bool(active)

    def __enter__(self):
        global _tuned_users
        if self.active:
            import torch.cuda.tunable as tun

            with _tuned_lock:
                _tuned_users += 1
                if _tuned_users == 1:
                    global _tuned_prev
                    _tuned_prev = (bool(tun.is_enabled()), bool(tun.tuning_is_enabled()))  # restored when the last forward leaves
                    tun.enable(True)
                    tun.tuning_enable(False)
        return self

    def __exit__(self, *exc):
        global _tuned_users
        if self.active:
            import torch.cuda.tunable as tun

            with _tuned_lock:
                _tuned_users -= 1
                if _tuned_users == 0:
                    tun.enable(_tuned_prev[0])
                    tun.tuning_enable(_tuned_prev[1])
        return False
