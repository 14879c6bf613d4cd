"""ctypes binding of ``libmtrssm_hip.so`` (C-ABI declared in ``include/mtrssm.h``).

The library is loaded from inside the package directory (built in-tree by
``make -C multimodal_mtrssm_amd/csrc`` / ``__graft_entry__.build()``).  There is no CPU fallback: if
the shared object is missing, or a kernel is asked to run on a non-GPU tensor, this module raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch
from torch import Tensor

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmtrssm_hip.so"

ACT_IDS = {"Identity": 0, "ReLU": 1, "ELU": 2, "Tanh": 3}

_f = C.c_float
_i = C.c_int32
_p = C.c_void_p


def _struct(name: str, fields: list[tuple[str, type]]) -> type:
    return type(name, (C.Structure,), {"_fields_": fields})


def _ptrs(*names: str) -> list[tuple[str, type]]:
    return [(n, _p) for n in names]


MrssmDims = _struct("MtrssmMrssmDims", [
    ("B", _i), ("T", _i), ("D", _i), ("H", _i), ("K", _i), ("C", _i), ("act", _i), ("post", _i),
    ("kl_w_post", _f), ("kl_w_prior", _f), ("rows_per_block", _i), ("threads", _i),
])
MrssmFwdWeights = _struct("MtrssmMrssmFwdWeights", _ptrs(
    "w1s_t", "w2_t", "b2", "wih_t", "bih", "whh_t", "bhh", "wh1_t", "b3", "w4", "b4", "wa2", "ba2", "wv2", "bv2"))
MrssmFwdIO = _struct("MtrssmMrssmFwdIO", _ptrs(
    "xa", "pa", "pv", "deter0", "stoch0", "u_post", "u_prior",
    "deter", "prior_logits", "prior_stoch", "post_logits", "post_stoch", "kl",
    "sv_h1", "sv_h2", "sv_gates", "sv_heads", "sv_la", "sv_lv"))
MrssmClusterWeights = _struct("MtrssmMrssmClusterWeights", _ptrs(
    "w1s_t", "wf_t", "bf", "whh_t", "bhh", "wh1_t", "b3", "w4", "b4", "wa2", "ba2", "wv2", "bv2"))
MrssmBwdWeights = _struct("MtrssmMrssmBwdWeights", _ptrs("w1s_t", "w2", "wih", "whh", "wh1", "w4", "wa2", "wv2"))
MrssmBwdIO = _struct("MtrssmMrssmBwdIO", _ptrs(
    "deter0", "deter", "prior_logits", "post_logits", "sv_h1", "sv_h2", "sv_gates", "sv_heads", "sv_la", "sv_lv",
    "g_deter", "g_post_stoch", "g_prior_stoch", "g_post_logits", "g_prior_logits", "g_kl",
    "g_deter0", "g_stoch0", "d_z1", "d_h2", "d_gi", "d_gh", "d_zh", "d_lp", "d_la", "d_lv"))

MmtrssmDims = _struct("MtrssmMmtrssmDims", [
    ("B", _i), ("T", _i), ("LD", _i), ("HD", _i), ("H", _i), ("KL", _i), ("CL", _i), ("KH", _i), ("CH", _i),
    ("act", _i), ("post", _i), ("tau_l", _f), ("tau_h", _f), ("keep_l", _f), ("keep_h", _f),
    ("kl_w_post", _f), ("kl_w_prior", _f), ("rows_per_block", _i), ("threads", _i),
])
MmtrssmFwdWeights = _struct("MtrssmMmtrssmFwdWeights", _ptrs(
    "wxl_s_t", "wdl_t", "wxh_t", "wdh_t", "bh", "wl1_t", "bl1", "wh1_t", "bh1",
    "wlp2", "blp2", "wa2", "ba2", "wv2", "bv2", "whp2", "bhp2", "whq2", "bhq2"))
MmtrssmFwdIO = _struct("MtrssmMmtrssmFwdIO", _ptrs(
    "xl", "pa", "pv", "deter_l0", "deter_h0", "hidden_l0", "hidden_h0", "stoch_l0", "stoch_h0",
    "u_post_l", "u_post_h", "u_prior_l", "u_prior_h",
    "deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "prior_stoch_l", "prior_stoch_h",
    "post_logits_l", "post_logits_h", "post_stoch_l", "post_stoch_h", "kl_l", "kl_h",
    "sv_l1", "sv_h1", "sv_la", "sv_lv"))
MmtrssmBwdWeights = _struct("MtrssmMmtrssmBwdWeights", _ptrs(
    "wxl_s_t", "wdl", "wxh_t", "wdh", "wl1", "wh1", "wlp2", "wa2", "wv2", "whp2", "whq2"))
MmtrssmBwdIO = _struct("MtrssmMmtrssmBwdIO", _ptrs(
    "deter_l0", "deter_h0", "deter_l", "deter_h", "prior_logits_l", "prior_logits_h", "post_logits_l", "post_logits_h",
    "sv_l1", "sv_h1", "sv_la", "sv_lv",
    "g_deter_l", "g_deter_h", "g_hidden_l", "g_hidden_h", "g_post_stoch_l", "g_post_stoch_h", "g_prior_stoch_l",
    "g_prior_stoch_h", "g_post_logits_l", "g_post_logits_h", "g_prior_logits_l", "g_prior_logits_h", "g_kl_l", "g_kl_h",
    "g_deter_l0", "g_deter_h0", "g_hidden_l0", "g_hidden_h0", "g_stoch_l0", "g_stoch_h0",
    "d_ul", "d_uh", "d_zl1", "d_zh1", "d_lpl", "d_la", "d_lv", "d_lph", "d_lqh"))

ConvGeom = _struct("MtrssmConvGeom", [(n, _i) for n in (
    "N", "C", "Hs", "Ws", "C2", "Cpad", "KH", "KW", "SS", "TS", "OFFY", "OFFX", "Hq", "Wq", "OS", "QY", "QX", "Ho", "Wo",
    "Cout", "CoutPad", "pre_act", "act", "mfma_split")])

Gemm = _struct("MtrssmGemm", _ptrs("A", "B", "C", "bias", "zgrad", "colsum") + [(n, _i) for n in (
    "M", "N", "R", "lda", "ldb", "ldc", "ldz", "a_rmajor", "b_rmajor", "act_a", "act_b", "act_out", "act_z", "accumulate", "split_r")]
    + [("tickets", _p), ("n_tickets", _i), ("mfma_split", _i)])

# every symbol include/mtrssm.h declares (tests/test_capi.py checks the header against this list)
SYMBOLS: dict[str, tuple[type | None, list[type]]] = {
    "mtrssm_version": (C.c_int, []),
    "mtrssm_last_error": (C.c_char_p, []),
    "mtrssm_last_kernel": (C.c_char_p, []),
    "mtrssm_mrssm_rollout_fwd": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmFwdWeights), C.POINTER(MrssmFwdIO), _p]),
    "mtrssm_mrssm_cluster_supported": (C.c_int, [C.POINTER(MrssmDims)]),
    "mtrssm_mrssm_cluster_workspace_bytes": (C.c_int64, [C.POINTER(MrssmDims)]),
    "mtrssm_mrssm_rollout_fwd_cluster": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmClusterWeights), C.POINTER(MrssmFwdIO), _p, C.c_int64, _p]),
    "mtrssm_mrssm_cluster_bwd_workspace_bytes": (C.c_int64, [C.POINTER(MrssmDims)]),
    "mtrssm_mrssm_rollout_bwd_cluster": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmClusterWeights), C.POINTER(MrssmBwdIO), _p, C.c_int64, _p]),
    "mtrssm_mrssm_wide_supported": (C.c_int, [C.POINTER(MrssmDims), _i]),
    "mtrssm_mrssm_wide_workspace_bytes": (C.c_int64, [C.POINTER(MrssmDims), _i]),
    "mtrssm_mrssm_wide_bwd_workspace_bytes": (C.c_int64, [C.POINTER(MrssmDims), _i]),
    "mtrssm_mrssm_rollout_fwd_wide": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmClusterWeights), C.POINTER(MrssmFwdIO), _i, _p, C.c_int64, _p]),
    "mtrssm_mrssm_rollout_bwd_wide": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmClusterWeights), C.POINTER(MrssmBwdIO), _i, _p, C.c_int64, _p]),
    "mtrssm_mrssm_rollout_bwd": (C.c_int, [C.POINTER(MrssmDims), C.POINTER(MrssmBwdWeights), C.POINTER(MrssmBwdIO), _p]),
    "mtrssm_mmtrssm_rollout_fwd": (C.c_int, [C.POINTER(MmtrssmDims), C.POINTER(MmtrssmFwdWeights), C.POINTER(MmtrssmFwdIO), _p]),
    "mtrssm_mmtrssm_rollout_bwd": (C.c_int, [C.POINTER(MmtrssmDims), C.POINTER(MmtrssmBwdWeights), C.POINTER(MmtrssmBwdIO), _p]),
    "mtrssm_mmtrssm_wide_supported": (C.c_int, [C.POINTER(MmtrssmDims), _i]),
    "mtrssm_mmtrssm_wide_workspace_bytes": (C.c_int64, [C.POINTER(MmtrssmDims), _i]),
    "mtrssm_mmtrssm_wide_bwd_workspace_bytes": (C.c_int64, [C.POINTER(MmtrssmDims), _i]),
    "mtrssm_mmtrssm_rollout_fwd_wide": (C.c_int, [C.POINTER(MmtrssmDims), C.POINTER(MmtrssmFwdWeights), C.POINTER(MmtrssmFwdIO), _i, _p, C.c_int64, _p]),
    "mtrssm_mmtrssm_rollout_bwd_wide": (C.c_int, [C.POINTER(MmtrssmDims), C.POINTER(MmtrssmBwdWeights), C.POINTER(MmtrssmBwdIO), _i, _p, C.c_int64, _p]),
    "mtrssm_conv_gather_gemm": (C.c_int, [C.POINTER(ConvGeom), _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mtrssm_conv_gather_gemm_pair": (C.c_int, [C.POINTER(ConvGeom), _p, _p, _p, _p, _p, _p, _p, _p,
                                               C.POINTER(ConvGeom), _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mtrssm_conv_gather_pair_merges": (C.c_int, [C.POINTER(ConvGeom), C.POINTER(ConvGeom), _i]),
    "mtrssm_residual_block_fwd_supported": (C.c_int, [C.POINTER(ConvGeom)]),
    "mtrssm_residual_block_fwd": (C.c_int, [C.POINTER(ConvGeom), _p, _p, _p, _p, _p, _p, _p, C.POINTER(ConvGeom), _p, _p, _p, _p, _p, _p, _p, _p]),
    "mtrssm_pack_conv_weight": (C.c_int, [_p, _i, _i, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _i, _i, _p, _p, _p]),
    "mtrssm_pack_conv_weights": (C.c_int, [_p, _i, _i, _p]),
    "mtrssm_unpack_conv_grads": (C.c_int, [_p, _i, _i, _p]),
    "mtrssm_conv_weight_grad": (C.c_int, [C.POINTER(ConvGeom), _p, _p, _p, _i, _p, _p, _p, C.c_int64, _p]),
    "mtrssm_conv_weight_grad_deferred": (C.c_int, [C.POINTER(ConvGeom), _p, _p, _p, _i, _p, _p, _p, C.c_int64, _p]),
    "mtrssm_conv_weight_grad_reduce": (C.c_int, [_p]),
    "mtrssm_conv_weight_grad_workspace_bytes": (C.c_int64, [C.POINTER(ConvGeom), _i]),
    "mtrssm_channel_sum": (C.c_int, [_p, _i, _i, _i, _p, _p]),
    "mtrssm_convt_k4s2_thin": (C.c_int, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p]),
    "mtrssm_convt_k4s2_band_supported": (C.c_int, [_i, _i, _i, _i, _i]),
    "mtrssm_convt_k4s2_band": (C.c_int, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p]),
    "mtrssm_convt_quad_supported": (C.c_int, [C.POINTER(ConvGeom)]),
    "mtrssm_convt_quad": (C.c_int, [C.POINTER(ConvGeom), _p, C.POINTER(C.c_void_p), _p, _p, _p, C.POINTER(ConvGeom), _p, C.POINTER(C.c_void_p),
                                    _p, _p, _p, _p]),
    "mtrssm_conv_tgather_thin": (C.c_int, [_i] * 11 + [_p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "mtrssm_episode_gather": (C.c_int, [_p, _p, _p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _f, _p, _p, _p]),
    "mtrssm_elbo_combine_fwd": (C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_float, C.c_float, _p, _p, _p, _p, _p]),
    "mtrssm_elbo_combine_bwd": (C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_float, C.c_float, _p, _p, _p, _p, _p]),
    "mtrssm_categorical_sample_fwd": (C.c_int, [_p, _p, C.c_int64, _i, _i, _p, _p, _p, _p]),
    "mtrssm_categorical_sample_bwd": (C.c_int, [_p, _p, _p, C.c_int64, _i, _i, _p, _p]),
    "mtrssm_gaussian_nll_fwd": (C.c_int, [_p, _p, C.c_int64, C.c_int64, _i, _p, _p]),
    "mtrssm_gaussian_nll_bwd": (C.c_int, [_p, _p, _p, C.c_int64, C.c_int64, _i, _p, _p]),
    "mtrssm_sumsq": (C.c_int, [_p, C.c_int64, _p, _p]),
    "mtrssm_adamw_step": (C.c_int, [_p, _p, _p, _p, C.c_int64, _p, _f, _f, _f, _f, _f, _f, _f, _i, _p]),
    "mtrssm_gemm": (C.c_int, [C.POINTER(Gemm), _p]),
    "mtrssm_gemm_group": (C.c_int, [C.POINTER(Gemm), _i, _p]),
    "mtrssm_clear": (C.c_int, [_p, C.c_int64, _p]),
    "mtrssm_adamw_prepare": (C.c_int, [_p, C.c_int64, _p, _p, _p, _f, _f, _p]),
    "mtrssm_adamw_apply": (C.c_int, [_p, _p, _p, _p, _p, C.c_int64, _p, _p, _p, _f, _f, _f, _f, _f, _f, _p]),
}

_LIB: C.CDLL | None = None


class MtrssmLibraryError(RuntimeError):
    """The HIP library is missing or a kernel call failed.  There is no fallback path."""


def load() -> C.CDLL:
    """Load ``libmtrssm_hip.so`` (once).  ``import torch`` has already mapped the HIP runtime."""
    global _LIB  # noqa: PLW0603
    if _LIB is not None:
        return _LIB
    path = Path(os.environ.get("MTRSSM_LIB", LIB_PATH))
    if not path.exists():
        msg = (f"{path} not found: build it with `make -C multimodal_mtrssm_amd/csrc` (or __graft_entry__.build()). "
               "multimodal_mtrssm_amd has no CPU or eager fallback for the rollout path.")
        raise MtrssmLibraryError(msg)
    lib = C.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        err = load().mtrssm_last_error()
        msg = f"{what} failed with code {rc}: {err.decode() if err else '?'}"
        raise MtrssmLibraryError(msg)


def ptr(t: Tensor | None) -> int | None:
    """Device pointer of a contiguous fp32 GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        msg = "multimodal_mtrssm_amd kernels run on MI355X only: got a CPU tensor (there is no CPU fallback)"
        raise MtrssmLibraryError(msg)
    if t.dtype != torch.float32 or not t.is_contiguous():
        msg = f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}"
        raise MtrssmLibraryError(msg)
    return t.data_ptr()


def raw_ptr(t: Tensor | None) -> int | None:
    """Device pointer of a GPU tensor of any dtype / stride (the callee is given the strides separately)."""
    if t is None:
        return None
    if not t.is_cuda:
        msg = "multimodal_mtrssm_amd kernels run on MI355X only: got a CPU tensor (there is no CPU fallback)"
        raise MtrssmLibraryError(msg)
    return t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class KernelTimers:
    """Optional HIP-event timing of the library's launches (used by bench.py for the roofline line).

    The kernels are enqueued on torch's CURRENT stream, so ``torch.cuda.Event`` records on that same
    stream bracket exactly the launch.  Rows are keyed by the device kernel's name as rocprofv3 prints it
    (``mtrssm_last_kernel()``), and carry the algorithmic FLOPs / bytes the caller states for the launch.
    Disabled (zero overhead) unless ``enable()`` was called.
    """

    def __init__(self) -> None:
        self.on = False
        self._rows: dict[str, list[tuple[torch.cuda.Event, torch.cuda.Event, float, float]]] = {}

    def enable(self) -> None:
        self.on = True
        self._rows = {}

    def disable(self) -> None:
        self.on = False

    def call(self, name: str, fn, *args, flops: float = 0.0, nbytes: float = 0.0) -> int:  # noqa: ANN001, ANN002
        if not self.on:
            return fn(*args)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        rc = fn(*args)
        end.record()
        kernel = load().mtrssm_last_kernel()
        key = kernel.decode() if kernel else name
        self._rows.setdefault(key, []).append((start, end, flops, nbytes))
        return rc

    def summary(self) -> dict[str, dict[str, float]]:
        """Per device kernel: launches, total / average milliseconds, algorithmic FLOPs and bytes (totals)."""
        torch.cuda.synchronize()
        out: dict[str, dict[str, float]] = {}
        for k, rows in self._rows.items():
            total = sum(s.elapsed_time(e) for s, e, _, _ in rows)
            out[k] = {"launches": len(rows), "total_ms": total, "avg_ms": total / len(rows),
                      "flops": sum(r[2] for r in rows), "bytes": sum(r[3] for r in rows)}
        return out


TIMERS = KernelTimers()


def fill(struct: C.Structure, **tensors: Tensor | None) -> C.Structure:
    """Set every pointer field from a tensor (missing / None -> NULL).

    The struct keeps a reference to each tensor (``_refs``): a temporary such as ``w.t().contiguous()``
    must stay allocated until the kernel that reads it has been ENQUEUED, otherwise the caching
    allocator may hand its block to the next ``torch.empty`` of the same call.
    """
    struct._refs = tensors  # noqa: SLF001
    for name, _ in struct._fields_:
        setattr(struct, name, ptr(tensors.get(name)))
    unknown = set(tensors) - {n for n, _ in struct._fields_}
    if unknown:
        msg = f"unknown fields for {type(struct).__name__}: {sorted(unknown)}"
        raise KeyError(msg)
    return struct
