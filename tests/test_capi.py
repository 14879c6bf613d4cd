"""CPU: the C-ABI library loads and exports every symbol ``include/mtrssm.h`` declares.

No compute call is made here (there is no GPU): argument validation returns before any launch.
"""

from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import pytest

from multimodal_mtrssm_amd import _lib

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "mtrssm.h").read_text()


def declared_functions() -> list[str]:
    names = re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(mtrssm_\w+)\s*\(", HEADER, flags=re.MULTILINE)
    assert len(names) >= 10
    return names


def test_library_exports_every_declared_symbol() -> None:
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/mtrssm.h but not exported"
    assert set(declared_functions()) == set(_lib.SYMBOLS), "ctypes table and header disagree"


def test_version_matches_header() -> None:
    want = int(re.search(r"#define MTRSSM_VERSION (\d+)", HEADER).group(1))
    assert _lib.load().mtrssm_version() == want


def _struct_fields(name: str) -> list[str]:
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", HEADER, flags=re.DOTALL).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.DOTALL)
    fields: list[str] = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(float|int32_t)\s*\*?", "", decl)
        fields += [f.strip().lstrip("*").strip() for f in decl.split(",")]
    return fields


@pytest.mark.parametrize("cls", [
    _lib.MrssmDims, _lib.MrssmFwdWeights, _lib.MrssmClusterWeights, _lib.MrssmFwdIO, _lib.MrssmBwdWeights, _lib.MrssmBwdIO,
    _lib.MmtrssmDims, _lib.MmtrssmFwdWeights, _lib.MmtrssmFwdIO, _lib.MmtrssmBwdWeights, _lib.MmtrssmBwdIO,
])
def test_ctypes_structs_mirror_the_header(cls: type) -> None:
    assert [n for n, _ in cls._fields_] == _struct_fields(cls.__name__)


def test_invalid_arguments_are_rejected_without_a_launch() -> None:
    lib = _lib.load()
    dims = _lib.MrssmDims(0, 4, 8, 8, 2, 2, 2, 1, 0.2, 0.8, 0, 0)  # B = 0
    rc = lib.mtrssm_mrssm_rollout_fwd(C.byref(dims), C.byref(_lib.MrssmFwdWeights()), C.byref(_lib.MrssmFwdIO()), None)
    assert rc == -1
    assert b"positive" in lib.mtrssm_last_error()
    dims = _lib.MrssmDims(2, 4, 8, 8, 2, 2, 2, 1, 0.2, 0.8, 0, 0)
    rc = lib.mtrssm_mrssm_rollout_fwd(C.byref(dims), C.byref(_lib.MrssmFwdWeights()), C.byref(_lib.MrssmFwdIO()), None)
    assert rc == -1
    assert b"null" in lib.mtrssm_last_error()
    mt = _lib.MmtrssmDims(2, 4, 8, 8, 8, 2, 2, 2, 2, 2, 1, 1.0, 4.0, 0.0, 0.75, 0.2, 0.8, 0, 0)  # tau_l = 1
    rc = lib.mtrssm_mmtrssm_rollout_fwd(C.byref(mt), C.byref(_lib.MmtrssmFwdWeights()), C.byref(_lib.MmtrssmFwdIO()), None)
    assert rc == -1
    assert b"tau" in lib.mtrssm_last_error()
    assert lib.mtrssm_sumsq(None, 10, None, None) == -1
    assert lib.mtrssm_gaussian_nll_fwd(None, None, 1, 1, 0, None, None) == -1
    assert lib.mtrssm_gemm(C.byref(_lib.Gemm()), None) == -1 and b"gemm" in lib.mtrssm_last_error()
    assert lib.mtrssm_adamw_prepare(None, 10, None, None, None, 0.9, 0.999, None) == -1
    assert lib.mtrssm_clear(None, 16, None) == -1
    assert lib.mtrssm_adamw_apply(None, None, None, None, None, 10, None, None, None, 1.0, 1.0, 0.9, 0.999, 1e-8, 1e-2, None) == -1
    # round-2 entries: argument errors are reported before anything is launched
    assert lib.mtrssm_conv_tgather_thin(1, 16, 8, 8, 9, 3, 3, 2, 1, 16, 16, None, None, None, 0, 0, None, None, None, None) == -1  # Cout > 8
    assert b"tgather" in lib.mtrssm_last_error()
    geoms = (_lib.ConvGeom * 4)()  # all-zero geometries: no instantiated shape
    assert lib.mtrssm_convt_quad_supported(geoms) == 0
    assert lib.mtrssm_convt_quad(geoms, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"convt_quad" in lib.mtrssm_last_error()
    g = _lib.Gemm()
    g.mfma_split = 1  # only 0 (fp32 MFMA) and 2 (two bf16 pieces) exist
    assert lib.mtrssm_gemm(C.byref(g), None) == -1
    # round-3 entries
    assert lib.mtrssm_conv_weight_grad_workspace_bytes(C.byref(_lib.ConvGeom()), 0) == -1  # all-zero geometry
    assert lib.mtrssm_conv_weight_grad(C.byref(_lib.ConvGeom()), None, None, None, 0, None, None, None, 0, None) == -1
    assert lib.mtrssm_conv_weight_grad_deferred(C.byref(_lib.ConvGeom()), None, None, None, 0, None, None, None, 0, None) == -1
    assert lib.mtrssm_conv_weight_grad_reduce(None) == 0  # nothing recorded: no launch
    assert lib.mtrssm_convt_k4s2_band_supported(3200, 16, 64, 16, 1) == 1 and lib.mtrssm_convt_k4s2_band_supported(3200, 16, 13, 37, 1) == 0
    assert lib.mtrssm_convt_k4s2_band(0, 16, 64, 16, 1, None, None, None, 1, 2, None, None) == -1
    assert lib.mtrssm_categorical_sample_fwd(None, None, 4, 6, 5, None, None, None, None) == -1
    assert lib.mtrssm_categorical_sample_bwd(None, None, None, 4, 6, 5, None, None) == -1
    assert lib.mtrssm_elbo_combine_fwd(None, None, None, None, 10, 1.0, 0.0, None, None, None, None, None) == -1
    assert lib.mtrssm_elbo_combine_bwd(None, None, None, None, 10, 1.0, 0.0, None, None, None, None, None) == -1
    assert lib.mtrssm_residual_block_fwd_supported(C.byref(_lib.ConvGeom())) == 0
    assert lib.mtrssm_residual_block_fwd(C.byref(_lib.ConvGeom()), *([None] * 7), None, *([None] * 7), None) == -1
    assert b"residual_block_fwd" in lib.mtrssm_last_error()

    def block_geom(n: int, mid: int, act: int = 2, pre: int = 1) -> C.Structure:
        g = _lib.ConvGeom()
        for k, v in {"N": n, "C": 64, "Hs": 8, "Ws": 8, "Cpad": 64, "KH": 3, "KW": 3, "SS": 1, "TS": 1, "OFFY": -1, "OFFX": -1, "Hq": 8, "Wq": 8,
                     "OS": 1, "Ho": 8, "Wo": 8, "Cout": mid, "CoutPad": mid, "pre_act": pre, "act": act, "mfma_split": 2}.items():
            setattr(g, k, v)
        return g

    assert lib.mtrssm_residual_block_fwd_supported(C.byref(block_geom(5, 128))) == 1
    assert lib.mtrssm_residual_block_fwd_supported(C.byref(block_geom(6, 64))) == 1
    assert lib.mtrssm_residual_block_fwd_supported(C.byref(block_geom(5, 64))) == 0   # two frames per tile: whole tiles only
    assert lib.mtrssm_residual_block_fwd_supported(C.byref(block_geom(6, 64, act=3))) == 0  # Tanh has no fused instance
    assert lib.mtrssm_residual_block_fwd_supported(C.byref(block_geom(6, 64, pre=0))) == 0  # not a residual block's first conv
    large = _lib.MrssmDims(32, 100, 1024, 1024, 16, 8, 2, 1, 0.2, 0.8, 0, 0)
    assert lib.mtrssm_mrssm_wide_supported(C.byref(large), 3) == 0  # no device here: the grid cannot be sized
    assert lib.mtrssm_mrssm_wide_workspace_bytes(C.byref(large), 3) > 60e6  # six bytes per weight of the ~10 M scan weights
    assert lib.mtrssm_mrssm_wide_bwd_workspace_bytes(C.byref(large), 2) > 40e6
    assert lib.mtrssm_mrssm_wide_workspace_bytes(C.byref(large), 4) == 0  # pieces: 2 or 3
    odd = _lib.MrssmDims(32, 100, 1000, 1024, 16, 8, 2, 1, 0.2, 0.8, 0, 0)  # D not a multiple of 16
    assert lib.mtrssm_mrssm_wide_workspace_bytes(C.byref(odd), 3) == 0
    assert lib.mtrssm_mrssm_rollout_fwd_wide(C.byref(large), C.byref(_lib.MrssmClusterWeights()), C.byref(_lib.MrssmFwdIO()), 3, None, 0, None) == -1
    assert b"wide" in lib.mtrssm_last_error()
    assert lib.mtrssm_mrssm_rollout_bwd_wide(C.byref(large), C.byref(_lib.MrssmClusterWeights()), C.byref(_lib.MrssmBwdIO()), 3, None, 0, None) == -1
    assert lib.mtrssm_mrssm_cluster_supported(C.byref(_lib.MrssmDims(64, 50, 200, 200, 6, 5, 2, 1, 0.2, 0.8, 0, 0))) == 0  # no CUs to count


def test_missing_library_fails_loudly(monkeypatch: pytest.MonkeyPatch, tmp_path: Path) -> None:
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("MTRSSM_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MtrssmLibraryError, match="no CPU or eager fallback"):
        _lib.load()
