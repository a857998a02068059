"""ctypes binding of libnvp_hip.so (include/nvp_hip.h).

The shared library is built in-tree by nvp_amd/csrc/build.sh (hipcc, gfx950).  There is
NO fallback: if the library is missing, or a tensor is not a contiguous fp32 HIP tensor,
the call raises - the product path never silently runs on anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NVP_HIP_LIB: load another build of the same library (A/B runs of kernel variants built by tools/build_variant.sh)
LIB_PATH = os.environ.get("NVP_HIP_LIB") or os.path.join(_HERE, "csrc", "libnvp_hip.so")

NVP_MAX_LEVELS = 16
COORDS_SORTED_BY_Y = 1
DZ_PLANES_READY = 2
SCATTER_SPARSE_ONLY = 8      # nvp_encode_bwd in two calls (needs DZ_PLANES_READY): sparse grid first ...
SCATTER_DENSE_ONLY = 16      # ... then the three dense planes
SCATTER_PRESORTED = 32       # nvp_encode_bwd_presort already ran on the workspace
GRID_POS_FMA, GRID_INTERP_FMA, GRID_CLAMP = 1, 2, 4      # nvp_levels.flags (include/nvp_hip.h)
HIDDEN = 128
TILE = 32


class Levels(C.Structure):
    """struct nvp_levels - geometry of one 2D multi-resolution dense grid."""
    _fields_ = [
        ("n_levels", C.c_int32),
        ("n_features", C.c_int32),
        ("scale", C.c_float * NVP_MAX_LEVELS),
        ("res", C.c_int32 * NVP_MAX_LEVELS),
        ("offset", C.c_int32 * (NVP_MAX_LEVELS + 1)),
        ("flags", C.c_int32),
    ]


class SparseShape(C.Structure):
    _fields_ = [("t_res", C.c_int32), ("x_res", C.c_int32), ("y_res", C.c_int32), ("n_features", C.c_int32)]


class MlpParams(C.Structure):
    _fields_ = [
        ("mod_w", C.c_void_p * 3), ("mod_b", C.c_void_p * 3),
        ("sir_w", C.c_void_p * 3), ("sir_b", C.c_void_p * 3),
        ("last_w", C.c_void_p), ("last_b", C.c_void_p),
    ]


MlpGrads = MlpParams  # identical layout (const-ness only differs in C)


class ScatterLm(C.Structure):
    """struct nvp_scatter_lm - device pointers into the scatter workspace (nvp_encode_bwd_prepare)."""
    _fields_ = [("dzs", C.c_void_p * 2), ("dzmax", C.c_void_p), ("sdzmax", C.c_void_p), ("scol0", C.c_int32), ("scols", C.c_int32)]


class AdamwSeg(C.Structure):
    """struct nvp_adamw_seg - one tensor of an AdamW step."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64)]

_p, _i64, _i32, _vp = C.c_void_p, C.c_int64, C.c_int32, C.c_void_p

# name -> argtypes; every entry point returns int unless listed in _RESTYPES
SIGNATURES = {
    "nvp_dense2d_fwd": [_p, _p, _p, _i64, C.POINTER(Levels), _vp],
    "nvp_dense2d_bwd": [_p, _p, _p, _i64, C.POINTER(Levels), _vp],
    "nvp_sparse3x3_fwd": [_p, _p, _p, _i64, C.POINTER(SparseShape), _vp],
    "nvp_sparse3x3_bwd": [_p, _p, _p, _i64, C.POINTER(SparseShape), _vp],
    "nvp_sparse3x3_inter_fwd": [_p, _p, _p, _i64, C.POINTER(SparseShape), _vp],
    "nvp_sparse_upsample2x_fwd": [_p, _p, C.POINTER(SparseShape), _vp],
    "nvp_sparse_upsample2x_bwd": [_p, _p, C.POINTER(SparseShape), _vp],
    "nvp_encode_fwd": [_p, _p, _p, _p, _p, _p, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels),
                       C.POINTER(SparseShape), C.c_int, _i32, _vp],
    "nvp_encode_mlp_fwd_supported": [C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape)],
    "nvp_encode_mlp_fwd": [_p, _p, _p, _p, _p, _p, C.POINTER(MlpParams), _p, _p, _p, _p, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels),
                           C.POINTER(SparseShape), C.c_int, _vp],
    "nvp_encode_bwd": [_p, _p, _i32, _p, _p, _p, _p, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels),
                       C.POINTER(SparseShape), _vp, _i64, _i32, _vp],
    "nvp_encode_bwd_workspace_bytes": [_i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape)],
    "nvp_dz_stride": [_i32],
    "nvp_rows_to_ptm": [_p, _p, _i64, _i32, _i32, _vp],
    "nvp_ptm_to_rows": [_p, _p, _i64, _i32, _i32, _vp],
    "nvp_mlp_pack_fwd": [C.POINTER(MlpParams), _p, _i32, _vp],
    "nvp_mlp_pack_bwd": [C.POINTER(MlpParams), _p, _i32, _vp],
    "nvp_mlp_fwd": [_p, _p, C.POINTER(MlpParams), _p, _p, _p, _i64, _i32, _vp],
    "nvp_mlp_bwd_dx": [_p, _p, _p, C.POINTER(MlpParams), _p, _p, _p, C.POINTER(ScatterLm), _i64, _i32, _vp],
    "nvp_encode_bwd_prepare": [_i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape), _vp, _i64,
                               C.POINTER(ScatterLm), _vp],
    "nvp_encode_bwd_presort": [_p, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape), _vp, _i64, _i32, _vp],
    "nvp_encode_bwd_sparse_adamw": [_p, _p, _i32, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape), _vp, _i64, _i32,
                                    _p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _vp],
    "nvp_encode_bwd_dense_adamw": [_p, _p, _i32, _i64, C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape), _vp, _i64, _i32,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.POINTER(C.c_int64), _vp],
    "nvp_dz_lm_supported": [_i32],
    "nvp_mlp_bwd_dw": [_p, _p, _p, _p, _p, C.POINTER(MlpParams), _p, _i32, C.POINTER(MlpGrads), _i64, _i32, _vp],
    "nvp_mse_u8": [_p, _p, _p, _p, _i64, _vp],
    "nvp_sample_order_workspace_bytes": [_i64, _i32],
    "nvp_sample_order_by_column": [_p, _p, _i64, _i32, _vp, _i64, _vp],
    "nvp_order_by_rows_workspace_bytes": [_i64, C.POINTER(Levels), C.POINTER(Levels)],
    "nvp_order_by_rows": [_p, _p, _i64, C.POINTER(Levels), C.POINTER(Levels), _vp, _i64, _vp],
    "nvp_sample_gather": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _vp],
    "nvp_adamw_step": [C.POINTER(AdamwSeg), _i32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, C.c_double, _vp],
    "nvp_packed_fwd_floats": [_i32],
    "nvp_packed_bwd_floats": [_i32],
    "nvp_dw_partial_floats": [_i32, _i32],
    "nvp_mlp_param_floats": [_i32],
    "nvp_latent_rows": [_i32],
    "nvp_version": [],
    "nvp_mlp_mfma_products": [],
}
_RESTYPES = {
    "nvp_packed_fwd_floats": _i64, "nvp_packed_bwd_floats": _i64, "nvp_dw_partial_floats": _i64, "nvp_sample_order_workspace_bytes": _i64, "nvp_order_by_rows_workspace_bytes": _i64,
    "nvp_mlp_param_floats": _i64, "nvp_latent_rows": _i32, "nvp_version": C.c_char_p,
    "nvp_encode_bwd_workspace_bytes": _i64, "nvp_dz_stride": _i32, "nvp_dz_lm_supported": _i32, "nvp_mlp_mfma_products": _i32, "nvp_encode_mlp_fwd_supported": _i32, "nvp_encode_mlp_fwd_bwd_supported": _i32,
}

# include/nvp_hip_experiments.h: entry points only libnvp_hip_experiments.so exports (bound when present; the product path never needs them)
EXPERIMENT_SIGNATURES = {
    "nvp_encode_mlp_fwd_bwd_supported": [C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape)],
    "nvp_encode_mlp_fwd_bwd": [_p, _p, _p, _p, _p, _p, _p, C.POINTER(MlpParams), _p, _p, _p, _p, _p, _p, _p, C.POINTER(ScatterLm), _i64,
                               C.POINTER(Levels), C.POINTER(Levels), C.POINTER(Levels), C.POINTER(SparseShape), _vp],
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libnvp_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run nvp_amd/csrc/build.sh "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    for name, argtypes in EXPERIMENT_SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def has_entry(name: str) -> bool:
    """True when the loaded library exports `name` (experiments-only entry points, include/nvp_hip_experiments.h)."""
    return hasattr(load(), name)


ERR_BADARG, ERR_UNSUPPORTED = -1, -2          # include/nvp_hip.h


class NvpHipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported configuration"}.get(rc, f"hipError {rc}")
        raise NvpHipError(f"{what} failed: {kind}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    """Device pointer of a contiguous HIP tensor; loud errors otherwise."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("nvp_amd kernels need tensors on a HIP device (got a CPU tensor); "
                           "there is no CPU path in the product - move the module/inputs with .cuda()")
    if t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr()


def ntiles(n: int) -> int:
    return (n + TILE - 1) // TILE


def _f32(x: float) -> float:
    """x rounded once to fp32 (as a Python float)."""
    return C.c_float(x).value


def grid_variant(cfg: dict):
    """(flags, scale_mode) of a dense-grid encoding_config.

    The tiny-cuda-nn fork the reference installs (README.md:30-32) is not under /root/reference, so the
    arithmetic details the reference does not pin are switchable (hard part 7):
      "variant": "tcnn" (default)  - published tiny-cuda-nn: pos = fmaf(scale, x, 0.5f), corner blend as an fma
                                     chain, scale = fp32 exp2f(l * log2f(per_level_scale)) * base - 1
                 "two_rounding"    - pos = fl(fl(x*scale) + 0.5f), separately rounded mul/add blend, scale =
                                     eval.py:28's double formula rounded once to fp32 (the round-1 behaviour)
      "border":  "wrap" (default, upstream `index % hashmap_size`) | "clamp"
      "scale_mode": "fp32_exp2" | "double" overrides the variant's scale arithmetic.
    Environment overrides (for whole-model experiments): NVP_DENSE_VARIANT, NVP_DENSE_BORDER, NVP_DENSE_SCALE."""
    variant = os.environ.get("NVP_DENSE_VARIANT") or cfg.get("variant", "tcnn")
    border = os.environ.get("NVP_DENSE_BORDER") or cfg.get("border", "wrap")
    if variant not in ("tcnn", "two_rounding"):
        raise ValueError(f"dense-grid variant {variant!r}: expected 'tcnn' or 'two_rounding'")
    if border not in ("wrap", "clamp"):
        raise ValueError(f"dense-grid border {border!r}: expected 'wrap' or 'clamp'")
    flags = (GRID_POS_FMA | GRID_INTERP_FMA) if variant == "tcnn" else 0
    if border == "clamp":
        flags |= GRID_CLAMP
    scale_mode = os.environ.get("NVP_DENSE_SCALE") or cfg.get("scale_mode", "fp32_exp2" if variant == "tcnn" else "double")
    if scale_mode not in ("fp32_exp2", "double"):
        raise ValueError(f"dense-grid scale_mode {scale_mode!r}: expected 'fp32_exp2' or 'double'")
    return flags, scale_mode


def make_levels(cfg: dict) -> Levels:
    """Host-side level table.  Resolutions and offsets: the reference's eval.py:28-35 arithmetic (pinned).
    `scale`: see grid_variant()."""
    lv = Levels()
    n_levels = int(cfg["n_levels"])
    if not 1 <= n_levels <= NVP_MAX_LEVELS:
        raise ValueError(f"n_levels must be in [1,{NVP_MAX_LEVELS}]")
    lv.n_levels = n_levels
    lv.n_features = int(cfg["n_features_per_level"])
    base = float(cfg.get("base_resolution", 16))
    pls = float(cfg["per_level_scale"])
    flags, scale_mode = grid_variant(cfg)
    lv.flags = flags
    log2_pls = _f32(math.log2(_f32(pls)))           # upstream: std::log2 of the float per_level_scale
    total = 0
    for lvl in range(n_levels):
        a = math.exp(lvl * math.log(pls)) * base - 1.0
        res = int(math.ceil(a) + 1)
        if scale_mode == "fp32_exp2":
            # grid_scale(): exp2f(level * log2_per_level_scale) * base_resolution - 1.0f, every step rounded to fp32
            e = _f32(2.0 ** _f32(float(lvl) * log2_pls))
            s = _f32(_f32(e * _f32(base)) - 1.0)
            if int(math.ceil(s) + 1) != res:
                raise ValueError(f"level {lvl}: fp32 scale {s} gives resolution {int(math.ceil(s) + 1)}, the reference's "
                                 f"eval.py arithmetic {res}; use scale_mode='double' for this encoding_config")
            lv.scale[lvl] = s
        else:
            lv.scale[lvl] = a        # ctypes rounds the double to fp32 once
        lv.res[lvl] = res
        lv.offset[lvl] = total
        total += res * res
    lv.offset[n_levels] = total
    return lv


def levels_n_params(lv: Levels) -> int:
    return int(lv.offset[lv.n_levels]) * int(lv.n_features)


def mlp_params_struct(tensors: Sequence[torch.Tensor]) -> MlpParams:
    """tensors in canonical order: mod_w0,mod_b0,mod_w1,mod_b1,mod_w2,mod_b2,
    sir_w0,sir_b0,sir_w1,sir_b1,sir_w2,sir_b2,last_w,last_b."""
    s = MlpParams()
    for k in range(3):
        s.mod_w[k] = ptr(tensors[2 * k])
        s.mod_b[k] = ptr(tensors[2 * k + 1])
        s.sir_w[k] = ptr(tensors[6 + 2 * k])
        s.sir_b[k] = ptr(tensors[6 + 2 * k + 1])
    s.last_w = ptr(tensors[12])
    s.last_b = ptr(tensors[13])
    return s


def side_stream(dev) -> "torch.cuda.Stream":
    """A side stream for work that should run UNDERNEATH the compute stream's kernels (sampler, coordinate-only scatter kernels,
    weight packing, early optimizer updates).  NVP_SIDE_PRIORITY (integer, default: unset = the default priority): a LOWER
    priority than the compute stream's (a larger number in HIP's convention) makes the dispatcher prefer the compute stream's
    workgroups whenever both have some ready."""
    pr = os.environ.get("NVP_SIDE_PRIORITY")
    if pr is None or pr == "":
        return torch.cuda.Stream(device=dev)
    return torch.cuda.Stream(device=dev, priority=int(pr))
