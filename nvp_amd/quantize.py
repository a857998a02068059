"""8-bit quantise -> de-quantise of the grids, the evaluation-time transform of the reference
(experiment_scripts/eval.py:19-82 `quantize_keyframes`, :84-109 `quantize_sparse_grid`; the
README's 0.901 bpp operating point).  Host-side tensor glue in the reference and here (stock
torch ops, any device); arithmetic order is the reference's so results are bit-identical:

    q  = uint8( 255 * (x - min) / (max - min) + 0.5 )          (float -> uint8 truncates)
    x' = (max - min) * (float(q) / 255) + min

with min/max taken per feature and per LEVEL for the keyframes, per feature over the whole
[T,X,Y] volume for the sparse grid."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L

UNIT = 2.0 ** 8 - 1.0          # eval.py:138 unit_multiplier


def _qdq(x: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    q = (x - lo) / (hi - lo)
    q = UNIT * q
    q = torch.clamp((q + 0.5).to(torch.uint8), 0, 255)
    d = q.to(torch.float32) / UNIT
    return (hi - lo) * d + lo


def quantize_keyframes(params: torch.Tensor, config: dict) -> nn.Parameter:
    """params: flat fp32 keyframe vector -> de-quantised copy as a fresh nn.Parameter (eval.py:163-172)."""
    lv = L.make_levels(config)
    F = lv.n_features
    feats = params.detach().clone().reshape(-1, F)
    for lvl in range(lv.n_levels):
        a, b = int(lv.offset[lvl]), int(lv.offset[lvl + 1])
        blk = feats[a:b]
        for d in range(F):
            col = blk[:, d]
            feats[a:b, d] = _qdq(col, torch.min(col), torch.max(col))
    return nn.Parameter(feats.reshape(-1))


def quantize_sparse_grid(emb: torch.Tensor, config: dict | None = None) -> nn.Parameter:
    """emb [T,X,Y,F] -> de-quantised copy, per-feature global min/max (eval.py:84-109)."""
    out = emb.detach().clone()
    for d in range(out.shape[-1]):
        ch = out[..., d]
        out[..., d] = _qdq(ch, torch.min(ch), torch.max(ch))
    return nn.Parameter(out)


def quantized_bpp(model: nn.Module, n_frames: int, height: int, width: int) -> float:
    """(8 bits x every parameter + 24 extra bits x MLP parameter) / pixels   (eval.py:189-190)."""
    n_all = sum(p.numel() for p in model.parameters())
    n_mlp = sum(p.numel() for p in model.wrapper.parameters())
    return (n_all * 8 + n_mlp * 24) / (n_frames * height * width)
