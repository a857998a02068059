"""SparseGrid - the 3D sparse positional-feature grid (reference sparsegrid.py:4-156).

Same constructor, parameter name (`embeddings` [T,X,Y,F]), init (U(-1e-4,1e-4),
sparsegrid.py:19-21) and methods (`forward`, `forward_inter`) as the reference; the gather
and its gradient scatter-add run in libnvp_hip.so (nvp_sparse3x3_*).  `embeddings` is read
at call time (eval.py:179 rebinds it).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .functional import SparseGrid3x3, SparseUpsample2x


class SparseGrid(nn.Module):
    def __init__(self, level_dim=2, x_resolution=300, y_resolution=300, t_resolution=600, upsample=False):
        super().__init__()
        self.level_dim = level_dim
        self.x_resolution = x_resolution
        self.y_resolution = y_resolution
        self.t_resolution = t_resolution
        self.embeddings = nn.Parameter(torch.empty(t_resolution, x_resolution, y_resolution, level_dim))
        self.upsample = upsample
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def _grid(self) -> torch.Tensor:
        """The grid the gather reads.  With upsample=True the reference bilinearly upsamples the (x,y) axes of the WHOLE grid by 2 on
        every call (sparsegrid.py:26-34: permute -> F.interpolate over [dim, T, X, Y] -> permute); here that is one HIP pass
        (functional.SparseUpsample2x: nvp_sparse_upsample2x_fwd, ATen's formula) and the gather / scatter run on its output; the
        gradient comes back through the pass's adjoint (nvp_sparse_upsample2x_bwd).  NVP_UPSAMPLE_ATEN=1 keeps the stock ATen route."""
        if not self.upsample:
            return self.embeddings
        if os.environ.get("NVP_UPSAMPLE_ATEN", "0") == "1":
            t = self.embeddings.permute(3, 0, 1, 2)
            t = torch.nn.functional.interpolate(t, scale_factor=2, mode='bilinear')
            return t.permute(1, 2, 3, 0).contiguous()
        return SparseUpsample2x.apply(self.embeddings)

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        return SparseGrid3x3.apply(inputs, self._grid(), False)

    def forward_inter(self, inputs: torch.Tensor) -> torch.Tensor:
        return SparseGrid3x3.apply(inputs, self._grid(), True)
