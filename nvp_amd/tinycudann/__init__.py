"""Drop-in for the one class NVP uses from the tiny-cuda-nn fork: `tinycudann.Encoding`
with otype "DenseGrid" (reference call sites modules.py:14-23,65-67; README.md:30-32).

Boundary contract:
  * `Encoding(n_input_dims=2, encoding_config=dict, seed=1337, dtype=None)` is an nn.Module
    with ONE registered parameter named `params` - flat fp32, length sum_l res_l^2 * F,
    level-major / cell-major / feature-innermost, no per-level padding (the layout
    eval.py:19-82 and compression.py:17-77 slice);
  * `.dtype == torch.float32` (asserted at modules.py:15), `.n_input_dims`, `.n_output_dims`;
  * `forward(x[N,2]) -> [N, n_levels*F]`; `self.params` is read at call time because
    eval.py:170-172 rebinds it to a fresh nn.Parameter.
The arithmetic runs in libnvp_hip.so (nvp_dense2d_fwd / nvp_dense2d_bwd).  Init is
U(-1e-4, 1e-4) from a generator seeded with `seed` (tcnn's default seed 1337 gives the
three planes identical initial values, as upstream does).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _lib as L
from ..functional import DenseGrid2D

__all__ = ["Encoding"]


class Encoding(nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None):
        super().__init__()
        if n_input_dims != 2:
            raise NotImplementedError("only the 2D DenseGrid used by NVP is implemented (n_input_dims=2)")
        otype = encoding_config.get("otype", "DenseGrid")
        if otype != "DenseGrid":
            raise NotImplementedError(f"encoding otype {otype!r} is outside NVP's path (DenseGrid only)")
        if dtype not in (None, torch.float32):
            raise NotImplementedError("the NVP path is fp32 end to end (reference modules.py:15)")
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.seed = seed
        self.dtype = torch.float32
        self.levels = L.make_levels(encoding_config)
        self.n_output_dims = self.levels.n_levels * self.levels.n_features
        n_params = L.levels_n_params(self.levels)
        gen = torch.Generator().manual_seed(seed)
        init = (torch.rand(n_params, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * 1e-4
        self.params = nn.Parameter(init)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return DenseGrid2D.apply(x, self.params, self.levels)

    def extra_repr(self) -> str:
        c = self.encoding_config
        return (f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, "
                f"n_levels={c.get('n_levels')}, F={c.get('n_features_per_level')}, n_params={self.params.numel()}")
