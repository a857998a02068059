"""NVP model assembly (reference modules.py:8-84) on the HIP kernels.

Same constructor (`NVP(out_features=3, encoding_config=cfg["nvp"], **ignored)`), attributes
(`keyframes_xy/yt/xt`, `sparse_grid`, `net`, `wrapper`), state_dict keys and
`forward(model_input, temporal_interp=False, params=None) -> {'model_out': [b,t,3]}`.
forward runs the fused path: one gather kernel writes the latent straight into the
pixel-tile-major layout the MFMA MLP consumes; backward is the mirrored chain.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import modulation
from . import tinycudann as tcnn
from .functional import NVPFused
from .sparsegrid import SparseGrid


class NVP(nn.Module):
    def __init__(self, out_features=3, encoding_config=None, **kwargs):
        super().__init__()
        cfg = encoding_config
        self.keyframes_xy = tcnn.Encoding(n_input_dims=2, encoding_config=cfg["2d_encoding_xy"])
        assert self.keyframes_xy.dtype == torch.float32
        self.keyframes_yt = tcnn.Encoding(n_input_dims=2, encoding_config=cfg["2d_encoding_yt"])
        assert self.keyframes_yt.dtype == torch.float32
        self.keyframes_xt = tcnn.Encoding(n_input_dims=2, encoding_config=cfg["2d_encoding_xt"])
        assert self.keyframes_xt.dtype == torch.float32

        c3 = cfg["3d_encoding"]
        self.sparse_grid = SparseGrid(level_dim=c3["n_features_per_level"], x_resolution=c3["x_resolution"],
                                      y_resolution=c3["y_resolution"], t_resolution=c3["t_resolution"],
                                      upsample=c3["upsample"])
        self.net = modulation.SirenNet(dim_in=1, dim_hidden=cfg["network"]["n_neurons"], dim_out=out_features,
                                       num_layers=cfg["network"]["n_hidden_layers"], w0_initial=30.)
        latent_dim = sum(cfg[k]["n_levels"] * cfg[k]["n_features_per_level"]
                         for k in ("2d_encoding_xy", "2d_encoding_yt", "2d_encoding_xt"))
        latent_dim += c3["n_features_per_level"] * 9
        self.latent_dim = latent_dim
        self.wrapper = modulation.SirenWrapper(self.net, latent_dim=latent_dim)
        if kwargs.get("verbose", os.environ.get("NVP_QUIET", "0") != "1"):
            print(self)          # as the reference does (modules.py:49); verbose=False (not a reference argument) or NVP_QUIET=1 silences it

    def forward(self, model_input, temporal_interp=False, params=None):
        timesteps = model_input['temporal_steps']
        b, t = timesteps.size(0), timesteps.size(1)
        steps = timesteps.reshape(b * t)
        coords = model_input['all_coords'].reshape(-1, 3)       # (t, x, y)
        out = NVPFused.apply(coords, steps,
                             self.keyframes_xy.params, self.keyframes_yt.params, self.keyframes_xt.params,
                             self.sparse_grid._grid(),            # == embeddings unless upsample=True
                             self.keyframes_xy.levels, self.keyframes_yt.levels, self.keyframes_xt.levels,
                             bool(temporal_interp), torch.is_grad_enabled(),
                             bool(model_input.get('sorted_by_y', False)),   # optional hint from nvp_amd's own sampler
                             model_input.get('nvp_hooks'),                  # optional functional.StepHooks of this call (harness.train_step)
                             *self.wrapper.mlp_tensors())
        return {'model_out': out.reshape((b, t, 3))}
