#!/usr/bin/env python3
"""Dump RGB + every gradient of one fixed fwd/bwd of the fused path to an .npz (A/B runs of kernel variants selected by
environment variables that libnvp_hip.so reads once per process: NVP_MLP_RING_FWD, NVP_MLP_RING_BWD, NVP_ENCODE_LDS).  tools/ab_libs.sh compares
the dumps bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import small_cfg  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

out, F, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = small_cfg(F=F, T=30, X=40, Y=40)
model = NVP(out_features=3, encoding_config=cfg).to(dev)
with torch.no_grad():
    for p in (model.keyframes_xy.params, model.keyframes_yt.params, model.keyframes_xt.params, model.sparse_grid.embeddings):
        p.normal_(0, 0.3)
g = torch.Generator().manual_seed(1)
coords = torch.rand((n, 3), generator=g)
coords[:, 2] = torch.randint(0, 1920, (n,), generator=g).float() / 1919
coords = coords[torch.argsort(coords[:, 2], stable=True)].unsqueeze(0).to(dev)
steps = torch.rand((1, n), generator=g).to(dev)
w = torch.randn((1, n, 3), generator=g).to(dev)
rgb = model({"all_coords": coords, "temporal_steps": steps, "sorted_by_y": True})["model_out"]
(rgb * w).sum().backward()
torch.cuda.synchronize()
d = {"rgb": rgb.detach().cpu().numpy()}
for k, p in model.named_parameters():
    d["g:" + k] = p.grad.cpu().numpy()
np.savez(out, **d)
print("dumped", out)
