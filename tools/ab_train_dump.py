#!/usr/bin/env python3
"""Dump every parameter and both Adam moments after four optimisation steps through harness.train_step (and the stages that ran) to an
.npz: A/B runs of step-level variants selected by the environment (NVP_HIP_LIB, NVP_TILE_FUSED, NVP_FUSED_*_ADAMW ...), compared bit for
bit by tests/test_gpu_parity.py.   usage: ab_train_dump.py OUT.npz N_PIXELS"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import small_cfg  # noqa: E402
from nvp_amd import functional, harness  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

out, n_px = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
cfg = small_cfg(F=2, T=6, X=20, Y=20)
video = torch.randint(0, 256, (6, 48, 64, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(dev)
torch.manual_seed(11)
model = NVP(out_features=3, encoding_config=cfg, verbose=False).to(dev)
data = harness.DeviceVideo(video, n_samples=n_px, seed=5)
opt, sched = harness.make_optimizer(model, total_steps=4)
timer = functional.TIMER = functional.KernelTimer()
losses = [float(harness.train_step(model, opt, sched, *data.sample())) for _ in range(4)]
functional.TIMER = None
torch.cuda.synchronize()
d = {"losses": np.array(losses), "stages": np.array(sorted(timer.summary()))}
for k, p in model.named_parameters():
    d["p:" + k] = p.detach().cpu().numpy()
    d["m:" + k] = opt.state[p]["exp_avg"].cpu().numpy()
    d["v:" + k] = opt.state[p]["exp_avg_sq"].cpu().numpy()
np.savez(out, **d)
print("dumped", out, d["stages"])
