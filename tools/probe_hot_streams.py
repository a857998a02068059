#!/usr/bin/env python3
"""Would the backward chain run faster if the forward's saved streams were still in the 256 MB memory-side cache when it reads them
(what a tile-fused forward + backward kernel would buy on the READ side)?  One wave-round of tiles (65 536 pixels:
168 MB of saved streams, 38 MB of latent) goes forward -> backward (a) back to back: the streams are the last thing written, (b) with
1 GB / 4 GB of unrelated writes in between (the situation of the full-size step, whose forward alone writes 3.9 GB).  Prints the HIP-event
span of every stage for both; identical kernels, identical data."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from nvp_amd import functional, harness  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = NVP(out_features=3, encoding_config=bench.make_cfg(2, 600), verbose=False).to(dev)
T, H, W = 600, 1080, 1920
video = torch.randint(0, 256, (T, H, W, 3), device=dev, dtype=torch.uint8)
data = harness.DeviceVideo(video, n_samples=n, seed=0, sort_by_y=True, prefetch=False)
pol = torch.empty(1 << 30, device=dev, dtype=torch.float32)        # 4 GB


def run(pollute_gb, reps=12):
    timer = functional.KernelTimer()
    for it in range(reps + 3):
        mi, gt = data.sample()
        functional.TIMER = timer if it >= 3 else None
        out = model(mi)["model_out"]
        loss = harness.image_mse_u8(out, gt["img"])
        if pollute_gb:
            pol[: pollute_gb * (1 << 28)].fill_(float(it))
        model.zero_grad(set_to_none=True)
        loss.backward()
        functional.TIMER = None
    torch.cuda.synchronize()
    return {k: round(v[0] * 1e3, 1) for k, v in timer.summary().items()}


for gb in (0, 1, 4, 0, 4):
    print(f"n = {n}, {gb} GB written between forward and backward: stage spans in us", run(gb))
