#!/usr/bin/env python3
"""Time NVP.forward (training mode: saved streams + latent written) and its no-grad form on BASELINE configs[1] / [2] through the module
surface, for whatever library NVP_HIP_LIB names (one process per library: it is chosen at load).  usage: fwd_time.py [s|l] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nvp_amd import _lib, harness  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "s"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = bench.WORKLOADS[cfgname]
T, H, W = wl["video"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = NVP(out_features=3, encoding_config=bench.make_cfg(wl["F"], T), verbose=False).to(dev)
video = torch.randint(0, 256, (T, H, W, 3), device=dev, dtype=torch.uint8)
data = harness.DeviceVideo(video, n_samples=bench.N_PX, seed=0, sort_by_y=True)
mi, gt = data.sample()
out = {}
for label, grad in (("train", True), ("nograd", False)):
    with torch.set_grad_enabled(grad):
        for _ in range(3):
            r = model(mi)["model_out"]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            r = model(mi)["model_out"]
        b.record()
        torch.cuda.synchronize()
        out[label] = a.elapsed_time(b) / reps
print(f"{os.path.basename(_lib.LIB_PATH):28s} {cfgname} forward call: train {out['train']:.3f} ms   no-grad {out['nograd']:.3f} ms   (checksum {float(r.double().sum()):.6f})", flush=True)
