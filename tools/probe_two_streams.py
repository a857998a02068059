#!/usr/bin/env python3
"""Probe: how much of a training step's time is idle GPU that a SECOND, independent step could use?  Two models, two half-size batches
(N/2 pixels each): their steps enqueued (a) back to back on one stream, (b) on two streams.  If (b) is not faster than (a), the kernels of a
step already fill the machine and chunking / pipelining a step (forward of chunk B under backward of chunk A) has nothing to win."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nvp_amd import harness  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["s"]
T, H, W = wl["video"]
torch.manual_seed(0)
video = torch.randint(0, 256, (T, H, W, 3), device=dev, dtype=torch.uint8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def make(n):
    m = NVP(out_features=3, encoding_config=bench.make_cfg(wl["F"], T), verbose=False).to(dev)
    o, s = harness.make_optimizer(m, total_steps=10000)
    d = harness.DeviceVideo(video, n_samples=n, seed=1, sort_by_y=True)
    return m, o, s, d


def run(jobs, streams):
    for _ in range(3):
        for (m, o, s, d), st in zip(jobs, streams):
            with torch.cuda.stream(st):
                mi, gt = d.sample()
                harness.train_step(m, o, s, mi, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for (m, o, s, d), st in zip(jobs, streams):
            with torch.cuda.stream(st):
                mi, gt = d.sample()
                harness.train_step(m, o, s, mi, gt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full = make(bench.N_PX)
s0 = torch.cuda.current_stream()
print(f"one full step (N = {bench.N_PX}):                         {run([full], [s0]):.3f} ms")
del full
torch.cuda.empty_cache()
a, b = make(bench.N_PX // 2), make(bench.N_PX // 2)
print(f"two half steps (2 models, N/2 each), one stream:          {run([a, b], [s0, s0]):.3f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two half steps (2 models, N/2 each), two streams:         {run([a, b], [s1, s2]):.3f} ms")
