"""Per-pixel stage time vs batch size: if the MLP kernels get much cheaper per pixel once all
streams fit the 256 MB Infinity Cache, HBM traffic (not MFMA issue) is what limits them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nvp_amd import functional, harness
from nvp_amd.modules import NVP
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = NVP(out_features=3, encoding_config=bench.CONFIG_NVP_S).to(dev)
T, H, W = 60, 1080, 1920
video = torch.randint(0, 256, (T, H, W, 3), device=dev, dtype=torch.uint8)
for n in [8192, 16384, 32768, 65536, 131072, 262144, 1245184]:
    data = harness.DeviceVideo(video, n_samples=n, seed=0)
    def step():
        mi, gt = data.sample()
        out = model(mi)["model_out"]
        loss = harness.image_mse_u8(out, gt["img"])
        model.zero_grad(set_to_none=True)
        loss.backward()
    for _ in range(3): step()
    functional.TIMER = functional.KernelTimer()
    torch.cuda.synchronize()
    reps = 20 if n < 200000 else 6
    for _ in range(reps): step()
    torch.cuda.synchronize()
    s = functional.TIMER.summary(); functional.TIMER = None
    print(n, {k: round(v[0] * 1e6 / n, 3) for k, v in s.items()}, "ns/px; ms:", {k: round(v[0], 3) for k, v in s.items()}, flush=True)
