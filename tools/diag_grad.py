"""Diagnostic: per-parameter / per-level gradient error of the HIP path vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import nvp_oracle as O
from test_gpu_parity import _nvp_pair, _grad_of
F, n = int(sys.argv[1]), int(sys.argv[2])
cfg, sd, model = _nvp_pair(F)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(n)
coords = torch.rand((1, n, 3), generator=gen)
coords[0, 0] = torch.tensor([1.0, 1.0, 1.0])
if n > 1: coords[0, 1] = torch.tensor([0.0, 0.0, 0.0])
T = cfg["3d_encoding"]["t_resolution"]
steps = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[torch.randint(0, T, (1, n), generator=gen)]
gt = torch.rand((1, n, 3), generator=gen) * 2 - 1
sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
ref = O.nvp_forward(coords, steps, sd_ref, cfg); O.image_mse(ref, gt).backward()
out = model({"all_coords": coords.to(dev), "temporal_steps": steps.to(dev)})["model_out"]
((out - gt.to(dev)) ** 2).mean().backward()
print("rgb err", float((out.detach().cpu() - ref.detach()).abs().max()))
scales, ress, offs = O.dense_grid_levels(cfg["2d_encoding_xy"])
for k in sd:
    got = _grad_of(model, k).cpu().numpy().astype(np.float64); want = sd_ref[k].grad.numpy().astype(np.float64)
    print(f"{k:45s} rel-to-max {np.abs(got-want).max()/(np.abs(want).max()+1e-30):.3e}  max|want| {np.abs(want).max():.3e}")
    if k.startswith("keyframes"):
        g2, w2 = got.reshape(-1, F), want.reshape(-1, F)
        for l in range(16):
            a, b = g2[offs[l]:offs[l+1]], w2[offs[l]:offs[l+1]]
            e = np.abs(a - b); i = int(e.max(axis=1).argmax())
            print(f"   L{l:2d} res {ress[l]:4d} maxerr {e.max():.3e} max|want| {np.abs(b).max():.3e} worst cell {i} = ({i % ress[l]},{i // ress[l]}) got {a[i]} want {b[i]} sum got {a.sum():.6e} want {b.sum():.6e}")
