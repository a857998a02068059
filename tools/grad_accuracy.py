"""How accurate are the HIP path's gradients?  One step of the small test configuration, three evaluations of the
same loss gradient: the oracle in float64 (reference value), the oracle in float32 (what the reference computes on
a CPU), and the HIP kernels.  Prints, per parameter tensor, the relative L2 error of the two float32 results
against float64.  Equal magnitudes mean the HIP path is as accurate as the reference's own arithmetic; trajectories
of two such implementations still separate over many steps (see tools/psnr_chaos_control.py).
    python tools/grad_accuracy.py [F] [n_pixels]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import nvp_oracle as O
from test_gpu_parity import _nvp_pair, _grad_of, _away_from_kinks
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
cfg, sd, model = _nvp_pair(F)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)
cand = torch.rand((2 * n + 64, 3), generator=gen)
coords = cand[_away_from_kinks(cand, sd, cfg, n, margin=1e-5)].unsqueeze(0)
T = cfg["3d_encoding"]["t_resolution"]
steps = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[torch.randint(0, T, (1, n), generator=gen)]
gt = torch.rand((1, n, 3), generator=gen) * 2 - 1

def oracle(dtype):
    s = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    out = O.nvp_forward(coords, steps.to(dtype), s, cfg)
    O.image_mse(out, gt.to(dtype)).backward()
    return out.detach(), {k: v.grad.double().numpy() for k, v in s.items()}

o64, g64 = oracle(torch.float64)
o32, g32 = oracle(torch.float32)
out = model({"all_coords": coords.to(dev), "temporal_steps": steps.to(dev)})["model_out"]
((out - gt.to(dev)) ** 2).mean().backward()
oh = out.detach().cpu().double()
print(f"pixels {n}  RGB max-abs error vs float64:  oracle fp32 {float((o32.double() - o64).abs().max()):.3e}   HIP {float((oh - o64).abs().max()):.3e}")
print(f"{'parameter':46s} {'oracle-fp32 relL2':>18s} {'HIP relL2':>12s}")
tot = [0.0, 0.0, 0.0]
for k in sd:
    ref = g64[k]; a = g32[k]; b = _grad_of(model, k).cpu().double().numpy()
    nr = np.linalg.norm(ref) + 1e-300
    ea, eb = np.linalg.norm(a - ref) / nr, np.linalg.norm(b - ref) / nr
    tot[0] += np.linalg.norm(a - ref) ** 2; tot[1] += np.linalg.norm(b - ref) ** 2; tot[2] += nr ** 2
    print(f"{k:46s} {ea:18.3e} {eb:12.3e}")
print(f"{'all parameters':46s} {np.sqrt(tot[0] / tot[2]):18.3e} {np.sqrt(tot[1] / tot[2]):12.3e}")
