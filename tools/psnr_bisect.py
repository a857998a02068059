#!/usr/bin/env python3
"""Why does the HIP trajectory end a few hundredths of a dB ABOVE the fp32 oracle's?

Train the 64x64x16 case (BASELINE.json configs[0]) on identical batches with
  Z   the oracle in float64 (parameters, activations, AdamW state) - the "exact" trajectory of the reference's algorithm,
  A   the oracle in fp32 (= the reference's arithmetic, pinned bit-exactly by the goldens),
  H   the HIP path with the product's AdamW kernel,
  Ht  the HIP path with torch.optim.AdamW (isolates the optimizer kernel),
and report A - Z, H - Z, Ht - Z (train PSNR, dB) at fixed steps.  If |H - Z| <= |A - Z| the HIP path follows the exact
trajectory at least as closely as the reference's own fp32 arithmetic, and the sign of H - A is the sign of Z - A."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nvp_oracle as O  # noqa: E402
from conftest import small_cfg  # noqa: E402
from util_parity import _load_state_into  # noqa: E402
from nvp_amd import harness  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402

STEPS = int(os.environ.get("NVP_PSNR_STEPS", "100"))
dev = torch.device("cuda:0")
T, H, W, n = 16, 64, 64, 8192
rows = []
for seed, n_levels in ((3, 12), (4, 12), (5, 12), (6, 12)):
    cfg = small_cfg(F=2, T=T, X=20, Y=20, n_levels=n_levels)
    sd = O.init_state(cfg, seed=seed)
    video = harness.procedural_video(T, H, W, torch.device("cpu"), seed=seed)
    flat = video.reshape(T, H * W, 3)

    def make_ref(dtype):
        ref = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.AdamW(list(ref.values()), lr=1e-2, weight_decay=0.001)
        return ref, opt, torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=STEPS, eta_min=1e-5)

    def make_hip(torch_adamw):
        m = NVP(out_features=3, encoding_config=cfg)
        _load_state_into(m, sd)
        m = m.to(dev)
        if torch_adamw:
            opt = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.001)
            sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=STEPS, eta_min=1e-5)
        else:
            opt, sch = harness.make_optimizer(m, total_steps=STEPS)
        return m, opt, sch

    Z, A = make_ref(torch.float64), make_ref(torch.float32)
    Hk, Ht = make_hip(False), make_hip(True)
    gen = torch.Generator().manual_seed(seed)
    traj = {"Z": [], "A": [], "H": [], "Ht": []}
    for it in range(STEPS):
        ti, pi, coords, tstep = O.sample_batch(T, H, W, n, gen)
        gt_u8 = flat[ti, pi].unsqueeze(0)
        for name, (ref, opt, sch) in (("Z", Z), ("A", A)):
            out = O.nvp_forward(coords.unsqueeze(0), tstep.unsqueeze(0), ref, cfg)
            loss = O.image_mse(out, O.normalise_gt(gt_u8).to(out.dtype))
            opt.zero_grad(); loss.backward(); opt.step(); sch.step()
            traj[name].append(10 * math.log10(4 / float(loss)))
        mi = {"all_coords": coords.unsqueeze(0).to(dev), "temporal_steps": tstep.unsqueeze(0).to(dev)}
        for name, (m, opt, sch) in (("H", Hk), ("Ht", Ht)):
            loss = harness.image_mse_u8(m(mi)["model_out"], gt_u8.to(dev))
            opt.zero_grad(); loss.backward(); opt.step(); sch.step()
            traj[name].append(10 * math.log10(4 / float(loss)))
    for s in (10, 30, 60, STEPS):
        if s <= STEPS:
            z = traj["Z"][s - 1]
            rows.append({"seed": seed, "step": s, "psnr_f64": round(z, 4), "A-Z": round(traj["A"][s - 1] - z, 4),
                         "H-Z": round(traj["H"][s - 1] - z, 4), "Ht-Z": round(traj["Ht"][s - 1] - z, 4),
                         "H-A": round(traj["H"][s - 1] - traj["A"][s - 1], 4)})
            print(json.dumps(rows[-1]), flush=True)
last = [r for r in rows if r["step"] == STEPS]
print(json.dumps({"summary": True, "steps": STEPS,
                  "mean_abs_A-Z": round(sum(abs(r["A-Z"]) for r in last) / len(last), 4),
                  "mean_abs_H-Z": round(sum(abs(r["H-Z"]) for r in last) / len(last), 4),
                  "mean_abs_Ht-Z": round(sum(abs(r["Ht-Z"]) for r in last) / len(last), 4),
                  "signs_H-A": [1 if r["H-A"] > 0 else -1 for r in last]}))
