#!/usr/bin/env python3
"""Long-horizon, full-size training trajectory of ONE build of libnvp_hip.so.

Trains BASELINE.json configs[1] (config_nvp_s, 1920x1080x600, N = 1 245 184 samples per step: the reference's sampler
dataio.py:104-120, loss / PSNR training.py:47-61, AdamW + cosine training.py:13-14) from a seeded init on seeded batches
and logs, every --every steps, the train PSNR of that step's batch and the full-frame evaluation PSNR (eval.py:243-256) of
--eval-frames frames.  Everything that decides the trajectory is seeded and lives on the device (model init: torch's CPU
generator; video, sampler: the device's Philox generator), so two processes that load DIFFERENT builds of the library
(NVP_HIP_LIB=...: the default fp16 x 2 split-operand MFMA build, the all-fp32-MFMA twin) see identical parameters and
identical batches, and any PSNR difference is the kernels' arithmetic.  --ulp starts every parameter <= 1 fp32 ulp away
(the envelope: how far two fp32 trainings of this model drift apart whatever computes them).

    python tools/long_horizon.py --steps 5000 --every 250 --tag f16x2 --out gpurun_out/lh_f16x2.jsonl
    NVP_HIP_LIB=$PWD/nvp_amd/csrc/libnvp_hip_fp32mfma.so python tools/long_horizon.py ... --tag fp32mfma
    ... --ulp 1 --tag fp32mfma_1ulp
    python tools/long_horizon.py --compare a.jsonl b.jsonl [c.jsonl ...]      # |first - others| per checkpoint
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ulp_perturb_(model, seed: int) -> None:
    """Every parameter moved by at most one fp32 ulp (a quarter up, a quarter down, random)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            v = p.detach().cpu()
            up = torch.rand(v.shape, generator=g) < 0.25
            dn = torch.rand(v.shape, generator=g) < 0.25
            w = torch.where(up, torch.nextafter(v, torch.full_like(v, float("inf"))), v)
            w = torch.where(dn & ~up, torch.nextafter(v, torch.full_like(v, float("-inf"))), w)
            p.copy_(w.to(p.device))


def load(path):
    recs = [json.loads(l) for l in open(path) if l.strip().startswith("{")]
    return {r["step"]: r for r in recs if "step" in r}, [r for r in recs if "summary" in r]


def compare(paths):
    runs = [load(p) for p in paths]
    base, _ = runs[0]
    tags = [(s[0]["tag"] if s else os.path.basename(p)) for (_, s), p in zip(runs, paths)]
    print("# |PSNR(%s) - PSNR(x)| in dB at equal step count, identical init and batches" % tags[0])
    print("# step  " + "  ".join(f"train:{t:<16s} eval:{t:<16s}" for t in tags[1:]))
    worst = {t: [0.0, 0.0] for t in tags[1:]}
    for step in sorted(base):
        row = [f"{step:6d}"]
        for (r, _), t in zip(runs[1:], tags[1:]):
            if step not in r:
                row.append(" " * 46)
                continue
            dt = abs(base[step]["train_psnr"] - r[step]["train_psnr"])
            de = abs(base[step]["eval_psnr"] - r[step]["eval_psnr"])
            worst[t][0], worst[t][1] = max(worst[t][0], dt), max(worst[t][1], de)
            row.append(f"{dt:22.4f} {de:21.4f}")
        print("  ".join(row))
    for t in tags[1:]:
        print(f"# max over checkpoints vs {t}: train {worst[t][0]:.4f} dB, eval {worst[t][1]:.4f} dB")
    last = max(base)
    print("# final: " + ", ".join(f"{t}: train {r[last]['train_psnr']:.4f} eval {r[last]['eval_psnr']:.4f}"
                                   for (r, _), t in zip(runs, tags) if last in r))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compare", nargs="+")
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--horizon", type=int, default=0, help="cosine horizon (default: --steps)")
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--video", default="natural", choices=["natural", "procedural"])
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--samples", type=int, default=1245184)
    ap.add_argument("--eval-frames", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ulp", type=int, default=0)
    ap.add_argument("--tag", default="run")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if args.compare:
        compare(args.compare)
        return

    from nvp_amd import _lib, harness
    from nvp_amd.modules import NVP
    from nvp_amd.train import config
    assert torch.cuda.is_available(), "needs a HIP device"
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    torch.manual_seed(args.seed)
    mk = harness.natural_video if args.video == "natural" else harness.procedural_video
    video = mk(args.frames, args.height, args.width, dev, seed=args.seed)
    T, H, W = (int(v) for v in video.shape[:3])
    model = NVP(out_features=3, encoding_config=config("s", T), verbose=False).to(dev)
    if args.ulp:
        ulp_perturb_(model, 1000 + args.ulp)
    data = harness.DeviceVideo(video, n_samples=args.samples, seed=args.seed, prefetch=True)
    frames = [int(round(i * (T - 1) / max(args.eval_frames - 1, 1))) for i in range(args.eval_frames)]
    n_slice = harness.eval_slices(H * W)
    opt, sched = harness.make_optimizer(model, total_steps=args.horizon or args.steps)

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(line + "\n")

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(args.steps):
        loss = harness.train_step(model, opt, sched, *data.sample())
        if (step + 1) % args.every == 0 or step + 1 == args.steps:
            emit({"tag": args.tag, "step": step + 1, "train_psnr": round(harness.train_psnr(loss), 5),
                  "eval_psnr": round(harness.eval_psnr(model, data, frames, n_slice=n_slice), 5)})
    torch.cuda.synchronize()
    emit({"summary": True, "tag": args.tag, "lib": _lib.LIB_PATH, "mfma_products": int(lib.nvp_mlp_mfma_products()),
          "steps": args.steps, "video": args.video, "geometry": [T, H, W], "samples": args.samples, "seed": args.seed,
          "ulp": args.ulp, "wall_s": round(time.perf_counter() - t0, 1),
          "param_checksum": float(sum(p.detach().double().sum() for p in model.parameters()))})


if __name__ == "__main__":
    main()
