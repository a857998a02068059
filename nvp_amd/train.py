#!/usr/bin/env python3
"""End-to-end encode (= train) + evaluate driver on the HIP path: the repo's own counterpart of the
reference's train_video.py -> training.train -> eval.py flow, without its
tensorboard / checkpoint / argparse plumbing.

    python -m nvp_amd.train --video procedural --frames 600 --height 1080 --width 1920 --seconds 90
    python -m nvp_amd.train --video /path/to/video.npy --steps 4000         # uint8 [T,H,W,3]
    python -m nvp_amd.train --video /data/UVG/Jockey --frames 600 --seconds 90 --log jockey_90s.jsonl      # directory of PNG frames
    python -m nvp_amd.train --video Jockey_1920x1080_120fps_420_8bit_YUV.yuv --height 1080 --width 1920 --frames 600 --seconds 90

Time-to-quality recipe (BASELINE.json north_star: the reference's 5-minute UVG-HD PSNR, 34.57 dB average over the seven
UVG clips at 0.901 bpp on a V100, README.md:92-100, in <= 90 s on one MI355X): run the third line on a real clip; every
--report-every steps a JSON line {step, seconds (training time only), train_psnr, eval_psnr, mpx_per_s} is printed (and
appended to --log), and the summary line carries the fp32 and the 8-bit-grid PSNR (what the README table reports) and
the bpp.  UVG frames are not shipped with this repo and cannot be fetched here: the committed 90-second log
(profiles/*train_90s*) is on the PROCEDURAL clip and says nothing about that target beyond the step rate.

What it reproduces from the reference: model = NVP(config_nvp_s|l with t_resolution = #frames),
N = 1 245 184 samples per step drawn as dataio.py:104-120, loss/PSNR as training.py:47-61,
AdamW(1e-2, wd 1e-3) + CosineAnnealingLR(T_max = total steps, eta_min 1e-5), per-frame eval PSNR on
[0,1] as eval.py:243-256, and the 8-bit grid quantisation of eval.py:19-109 for the final number.
Prints one JSON line per report and a final summary line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nvp_amd import harness, quantize  # noqa: E402
from nvp_amd.modules import NVP  # noqa: E402


def config(size: str, t_res: int) -> dict:
    F = {"s": 2, "l": 4}[size]          # the only difference between config_nvp_s.json and config_nvp_l.json
    enc = {"otype": "DenseGrid", "n_levels": 16, "n_features_per_level": F, "log2_hashmap_size": 32,
           "base_resolution": 16, "per_level_scale": 1.35}
    return {"2d_encoding_xy": dict(enc), "2d_encoding_xt": dict(enc), "2d_encoding_yt": dict(enc),
            "3d_encoding": {"otype": "SparseGrid", "n_features_per_level": F, "x_resolution": 300, "y_resolution": 300,
                            "t_resolution": t_res, "upsample": False},
            "network": {"n_neurons": 128, "n_hidden_layers": 3}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["s", "l"], default="s")
    ap.add_argument("--video", default="procedural", help="'procedural' (smooth sinusoid pattern), 'natural' (synthetic clip with natural-image statistics: "
                                                          "1/f spectrum, edges, camera + object motion, grain - NOT UVG), a .npy file with uint8 [T,H,W,3], "
                                                          "a directory of PNG frames, or a raw 4:2:0 .yuv")
    ap.add_argument("--target-psnr", type=float, default=34.57, help="report the first step / second at which the evaluation PSNR (fp32 grids, and "
                                                                     "8-bit grids with --eval-8bit) reaches this value (README.md:96: 34.57 dB)")
    ap.add_argument("--natural-alpha", type=float, default=None, help="--video natural: spectral slope of the scene texture (amplitude ~ 1/f^alpha)")
    ap.add_argument("--natural-grain", type=float, default=None, help="--video natural: per-frame Gaussian grain, in 8-bit levels")
    ap.add_argument("--eval-8bit", action="store_true", help="also evaluate with 8-bit de-quantised grids (eval.py:163-179) at every report")
    ap.add_argument("--log", default="", help="append the JSON report lines to this file")
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--steps", type=int, default=0, help="total optimisation steps (cosine horizon)")
    ap.add_argument("--seconds", type=float, default=0, help="alternative to --steps: wall-clock budget; the cosine "
                                                             "horizon is set from a short calibration of the step rate")
    ap.add_argument("--report-every", type=int, default=500)
    ap.add_argument("--eval-frames", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "nvp_amd.train needs a HIP device"
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)

    if args.video == "procedural":
        video = harness.procedural_video(args.frames, args.height, args.width, dev, seed=args.seed)
        source = "procedural moving pattern (UVG frames are not shipped)"
    elif args.video == "natural":
        kw = {k: v for k, v in (("alpha", args.natural_alpha), ("grain", args.natural_grain)) if v is not None}
        video = harness.natural_video(args.frames, args.height, args.width, dev, seed=args.seed, **kw)
        source = "synthetic clip with natural-image statistics (harness.natural_video" + (f", {kw}" if kw else "") + ") - NOT UVG"
    else:
        video = harness.load_video(args.video, args.frames, args.height, args.width).to(dev)
        source = args.video
    T, H, W = (int(v) for v in video.shape[:3])
    model = NVP(out_features=3, encoding_config=config(args.config, T), verbose=False).to(dev)
    data = harness.DeviceVideo(video, seed=args.seed, prefetch=True)
    frames = [int(round(i * (T - 1) / max(args.eval_frames - 1, 1))) for i in range(args.eval_frames)]
    n_slice = harness.eval_slices(H * W)

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if args.log:
            with open(args.log, "a") as f:
                f.write(line + "\n")

    total = args.steps
    if total <= 0:
        # calibrate the step rate on throw-away optimiser state, then fix the cosine horizon
        opt, sched = harness.make_optimizer(model, total_steps=1000)
        state = {k: v.clone() for k, v in model.state_dict().items()}
        for _ in range(5):
            harness.train_step(model, opt, sched, *data.sample())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            harness.train_step(model, opt, sched, *data.sample())
        torch.cuda.synchronize(); rate = 20 / (time.perf_counter() - t0)
        model.load_state_dict(state)
        total = max(int(rate * args.seconds * 0.97), 10)
    opt, sched = harness.make_optimizer(model, total_steps=total)

    cfg = config(args.config, T)
    kf_keys = (("keyframes_xy", "2d_encoding_xy"), ("keyframes_xt", "2d_encoding_xt"), ("keyframes_yt", "2d_encoding_yt"))

    def eval_8bit() -> float:
        """evaluation PSNR with the grids 8-bit quantised / de-quantised (eval.py:163-179 rebinds the parameters the same way);
        the fp32 parameters are put back afterwards"""
        with torch.no_grad():
            keep = [(getattr(model, name), getattr(model, name).params) for name, _ in kf_keys]
            keep_emb = model.sparse_grid.embeddings
            try:
                for (enc, p), (_, key) in zip(keep, kf_keys):
                    enc.params = quantize.quantize_keyframes(p.detach(), cfg[key])
                model.sparse_grid.embeddings = quantize.quantize_sparse_grid(keep_emb.detach())
                return harness.eval_psnr(model, data, frames, n_slice=n_slice)
            finally:
                for enc, p in keep:
                    enc.params = p
                model.sparse_grid.embeddings = keep_emb

    torch.cuda.synchronize()
    train_s, seg = 0.0, time.perf_counter()          # evaluation time is excluded from the encode clock
    reached = {}                                     # first report at which the target PSNR was met
    for step in range(total):
        loss = harness.train_step(model, opt, sched, *data.sample())
        if (step + 1) % args.report_every == 0 or step + 1 == total:
            torch.cuda.synchronize()
            train_s += time.perf_counter() - seg
            rec = {"step": step + 1, "seconds": round(train_s, 2), "train_psnr": round(harness.train_psnr(loss), 3),
                   "eval_psnr": round(harness.eval_psnr(model, data, frames, n_slice=n_slice), 3),
                   "mpx_per_s": round((step + 1) * data.n / train_s / 1e6, 2)}
            if args.eval_8bit:
                rec["eval_psnr_8bit_grids"] = round(eval_8bit(), 3)
            for key, name in (("eval_psnr", "fp32"), ("eval_psnr_8bit_grids", "8bit_grids")):
                if key in rec and rec[key] >= args.target_psnr and name not in reached:
                    reached[name] = {"step": step + 1, "seconds": rec["seconds"], "psnr": rec[key]}
            emit(rec)
            torch.cuda.synchronize()
            seg = time.perf_counter()
    encode_s = train_s

    psnr_fp32 = harness.eval_psnr(model, data, frames, n_slice=n_slice)
    psnr_q8 = eval_8bit()
    emit({"summary": True, "video": source, "frames": T, "height": H, "width": W, "config": "nvp_" + args.config,
                      "steps": total, "encode_seconds": round(encode_s, 2), "eval_frames": frames,
                      "psnr_fp32": round(psnr_fp32, 3), "psnr_8bit_grids": round(psnr_q8, 3),
                      "bpp_8bit": round(quantize.quantized_bpp(model, T, H, W), 4),
                      "mpx_per_s": round(total * data.n / encode_s / 1e6, 2),
                      "target_psnr": args.target_psnr, "target_reached": reached})


if __name__ == "__main__":
    main()
