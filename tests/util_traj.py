"""The ORACLE's half of the free-running PSNR-at-equal-steps tests (tests/test_gpu_zz_trajectories.py), as a script a background
process runs (tests/util_background.py): the reference's training loop (training.py:13-14,47-76: AdamW + cosine, zero_grad /
backward / step / sched.step, train PSNR 10 log10(4 / mse)) on the reference's sampler (dataio.py:104-120) over a small clip,
for one or several (seed, steps, levels) specs, optionally with the same training started <= 1 ulp away (the envelope), plus the
full-frame evaluation PSNR (eval.py:243-256) of the final parameters.  The GPU side of a test draws the same batches from the
same seeded generator and loads the clip this process saved.  Runs under util_windows.oracle_env() (fixed thread count,
MKL_CBWR) with torch's deterministic algorithms: the same trajectory on every box.  Test infrastructure only."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

T, H, W, N_BATCH = 16, 64, 64, 8192                 # defaults (BASELINE.json configs[0] geometry); a spec may override them
FRAMES = (0, 7, 15)


def geometry(spec: dict):
    """(T, H, W, pixels per batch, sparse X = Y, eval frames) of a spec"""
    t = spec.get("T", T)
    return t, spec.get("H", H), spec.get("W", W), spec.get("n_batch", N_BATCH), spec.get("sparse_xy", 20), tuple(spec.get("frames", (0, t // 2 - 1, t - 1)))


def problem(spec: dict):
    """(cfg, initial oracle state, u8 clip [T,H,W,3]) of a spec {"seed", "n_levels", "clip", "video_seed", ...}"""
    from conftest import small_cfg
    import nvp_oracle as O
    from nvp_amd import harness
    T, H, W, _, sxy, _ = geometry(spec)
    cfg = small_cfg(F=spec.get("F", 2), T=spec.get("sparse_t", T), X=sxy, Y=sxy, n_levels=spec["n_levels"])
    sd = O.init_state(cfg, seed=spec["seed"])                       # reference init distributions
    if spec.get("clip", "procedural") == "natural":
        video = harness.natural_video(T, H, W, torch.device("cpu"), seed=spec["video_seed"], grain=4.0)
    else:
        video = harness.procedural_video(T, H, W, torch.device("cpu"), seed=spec["video_seed"])
    return cfg, sd, video


def oracle_trajectory(spec: dict, out_dir: str) -> dict:
    import nvp_oracle as O
    from conftest import ORACLE_TRAIN_THREADS, oracle_determinism
    from util_windows import ulp_perturbed
    cfg, sd, video = problem(spec)
    T, H, W, N_BATCH, _, FRAMES = geometry(spec)
    torch.save(video, os.path.join(out_dir, f"video_{spec['name']}.pt"))
    flat = video.reshape(T, H * W, 3)
    steps_total = spec["steps"]

    def make_ref(state):
        ref = {k: v.clone().requires_grad_(True) for k, v in state.items()}
        opt = torch.optim.AdamW(list(ref.values()), lr=1e-2, weight_decay=0.001)                       # training.py:13
        return ref, opt, torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps_total, eta_min=1e-5), []

    runs = [make_ref(sd)]
    if spec.get("ulp_twin"):
        runs.append(make_ref(ulp_perturbed(sd, spec["seed"] + 1000)))
    gen = torch.Generator().manual_seed(spec["gen_seed"])
    with oracle_determinism(ORACLE_TRAIN_THREADS):
        for _ in range(steps_total):
            ti, pi, coords, tstep = O.sample_batch(T, H, W, N_BATCH, gen)          # the reference's sampler order
            gt = O.normalise_gt(flat[ti, pi].unsqueeze(0))
            for ref, opt, sch, acc in runs:
                loss = O.image_mse(O.nvp_forward(coords.unsqueeze(0), tstep.unsqueeze(0), ref, cfg), gt)   # training.py:50-76 order
                opt.zero_grad(); loss.backward(); opt.step(); sch.step()
                acc.append(10 * math.log10(4 / float(loss.detach())))                                     # training.py:58

        def eval_ref(ref):
            with torch.no_grad():                                                                          # eval.py:243-256 on full frames
                mg, ps = O.get_mgrid_2d(H, W), []
                for f in FRAMES:
                    c = torch.cat((torch.linspace(0, 1, T)[f].expand(H * W, 1), mg), dim=1).unsqueeze(0)
                    s_ = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[f].expand(1, H * W)
                    img = torch.clamp((O.nvp_forward(c, s_, {k: v.detach() for k, v in ref.items()}, cfg) + 1) / 2, 0, 1)
                    ps.append(10 * math.log10(1 / float(((img.reshape(-1, 3) - flat[f].float() / 255.0) ** 2).mean())))
                return sum(ps) / len(ps)
        evs = [eval_ref(r[0]) for r in runs]
    return {"spec": spec, "psnr": runs[0][3], "psnr_1ulp": runs[1][3] if len(runs) > 1 else None, "eval": evs[0],
            "eval_1ulp": evs[1] if len(runs) > 1 else None, "threads": torch.get_num_threads()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--specs", required=True, help="JSON list of specs; one result file traj_<name>.json per spec, written as each finishes")
    a = ap.parse_args()
    from conftest import ORACLE_TRAIN_THREADS
    torch.set_num_threads(ORACLE_TRAIN_THREADS)
    os.makedirs(a.dir, exist_ok=True)
    for spec in json.loads(a.specs):
        r = oracle_trajectory(spec, a.dir)
        tmp = os.path.join(a.dir, f"traj_{spec['name']}.json.tmp")
        with open(tmp, "w") as f:
            json.dump(r, f)
        os.replace(tmp, os.path.join(a.dir, f"traj_{spec['name']}.json"))
