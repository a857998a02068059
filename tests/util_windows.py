"""PSNR at equal step count along a whole cosine schedule, walked in windows from the ORACLE's state - with the controls in the run.

north_star: "PSNR within +-0.02 dB at equal step count".  The problem (BASELINE.json configs[0] geometry: 64x64x16, config_nvp_s
values, 12 keyframe levels, 8 192-pixel batches of a clip with natural-image statistics, the reference's sampler / loss / AdamW +
cosine: dataio.py:104-120, training.py:13-14,47-76) is chaotic while the learning rate is high: two fp32 trainings that differ by
one ulp drift apart whatever computes them.  So four trainings take the same batches and are compared at every window end:

  A         the oracle, free-running over the whole schedule (the trajectory that defines the windows)
  B_k       the oracle again, restarted at every window start from A's state moved by <= 1 ulp  (the in-run ENVELOPE)
  product   libnvp_hip.so (fp16x2 split-operand MFMA), restarted at every window start from A's state
  twin      libnvp_hip_fp32mfma.so (every MLP GEMM on v_mfma_f32_32x32x2_f32), likewise - in a subprocess (NVP_HIP_LIB is read once)

`oracle_walk` writes A's state (parameters, both Adam moments, step count) at every window start and the learning rate of every
step to a directory; `hip_walk` (this file run as a script, or imported) replays the windows on whatever library the process
loaded.  The batches are redrawn from the same seeded generator on both sides.  Test infrastructure only."""
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

T, H, W, N_BATCH, N_LEVELS = 16, 64, 64, 8192, 12
FRAMES = (0, 7, 15)


def problem(seed: int):
    from conftest import small_cfg
    import nvp_oracle as O
    from nvp_amd import harness
    cfg = small_cfg(F=2, T=T, X=20, Y=20, n_levels=N_LEVELS)
    sd = O.init_state(cfg, seed=seed)
    video = harness.natural_video(T, H, W, torch.device("cpu"), seed=seed, grain=4.0)         # u8 [T,H,W,3], generated on the CPU
    return cfg, sd, video


def ulp_perturbed(sd, seed):
    """Every parameter moved by at most one fp32 ulp (half of the elements, random direction)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.detach()
        up = torch.rand(v.shape, generator=g) < 0.25
        dn = torch.rand(v.shape, generator=g) < 0.25
        w = torch.where(up, torch.nextafter(v, torch.full_like(v, float("inf"))), v)
        out[k] = torch.where(dn & ~up, torch.nextafter(v, torch.full_like(v, float("-inf"))), w)
    return out


def _psnr(loss) -> float:
    return 10 * math.log10(4 / float(loss))                 # training.py:58


def oracle_walk(seed: int, steps_total: int, window: int, n_controls: int, out_dir: str):
    """-> {"psnr": [...], "controls": [[...]] * n_controls, "eval": dB}; writes win_###.pt and meta.json into out_dir."""
    import nvp_oracle as O
    from conftest import ORACLE_TRAIN_THREADS, oracle_determinism
    cfg, sd, video = problem(seed)
    flat = video.reshape(T, H * W, 3)
    os.makedirs(out_dir, exist_ok=True)
    torch.save(video, os.path.join(out_dir, "video.pt"))      # the replaying processes take THIS clip (no second synthesis under other thread settings)
    with oracle_determinism(ORACLE_TRAIN_THREADS):
        ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt = torch.optim.AdamW(list(ref.values()), lr=1e-2, weight_decay=0.001)                       # training.py:13
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps_total, eta_min=1e-5)          # training.py:14
        gen = torch.Generator().manual_seed(seed)
        pa, pb, lrs = [], [[] for _ in range(n_controls)], []
        ctl = [None] * n_controls
        for it in range(steps_total):
            if it % window == 0:
                state = {"it": it, "params": {k: v.detach().clone() for k, v in ref.items()},
                         "m": {k: opt.state[v]["exp_avg"].clone() for k, v in ref.items()} if it else None,
                         "v": {k: opt.state[v]["exp_avg_sq"].clone() for k, v in ref.items()} if it else None}
                torch.save(state, os.path.join(out_dir, f"win_{it // window:03d}.pt"))
                for c in range(n_controls):              # control c := A's state, parameters moved by <= 1 ulp, A's moments and step count
                    rb = {k: v.clone().requires_grad_(True) for k, v in ulp_perturbed(state["params"], 1000 * (c + 1) + seed + it).items()}
                    ob = torch.optim.AdamW(list(rb.values()), lr=1e-2, weight_decay=0.001)
                    if it:
                        for k, v in rb.items():
                            sa = opt.state[ref[k]]
                            ob.state[v] = {"step": sa["step"].clone(), "exp_avg": sa["exp_avg"].clone(), "exp_avg_sq": sa["exp_avg_sq"].clone()}
                    ctl[c] = (rb, ob)
            lr = opt.param_groups[0]["lr"]
            lrs.append(lr)
            ti, pi, coords, tstep = O.sample_batch(T, H, W, N_BATCH, gen)                              # the reference's sampler order
            gt = O.normalise_gt(flat[ti, pi].unsqueeze(0))
            loss = O.image_mse(O.nvp_forward(coords.unsqueeze(0), tstep.unsqueeze(0), ref, cfg), gt)   # training.py:50-76 order
            opt.zero_grad(); loss.backward(); opt.step(); sch.step()
            pa.append(_psnr(loss.detach()))
            for c, (rb, ob) in enumerate(ctl):
                for g_ in ob.param_groups:
                    g_["lr"] = lr
                lb = O.image_mse(O.nvp_forward(coords.unsqueeze(0), tstep.unsqueeze(0), rb, cfg), gt)
                ob.zero_grad(); lb.backward(); ob.step()
                pb[c].append(_psnr(lb.detach()))
        with torch.no_grad():                                                                          # eval.py:243-256 on full frames
            mg, ev = O.get_mgrid_2d(H, W), []
            for f in FRAMES:
                c_ = torch.cat((torch.linspace(0, 1, T)[f].expand(H * W, 1), mg), dim=1).unsqueeze(0)
                s_ = torch.linspace(0.5 / T, 1 - 0.5 / T, T)[f].expand(1, H * W)
                img = torch.clamp((O.nvp_forward(c_, s_, {k: v.detach() for k, v in ref.items()}, cfg) + 1) / 2, 0, 1)
                ev.append(10 * math.log10(1 / float(((img.reshape(-1, 3) - flat[f].float() / 255.0) ** 2).mean())))
    res = {"seed": seed, "steps": steps_total, "window": window, "psnr": pa, "controls": pb, "eval": sum(ev) / len(ev), "lr": lrs,
           "threads": torch.get_num_threads()}
    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump({"seed": seed, "steps": steps_total, "window": window, "lr": lrs}, f)
    return res


def hip_walk(out_dir: str, free: bool = False):
    """Replay the windows of out_dir on the HIP library this process loaded.
    -> {"psnr": per-step train PSNR of the window-synced model, "eval": its final full-frame PSNR, "free": ..., "lib", "mfma_products"}"""
    import nvp_oracle as O
    from nvp_amd import _lib, harness
    from nvp_amd.modules import NVP
    from nvp_amd.optim import AdamW as NvpAdamW
    from util_parity import _load_state_into
    meta = json.load(open(os.path.join(out_dir, "meta.json")))
    seed, steps_total, window, lrs = meta["seed"], meta["steps"], meta["window"], meta["lr"]
    cfg, sd, _ = problem(seed)
    video = torch.load(os.path.join(out_dir, "video.pt"))
    flat = video.reshape(T, H * W, 3)
    dev = torch.device("cuda:0")

    def make():
        m = NVP(out_features=3, encoding_config=cfg, verbose=False)
        _load_state_into(m, sd)
        m = m.to(dev)
        o, _ = harness.make_optimizer(m, total_steps=steps_total)          # the product's optimiser: nvp_adamw_step
        assert isinstance(o, NvpAdamW)
        return m, o

    def param_of(m, key):
        obj = m
        parts = key.split(".")
        for p_ in parts[:-1]:
            obj = obj[int(p_)] if p_.isdigit() else getattr(obj, p_)
        return getattr(obj, parts[-1])

    def resync(m, o, st):
        _load_state_into(m, st["params"])
        for k in st["params"]:
            s_ = o._state_of(param_of(m, k))
            s_["step"] = st["it"]
            if st["it"]:
                s_["exp_avg"].copy_(st["m"][k])
                s_["exp_avg_sq"].copy_(st["v"][k])
            else:
                s_["exp_avg"].zero_()
                s_["exp_avg_sq"].zero_()

    models = [make()] + ([make()] if free else [])
    acc = [[] for _ in models]
    gen = torch.Generator().manual_seed(seed)
    for it in range(steps_total):
        if it % window == 0:
            resync(*models[0], torch.load(os.path.join(out_dir, f"win_{it // window:03d}.pt")))
        ti, pi, coords, tstep = O.sample_batch(T, H, W, N_BATCH, gen)
        mi = {"all_coords": coords.unsqueeze(0).to(dev), "temporal_steps": tstep.unsqueeze(0).to(dev)}
        gtd = flat[ti, pi].unsqueeze(0).to(dev)
        for (m, o), a in zip(models, acc):
            for g_ in o.param_groups:
                g_["lr"] = lrs[it]                     # the oracle's schedule value of this step
            loss = harness.image_mse_u8(m(mi)["model_out"], gtd)
            o.zero_grad(); loss.backward(); o.step()
            a.append(_psnr(loss))
    data = harness.DeviceVideo(video.to(dev), n_samples=N_BATCH, seed=0)
    ev = [harness.eval_psnr(m, data, frames=list(FRAMES), n_slice=4) for m, _ in models]
    return {"psnr": acc[0], "eval": ev[0], "free": acc[1] if free else None, "eval_free": ev[1] if free else None,
            "lib": _lib.LIB_PATH, "mfma_products": int(_lib.load().nvp_mlp_mfma_products())}


def window_verdicts(gp, gt, env, abs_bound=0.02, calm=0.01, twin_margin=0.005, env_factor=2.0, hot_cap=0.1, max_hot_fraction=0.25):
    """The rule of the windowed test, per window end (all in dB): gp = |product - oracle|, gt = |fp32-MFMA twin - oracle|, env = the
    in-run envelope (largest |1-ulp oracle - oracle| over the controls).
      gp <= 0.02: north_star's bound holds - nothing else is asked, in any window;
      gp  > 0.02 in a calm window (env <= 0.01: the oracle reproduces itself there): a violation;
      gp  > 0.02 in a hot window (env > 0.01: the oracle does NOT reproduce itself to 0.01 dB there): the product is held to what
                 fp32 arithmetic shows in the same window - gp <= gt + 0.005 and gp <= 2 env - else a violation;
      ADVICE r5: the hot-window tolerance scales with the run's OWN envelope and the twin shares the scatter / AdamW / gather code with the
      product, so a defect common to both could ride a drifting envelope.  Two absolute guards: gp <= hot_cap (0.1 dB) in ANY window, and
      at most max_hot_fraction (a quarter) of the windows may be hot at all - most of the schedule is held to the plain 0.02 dB
      (measured: 1 hot window of 20 for seeds 7, 8, 9).
    -> list of (window index, cause) for every violation; the cause names the arithmetic (fp16x2 split) only when the fp32-MFMA twin
    stays inside the bound the product left."""
    bad = []
    n_hot = sum(1 for e in env if e > calm)
    if len(env) >= 4 and n_hot > max_hot_fraction * len(env):
        bad.append((-1, f"{n_hot} of {len(env)} windows are hot (envelope > {calm}): the tolerant rule would cover more than {max_hot_fraction:.0%} of the schedule"))
    for w, (p, t, e) in enumerate(zip(gp, gt, env)):
        if p <= abs_bound:
            continue
        if p > hot_cap:
            bad.append((w, f"product gap {p:.4f} > the absolute ceiling {hot_cap} dB (envelope {e:.4f}, twin gap {t:.4f})"))
        elif e <= calm:
            bad.append((w, f"calm window (envelope {e:.4f}): product gap {p:.4f} > {abs_bound}; fp32-MFMA twin gap {t:.4f} -> "
                           + ("the twin leaves too: HIP-vs-ATen fp32 summation order, not the fp16x2 split" if t > abs_bound else
                              "the twin stays: the fp16x2 split-operand arithmetic is the cause")))
        elif p > t + twin_margin and p > env_factor * e:
            bad.append((w, f"hot window (envelope {e:.4f}): product gap {p:.4f} > twin gap {t:.4f} + {twin_margin} and > {env_factor} x envelope -> the fp16x2 split"))
        elif p > t + twin_margin:
            bad.append((w, f"hot window (envelope {e:.4f}): product gap {p:.4f} > twin gap {t:.4f} + {twin_margin} (within {env_factor} x envelope)"))
        elif p > env_factor * e:
            bad.append((w, f"hot window (envelope {e:.4f}): product gap {p:.4f} > {env_factor} x envelope (twin gap {t:.4f}: the twin leaves too)"))
    return bad


def oracle_env(threads=None):
    """Environment of the oracle's process: MKL's conditional numerical reproducibility on one code path (run-to-run and
    alignment-independent sgemm results), a fixed thread count for OpenMP and MKL.  Set before the process starts - MKL reads
    MKL_CBWR once - which is why the oracle of this comparison runs in a process of its own."""
    from conftest import ORACLE_TRAIN_THREADS
    n = str(threads or ORACLE_TRAIN_THREADS)
    env = dict(os.environ)
    env.update(MKL_CBWR="AVX2", OMP_NUM_THREADS=n, MKL_NUM_THREADS=n, MKL_DYNAMIC="FALSE", OMP_DYNAMIC="FALSE", NVP_ORACLE_TRAIN_THREADS=n)
    env.pop("NVP_HIP_LIB", None)
    return env


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--free", type=int, default=0)
    ap.add_argument("--oracle", type=int, default=0, help="1: walk the ORACLE (CPU) and write the window files; 0: replay them on the HIP library")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--window", type=int, default=50)
    ap.add_argument("--controls", type=int, default=2)
    a = ap.parse_args()
    if a.oracle:
        from conftest import ORACLE_TRAIN_THREADS
        torch.set_num_threads(ORACLE_TRAIN_THREADS)
        r = oracle_walk(a.seed, a.steps, a.window, a.controls, a.dir)
    else:
        r = hip_walk(a.dir, free=bool(a.free))
    with open(a.out, "w") as f:
        json.dump(r, f)
