"""Trajectory tests (-m gpu): PSNR at equal step count over hundreds of optimisation steps, the HIP path against the ORACLE.

Collected LAST (tests/conftest.py sorts `zz_trajectories` behind everything else): these compare two trainings of a chaotic
problem, their margins are statistical, and a margin must never stand in front of the deterministic parity tests
(test_gpu_parity.py, test_gpu_real_configs.py, test_gpu_twin.py, test_gpu_dp2.py) under `-x`.

The checker is made reproducible: the oracle's trainings run with a FIXED intra-op thread count and torch's deterministic
algorithms (conftest.oracle_determinism); the 1 000-step oracle walks in a process of its own with MKL's conditional numerical
reproducibility switched on (util_windows.oracle_env).  The HIP side is bit-reproducible by construction (fixed-point scatter,
ordered dW reduction).  Every test prints ONE summary line (`PSNR-PARITY ...`) so the numbers reach the driver's tail.

Round 6: the oracle's trainings depend on nothing the GPU computes, so they no longer run INSIDE the tests:
conftest.pytest_collection_finish starts them as background processes (`background_jobs()` below, tests/util_background.py,
tests/util_traj.py, tests/util_windows.py --oracle 1) underneath the deterministic parity tests; a test only replays the same
seeded batches on the HIP path and compares.  Every oracle process runs under util_windows.oracle_env() - bit-identical
results whenever and wherever it runs, so moving it changes no verdict."""
import json
import os
import subprocess
import sys

import pytest
import torch

import nvp_oracle as O
import util_background as background
from conftest import ORACLE_TRAIN_THREADS, ORACLE_WALK_THREADS, ROOT, report, say, small_cfg
from util_parity import _load_state_into
from util_windows import oracle_env, window_verdicts

pytestmark = pytest.mark.gpu

TWIN = os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip_fp32mfma.so")


def dev():
    return torch.device("cuda:0")


def _say(line: str) -> None:
    say("PSNR-PARITY " + line)


# ----------------------------------------------------------------------------------------
# the oracle's trainings: background jobs (started by conftest.pytest_collection_finish, or on demand)
# ----------------------------------------------------------------------------------------
WINDOW_SEEDS = [int(v) for v in os.environ.get("NVP_PSNR_SEEDS", os.environ.get("NVP_PSNR_SEED", "7,8")).split(",")]
STEPS_100 = int(os.environ.get("NVP_PSNR_STEPS", "100"))
STEPS_FULL = int(os.environ.get("NVP_PSNR_STEPS_FULL", "50"))
STEPS_LONG = int(os.environ.get("NVP_PSNR_STEPS_LONG", "1000"))
WINDOW = int(os.environ.get("NVP_PSNR_WINDOW", "50"))
SPECS_100 = [{"name": f"s{seed}", "seed": seed, "video_seed": seed, "gen_seed": seed, "steps": STEPS_100, "n_levels": 12, "ulp_twin": True,
              "clip": "procedural"} for seed in (3, 4, 5)]
# a 64 x larger problem than configs[0]: 32 x 256 x 256 clip, 65 536-pixel batches, all 16 levels, a 64 x 64 x 32 sparse grid
SPEC_MID = {"name": "mid16", "seed": 6, "video_seed": 6, "gen_seed": 6, "steps": int(os.environ.get("NVP_PSNR_STEPS_MID", "30")), "n_levels": 16,
            "ulp_twin": False, "clip": "natural", "T": 32, "H": 256, "W": 256, "n_batch": 65536, "sparse_xy": 64}
SPEC_FULL = {"name": "full16", "seed": 3, "video_seed": 1, "gen_seed": 0, "steps": STEPS_FULL, "n_levels": 16, "ulp_twin": False, "clip": "procedural"}


SPECS_16 = [SPEC_FULL, SPEC_MID]
# the REAL model and batch of BASELINE.json configs[1] - config_nvp_s grids (16 levels, sparse grid 600 x 300 x 300: 135.8 M parameters), N = 1 245 184
# samples per step - on a small 600-frame clip (64 x 64 pixels: the coordinates still span every grid); a CPU step of it takes tens of seconds
SPEC_REAL = {"name": "real_s", "seed": 9, "video_seed": 9, "gen_seed": 9, "steps": int(os.environ.get("NVP_PSNR_STEPS_REAL", "6")), "n_levels": 16,
             "ulp_twin": False, "clip": "procedural", "T": 600, "H": 64, "W": 64, "n_batch": 1245184, "sparse_xy": 300}


# ... and config_nvp_l (F = 4: 228-row latent, 163.5 M parameters with the 300-frame sparse grid of configs[2] / [3]), three steps
SPEC_REAL_L = {"name": "real_l", "seed": 10, "video_seed": 10, "gen_seed": 10, "steps": int(os.environ.get("NVP_PSNR_STEPS_REAL_L", "3")), "n_levels": 16, "F": 4,
               "ulp_twin": False, "clip": "procedural", "T": 300, "H": 64, "W": 64, "n_batch": 1245184, "sparse_xy": 300}
SPECS_REAL = [SPEC_REAL, SPEC_REAL_L]


def _windows_argv(seed):
    d = background.job_dir(f"windows_seed{seed}")
    return [os.path.join(ROOT, "tests", "util_windows.py"), "--oracle", "1", "--dir", d, "--out", os.path.join(d, "oracle.json"),
            "--seed", str(seed), "--steps", str(STEPS_LONG), "--window", str(WINDOW), "--controls", "2"]


def _traj_argv(name, specs):
    return [os.path.join(ROOT, "tests", "util_traj.py"), "--dir", background.job_dir(name), "--specs", json.dumps(specs)]


def background_jobs(nodeids):
    """Start the oracle processes the selected tests will ask for - the longest first (conftest calls this once collection is done)."""
    ids = " ".join(nodeids)
    if "test_psnr_tracks_the_oracle_along_a_1000_step_schedule" in ids:
        for seed in WINDOW_SEEDS:
            if f"schedule[{seed}]" in ids:
                background.start(f"windows_seed{seed}", _windows_argv(seed), env=oracle_env(ORACLE_WALK_THREADS))
    if "test_psnr_at_equal_steps_full_levels" in ids or "test_psnr_at_equal_steps_larger_problem" in ids:
        background.start("traj_16", _traj_argv("traj_16", SPECS_16))          # the two 16-level trainings one after the other in ONE process (CPU quota)
    if "test_psnr_at_equal_steps_matches_oracle" in ids:
        background.start("traj100", _traj_argv("traj100", SPECS_100))
    if "test_psnr_at_equal_steps_real_config_size" in ids:
        background.start("traj_real", _traj_argv("traj_real", SPECS_REAL))


def _oracle_trajectory(job, specs, spec):
    """(result dict, u8 clip) of one spec of a util_traj.py job"""
    d = background.result(job, _traj_argv(job, specs))
    return json.load(open(os.path.join(d, f"traj_{spec['name']}.json"))), torch.load(os.path.join(d, f"video_{spec['name']}.pt"))


# ----------------------------------------------------------------------------------------
# PSNR at equal step count: the HIP path and the oracle trained on IDENTICAL batches
# (BASELINE.json configs[0]: 64x64x16 synthetic RGB, config_nvp_s values; north_star: +-0.02 dB)
# ----------------------------------------------------------------------------------------
def _hip_trajectory(spec, video, log=None):
    """The HIP half: the product's modules and its own AdamW kernel trained on the batches util_traj.oracle_trajectory drew (same
    seeded generator, the reference's sampler order) from the same init; returns per-step train PSNRs (training.py:58) and the final
    parameters' full-frame eval PSNR (eval.py:243-256)."""
    import math
    from nvp_amd import harness
    from nvp_amd.modules import NVP
    from nvp_amd.optim import AdamW as _NvpAdamW
    from util_traj import geometry
    T, H, W, N_BATCH, sxy, FRAMES = geometry(spec)
    cfg = small_cfg(F=spec.get("F", 2), T=spec.get("sparse_t", T), X=sxy, Y=sxy, n_levels=spec["n_levels"])
    sd = O.init_state(cfg, seed=spec["seed"])               # reference init distributions
    model = NVP(out_features=3, encoding_config=cfg)
    _load_state_into(model, sd)
    model = model.to(dev())
    flat = video.reshape(T, H * W, 3)
    opt_g, sch_g = harness.make_optimizer(model, total_steps=spec["steps"])      # the product's optimiser: nvp_adamw_step + cosine
    assert isinstance(opt_g, _NvpAdamW)
    gen = torch.Generator().manual_seed(spec["gen_seed"])
    pg = []
    for it in range(spec["steps"]):
        ti, pi, coords, tstep = O.sample_batch(T, H, W, N_BATCH, gen)          # the reference's sampler order
        mi = {"all_coords": coords.unsqueeze(0).to(dev()), "temporal_steps": tstep.unsqueeze(0).to(dev())}
        loss_g = harness.image_mse_u8(model(mi)["model_out"], flat[ti, pi].unsqueeze(0).to(dev()))
        opt_g.zero_grad(); loss_g.backward(); opt_g.step(); sch_g.step()
        pg.append(10 * math.log10(4 / float(loss_g.detach())))
        if log:
            with open(log, "a") as f:
                f.write(f'{{"seed": {spec["seed"]}, "step": {it + 1}, "psnr_hip": {pg[-1]:.4f}}}\n')
    data = harness.DeviceVideo(video.to(dev()), n_samples=N_BATCH, seed=0)
    return pg, harness.eval_psnr(model, data, frames=list(FRAMES), n_slice=4)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_psnr_at_equal_steps_matches_oracle(seed):
    """north_star: PSNR within +-0.02 dB at equal step count - 100 steps (NVP_PSNR_STEPS), three seeds, identical batches, the
    product's own AdamW kernel, plus the full-frame evaluation PSNR of the final parameters.

    Each run also trains the oracle started <= 1 ulp away from itself: two fp32 trainings of this model drift apart whatever
    computes them (sine layers with w0 = 30 amplify rounding differences); that envelope is REPORTED next to the gap
    (measured on MI355X: gap 0.008-0.013 dB, envelope 0.006-0.011 dB, signed final differences -0.008 ... +0.011 dB: no
    systematic sign - profiles/r02_parity_report.jsonl; against a float64 training the HIP path is closer than the fp32 oracle,
    profiles/r02_psnr_bisect_f64_f32_hip.txt).  12 of the 16 keyframe levels (0.36 M cells per
    plane instead of 4.6 M) keep the CPU trainings of the checker affordable; the 16-level model is covered by
    test_psnr_at_equal_steps_full_levels.  The oracle's two trainings run in the background job `traj100`."""
    import math
    spec = next(s_ for s_ in SPECS_100 if s_["seed"] == seed)
    orc, video = _oracle_trajectory("traj100", SPECS_100, spec)
    pa, pb, ev_a, ev_b = orc["psnr"], orc["psnr_1ulp"], orc["eval"], orc["eval_1ulp"]
    pg, ev_g = _hip_trajectory(spec, video, log=os.environ.get("NVP_PSNR_LOG"))
    steps_total = spec["steps"]
    assert len(pa) == len(pb) == len(pg) == steps_total and orc["threads"] == ORACLE_TRAIN_THREADS
    assert pg[-1] > 10 * math.log10(4 / 0.34) + 3, "training did not make progress"
    gap = [abs(a - g) for a, g in zip(pa, pg)]
    env = [abs(a - b) for a, b in zip(pa, pb)]
    report("psnr_equal_steps", seed=seed, n_levels=12, steps=steps_total, gap30=max(gap[:30]), gap=max(gap), envelope=max(env),
           final_hip_minus_oracle=pg[-1] - pa[-1], final_1ulp_minus_oracle=pb[-1] - pa[-1],
           eval_hip_minus_oracle=ev_g - ev_a, eval_1ulp_minus_oracle=ev_b - ev_a, final_psnr=pa[-1])
    # north_star: +-0.02 dB, UNCONDITIONAL (measured 0.007-0.011 dB train, 0.001-0.004 dB eval).  The oracle's drift against its
    # own 1-ulp twin on the same batches is reported next to it (envelope), it does not widen the bound.
    _say(f"100-step seed={seed} train_gap_max={max(gap):.4f} envelope_max={max(env):.4f} final_hip-oracle={pg[-1] - pa[-1]:+.4f} "
         f"final_1ulp-oracle={pb[-1] - pa[-1]:+.4f} eval_hip-oracle={ev_g - ev_a:+.4f} eval_1ulp-oracle={ev_b - ev_a:+.4f} dB")
    assert max(gap) <= 0.02, f"train-PSNR gap {max(gap):.4f} dB over {steps_total} steps (1-ulp envelope {max(env):.4f} dB)"
    assert abs(ev_g - ev_a) <= 0.02, f"eval-PSNR gap {abs(ev_g - ev_a):.4f} dB (1-ulp control {abs(ev_b - ev_a):.4f})"


def _walk_windows(seed):
    """oracle (background process, reproducible) -> product (this process) and fp32-MFMA twin (own process, at the same time: the
    walks are chains of small launches with host round trips in between - two of them share the GPU without slowing each other
    much); returns the three results"""
    from util_windows import hip_walk
    background.start(f"windows_seed{seed}", _windows_argv(seed), env=oracle_env(ORACLE_WALK_THREADS))        # (no-op when collection started it)
    d = background.result(f"windows_seed{seed}")
    orc = json.load(open(os.path.join(d, "oracle.json")))
    env = dict(os.environ)
    env["NVP_HIP_LIB"] = TWIN
    tw = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "util_windows.py"), "--dir", d, "--out", os.path.join(d, "twin.json")],
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        prod = hip_walk(d, free=True)
        _, err = tw.communicate(timeout=1500)
    finally:
        if tw.poll() is None:
            tw.kill()
    assert tw.returncode == 0, f"twin walk: {err[-2000:]}"
    twin = json.load(open(os.path.join(d, "twin.json")))
    for f in os.listdir(d):
        if f.endswith(".pt"):
            os.remove(os.path.join(d, f))
    return orc, prod, twin


@pytest.mark.parametrize("seed", WINDOW_SEEDS)
def test_psnr_tracks_the_oracle_along_a_1000_step_schedule(seed):
    """The HIP path against the ORACLE over a whole cosine schedule of 1 000 steps (NVP_PSNR_STEPS_LONG) at the small size, walked in
    50-step windows (NVP_PSNR_WINDOW) from the oracle's state, WITH THE CONTROLS IN THE RUN (tests/util_windows.py): at every window
    start the product build, the fp32-MFMA twin build (subprocess) and two 1-ulp copies of the oracle are set to the oracle's
    parameters, Adam moments and step count; all take the oracle's batches and learning rates; train PSNR (training.py:58) is
    compared at every window end.

    Asserted (util_windows.window_verdicts): north_star's +-0.02 dB at every window end; a window end beyond it is a violation when
    the oracle reproduces ITSELF to 0.01 dB in that window (the two 1-ulp controls), and otherwise only tolerated when the product
    stays with the fp32-MFMA twin (+0.005 dB) and inside twice the controls' envelope.  The final full-frame evaluation PSNR
    (eval.py:243-256) of the synced model is held to +-0.02 dB.  A free-running product model is reported and bounded loosely (a
    real divergence still fails).  A failure names its cause (fp16x2 split vs HIP-vs-ATen summation order vs chaos beyond the
    controls) and carries all three series; a pass prints them too (PSNR-PARITY line).

    Reproducibility: the oracle walks in a process of its own (fixed thread count, deterministic algorithms, MKL_CBWR); measured on
    MI355X hosts its trajectory is bit-identical box to box, and the HIP side is bit-reproducible by construction - so this test's
    outcome for a given seed is the same on every box (profiles/r05_psnr_windows_seeds_boxes.txt: seeds 7, 8, 9 on several boxes).
    Two seeds are graded (NVP_PSNR_SEEDS, default "7,8": seed 8 has the largest product gap measured, 0.0185 dB in its one hot
    window); their oracle walks run as background processes from the start of the session."""
    import math
    steps_total, window = STEPS_LONG, WINDOW
    assert os.path.exists(TWIN), f"{TWIN} missing: run nvp_amd/csrc/build.sh"
    orc, prod, twin = _walk_windows(seed)
    assert prod["mfma_products"] == 3 and twin["mfma_products"] == 1, (prod["lib"], twin["lib"])
    pa = orc["psnr"]
    ends = [min(s + window, steps_total) - 1 for s in range(0, steps_total, window)]
    gp = [abs(pa[e] - prod["psnr"][e]) for e in ends]
    gt = [abs(pa[e] - twin["psnr"][e]) for e in ends]
    env = [max(abs(pa[e] - c[e]) for c in orc["controls"]) for e in ends]
    gap_free = [abs(a - b) for a, b in zip(pa, prod["free"])]
    ev_gap, ev_gap_twin = abs(prod["eval"] - orc["eval"]), abs(twin["eval"] - orc["eval"])
    fmt = lambda xs: "[" + " ".join(f"{x:.4f}" for x in xs) + "]"          # noqa: E731
    series = f"product={fmt(gp)} twin={fmt(gt)} envelope={fmt(env)}"
    report("psnr_equal_steps_windows", seed=seed, steps=steps_total, window=window, product=gp, twin=gt, envelope=env, eval_product_gap=ev_gap,
           eval_twin_gap=ev_gap_twin, free_final_gap=gap_free[-1], free_max_gap=max(gap_free), free_argmax=gap_free.index(max(gap_free)) + 1,
           final_psnr_oracle=pa[-1], oracle_threads=orc["threads"])
    bad = window_verdicts(gp, gt, env)
    _say(f"windows seed={seed} steps={steps_total} window={window} max_product={max(gp):.4f} max_twin={max(gt):.4f} max_envelope={max(env):.4f} "
         f"hot_windows={sum(e > 0.01 for e in env)} violations={len(bad)} eval_gap={ev_gap:.5f} free_max={max(gap_free):.3f} free_final={gap_free[-1]:.3f} dB | {series}")
    assert pa[-1] > 10 * math.log10(4 / 0.34) + 6, "training did not make progress"
    assert not bad, "; ".join(f"window {w} (steps {w * window}-{ends[w] + 1}): {why}" for w, why in bad) + " | " + series
    assert ev_gap <= 0.02, f"final eval-PSNR gap {ev_gap:.4f} dB (twin: {ev_gap_twin:.4f})"
    assert max(gap_free) <= 1.0 and gap_free[-1] <= 0.3, (f"free-running HIP trajectory left the oracle's: {max(gap_free):.3f} dB at step "
                                                         f"{gap_free.index(max(gap_free)) + 1}, {gap_free[-1]:.3f} dB at the end")


def test_psnr_at_equal_steps_full_levels():
    """The same check on the full 16-level keyframes (config_nvp_s values, BASELINE.json configs[0]) over 50 steps, without the
    1-ulp twin (each CPU step of the checker updates 27.8 M parameters; background job `traj_16`)."""
    import math
    orc, video = _oracle_trajectory("traj_16", SPECS_16, SPEC_FULL)
    pg, _ = _hip_trajectory(SPEC_FULL, video)
    pa = orc["psnr"]
    assert len(pa) == len(pg) == SPEC_FULL["steps"]
    gap = [abs(a_ - g_) for a_, g_ in zip(pa, pg)]
    report("psnr_equal_steps_full", steps=SPEC_FULL["steps"], gap=max(gap), final_gap=gap[-1], final_psnr=pa[-1])
    assert pg[-1] > 10 * math.log10(4 / 0.34) + 3, "training did not make progress"
    _say(f"50-step 16-level train_gap_max={max(gap):.4f} final_gap={gap[-1]:.4f} dB")
    assert max(gap) <= 0.02, f"train-PSNR gap {max(gap):.4f} dB"


def test_psnr_at_equal_steps_larger_problem():
    """PSNR at equal step count against the ORACLE on a problem 64 x the size of configs[0] (oracle trajectories existed only at
    64 x 64 x 16 with 8 192-pixel batches): a 32 x 256 x 256 clip with natural-image statistics, 65 536-pixel batches, all 16 keyframe levels,
    a 64 x 64 x 32 sparse grid, 30 steps (NVP_PSNR_STEPS_MID; background job `traj_16`: a CPU step of this size takes seconds on the two
    threads a background training gets).  +-0.02 dB on the train PSNR at every step and on the final full-frame evaluation PSNR."""
    import math
    orc, video = _oracle_trajectory("traj_16", SPECS_16, SPEC_MID)
    pg, ev_g = _hip_trajectory(SPEC_MID, video)
    pa = orc["psnr"]
    assert len(pa) == len(pg) == SPEC_MID["steps"]
    gap = [abs(a_ - g_) for a_, g_ in zip(pa, pg)]
    report("psnr_equal_steps_mid", steps=SPEC_MID["steps"], gap=max(gap), final_gap=gap[-1], eval_gap=abs(ev_g - orc["eval"]), final_psnr=pa[-1], first_psnr=pa[0])
    _say(f"{SPEC_MID['steps']}-step 16-level 256x256x32 batch=65536 train_gap_max={max(gap):.4f} final_gap={gap[-1]:.4f} "
         f"eval_hip-oracle={ev_g - orc['eval']:+.4f} dB (oracle PSNR {pa[0]:.2f} -> {pa[-1]:.2f})")
    assert pg[-1] > pg[0] + 1.0, "training did not make progress"
    assert max(gap) <= 0.02, f"train-PSNR gap {max(gap):.4f} dB"
    assert abs(ev_g - orc["eval"]) <= 0.02, f"eval-PSNR gap {abs(ev_g - orc['eval']):.4f} dB"


@pytest.mark.parametrize("which", ["s", "l"])
def test_psnr_at_equal_steps_real_config_size(which):
    """PSNR at equal step count against the ORACLE with the REAL models and batch of BASELINE.json configs[1] and configs[2] / [3] (VERDICT r5:
    at full size the trajectory comparison was HIP against HIP only): config_nvp_s with its real grids (135.8 M parameters; six steps,
    NVP_PSNR_STEPS_REAL) and config_nvp_l (F = 4, 228-row latent, 163.5 M parameters; three steps), N = 1 245 184 samples per step, the
    reference's loop - sampler, image_mse, AdamW + cosine - on small 600- / 300-frame clips; the oracle trains in the background job
    `traj_real` (tens of seconds per step).  +-0.02 dB on the train PSNR of every step and on the evaluation PSNR of the final parameters."""
    spec = SPEC_REAL if which == "s" else SPEC_REAL_L
    orc, video = _oracle_trajectory("traj_real", SPECS_REAL, spec)
    pg, ev_g = _hip_trajectory(spec, video)
    pa = orc["psnr"]
    assert len(pa) == len(pg) == spec["steps"]
    gap = [abs(a_ - g_) for a_, g_ in zip(pa, pg)]
    report("psnr_equal_steps_real", config=which, steps=spec["steps"], gap=max(gap), final_gap=gap[-1], eval_gap=abs(ev_g - orc["eval"]), final_psnr=pa[-1], first_psnr=pa[0])
    _say(f"{spec['steps']}-step config_nvp_{which} real grids N=1245184 train_gap_max={max(gap):.6f} eval_hip-oracle={ev_g - orc['eval']:+.6f} dB "
         f"(oracle PSNR {pa[0]:.3f} -> {pa[-1]:.3f})")
    assert pa[-1] > pa[0], "training did not make progress"
    assert max(gap) <= 0.02, f"train-PSNR gap {max(gap):.4f} dB"
    assert abs(ev_g - orc["eval"]) <= 0.02, f"eval-PSNR gap {abs(ev_g - orc['eval']):.4f} dB"


# ----------------------------------------------------------------------------------------
# FULL SIZE (configs[1]): the fp16x2 build against the fp32-MFMA twin over a 1 000-step schedule
# ----------------------------------------------------------------------------------------
def _start(tag, lib, steps, every, out, ulp=0):
    env = dict(os.environ)
    env.pop("NVP_HIP_LIB", None)
    if lib:
        env["NVP_HIP_LIB"] = lib
    if os.path.exists(out):
        os.remove(out)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "long_horizon.py"), "--steps", str(steps), "--every", str(every),
           "--video", "natural", "--tag", tag, "--out", out, "--ulp", str(ulp)]
    return tag, out, subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)


def _finish(job):
    tag, out, proc = job
    try:
        _, err = proc.communicate(timeout=1500)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, f"{tag}: {err[-2000:]}"
    recs = [json.loads(line) for line in open(out) if line.startswith("{")]
    return {x["step"]: x for x in recs if "step" in x}, [x for x in recs if x.get("summary")][0]


def test_fp16x2_split_tracks_the_fp32_mfma_twin_at_full_size(tmp_path):
    """Long-horizon, FULL-SIZE parity of the split-operand 16-bit MFMA arithmetic (-m gpu).

    The default build issues every fp32 MLP product as three fp16 MFMA products of a scaled hi+lo operand split
    (nvp_amd/csrc/mlp_b3.h); its twin libnvp_hip_fp32mfma.so (same sources, -DNVP_FWD_B3=0 -DNVP_BWD_B3=0 -DNVP_DW_B3=0, built
    next to it by nvp_amd/csrc/build.sh) runs the same GEMMs on v_mfma_f32_32x32x2_f32, which is bit-equal to an ordered fmaf
    chain.  Both are trained here on BASELINE.json configs[1] - config_nvp_s, 1920x1080x600, N = 1 245 184 samples per step, the
    reference's sampler / loss / AdamW + cosine (dataio.py:104-120, training.py:13-14,47-76) - from the same init on the same
    batches for NVP_LH_STEPS (default 1000) steps, in one process each (tools/long_horizon.py; the library is chosen at load
    time), together with the twin started <= 1 fp32 ulp away (the envelope).

    What is asserted.  At the END of the schedule (the cosine has annealed the step size) train PSNR (training.py:58) and
    full-frame evaluation PSNR (eval.py:243-256) of the two builds agree to the north-star bound of +-0.02 dB, unconditionally.
    At the intermediate checkpoints the optimiser runs at lr ~ 1e-2 and the instantaneous PSNR of ANY trajectory of this model
    jitters by ~0.1 dB from step to step (profiles/r03_long_horizon_5000_compare.txt: the fp32-MFMA build against its own 1-ulp
    twin differs by up to 0.10 dB train / 0.19 dB eval during the first 2000 of 5000 steps, and by <= 0.006 dB from step 2250 on;
    the fp16x2 build sits inside that envelope: 0.15 / 0.19 early, <= 0.005 dB from step 2250 on) - there the gap must stay within
    max(0.02 dB, 3 x the envelope measured in the same test), and the numbers are reported."""
    assert os.path.exists(TWIN), f"{TWIN} missing: run nvp_amd/csrc/build.sh (it builds the fp32-MFMA twin next to libnvp_hip.so)"
    steps = int(os.environ.get("NVP_LH_STEPS", "1000"))
    every = int(os.environ.get("NVP_LH_EVERY", "250"))
    # the three trainings are independent processes (13 GB of the 288 GB each): started together, their start-up, clip synthesis and
    # evaluation phases overlap; every one of them is bit-reproducible, so sharing the GPU changes no number
    jobs = [_start("f16x2", None, steps, every, str(tmp_path / "a.jsonl")), _start("fp32mfma", TWIN, steps, every, str(tmp_path / "b.jsonl")),
            _start("fp32mfma_1ulp", TWIN, steps, every, str(tmp_path / "c.jsonl"), ulp=1)]
    try:
        (a, sa), (b, sb), (c, sc) = (_finish(j) for j in jobs)
    finally:
        for _, _, pr in jobs:
            if pr.poll() is None:
                pr.kill()
    # the processes really ran different arithmetic on the same problem
    assert sa["mfma_products"] == 3 and sb["mfma_products"] == 1 and sc["mfma_products"] == 1, (sa["mfma_products"], sb["mfma_products"])
    assert sa["samples"] == sb["samples"] == 1245184 and sa["geometry"] == sb["geometry"] == [600, 1080, 1920]
    assert sorted(a) == sorted(b) == sorted(c) and max(a) == steps
    gap = {k: [abs(a[s_][k] - b[s_][k]) for s_ in sorted(a)] for k in ("train_psnr", "eval_psnr")}
    env = {k: [abs(c[s_][k] - b[s_][k]) for s_ in sorted(a)] for k in ("train_psnr", "eval_psnr")}
    report("long_horizon", steps=steps, checkpoints=sorted(a), train_gap=gap["train_psnr"], eval_gap=gap["eval_psnr"],
           train_envelope=env["train_psnr"], eval_envelope=env["eval_psnr"], final_train=a[steps]["train_psnr"], final_eval=a[steps]["eval_psnr"],
           wall_f16x2=sa["wall_s"], wall_fp32mfma=sb["wall_s"])
    _say(f"full-size {steps} steps fp16x2-vs-fp32mfma train_gap={gap['train_psnr']} eval_gap={gap['eval_psnr']} envelope_train={env['train_psnr']} envelope_eval={env['eval_psnr']}")
    assert a[steps]["eval_psnr"] > a[min(a)]["eval_psnr"] and a[steps]["eval_psnr"] > 20.0, "training did not make progress"
    for k in ("train_psnr", "eval_psnr"):
        assert gap[k][-1] <= 0.02, f"final {k} gap {gap[k][-1]:.4f} dB between the fp16x2 build and the fp32-MFMA twin after {steps} steps"
        bound = max(0.02, 3.0 * max(env[k]))
        assert max(gap[k]) <= bound, (f"{k} gap {max(gap[k]):.4f} dB at an intermediate checkpoint exceeds max(0.02, 3 x the 1-ulp envelope "
                                      f"{max(env[k]):.4f}) dB; gaps {gap[k]}, envelope {env[k]}")
