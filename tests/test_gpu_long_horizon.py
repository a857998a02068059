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
max(0.02 dB, 3 x the envelope measured in the same test), and the numbers are reported.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, report

pytestmark = pytest.mark.gpu

TWIN = os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip_fp32mfma.so")


def _run(tag, lib, steps, every, out, ulp=0):
    env = dict(os.environ)
    env.pop("NVP_HIP_LIB", None)
    if lib:
        env["NVP_HIP_LIB"] = lib
    if os.path.exists(out):
        os.remove(out)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "long_horizon.py"), "--steps", str(steps), "--every", str(every),
           "--video", "natural", "--tag", tag, "--out", out, "--ulp", str(ulp)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, f"{tag}: {r.stderr[-2000:]}"
    recs = [json.loads(line) for line in open(out) if line.startswith("{")]
    return {x["step"]: x for x in recs if "step" in x}, [x for x in recs if x.get("summary")][0]


def test_fp16x2_split_tracks_the_fp32_mfma_twin_at_full_size(tmp_path):
    assert os.path.exists(TWIN), f"{TWIN} missing: run nvp_amd/csrc/build.sh (it builds the fp32-MFMA twin next to libnvp_hip.so)"
    steps = int(os.environ.get("NVP_LH_STEPS", "1000"))
    every = int(os.environ.get("NVP_LH_EVERY", "250"))
    a, sa = _run("f16x2", None, steps, every, str(tmp_path / "a.jsonl"))
    b, sb = _run("fp32mfma", TWIN, steps, every, str(tmp_path / "b.jsonl"))
    c, sc = _run("fp32mfma_1ulp", TWIN, steps, every, str(tmp_path / "c.jsonl"), ulp=1)
    # the processes really ran different arithmetic on the same problem
    assert sa["mfma_products"] == 3 and sb["mfma_products"] == 1 and sc["mfma_products"] == 1, (sa["mfma_products"], sb["mfma_products"])
    assert sa["samples"] == sb["samples"] == 1245184 and sa["geometry"] == sb["geometry"] == [600, 1080, 1920]
    assert sorted(a) == sorted(b) == sorted(c) and max(a) == steps
    gap = {k: [abs(a[s_][k] - b[s_][k]) for s_ in sorted(a)] for k in ("train_psnr", "eval_psnr")}
    env = {k: [abs(c[s_][k] - b[s_][k]) for s_ in sorted(a)] for k in ("train_psnr", "eval_psnr")}
    report("long_horizon", steps=steps, checkpoints=sorted(a), train_gap=gap["train_psnr"], eval_gap=gap["eval_psnr"],
           train_envelope=env["train_psnr"], eval_envelope=env["eval_psnr"], final_train=a[steps]["train_psnr"], final_eval=a[steps]["eval_psnr"],
           wall_f16x2=sa["wall_s"], wall_fp32mfma=sb["wall_s"])
    assert a[steps]["eval_psnr"] > a[min(a)]["eval_psnr"] and a[steps]["eval_psnr"] > 20.0, "training did not make progress"
    for k in ("train_psnr", "eval_psnr"):
        assert gap[k][-1] <= 0.02, f"final {k} gap {gap[k][-1]:.4f} dB between the fp16x2 build and the fp32-MFMA twin after {steps} steps"
        bound = max(0.02, 3.0 * max(env[k]))
        assert max(gap[k]) <= bound, (f"{k} gap {max(gap[k]):.4f} dB at an intermediate checkpoint exceeds max(0.02, 3 x the 1-ulp envelope "
                                      f"{max(env[k]):.4f}) dB; gaps {gap[k]}, envelope {env[k]}")
