"""CPU tests of the trajectory checker itself (tests/util_windows.py): the verdict rule, and that the oracle walk is reproducible
run to run under the environment the graded test gives it."""
import json
import os
import subprocess
import sys

from conftest import ORACLE_TRAIN_THREADS, ROOT
from util_windows import oracle_env, window_verdicts


def test_window_rule():
    calm, hot = 0.004, 0.03
    assert window_verdicts([0.019], [0.001], [calm]) == []                       # north_star's bound where the oracle reproduces itself
    assert window_verdicts([0.0185], [0.0023], [0.0123]) == []                   # ... and a hot window is never STRICTER than the calm bound (seed 8, window 5 on MI355X)
    bad = window_verdicts([0.021], [0.001], [calm])
    assert len(bad) == 1 and "fp16x2 split-operand arithmetic is the cause" in bad[0][1]
    bad = window_verdicts([0.021], [0.025], [calm])
    assert len(bad) == 1 and "the twin leaves too" in bad[0][1]
    assert window_verdicts([0.034], [0.03], [hot]) == []                         # hot window: with the twin (+0.005) and inside 2 x envelope
    assert len(window_verdicts([0.036], [0.03], [hot])) == 1                     # leaves the twin
    assert len(window_verdicts([0.07], [0.07], [hot])) == 1                      # with the twin but outside 2 x envelope
    assert [w for w, _ in window_verdicts([0.0, 0.5, 0.0], [0.0] * 3, [0.0, 0.1, 0.0])] == [1]
    # absolute guards - a gap above 0.1 dB is a violation however hot the window and however close the twin ...
    bad = window_verdicts([0.15], [0.15], [0.2])
    assert len(bad) == 1 and "absolute ceiling" in bad[0][1]
    assert window_verdicts([0.09], [0.088], [0.05]) == []
    # ... and the tolerant rule may cover at most a quarter of the schedule
    env = [0.03] * 6 + [0.001] * 14
    bad = window_verdicts([0.001] * 20, [0.001] * 20, env)
    assert len(bad) == 1 and bad[0][0] == -1 and "6 of 20 windows are hot" in bad[0][1]
    assert window_verdicts([0.001] * 20, [0.001] * 20, [0.03] * 5 + [0.001] * 15) == []


def test_oracle_walk_is_reproducible(tmp_path):
    outs = []
    for k in range(2):
        d = str(tmp_path / f"w{k}")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "util_windows.py"), "--oracle", "1", "--dir", d, "--out", os.path.join(d, "o.json"),
                            "--seed", "7", "--steps", "4", "--window", "2", "--controls", "1"], cwd=ROOT, env=oracle_env(), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.load(open(os.path.join(d, "o.json"))))
        assert sorted(f for f in os.listdir(d) if f.startswith("win_")) == ["win_000.pt", "win_001.pt"]
    assert outs[0]["psnr"] == outs[1]["psnr"] and outs[0]["controls"] == outs[1]["controls"] and outs[0]["eval"] == outs[1]["eval"]
    assert outs[0]["threads"] == ORACLE_TRAIN_THREADS and len(outs[0]["lr"]) == 4


def test_background_oracle_job_runs_and_is_reproducible():
    """tests/util_background.py + tests/util_traj.py (the oracle's free-running trainings as background processes of the GPU suite): a tiny
    job started twice under different names gives identical numbers (fixed threads, deterministic algorithms, MKL_CBWR), writes the clip the
    GPU side loads, reports failures with the job's stderr, and cleanup() leaves no process behind."""
    import pytest
    import torch
    import util_background as bg
    spec = [{"name": "t", "seed": 3, "video_seed": 3, "gen_seed": 3, "steps": 3, "n_levels": 12, "ulp_twin": True, "clip": "procedural"}]
    try:
        outs = []
        for name in ("job_a", "job_b"):
            d = bg.start(name, [os.path.join(ROOT, "tests", "util_traj.py"), "--dir", bg.job_dir(name), "--specs", json.dumps(spec)])
            assert bg.start(name, ["never", "run"]) == d                       # idempotent: the second start is ignored
        for name in ("job_a", "job_b"):
            d = bg.result(name, timeout=600)
            outs.append(json.load(open(os.path.join(d, "traj_t.json"))))
            assert tuple(torch.load(os.path.join(d, "video_t.pt")).shape) == (16, 64, 64, 3)
        assert outs[0]["psnr"] == outs[1]["psnr"] and outs[0]["psnr_1ulp"] == outs[1]["psnr_1ulp"] and outs[0]["eval"] == outs[1]["eval"]
        assert len(outs[0]["psnr"]) == 3 and outs[0]["threads"] == ORACLE_TRAIN_THREADS
        bg.start("job_bad", [os.path.join(ROOT, "tests", "util_traj.py"), "--dir", bg.job_dir("job_bad"), "--specs", "not json"])
        with pytest.raises(AssertionError, match="job_bad"):
            bg.result("job_bad", timeout=300)
        with pytest.raises(KeyError):
            bg.result("never_started")
    finally:
        procs = [j["proc"] for j in bg._JOBS.values()]
        bg.cleanup()
        assert all(p.poll() is not None or p.wait(10) is not None for p in procs) and not bg._JOBS
