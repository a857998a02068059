import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_THREADS = 8
# Thread count of the oracle's background TRAININGS (util_windows.py --oracle, util_traj.py).  The GPU boxes give a container a CPU
# quota of 16 cores (cgroup cpu.max) on a 256-thread host: a training of the small problem (8 192-pixel batches) gains nothing from 16
# threads (fork-join per small op), and several trainings share the quota.  Fixed (not host-dependent) for the same reason as above.
# Measured on an MI355X host (gpurun_out/r06_oracle_threads.txt -> profiles/): 100 steps of the windowed walk with two controls take 20.7 /
# 22.1 / 24.1 / 29.5 s on 16 / 8 / 4 / 2 threads; four walks at once on 4 threads each: 25.1 s for all four.  Two threads per training
# leave half of the quota to the foreground tests while four trainings run underneath them.
ORACLE_TRAIN_THREADS = int(os.environ.get("NVP_ORACLE_TRAIN_THREADS", "2"))
# ... the two 1 000-step windowed walks (3 000 oracle steps each: the suite's critical path) get four.  The thread count is a scheduling
# choice only: under deterministic algorithms + MKL_CBWR the walk's numbers came out identical to the last digit on 16 and on 2 threads
# (profiles/r05_psnr_windows_seeds_boxes.txt vs gpurun r06b: max product gap 0.0066 / 0.0185 dB, envelope 0.0207 / 0.0123 for seeds 7 / 8).
ORACLE_WALK_THREADS = int(os.environ.get("NVP_ORACLE_WALK_THREADS", "4"))


class oracle_determinism:
    """Context for the oracle's TRAININGS (chaotic comparisons): fixed intra-op thread count and torch's deterministic algorithms,
    so the checker's trajectory is reproducible run to run and box to box.  Scoped, because deterministic mode makes some ATen
    device kernels used elsewhere in the suite raise."""

    def __init__(self, threads=None):
        self.threads = threads or ORACLE_THREADS

    def __enter__(self):
        import torch
        self._n, self._det = torch.get_num_threads(), torch.are_deterministic_algorithms_enabled()
        torch.set_num_threads(self.threads)
        torch.use_deterministic_algorithms(True)
        return self

    def __exit__(self, *exc):
        import torch
        torch.use_deterministic_algorithms(self._det)
        torch.set_num_threads(self._n)
        return False


_CONFIG = None


def say(line: str) -> None:
    """One summary line on the REAL stdout, past pytest's capture, for passing tests too: the driver keeps the tail of the run's output,
    and the numbers behind a statistical or arithmetic claim belong there."""
    capman = _CONFIG.pluginmanager.getplugin("capturemanager") if _CONFIG is not None else None
    if capman is not None:
        with capman.global_and_fixture_disabled():
            print("\n" + line, flush=True)
    else:
        print("\n" + line, flush=True)


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is the CHECKER: its results must not depend on the host it runs on.  ATen's CPU kernels partition their
    # reductions by the intra-op thread count, so the count is FIXED (not "whatever the box has, capped"): the builder's box and the
    # driver's box then walk the same oracle trajectory.  8: the GPU boxes give the container a quota of 16 CPUs (cgroup cpu.max) of a
    # 256-thread host, the oracle's background trainings (util_background.py) take half of it, and bench.py's thread sweep measures the
    # oracle's forward + backward fastest on 8 threads there (1.53 s against 1.81 s on 16, 27 s on 128 for 155 648 pixels).
    import torch
    torch.set_num_threads(ORACLE_THREADS)
    os.environ.setdefault("NVP_QUIET", "1")        # NVP.__init__ prints the module tree like the reference (modules.py:49): keep the driver's tail for numbers


def pytest_collection_modifyitems(config, items):
    # Deterministic parity first: every multi-hundred-step TRAJECTORY test (statistical margins on a chaotic comparison) is collected
    # after everything else, whatever its file is called - a margin can then never hide a deterministic test behind `-x`.
    # Before them: golden-vector and oracle parity (test_gpu_parity), the real configuration sizes, the twin, then the multi-process tests.
    order = ("test_gpu_parity", "test_gpu_real_configs", "test_gpu_twin", "test_gpu_dp2")
    # inside the trajectory group: first what needs no oracle job (the full-size fp16x2-vs-twin run), last what waits for the longest ones
    # (the 1 000-step windowed walks): the background trainings started at collection time have the whole suite to finish in
    zz = ("test_fp16x2_split_tracks", "test_psnr_at_equal_steps_matches_oracle", "test_psnr_at_equal_steps_full_levels", "test_psnr_at_equal_steps_larger_problem",
          "test_psnr_at_equal_steps_real_config_size", "test_psnr_tracks_the_oracle")

    def rank(it):
        if "zz_trajectories" in it.nodeid:
            return (len(order) + 1, next((i for i, n in enumerate(zz) if n in it.nodeid), len(zz)))
        return (next((i for i, n in enumerate(order) if n in it.nodeid), len(order)), 0)
    items.sort(key=rank)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_collection_finish(session):
    """The CPU oracle's long trainings start NOW, as background processes underneath the deterministic parity tests (VERDICT r5 item 3;
    tests/util_background.py): the trajectory tests, collected last, only pick their results up."""
    import torch
    if not torch.cuda.is_available() or session.config.option.collectonly:
        return
    ids = [it.nodeid for it in session.items if "zz_trajectories" in it.nodeid and not any(m.name == "skip" for m in it.iter_markers())]
    if ids:
        import test_gpu_zz_trajectories as zz
        zz.background_jobs(ids)


def pytest_sessionfinish(session, exitstatus):
    if "util_background" in sys.modules:
        sys.modules["util_background"].cleanup()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def small_cfg(F=2, T=8, X=9, Y=7, n_levels=16):
    """config_nvp_s / config_nvp_l values (reference config/*.json) with a small sparse grid."""
    enc = {"otype": "DenseGrid", "n_levels": n_levels, "n_features_per_level": F, "log2_hashmap_size": 32,
           "base_resolution": 16, "per_level_scale": 1.35}
    return {
        "2d_encoding_xy": dict(enc), "2d_encoding_xt": dict(enc), "2d_encoding_yt": dict(enc),
        "3d_encoding": {"otype": "SparseGrid", "n_features_per_level": F, "x_resolution": X, "y_resolution": Y,
                        "t_resolution": T, "upsample": False},
        "network": {"n_neurons": 128, "n_hidden_layers": 3},
    }


def full_cfg(F=2, t_res=600, **enc_extra):
    """The VALUES of the reference's config/config_nvp_s.json (F=2) / config_nvp_l.json (F=4): 16 levels, base 16, scale 1.35,
    sparse grid 300 x 300 x t_res (README.md:55 sets t_resolution to the clip's frame count), 128 x 3 network."""
    cfg = small_cfg(F=F, T=t_res, X=300, Y=300)
    for k in ("2d_encoding_xy", "2d_encoding_xt", "2d_encoding_yt"):
        cfg[k].update(enc_extra)
    return cfg


def relerr_max(a, b):
    """max |a - b| / max |b|"""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def relerr_l2(a, b):
    """||a - b||_2 / ||b||_2"""
    import numpy as np
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-300))


def report(test: str, **vals):
    """Append measured errors to gpurun_out/parity_report.jsonl (NVP_PARITY_REPORT=1): the numbers the tolerances in the
    GPU tests are set from (~3x the measured error)."""
    if os.environ.get("NVP_PARITY_REPORT", "0") != "1":
        return
    import json
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"), "a") as f:
        def plain(v):
            if isinstance(v, (list, tuple)):
                return [plain(x) for x in v]
            try:
                return float(v) if not isinstance(v, str) else v
            except (TypeError, ValueError):
                return str(v)
        f.write(json.dumps({"test": test, **{k: plain(v) for k, v in vals.items()}}) + "\n")
