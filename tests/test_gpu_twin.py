"""The yardstick is checked too (-m gpu).

tests/test_gpu_zz_trajectories.py (full-size test) bounds the default build's split-operand arithmetic against libnvp_hip_fp32mfma.so (the same
sources with every MLP GEMM on v_mfma_f32_32x32x2_f32).  That twin is only a yardstick if it passes the ORACLE tests itself:
here the golden-vector MLP test, the whole-path forward/backward test and the real-config test of configs[1] run again in a
subprocess whose NVP_HIP_LIB points at the twin (the library is chosen at load time, once per process)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TWIN = os.path.join(ROOT, "nvp_amd", "csrc", "libnvp_hip_fp32mfma.so")
ORACLE_TESTS = [
    "tests/test_gpu_parity.py::test_mlp_golden",
    "tests/test_gpu_parity.py::test_nvp_forward_backward_vs_oracle[2-4096]",
    "tests/test_gpu_parity.py::test_nvp_forward_backward_vs_oracle[4-2048]",
    "tests/test_gpu_parity.py::test_e2e_minus_keyframes_golden_and_trajectory",
    "tests/test_gpu_real_configs.py::test_real_config_forward_backward_vs_oracle[configs1_nvp_s_1080p_x600]",
]


def _env():
    env = dict(os.environ)
    env["NVP_HIP_LIB"] = TWIN
    env.pop("NVP_PARITY_REPORT", None)
    return env


def test_twin_is_the_fp32_mfma_build():
    assert os.path.exists(TWIN), f"{TWIN} missing: run nvp_amd/csrc/build.sh"
    r = subprocess.run([sys.executable, "-c", "from nvp_amd import _lib; l = _lib.load(); print(_lib.LIB_PATH, l.nvp_mlp_mfma_products())"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    path, products = r.stdout.split()[-2:]
    assert path == TWIN and products == "1", r.stdout


def test_fp32_mfma_twin_passes_the_oracle_tests():
    assert os.path.exists(TWIN), f"{TWIN} missing: run nvp_amd/csrc/build.sh"
    # (no --timeout flag: it belongs to the pytest-timeout plugin, which a GPU image need not carry; subprocess.run bounds the run)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + ORACLE_TESTS,
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1000:])
    tail = r.stdout.strip().splitlines()[-1]
    assert " passed" in tail and "skipped" not in tail and "deselected" not in tail, tail
