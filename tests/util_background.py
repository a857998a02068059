"""Background jobs of the GPU suite: the CPU ORACLE's long trainings, started when collection has finished and running in
processes of their own UNDERNEATH the deterministic parity tests (the suite's wall clock was dominated by
oracle trainings the GPU tests only wait for).

The trajectory tests (tests/test_gpu_zz_trajectories.py, collected last) compare the HIP path with oracle trainings of
hundreds of steps.  Those trainings depend on nothing the GPU computes: every one is a seeded, deterministic CPU program
(fixed thread count, torch's deterministic algorithms, MKL_CBWR: util_windows.oracle_env) whose result is bit-identical
wherever and whenever it runs.  So they are launched at the start of the session and the tests pick the results up:

    start(name, argv)      idempotent; argv is run with the oracle's environment, stdout / stderr into the job's directory
    result(name)           waits for the job (starting it now if nobody did: a test run on its own), returns its directory

Nothing here touches the GPU or the product library.  Test infrastructure only."""
from __future__ import annotations

import atexit
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_JOBS = {}          # name -> {"proc", "dir", "argv", "t0"}
_BASE = None


def _base() -> str:
    global _BASE
    if _BASE is None:
        _BASE = tempfile.mkdtemp(prefix="nvp_oracle_jobs_")
        atexit.register(cleanup)
    return _BASE


def job_dir(name: str) -> str:
    d = os.path.join(_base(), name)
    os.makedirs(d, exist_ok=True)
    return d


def start(name: str, argv: list, env: dict = None) -> str:
    """Start `python argv...` as job `name` unless it already runs / ran; returns the job's directory."""
    if name in _JOBS:
        return _JOBS[name]["dir"]
    from util_windows import oracle_env
    d = job_dir(name)
    out = open(os.path.join(d, "stdout.txt"), "w")
    err = open(os.path.join(d, "stderr.txt"), "w")
    e = dict(env or oracle_env())
    e["HIP_VISIBLE_DEVICES"] = ""                   # the oracle is a CPU program: it must not create a HIP context next to the tests'
    p = subprocess.Popen([sys.executable] + list(argv), cwd=ROOT, env=e, stdout=out, stderr=err)
    _JOBS[name] = {"proc": p, "dir": d, "argv": list(argv), "t0": time.time()}
    return d


def result(name: str, argv: list = None, timeout: float = 3000.0) -> str:
    """Directory of the finished job `name`; starts it first if it was never started (then `argv` is required)."""
    if name not in _JOBS:
        if argv is None:
            raise KeyError(f"background job {name!r} was never started and no command was given")
        start(name, argv)
    j = _JOBS[name]
    try:
        rc = j["proc"].wait(timeout=max(1.0, timeout - (time.time() - j["t0"])))
    except subprocess.TimeoutExpired:
        j["proc"].kill()
        raise AssertionError(f"background oracle job {name!r} did not finish within {timeout:.0f} s: {' '.join(j['argv'])}")
    j["waited_s"] = round(time.time() - j["t0"], 1)
    if rc != 0:
        tail = open(os.path.join(j["dir"], "stderr.txt")).read()[-2000:]
        raise AssertionError(f"background oracle job {name!r} failed (rc {rc}): {tail}")
    return j["dir"]


def cleanup() -> None:
    for j in _JOBS.values():
        if j["proc"].poll() is None:
            j["proc"].kill()                         # exactly the processes started here
    _JOBS.clear()
    global _BASE
    if _BASE and os.path.isdir(_BASE):
        shutil.rmtree(_BASE, ignore_errors=True)
    _BASE = None
