"""bench.py's supervisor: a worker that stops announcing progress is killed (exact PID) and reported as stalled; a worker
that exits is reported with its return code.  CPU-only: the workers here are tiny scripts speaking the heartbeat protocol."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

WORKER = r"""
import os, sys, time
fd = int(os.environ["PASNL_BENCH_HEARTBEAT_FD"])
for phase, secs, nap in {plan}:
    os.write(fd, f"{{phase}} {{secs}}\n".encode())
    time.sleep(nap)
sys.exit({rc})
"""


def run(plan, rc=0, first=5.0):
    t0 = time.perf_counter()
    out = bench.watch([sys.executable, "-c", WORKER.format(plan=plan, rc=rc)], os.environ, first_allowance=first)
    return out, time.perf_counter() - t0


def test_worker_that_finishes_returns_its_code():
    assert run([("setup", 5, 0.05), ("run", 5, 0.05), ("post", 5, 0.0)])[0] == (0, None)
    assert run([("setup", 5, 0.0)], rc=7)[0] == (7, None)


def test_stalled_worker_is_killed_and_phase_reported():
    (rc, phase), dt = run([("setup", 5, 0.05), ("run", 0.5, 60)])
    assert rc is None and phase == "run"
    assert dt < 10, "the supervisor must not wait for the worker's own sleep"


def test_silent_worker_hits_the_start_allowance():
    rc, phase = bench.watch([sys.executable, "-c", "import time; time.sleep(60)"], os.environ, first_allowance=0.5)
    assert rc is None and phase == "start"


def test_allowance_follows_the_latest_heartbeat():
    # a long allowance announced first, then a short one: the short one governs
    (rc, phase), dt = run([("setup", 30, 0.05), ("run", 0.3, 60)])
    assert (rc, phase) == (None, "run") and dt < 10


def test_bench_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip(), "no GPU: no JSON line, non-zero exit"


@pytest.mark.gpu
def test_stalled_pipelined_run_falls_back_to_serial():
    """The first worker hangs in the watched region (test hook); the supervisor kills it and the serial retry delivers
    the JSON line, marked as a retry."""
    env = dict(os.environ, PASNL_BENCH_FAKE_STALL="run", PASNL_BENCH_STALL_SCALE="0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["config"]["pipeline"] == "serial" and line["config"]["retry_of_stalled_phase"] == "run"
    assert "retrying with --pipeline serial" in p.stderr
    assert line["value"] > 0
