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


def _bench(*flags, env=None, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})))


def test_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` (no torchrun): the launcher forks two ranks with RANK / WORLD_SIZE / MASTER_* set, they
    rendezvous (gloo here, RCCL on the GPU box), all-gather every step, and rank 0 prints ONE line that says n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "none",
                        "--steps", "3", "--warmup", "1", "--batch", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["allreduce_check"] == 3.0
    assert line["config"]["global_batch"] == 16 and line["steps"] == 3 and line["warmup"] == 1


def test_same_worker_under_torch_distributed_run():
    """The driver's launch form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ..."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--model", "none", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_refuses_instead_of_degrading():
    import torch

    p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()  # --gpus 2 inside a 1-rank job
    if torch.cuda.device_count() < 2:
        p = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
        assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()  # more ranks than visible devices


@pytest.mark.gpu
def test_stalled_run_is_killed_not_hung():
    """The worker hangs in the watched region (test hook); the supervisor kills it (exact PID), prints no JSON line and
    exits with code 3 long before the worker's own sleep would end."""
    env = dict(os.environ, PASNL_BENCH_FAKE_STALL="run", PASNL_BENCH_STALL_SCALE="0.1")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--no-others"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "all workers were killed" in p.stderr and time.perf_counter() - t0 < 400


@pytest.mark.gpu
def test_rccl_path_with_one_rank():
    """Everything of the multi-rank path but the wire, on one GPU: RCCL init, the all-reduce check, the per-step
    all-gather of the logits (captured forward + collective on the same stream), barriers, max-over-ranks timing."""
    p = _bench("--gpus", "1", "--force-dist", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["rccl_ranks"] == 1 and cfg["allreduce_check"] == 1.0
    assert cfg["gathered_rows_match_local"] is True and cfg["outputs_agree"] is True and line["value"] > 0


@pytest.mark.gpu
def test_rccl_path_sem_seg_res_gather_is_asynchronous():
    """configs[4] on the multi-rank path with one rank: the per-step all-gather moves the (8, 10240 x 20) segmentation logits
    (6.5 MB per rank, not the 10 KB of the classifier) behind the replayed graph, and the host only ENQUEUES a step: issuing
    the steps takes a fraction of the time the GPU needs for them, so there is no host synchronisation inside a step."""
    p = _bench("--gpus", "1", "--force-dist", "--model", "sem_seg_res", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 1 and cfg["gathered_rows_match_local"] is True and cfg["outputs_agree"] is True
    assert cfg["global_batch"] == 8 and "sem_seg_res" in line["metric"]
    assert cfg["enqueue_ms_per_step"] < 0.5 * line["ms_per_step"], (cfg["enqueue_ms_per_step"], line["ms_per_step"])


@pytest.mark.gpu
def test_two_ranks_over_rccl():
    """Two ranks on two GPUs (skipped on a 1-GPU box): n_gpus, RCCL world size, all-reduce over both ranks, every rank's
    rows of the gathered logits equal its own logits, throughput counts both shards."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    p = _bench("--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["rccl_ranks"] == 2 and cfg["allreduce_check"] == 3.0 and cfg["global_batch"] == 128
    assert cfg["gathered_rows_match_local"] is True and cfg["shards_differ"] is True  # per-rank seeds, rows arrive intact
    one = _bench("--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-others", timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    v1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])["value"]
    assert line["value"] > 1.2 * v1, (line["value"], v1)  # two shards in about the time of one


@pytest.mark.parametrize("shape,width", [("cls", 40), ("sem_seg_res", 10240 * 20)])
def test_eight_ranks_protocol_line_carries_every_multi_rank_key(shape, width):
    """VERDICT r05 #8 (first-run insurance for the 8-GPU scaling bench): `python bench.py --gpus 8 --model none` starts eight
    ranks (gloo here), every step all-gathers stand-in logits of the model's width, and rank 0's ONE line already carries what
    the real 8-GPU line carries: n_gpus, the max-over-ranks step time, every rank's own step time, the gathered-rows and
    shards-differ checks, the NUMA binding fields, the all-reduce over all ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--model", "none",
                        "--proto-shape", shape, "--steps", "3", "--warmup", "1", "--batch", "2"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["rccl_ranks"] == 8 and cfg["allreduce_check"] == 36.0 and cfg["global_batch"] == 16
    assert len(cfg["per_rank_ms_per_step"]) == 8 and all(v > 0 for v in cfg["per_rank_ms_per_step"])
    assert line["ms_per_step"] >= max(cfg["per_rank_ms_per_step"]) - 1e-3   # the line's figure is the slowest rank's
    assert cfg["gathered_rows_match_local"] is True and cfg["shards_differ"] is True
    assert "numa_node" in cfg and "cpus_bound" in cfg and str(width) in cfg["parallelism"]
    assert line["scaling"] == "weak" and line["value"] > 0 and line["unit"] == "point-clouds/s"


@pytest.mark.gpu
def test_one_rank_through_the_multi_rank_path_agrees_with_the_plain_line():
    """--gpus 1 --force-dist (RCCL init, per-step all-gather, barriers, max-over-ranks) against the plain N = 1 line of the same
    process layout: within 2 %, so that the first SCALE record is comparable with BENCH by construction."""
    a = _bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-others", timeout=900)
    b = _bench("--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-others", timeout=900)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1500:], b.stderr[-1500:])
    la = json.loads([l for l in a.stdout.splitlines() if l.startswith("{")][-1])
    lb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    ma, mb = la["ms_per_step_blocks"]["median"], lb["ms_per_step_blocks"]["median"]
    assert abs(ma - mb) / ma < 0.02, (ma, mb)
    assert len(lb["config"]["per_rank_ms_per_step"]) == 1
