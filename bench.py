#!/usr/bin/env python
"""bench.py -- ModelNet40 pointasnl_cls forward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by torch.distributed.run, one rank per GPU, RCCL; `python bench.py --gpus N` starts the ranks itself)

A "step" is one inference forward of models/pointasnl_cls.py over one batch of synthetic clouds
(B=64 x 1024 x 3 per GPU, BASELINE.json configs[1]); inputs are resident in HBM before the timed region.  The forward
is captured into ONE HIP graph and replayed (what can overlap inside a forward is forked onto side streams by the
models).  With N>1 every rank owns its own 64 clouds (weak scaling, no data-path collective) and the per-shard logits
are all-gathered over RCCL each step.

Rank 0 prints ONE JSON line.  `roofline` describes the hand-written kernel that takes the largest share of the
step: algorithmic bytes/flops / its average launch duration, measured with HIP events on the
launch stream in an event-instrumented pass of the same forward.  `cpu_baseline` times the CPU restatement of
the same forward on the host cores over a bounded sample.  `kernels` lists every hand-written kernel the same way.
At N=1 the same process then measures the other BASELINE configurations the same way (captured, replayed, timed,
checked against the eager outputs) -> `other_configs`: configs[2] (cls --AS, 10 outliers per cloud), configs[3]
(pointasnl_sem_seg, 16 x 8192) and the per-GPU share of configs[4] (pointasnl_sem_seg_res, 8 x 10240); and the
north-star operator sweep -> `ball_query_sweep` (query_ball_point at (B,1024)/(B,512), nsample 32, B = 64..4096).

Every rank's measurement runs in a worker process under a thin supervisor (`supervise`): the worker reports its phase
over a pipe, and a job in which a worker makes no progress within the phase's allowance is killed (exact PIDs) instead
of hanging the caller.
"""
import argparse
import json
import os
import select
import signal
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and the forks of a captured
# forward that land on one queue serialise.  With four queues the mapping that RCCL's own streams leave for the forward is a bad one:
# the SAME captured step took 1.75-1.80 ms instead of 1.32 once a process group existed -- no collective in the step --, which the
# driver's N = 2 run would have shown as a 25 % scaling loss.  With three queues every workload runs as with four (cls 1.322 /
# 1.322, sem_seg 4.39 / 4.37, sem_seg_res 2.376 / 2.376 ms) and RCCL's presence changes nothing (1.32; sem_seg_res 2.375):
# profiles/r06_hw_queues.txt.  Read by the HIP runtime when it initialises, so main() sets it before torch is imported -- in
# MULTI-RANK runs only (a plain N = 1 run keeps the runtime's default, under which every number of rounds 1-5 was taken; a value the
# caller exported wins).  Recorded in config.hip_hw_queues.
MULTI_RANK_HW_QUEUES = "3"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 matrix (= vector) peak


def synth_clouds(seed, b, n):
    """uniform in the unit ball, then pc_normalize (zero mean, max norm 1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    v = rng.standard_normal((b, n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    pc = v * rng.random((b, n, 1)) ** (1 / 3)
    pc -= pc.mean(axis=1, keepdims=True)
    pc /= np.linalg.norm(pc, axis=-1).max(axis=1)[:, None, None]
    return pc.astype(np.float32)


def normalize_data(batch):
    """provider.normalize_data (utils/provider.py:8-24): per cloud, centroid to the origin, max norm 1."""
    batch = np.asarray(batch, dtype=np.float64)
    batch = batch - batch.mean(axis=1, keepdims=True)
    return batch / np.sqrt((batch ** 2).sum(-1)).max(axis=1)[:, None, None]


def add_noise(pc, noise, seed):
    """configs[2]: overwrite the first `noise` points per cloud with uniform outliers, normalised among themselves
    like the reference does (test.py:128-132: np.random.random((bsize, noise, 3)) -> provider.normalize_data)."""
    if noise <= 0:
        return pc
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    pc = pc.copy()
    pc[:, :noise, :] = normalize_data(rng.random((pc.shape[0], noise, 3))).astype(np.float32)
    return pc


def synth_scannet(seed, b, n):
    """SURVEY 8(d) C4 (configs[3]): xyz uniform in a 1.5 x 1.5 x 3 m block then normalize_data (train_scannet.py:301),
    rgb uniform in [0,1) -> (b, n, 6) float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = rng.random((b, n, 3)) * np.array([1.5, 1.5, 3.0])
    rgb = rng.random((b, n, 3))
    return np.concatenate([normalize_data(xyz), rgb], axis=-1).astype(np.float32)


def synth_kitti(seed, b, n):
    """SURVEY 8(d) C5 (configs[4]): per cloud the n points nearest to a random centre of a synthetic lidar scan
    (ground z ~ N(0, 0.02) with 1/r density + a few vertical walls), voxel-snapped at 0.06 m to one point per voxel
    (mimics grid_subsampling + crop_pc, semantic_kitti_dataset_grid.py:265-286); metres, NOT normalised -> (b, n, 3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((b, n, 3), np.float32)
    for i in range(b):
        m = 6 * n
        r = 2.0 + 38.0 * rng.random(m) ** 2          # dense near the sensor
        th = rng.random(m) * 2 * np.pi
        pts = np.stack([r * np.cos(th), r * np.sin(th), rng.standard_normal(m) * 0.02], 1)
        for _ in range(6):                            # walls / car sides: vertical rectangles
            c = (rng.random(2) - 0.5) * 40
            d = rng.random() * np.pi
            u = (rng.random(m // 12) - 0.5) * 8
            w = np.stack([c[0] + u * np.cos(d), c[1] + u * np.sin(d), rng.random(m // 12) * 2.0], 1)
            pts = np.concatenate([pts, w])
        vox = np.floor(pts / 0.06).astype(np.int64)
        _, first = np.unique(vox, axis=0, return_index=True)
        pts = pts[np.sort(first)]
        centre = pts[rng.integers(0, len(pts))]
        near = np.argsort(((pts - centre) ** 2).sum(1), kind="stable")[:n]
        sel = pts[near[rng.permutation(len(near))]]   # crop_pc shuffles the selection
        if len(sel) < n:
            sel = np.concatenate([sel, sel[rng.integers(0, len(sel), n - len(sel))]])
        out[i] = sel.astype(np.float32)
    return out


# ---- algorithmic work per launch, SURVEY.md 8(d).  ints = the integer arguments of the C-ABI call.
def algorithmic(symbol, ints):
    """-> (bytes, flops, bound) for one launch"""
    if symbol in ("pasnl_farthest_point_sample", "pasnl_farthest_point_sample_gather"):
        b, n, m = ints
        out = 4 * b * m + (12 * b * m if symbol.endswith("_gather") else 0)
        return 12 * b * n + out, 10 * b * n * m, "latency"  # m dependent rounds per cloud: HBM fraction ~0 by construction
    if symbol == "pasnl_gather_point":
        b, n, m = ints
        return 28 * b * m, 0, "hbm"
    if symbol == "pasnl_group_point":
        b, n, c, m, ns = ints
        return 4 * b * (n * c + m * ns + m * ns * c), 0, "hbm"
    if symbol in ("pasnl_knn_batch", "pasnl_knn_batch_ws", "pasnl_knn_batch_ws_bg", "pasnl_knn_batch_ref"):
        b, n, m, k = ints[:4]
        return 12 * b * (n + m) + 4 * b * m * k, 8 * b * n * m, "valu"  # a search: vector-issue bound
    if symbol == "pasnl_knn_batch_tree":
        b, n, m, k = ints[:4]
        return 12 * b * (n + m) + 4 * b * m * k, 8 * b * n * m, "valu"
    if symbol in ("pasnl_dense_splitk", "pasnl_dense_splitk_workspace"):
        rows, k, n = ints[:3]
        return 4 * (rows * k + k * n + n + rows * n), 2 * rows * k * n, "mfma"
    if symbol == "pasnl_knn_crop":
        b, n = ints[:2]
        kcap = ints[3]
        return 12 * b * n + 4 * b * kcap, 8 * b * n, "hbm"  # one pass over the scan; the five key passes stay in L2
    if symbol == "pasnl_query_ball_point":
        b, n, m, ns = ints
        return 12 * b * (n + m) + 4 * b * m * (ns + 1), 10 * b * n * m, "hbm"
    if symbol == "pasnl_three_nn":
        b, n, m = ints
        return 12 * b * (n + m) + 24 * b * n, 8 * b * n * m, "valu"
    if symbol == "pasnl_three_interpolate":
        b, m, c, n = ints
        return 24 * b * n + 4 * b * c * (m + n), 5 * b * n * c, "hbm"
    if symbol == "pasnl_three_weights":
        (rows,) = ints
        return 24 * rows, 8 * rows, "hbm"
    if symbol in ("pasnl_nl_attention", "pasnl_nl_attention_ws"):
        b, p, n, cb = ints[:4]
        return 4 * b * cb * (2 * p + 2 * n), 4 * b * p * n * cb + 5 * b * p * n, "mfma"
    if symbol == "pasnl_as_attention":
        g, as_, cb = ints
        return 4 * g * as_ * cb * 4, 4 * g * as_ * as_ * cb + 5 * g * as_ * as_, "mfma"
    if symbol == "pasnl_sa_group":
        b, n, c, m, k = ints
        return 4 * b * (3 * n + n * c + m * k + 3 * m + m * k * (6 + c) + m * (6 + c)), 0, "hbm"
    if symbol == "pasnl_sa_local_cell":
        g, k, w, c1, c2 = ints
        # reads the grouped points once, writes (c2 x 32) per group; weights are LDS-resident
        return 4 * (g * k * w + g * c2 * 32 + w * c1 + c1 * c2), g * (2 * k * (w * c1 + c1 * c2 + 3 * 32) + 2 * c2 * k * 32), "mfma"
    if symbol in ("pasnl_sa_cell", "pasnl_sa_cell_centre0", "pasnl_sa_cell_packed"):
        b, n, c, m, k, c1, c2 = ints
        g, w = b * m, 6 + c
        # reads the tables, the indices and the centres once; writes (c2 x 32) per group + the skip maxima
        # (centre0: no centre table to read; the centres and neighbour 0's feature rows are written instead)
        centres = 3 * g if symbol != "pasnl_sa_cell_centre0" else 3 * g + (3 + c) * g  # (_packed: the centre mode is a pointer argument)
        return (4 * (b * n * (3 + c) + g * k + centres + g * c2 * 32 + g * w + w * c1 + c1 * c2),
                g * (2 * k * (w * c1 + c1 * c2 + 3 * 32) + 2 * c2 * k * 32), "mfma")
    if symbol in ("pasnl_sa_tail", "pasnl_sa_tail_cat", "pasnl_sa_tail_res", "pasnl_sa_tail_packed"):
        rows, w, cb, c = ints[:4]
        # (the packed form's optional concat rows / residual are pointer arguments: not counted -- a lower bound of its bytes)
        extra = rows * (c + 4 + 3) if symbol.endswith("_cat") else (rows * c if symbol.endswith("_res") else 0)
        return 4 * (rows * (2 * c + w + cb) + c * (w + cb + c) + extra), 2 * rows * c * (w + cb + c), "mfma"
    if symbol == "pasnl_dense_rows":
        rows, k, n, _relu = ints
        return 4 * (rows * k + k * n + n + rows * n), 2 * rows * k * n, "latency"  # ~0.1 GFLOP: bound by its own start-up
    if symbol == "pasnl_narrow_project2":
        r0, k0, n0, r1, k1, n1 = ints
        return 4 * (r0 * (k0 + n0) + r1 * (k1 + n1)), 2 * (r0 * k0 * n0 + r1 * k1 * n1), "hbm"
    if symbol in ("pasnl_decode_cell", "pasnl_decode_cell_tiled"):
        b, n, c, k = ints
        return 4 * b * n * (3 + c + k + (3 + c) * 32), 2 * b * n * k * ((3 + c) * 32 + 3 * 32), "hbm"
    if symbol == "pasnl_mlp3_max_pool":
        b, n, k0, c1, c2, c3 = ints[:6]
        return 4 * (b * n * k0 + k0 * c1 + c1 * c2 + c2 * c3 + c1 + c2 + c3 + b * c3), 2 * b * n * (k0 * c1 + c1 * c2 + c2 * c3), "mfma"
    if symbol in ("pasnl_max_pool_rows", "pasnl_max_pool_rows_strided"):
        b, n, c = ints[:3]
        return 4 * b * c * (n + 1), 0, "hbm"
    if symbol == "pasnl_as_gather":
        b, n, c, m, k, as_ = ints
        return 4 * b * (n * (3 + c) + m * as_ + m * as_ * (6 + c)), 0, "hbm"
    if symbol == "pasnl_take_neighbor0":
        b, n, c, m, k = ints
        return 4 * b * m * (1 + 2 * (3 + c) + 3), 0, "hbm"
    if symbol in ("pasnl_as_cell_narrow", "pasnl_as_cell_wide", "pasnl_as_cell_wide_ld"):
        g, as_, cb, w, ch = ints[:5]
        tail = 4 * as_ * as_ * cb + 5 * as_ * as_ + 2 * as_ * cb * 32 + 2 * as_ * 32 * (1 + ch) + 2 * as_ * (3 + ch)
        if symbol == "pasnl_as_cell_narrow":  # reads the gathered rows; K, V, Q never reach memory
            return 4 * g * (as_ * w + 3 + ch), g * (2 * as_ * w * 3 * cb + tail), "mfma"
        return 4 * g * (as_ * 3 * cb + as_ * w + 3 + ch), g * tail, "mfma"
    if symbol == "pasnl_as_reweight":
        g, as_, ns, ch = ints
        return 4 * g * (as_ * (1 + ch) + as_ * (3 + ch) + 3 + ch), 4 * g * as_ * (1 + ch), "hbm"
    return 0, 0, "hbm"


def kernel_table(records):
    """records: list of (symbol, ints, e0, e1) over several steps -> per (symbol, ints) averages"""
    agg = {}
    for sym, ints, e0, e1 in records:
        key = (sym, ints)
        a = agg.setdefault(key, [0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
    rows = []
    for (sym, ints), (ms, cnt) in agg.items():
        by, fl, bound = algorithmic(sym, ints)
        avg_s = ms / cnt * 1e-3
        row = {"kernel": sym, "dims": list(ints), "launches": cnt, "avg_us": round(avg_s * 1e6, 2), "bound": bound,
               "alg_bytes": by, "alg_flops": fl, "GB/s": round(by / avg_s / 1e9, 2),
               "TFLOP/s": round(fl / avg_s / 1e12, 3)}
        if bound in ("valu", "latency"):  # the roof these are bound by: the vector issue ports (committed counter pass, or null)
            row["valu_frac"] = measured_valu_frac(sym, ints)
        rows.append(row)
    rows.sort(key=lambda r: -r["avg_us"] * 1.0)
    return rows


def cpu_baseline(pc_all, params, adaptive, seconds_budget=20.0):
    """The same forward on the host cores, the way BASELINE.md 3 lays it out: the reference's OWN kNN (knn_.cxx + nanoflann,
    OpenMP over the batch like knn_batch(omp=True); oracle/_ref/libref_knn.so) where that build travelled with the tree, C
    ports of the ops the reference only has as CUDA kernels (FPS, gathers; OpenMP over the batch), and torch-CPU fp32 GEMMs
    on the host threads for the dense layers and the attention (oracle/cells_torch.py).  The sample is the SAME batch the GPU
    measurement runs on (all B clouds), repeated for about `seconds_budget` seconds.  A baseline, not a target."""
    import torch

    from oracle import cells_torch, ops, ref

    cores = os.cpu_count() or 1
    sample = pc_all
    bsz = sample.shape[0]
    ops.set_threads(min(cores, bsz))  # OpenMP over the batch: more threads than clouds only spin
    # give the CPU its best configuration: the intra-op thread count that runs this forward fastest (all logical cores
    # is NOT it on a 2-socket host with (B*P*K, C<=134) GEMMs)
    best = (None, 1e30)
    for t in sorted({min(cores, c) for c in (16, 32, 64)}):
        torch.set_num_threads(t)
        cells_torch.cls_forward(sample, params, adaptive_sample=adaptive)  # warm-up (thread pools, BLAS)
        t0 = time.perf_counter()
        cells_torch.cls_forward(sample, params, adaptive_sample=adaptive)
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (t, dt)
    torch.set_num_threads(best[0])
    one = best[1]
    reps = max(3, min(30, int(seconds_budget / max(one, 1e-3))))
    ts, pieces = [], {}
    for _ in range(reps):
        cells_torch.TIMES = {}
        t0 = time.perf_counter()
        cells_torch.cls_forward(sample, params, adaptive_sample=adaptive)
        ts.append(time.perf_counter() - t0)
        for k, v in cells_torch.TIMES.items():
            pieces.setdefault(k, []).append(v)
    cells_torch.TIMES = None
    med = float(np.median(ts))
    # the reference's single-threaded CPU op of the decoder path, timed as a piece (tf_interpolate.cpp:60-103, 1 thread as in
    # the TF op) next to the GPU kernel's line in `kernels`: ScanNet fa_layer3 shape, 16 x (1024 unknown, 256 known)
    nn = None
    if ref.available("libref_interp.so"):
        a, b_ = synth_clouds(7, 16, 1024), synth_clouds(8, 16, 256)
        t0 = time.perf_counter()
        ref.three_nn(a, b_)
        nn = {"shape": [16, 1024, 256], "seconds": round(time.perf_counter() - t0, 4), "threads": 1, "kind": "reference"}
    return {"value": round(bsz / med, 2), "unit": "point-clouds/s", "cores": int(torch.get_num_threads()),
            "kind": "reference" if ref.available("libref_knn.so") else "port",
            "sample": f"{reps} forwards of the GPU's own B={bsz}x{sample.shape[1]} batch, median {med * 1e3:.0f} ms each",
            "how": f"kNN = the reference's knn_.cxx + nanoflann, OpenMP over the batch"
                   f"{'' if ref.available('libref_knn.so') else ' (C PORT: oracle/_ref absent)'}; FPS / gathers = C port (the reference "
                   f"has no CPU kernel); dense + attention = torch CPU fp32, {torch.get_num_threads()} threads of {cores} logical cores",
            "pieces_ms": {k: round(float(np.median(v)) * 1e3, 3) for k, v in pieces.items()},
            "three_nn_reference_1thread": nn}


def _sha256(path):
    import hashlib

    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def csrc_digests():
    d = os.path.join(ROOT, "pointasnl_amd", "csrc")
    return {f: _sha256(os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp", ".inc"))}


# kernels that live in another file than the entry point that launches them
EXTRA_SOURCES = {"pasnl_query_ball_point": ["ball_grid.hip", "sortnet.inc"], "pasnl_knn_batch_ws": ["grouping.hip"],
                 "pasnl_knn_batch_ref": ["grouping.hip", "knn_grid.hip", "knn_tree.hip"]}


def measured_traffic(symbol, dims):
    """HBM bytes per launch of (symbol, dims) from the committed PMC pass (profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, profiles/collect_traffic.sh + pmc_to_traffic.py) -> (bytes or None, provenance).  The file
    records the digests of the kernel sources it was collected on; if the file that defines `symbol` (or common.hpp) has
    changed since, the number is NOT reported (a stale traffic figure reads as a measurement when it is not one)."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, "profiles/traffic.json absent"
    data = json.load(open(tfile))
    key = symbol + ":" + ",".join(map(str, dims))
    src = data.get("_source") or {}
    if key not in data:
        return None, f"{key} not in profiles/traffic.json (collected at {src.get('commit', '?')})"
    now, then = csrc_digests(), src.get("csrc_sha256", {})
    owner = next((f for f in now if f.endswith(".hip") and f'extern "C" int {symbol}(' in
                  open(os.path.join(ROOT, "pointasnl_amd", "csrc", f)).read()), None)
    changed = [f for f in [owner, "common.hpp"] + EXTRA_SOURCES.get(symbol, []) if f and now.get(f) != then.get(f)]
    if changed:
        return None, f"stale: {', '.join(changed)} changed since profiles/traffic.json was collected at {src.get('commit', '?')}"
    return data[key], f"profiles/traffic.json@{src.get('commit', '?')} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; {owner} unchanged since)"


def measured_valu_frac(symbol, dims):
    """Fraction of the vector-issue roof a launch of (symbol, dims) reached -- SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles),
    from the same committed counter collection as the traffic (third pass of profiles/collect_traffic.sh), under the same
    staleness rule -> fraction or None.  What the rows labelled "valu" / "latency" are bound by (VERDICT r05 missing 5)."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None
    data = json.load(open(tfile))
    v = (data.get("_valu_frac") or {}).get(symbol + ":" + ",".join(map(str, dims)))
    if v is None or measured_traffic(symbol, dims)[0] is None:  # (absent, or the kernel's source changed since)
        return None
    return v



# ---------------------------------------------------------------------------------------------------------------
# supervisor: progress-watched worker processes
# ---------------------------------------------------------------------------------------------------------------
HEARTBEAT_ENV = "PASNL_BENCH_HEARTBEAT_FD"
# environment the supervisor / its tests use to watch the worker: not tuning switches (config.switches lists the others)
SUPERVISOR_ENV = ("PASNL_BENCH_HEARTBEAT_FD", "PASNL_BENCH_STALL_SCALE", "PASNL_BENCH_SUPERVISE", "PASNL_BENCH_WATCHDOG", "PASNL_BENCH_FAKE_STALL")
# seconds without a heartbeat that count as a stall, per phase the worker announces.  start: interpreter + first
# `import torch` on a fresh box (minutes) + RCCL init; setup: eager forwards, BLAS heuristics, graph capture;
# run: warm-up + timed steps (+ per step, see beat()); post: event-instrumented pass, the other configurations,
# the operator sweep, the CPU baseline sample
ALLOWANCE = {"start": 900.0, "setup": 600.0, "run": 120.0, "post": 900.0}


def beat(phase, extra=0.0):
    """worker side: announce a phase; the supervisor expects the next announcement within ALLOWANCE[phase]+extra s"""
    fd = os.environ.get(HEARTBEAT_ENV)
    if fd:
        scale = float(os.environ.get("PASNL_BENCH_STALL_SCALE", "1"))  # tests shrink the allowances
        os.write(int(fd), f"{phase} {(ALLOWANCE[phase] + extra) * scale:.3f}\n".encode())


def watch_many(jobs, first_allowance=ALLOWANCE["start"]):
    """Run every (cmd, env) of `jobs` with its own heartbeat pipe.  Returns (returncode, None) when all workers have exited
    (the first non-zero code wins, and the other workers are killed as soon as one fails), or (None, stalled_phase) after
    killing every worker because one of them made no progress within the allowance of the phase it last announced."""
    procs, pipes = [], {}
    for cmd, env in jobs:
        r, w = os.pipe()
        env = dict(env)
        env[HEARTBEAT_ENV] = str(w)
        proc = subprocess.Popen(cmd, env=env, pass_fds=(w,))
        os.close(w)
        procs.append(proc)
        pipes[r] = {"proc": proc, "phase": "start", "deadline": time.monotonic() + first_allowance, "buf": b""}

    def kill_all():
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()

    def forward_signal(signum, _frame):  # the driver stops the supervisor: take the workers along
        kill_all()
        sys.exit(128 + signum)

    old = {sig: signal.signal(sig, forward_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        while pipes:
            now = time.monotonic()
            late = [st for st in pipes.values() if st["deadline"] <= now]
            if late:
                kill_all()
                return None, late[0]["phase"]
            ready, _, _ = select.select(list(pipes), [], [], max(0.0, min(st["deadline"] for st in pipes.values()) - now))
            for r in ready:
                st = pipes[r]
                data = os.read(r, 4096)
                if not data:  # every write end closed: this worker has exited
                    code = st["proc"].wait()
                    os.close(r)
                    del pipes[r]
                    if code != 0 and rc == 0:
                        rc = code
                        kill_all()  # a rank that failed leaves the others waiting in a collective
                    continue
                st["buf"] += data
                lines = st["buf"].split(b"\n")
                st["buf"] = lines.pop()
                if lines:
                    name, secs = lines[-1].decode().split()
                    st["phase"], st["deadline"] = name, time.monotonic() + float(secs)
        return rc, None
    finally:
        for r in pipes:
            os.close(r)
        for sig, h in old.items():
            signal.signal(sig, h)


def watch(cmd, env, first_allowance=ALLOWANCE["start"]):
    """One worker: (returncode, None), or (None, stalled_phase) after killing it."""
    return watch_many([(cmd, env)], first_allowance)


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flag(argv, name, default):
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def supervise(argv):
    """Launcher + supervisor.  `python bench.py --gpus N` starts N ranks ITSELF (one process per GPU: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT); under torch.distributed.run (WORLD_SIZE already set) it is one
    rank of that job.  Every worker is watched through its heartbeat pipe; if one stalls, all are killed (exit code 3)."""
    base = [sys.executable, os.path.abspath(__file__), "--worker"]
    scale = float(os.environ.get("PASNL_BENCH_STALL_SCALE", "1"))
    n = int(_flag(argv, "--gpus", "1"))
    if "WORLD_SIZE" in os.environ or n == 1:
        jobs = [(base + argv, os.environ)]
    else:
        port = _free_port()
        jobs = [(base + argv, dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                                   MASTER_PORT=str(port))) for r in range(n)]
    rc, stalled = watch_many(jobs, ALLOWANCE["start"] * scale)
    if stalled is None:
        return rc
    print(f"bench.py: a worker made no progress in phase '{stalled}'; all workers were killed", file=sys.stderr, flush=True)
    return 3


def protocol_only(args, rank, world):
    """--model none: everything of a bench run except the model -- rendezvous, NUMA binding, barriers, a per-step all-gather of
    stand-in logits of the chosen model's width (--proto-shape cls: (B,40); sem_seg_res: (B, 10240 x 20)), max-over-ranks timing,
    the per-rank times, the gathered-rows / shards-differ checks, an all-reduce sanity value and the JSON line with every key the
    multi-rank line of a real run carries.  Runs on CPU with gloo: the first 8-GPU run then exercises nothing new but the wire."""
    import torch
    import torch.distributed as dist

    from pointasnl_amd import sharding

    dev = "cpu"
    numa_node, cpus_bound = None, 0
    if args.backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dev = torch.device("cuda", torch.cuda.current_device())
        numa_node, cpus_bound = sharding.bind_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")))
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    B = args.batch
    width = 40 if args.proto_shape == "cls" else 10240 * 20
    logits = torch.full((B, width), float(rank + 1), device=dev)
    gather = sharding.LogitsGather(world, B, width, dev, force=multi)
    beat("run", 0.25 * (args.warmup + args.steps))
    for _ in range(args.warmup):
        gather.all_gather(logits)
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = gather.all_gather(logits)
    if multi:
        dist.barrier()
    mine = time.perf_counter() - t0
    elapsed = mine
    beat("post")
    check, ranks, per_rank, gathered_ok, shards_differ = float(rank + 1), 1, [round(mine / args.steps * 1e3, 4)], None, None
    if multi:
        t = torch.tensor([elapsed, float(rank + 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        elapsed, check, ranks = float(t[0]), float(t[1]), dist.get_world_size()
        every = torch.empty((world,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, torch.tensor([mine], dtype=torch.float64, device=dev))
        per_rank = [round(float(v) / args.steps * 1e3, 4) for v in every]
        # the checks of a real multi-rank run: this rank's rows of the gathered tensor are its own logits, every rank's rows carry
        # that rank's checksum, and the shards differ
        gathered_ok = bool(torch.equal(out[rank * B:(rank + 1) * B], logits))
        sums = torch.empty((world,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(sums, logits.double().sum().reshape(1))
        gathered_ok = gathered_ok and bool(torch.allclose(sums, out.view(world, -1).double().sum(1), rtol=1e-9, atol=0.0))
        shards_differ = bool(len(set(sums.tolist())) == world)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "protocol only (no model)", "value": round(world * B * args.steps / elapsed, 2),
                          "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"none ({args.proto_shape}-shaped logits)", "backend": args.backend, "rccl_ranks": ranks,
                                     "allreduce_check": check, "global_batch": world * B, "per_rank_ms_per_step": per_rank,
                                     "gathered_rows_match_local": gathered_ok, "shards_differ": shards_differ,
                                     "numa_node": numa_node, "cpus_bound": cpus_bound,
                                     "parallelism": f"batch-shard x{world}, all-gather of logits ({B} x {width} per rank)"}}), flush=True)


# ---------------------------------------------------------------------------------------------------------------
# one configuration: inputs, capture, timed replays, agreement with the eager forward, per-kernel pass
# ---------------------------------------------------------------------------------------------------------------
_FORWARD_STREAM = None
SETTLE_REPLAYS = 40  # untimed replays of the captured step between the capture and the W warm-up steps (config.settle_replays)
SPLIT_PREFIX = os.environ.get('PASNL_BENCH_SPLIT_PREFIX', '1') != '0'  # tuning switch: sem_seg_res prefix as two plain branches
PREFIX_L2 = os.environ.get('PASNL_BENCH_PREFIX_L2', '1') != '0'   # the classifier's prefix (no adaptive sampling) carries BOTH levels' searches and is
#   forked at the START of the step, not lazily: 1.320 -> 1.275 ms (prefix of layer 1 alone, forked behind layer 2's cell: 1.320; both
#   levels forked there: 1.355 -- it then ends after the head; at the start but lazily: 1.307)
FORK_AT_DEFAULT = os.environ.get("PASNL_BENCH_FORK_AT", "cell2")  # cls: where the next batch's prefix is forked (head / conv2 / cell2)
SELF_KNN_PREFIX = os.environ.get('PASNL_BENCH_SELF_KNN', '0') != '0'    # tuning switch: cls / sem_seg prefix = sampler || self-kNN, then a row gather (measured: 1.335-1.349 vs 1.315 ms)
# the next batch's self-kNN (large clouds: the grid-pruned search) as a background job on at most this many workgroups (0 = one wave
# per query): its usual grid of ~20 000 workgroups leaves the forward's own small kernels waiting for slots (sa_tail: 10 -> 225 us)
PREFETCH_KNN_WGS = int(os.environ.get('PASNL_BENCH_PREFETCH_KNN_WGS', '1024')) or None
PREFETCH_SLOTS = tuple(int(v) for v in os.environ.get('PASNL_BENCH_PREFETCH_SLOTS', '3,4').split(','))  # side streams of the prefetch
WORKLOADS = {
    1: dict(model="cls", AS=False, noise=0, batch=64, points=1024, name="configs[1]: ModelNet40 pointasnl_cls, 1024 pts"),
    2: dict(model="cls", AS=True, noise=10, batch=64, points=1024,
            name="configs[2]: ModelNet40 pointasnl_cls --AS, 1024 pts + 10 outliers per cloud"),
    3: dict(model="sem_seg", AS=False, noise=0, batch=16, points=8192, synth="scannet", feature_channel=3,
            name="configs[3]: ScanNet pointasnl_sem_seg, 8192 pts (xyz + rgb, synthetic 1.5 x 1.5 x 3 m blocks)"),
    4: dict(model="sem_seg_res", AS=False, noise=0, batch=8, points=10240, synth="kitti", feature_channel=0,
            name="configs[4]: SemanticKITTI pointasnl_sem_seg_res, 10240 pts (synthetic lidar-like scans in metres; one GPU's "
                 "share of the batch)"),
}


def switch_target(target):
    """"module.FLAG" (a module of pointasnl_amd.utils, or _hip) -> (module, flag)."""
    import importlib
    mod, flag = target.rsplit(".", 1)
    m = importlib.import_module("pointasnl_amd." + mod if mod == "_hip" else "pointasnl_amd.utils." + mod)
    if not hasattr(m, flag):
        raise AttributeError(f"{m.__name__} has no {flag}")
    return m, flag


# Explicit MODES of the path, measured next to the default one (`modes` in the line; N = 1 only): the same workload with a
# module-level switch of the host mirror set -- the arithmetic / the order they change is named in `what`, and the deviation
# of the eager logits from the default mode's is measured, not assumed.
MODES = [
    dict(tag="bf16x3", cfg=3, switches={"tf_util.DENSE_BF16X3": True},
         what="the long GEMMs (>= 256 output tiles, K >= 512) as six bf16 matrix products of three-term operand splits, fp32 "
              "accumulation: fp32-grade (1e-5 of scale against fp64), not the fp32 chain's bits", dtype="f32 (bf16x3 products)"),
    dict(tag="canonical_tie_order", cfg=1, switches={"pointasnl_util.KNN_TIE_ORDER": "index"},
         what="neighbour lists in canonical (distance, index) order: the default (the reference's order among equal distances) minus "
              "its tie flags, the tie paths of the few listed queries (chance ties) and the tree kernels that return at once -- the "
              "price of the default", dtype="f32"),
    dict(tag="canonical_tie_order_cfg2", cfg=2, switches={"pointasnl_util.KNN_TIE_ORDER": "index"},
         what="configs[2] with canonical neighbour order: random-weight adaptive sampling collapses neighbouring points into duplicates, "
              "whose order in the reference is their leaf's reading order -- the default builds those clouds' trees", dtype="f32"),
    dict(tag="canonical_tie_order_cfg3", cfg=3, switches={"pointasnl_util.KNN_TIE_ORDER": "index"},
         what="configs[3] with canonical neighbour order: what the default's tie handling costs on 8192-point clouds (set-form tie paths, "
              "one record-moving descent where two tied points share a leaf)", dtype="f32"),
    dict(tag="canonical_tie_order_cfg4", cfg=4, switches={"pointasnl_util.KNN_TIE_ORDER": "index"},
         what="configs[4] with canonical neighbour order: the crops are padded by resampling (duplicated points: the tree of that cloud at "
              "the second level), the 10240-point input level meets a chance tie per step (tie paths on the cloud in global memory)",
         dtype="f32"),
    dict(tag="lattice_default_tie_order", cfg=1, switches={}, lattice=8,
         what="the DEFAULT on clouds made of ties: coordinates snapped to multiples of 1/8 (SURVEY 8(d) lattice stress set) -- nearly "
              "every query is flagged and goes through the rebuilt KD-tree (build + leaf-order search inside the graph)", dtype="f32"),
    dict(tag="every_query_through_the_tree", cfg=1, switches={"pointasnl_util.KNN_TIE_ORDER": "nanoflann"},
         what="the checker of the default: EVERY query of every cloud through the rebuilt KD-tree (round 5's reference_tie_order "
              "mode)", dtype="f32"),
]


def make_input(cfg_index, spec, rank):
    seed = 1234 + cfg_index + 100 * rank
    if spec.get("synth") == "scannet":
        return synth_scannet(seed, spec["batch"], spec["points"])
    if spec.get("synth") == "kitti":
        return synth_kitti(seed, spec["batch"], spec["points"])
    pc = synth_clouds(seed, spec["batch"], spec["points"])
    if spec["AS"]:
        pc = add_noise(pc, spec["noise"], seed)
    if spec.get("lattice"):  # (modes) the lattice stress set: distance ties everywhere
        pc = (np.round(pc * spec["lattice"]) / spec["lattice"]).astype(np.float32)
    return pc


def roofline_of(rows):
    """The roofline block of the hand-written kernel with the largest time per step among the bandwidth- / matrix-bound ones
    (latency- and issue-bound searches have no meaningful fraction of either roof; they are listed in `kernels`)."""
    cand = [r for r in rows if r["bound"] in ("mfma", "hbm")]
    if not cand:
        return None
    dom = max(cand, key=lambda r: r["avg_us"] * r["launches"])
    if dom["bound"] == "mfma":
        ach, peak, unit = dom["TFLOP/s"], F32_MFMA_PEAK_TF, "TFLOP/s"
    else:
        ach, peak, unit = dom["GB/s"], HBM_PEAK_GBS, "GB/s"
    traffic, traffic_source = measured_traffic(dom["kernel"], dom["dims"])
    if traffic is None:
        print(f"bench.py: roofline.traffic not reported: {traffic_source}", file=sys.stderr)
    return {"kernel": dom["kernel"], "dims": dom["dims"], "bound": dom["bound"], "achieved": ach, "peak": peak, "unit": unit,
            "frac": round(ach / peak, 5), "traffic": traffic, "traffic_source": traffic_source, "avg_us": dom["avg_us"],
            "alg_bytes": dom["alg_bytes"], "alg_flops": dom["alg_flops"]}


def run_config(cfg_index, spec, *a, **kw):
    """_run_config with the spec's mode switches (module flags of the host mirror) restored afterwards."""
    switched = []
    try:
        return _run_config(cfg_index, spec, *a, _switched=switched, **kw)
    finally:
        for m, flag, old in reversed(switched):
            setattr(m, flag, old)


def _run_config(cfg_index, spec, steps, warmup, rank=0, world=1, multi=False, graph=True, kernel_pass=True, announce=True,
                pipeline="serial", extra_blocks=0, _switched=None):
    """Measure one workload on the current device -> dict.  The timed region is `steps` forwards bracketed by
    (barrier +) torch.cuda.synchronize() on both sides, max over ranks.

    pipeline = "serial": one captured graph per forward, replayed back to back on one stream.
    pipeline = "prefetch": a serving loop over a stream of batches with two input buffers.  The search prefix of a forward
    (farthest point sampling + the gather + the kNN of the first set-abstraction layer) reads coordinates only and keeps
    64 of the 256 CUs busy for ~0.2 ms of dependent rounds; so the graph of step k computes the rest of forward k (from the
    prefix results step k-1 left in a buffer) and, on a side stream of the same graph, the prefix of forward k+1 on the OTHER
    input buffer.  Every step still does one full forward's worth of work, every output is the output of a complete forward
    on its own input (checked bit for bit against the eager forward of both buffers), and only hand-written kernels run on
    the side stream (two concurrent vendor Stream-K GEMMs can dead-lock).
    The classifier without adaptive sampling: layer 2's points are layer 1's sampled INPUT points, so its search reads the input
    cloud alone as well -- the prefix carries both levels' searches (FPS 1024 -> 512, kNN, FPS 512 -> 128, kNN) and is forked at the
    start of the step: no search is left on the forward's chain (PREFIX_L2)."""
    import importlib

    import torch
    import torch.distributed as dist

    from pointasnl_amd import _hip, sharding, tf_sampling
    from pointasnl_amd.utils import pointasnl_util, tf_util

    model = importlib.import_module(f"pointasnl_amd.models.pointasnl_{spec['model']}")
    # (tuning switch "model,prefix": are layer 2's search / the next batch's prefix enqueued behind the forward's next kernel;
    # cls only.  Measured: 1.315 -> 1.288 ms without adaptive sampling; the segmentation models lose (their samplers are ~1 ms
    # chains that have to start at once): 4.12 -> 5.0 and 2.27 -> 3.16 ms)
    lz = os.environ.get("PASNL_BENCH_LAZY_FORK")
    prefix_l2 = PREFIX_L2 and spec["model"] == "cls" and not spec.get("AS")  # both levels' searches in the prefix, forked at the start
    lazy_model, lazy_prefix = (tuple(v == "1" for v in lz.split(",")) if lz else (None, spec["model"] == "cls" and not spec.get("AS") and not prefix_l2))
    if os.environ.get("PASNL_BENCH_GROUP_ALL") is not None:  # (tuning switch: which group_all modules take the fused kernel)
        from pointasnl_amd.utils import pointnet_util
        pointnet_util.GROUP_ALL_FUSED = tuple(int(v) for v in os.environ["PASNL_BENCH_GROUP_ALL"].split(",") if v)
    B, N = spec["batch"], spec["points"]
    pc = make_input(cfg_index, spec, rank)
    x = torch.from_numpy(pc).cuda()
    store = tf_util.set_store(tf_util.VariableStore(seed=1234))  # identical weights on every rank
    fch = spec.get("feature_channel", 0)

    fork_at = FORK_AT_DEFAULT  # (tuning switch PASNL_BENCH_FORK_AT; cls only)

    def forward(xin=None, search=None, before_head=None):
        xin = x if xin is None else xin
        if spec["model"] == "cls":
            logits, _ = model.get_model(xin, is_training=False, adaptive_sample=spec["AS"], search=search, before_head=before_head,
                                        fork_at=fork_at, lazy_fork=lazy_model)
            return logits
        logits, _ = model.get_model(xin, False, 20, feature_channel=fch, search=search, before_head=before_head)
        return logits.reshape(B, -1)

    width = 40 if spec["model"] == "cls" else N * 20
    gather = sharding.LogitsGather(world, B, width, x.device, force=multi) if multi else None
    if announce:
        beat("setup")
    mode_dev = None
    if spec.get("switches") is not None:  # a MODE: the default mode's eager logits first, then the switches (restored on the way out)
        with torch.no_grad():
            ref_default = forward().clone()
        for target, value in spec["switches"].items():
            m, flag = switch_target(target)
            _switched.append((m, flag, getattr(m, flag)))
            setattr(m, flag, value)
        with torch.no_grad():
            got = forward()
        mode_dev = {"max_abs_dev_of_logits": float((got - ref_default).abs().max()), "logits_scale": float(ref_default.abs().max()),
                    "argmax_agree": float((got.reshape(-1, 20 if spec["model"] != "cls" else 40).argmax(1) ==
                                           ref_default.reshape(-1, 20 if spec["model"] != "cls" else 40).argmax(1)).float().mean())}
    with torch.no_grad():
        # ---- warm-up (eager: creates weights, BLAS workspaces), then capture
        # ONE forward stream for every workload of the process (a new stream per workload lands on another hardware queue
        # and changes which side streams it collides with: measured 1.70 -> 2.07 ms on configs[2])
        global _FORWARD_STREAM
        if _FORWARD_STREAM is None:
            _FORWARD_STREAM = torch.cuda.Stream(priority=-1)  # outranks the prefetching side streams
        side = _FORWARD_STREAM
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                out = forward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, outs, xs = [], [], [x]
        if graph and pipeline == "prefetch":
            spec2 = dict(spec)
            x2 = torch.from_numpy(make_input(cfg_index + 50, spec2, rank)).cuda()  # the batch behind the current one
            xs = [x, x2]
            first = model.first_layer(N)

            def xyz_of(t):
                return t if t.shape[2] == 3 else t[:, :, :3].contiguous()

            nsamp, npnt = first["nsample"], first["npoint"]
            res = spec["model"] == "sem_seg_res"

            def prefix(t, kf=None, into=None):
                """The coordinate-only searches at the head of a forward, as the model itself computes them (bit-identical
                results: exact kNN, the same sampler): cls / sem_seg -- layer1's sa_search = FPS + gather + kNN of the sampled
                points -> [new_xyz, idx]; sem_seg_res -- layer0's self-kNN (kf: already running on another side stream, or
                computed here) and layer1's FPS, whose neighbour lists are rows of that kNN -> [k_all, new_xyz, idx].
                into: the hand-over buffers of the step -- the kernels write them directly (no copy behind them)."""
                xyz = xyz_of(t)
                o = into if into is not None else [None, None, None]
                mw = PREFETCH_KNN_WGS if into is not None else None  # inside a step's graph the search is a background job
                if not res:
                    _, new_xyz = tf_sampling.farthest_point_sample_gather(npnt, xyz, out=(None, o[0]))
                    r = [new_xyz, pointasnl_util.knn_query(nsamp, xyz, new_xyz, out=o[1], max_workgroups=mw)]
                    if prefix_l2:
                        # layer 2's search as well: its points are layer 1's sampled input points (no adaptive sampling)
                        o2 = into[2:4] if into is not None else [None, None]
                        _, xyz2 = tf_sampling.farthest_point_sample_gather(128, new_xyz, out=(None, o2[0]))
                        r += [xyz2, pointasnl_util.knn_query(64, new_xyz, xyz2, out=o2[1])]
                    return r
                fps_idx, new_xyz = tf_sampling.farthest_point_sample_gather(N // 8, xyz, out=(None, o[1]))
                if kf is not None:
                    k_all = kf.get()
                    if o[0] is not None:
                        o[0].copy_(k_all)  # (the self-kNN ran on another branch into a buffer of its own)
                        k_all = o[0]
                else:
                    k_all = pointasnl_util.knn_query(32, xyz, xyz, out=o[0], max_workgroups=mw)
                return [k_all, new_xyz, pointasnl_util._gather_index_rows(k_all, fps_idx, out=o[2])]

            def as_search(t, bufs):
                if not res:
                    if len(bufs) == 4:
                        return {1: (bufs[0], None, bufs[1]), 2: (bufs[2], None, bufs[3])}
                    return (bufs[0], None, bufs[1])
                return {0: (xyz_of(t), None, bufs[0]), 1: (bufs[1], None, bufs[2])}

            # the hand-over buffers between consecutive steps, one set per input buffer
            S = [[r.clone() for r in prefix(t_in)] for t_in in xs]
            torch.cuda.synchronize()

            def body(cur, nxt):
                fk, late = [], []

                def fork():  # sibling forks from the forward's own stream, each joined to it (a fork of a fork crashes
                    #          hipStreamEndCapture on ROCm 7.2)
                    if res and SPLIT_PREFIX:
                        # sem_seg_res: the sampler and the self-kNN each ALONE on a side branch, their join (the rows of the
                        # sampled points + the hand-over copies) on the forward's stream once both are back -- a branch that
                        # waits for another branch inside misleads the graph executor's placement (EXPERIMENTS.md, round 4)
                        # (both branches write the hand-over buffers of the next step themselves)
                        ff = pointasnl_util.Forked(lambda: tf_sampling.farthest_point_sample_gather(N // 8, xyz_of(xs[nxt]), out=(None, S[nxt][1])), slot=PREFETCH_SLOTS[0])
                        kf = pointasnl_util.Forked(lambda: pointasnl_util.knn_query(32, xyz_of(xs[nxt]), xyz_of(xs[nxt]), out=S[nxt][0], max_workgroups=PREFETCH_KNN_WGS), slot=PREFETCH_SLOTS[1])
                        fk.extend([ff, kf])
                        late.append((ff, kf, S[nxt][2]))
                        return
                    if not res and SELF_KNN_PREFIX:
                        # cls / sem_seg: the neighbour lists of the sampled points as ROWS of the cloud's self-kNN (the queries are
                        # support points: the same distances, the same (distance, index) order, bit for bit -- what sem_seg_res
                        # does by construction).  Twice the search, but beside the sampler instead of BEHIND it: the step
                        # ends with the sampler (+ a row gather), not with sampler + kNN
                        ff = pointasnl_util.Forked(lambda: tf_sampling.farthest_point_sample_gather(npnt, xyz_of(xs[nxt]), out=(None, S[nxt][0])), slot=PREFETCH_SLOTS[0])
                        kf = pointasnl_util.Forked(lambda: pointasnl_util.knn_query(nsamp, xyz_of(xs[nxt]), xyz_of(xs[nxt])), slot=PREFETCH_SLOTS[1])
                        fk.extend([ff, kf])
                        late.append((ff, kf, S[nxt][1]))
                        return
                    kf = pointasnl_util.Forked(lambda: pointasnl_util.knn_query(32, xyz_of(xs[nxt]), xyz_of(xs[nxt]), max_workgroups=PREFETCH_KNN_WGS), slot=PREFETCH_SLOTS[1]) \
                        if res else None

                    def run_prefix():
                        prefix(xs[nxt], kf, into=S[nxt])
                        return True
                    fk.append(pointasnl_util.Forked(run_prefix, slot=PREFETCH_SLOTS[0], lazy=lazy_prefix))
                    if kf is not None:
                        fk.append(kf)
                if os.environ.get("PASNL_BENCH_PREFETCH_AT", "start" if prefix_l2 else spec.get("prefetch_at", "head")) == "start":
                    fork()
                o = forward(xs[cur], search=as_search(xs[cur], S[cur]), before_head=None if fk else fork)
                for f in fk:
                    f.get()  # join: the graph ends when everything has finished
                for ff, kf, dst in late:
                    (fps_idx, _), k_all = ff.get(), kf.get()
                    pointasnl_util._gather_index_rows(k_all, fps_idx, out=dst)
                return o

            for cur in (0, 1):
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    o = body(cur, 1 - cur)
                torch.cuda.current_stream().wait_stream(side)
                graphs.append(g)
                outs.append(o)
            out = outs[0]
        elif graph:
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread must not be able to invalidate the capture
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                out = forward()
            torch.cuda.current_stream().wait_stream(side)
            graphs, outs = [g], [out]
        step_no = [0]

        gather_mode = os.environ.get("PASNL_BENCH_GATHER", "async")  # tuning switch: async (default) / sync (round 5's blocking form) / none

        def step():
            if graphs:
                i = step_no[0] % len(graphs)
                step_no[0] += 1
                with torch.cuda.stream(side):
                    if gather is not None and gather_mode == "async":
                        gather.before_reuse(i)  # the all-gather that read outs[i] len(graphs) steps ago
                    graphs[i].replay()
                    if gather is not None:
                        if gather_mode == "async":
                            gather.all_gather_async(outs[i], i)
                        elif gather_mode == "sync":
                            gather.all_gather(outs[i])
                return outs[i]
            o = forward()
            if gather is not None:
                gather.all_gather(o)
            return o

        if multi:
            dist.barrier()  # ranks enter the watched region together, so a stall expires every rank's allowance together
        if announce:
            beat("run", 0.25 * (warmup + steps))
            if os.environ.get("PASNL_BENCH_FAKE_STALL") == "run":  # supervisor test hook: the worker hangs in the watched region
                time.sleep(1e6)
        if announce and graphs:
            # settle (part of the set-up, not of the W warm-up steps): the first block of replays after the capture ran 1-2 % slower
            # than the blocks behind it on every box (clocks / power state after the eager and capture phases) -- a fixed number of
            # untimed replays before the contract's warm-up; an even count, so that buffer 0 is still next
            for _ in range(SETTLE_REPLAYS):
                step()
            torch.cuda.synchronize()
        for _ in range(warmup + (warmup + len(graphs)) % max(1, len(graphs))):  # an even number of steps: buffer 0 is next
            step()
        # ---- timed region: barrier + sync on both sides, max over ranks
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        enqueued = time.perf_counter() - t0  # host time to issue the K steps (replay + collective): no host sync inside a step
        torch.cuda.synchronize()
        if os.environ.get("PASNL_BENCH_TRACE_ONLY"):  # profiling hook (tools/sessions/session_tl.sh): a kernel trace that ends with the timed
            sys.exit(0)                               # replays, not with the serial comparison and the eager per-kernel pass
        if multi:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if announce:
            beat("post")
        per_rank_ms = [elapsed / steps * 1e3]
        if multi:
            every = torch.empty((world,), dtype=torch.float64, device=x.device)
            dist.all_gather_into_tensor(every, torch.tensor([elapsed], dtype=torch.float64, device=x.device))
            per_rank_ms = [float(v) / steps * 1e3 for v in every]
            t = torch.tensor([elapsed], dtype=torch.float64, device=x.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # ---- the spread of the figure: `extra_blocks` further blocks of K steps, each bracketed like the timed region (VERDICT r05
        # weak 8: one 20-step block is 26 ms on boxes that differ by 15 %).  `value` stays the contract's EXACTLY-K-steps block above.
        block_ms = [elapsed / steps * 1e3]
        for _ in range(extra_blocks):
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(steps + steps % max(1, len(graphs))):  # (whole rounds over the buffers: buffer 0 is next again)
                last = step()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            eb = time.perf_counter() - tb
            if multi:
                t = torch.tensor([eb], dtype=torch.float64, device=x.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                eb = float(t.item())
            block_ms.append(eb / (steps + steps % max(1, len(graphs))) * 1e3)
        gathered_ok, shards_differ = None, None
        if gather is not None:
            # this rank's rows of the gathered logits are its own logits, and EVERY rank's rows carry that rank's logits: the
            # ranks exchange a checksum of their local logits and compare it with the checksum of their rows as received here
            gathered_ok = bool(torch.equal(gather.out[rank * B:(rank + 1) * B], last))
            mine = last.double().sum().reshape(1)
            sums = torch.empty((world,), dtype=torch.float64, device=x.device)
            dist.all_gather_into_tensor(sums, mine)
            got = gather.out.view(world, -1).double().sum(1)
            gathered_ok = gathered_ok and bool(torch.allclose(sums, got, rtol=1e-9, atol=0.0))
            shards_differ = bool(len(set(sums.tolist())) == world)  # different seeds per rank -> different logits
        # the replayed graphs compute the same function of the same input as the plain eager forward: bit-identical, for every
        # input buffer (prefetch: the last two steps left the outputs of both buffers)
        agree = None
        if graphs:
            for i in range(len(graphs)):  # a run of fewer steps than buffers: every graph has to have produced its output once
                if step_no[0] + i < len(graphs):
                    with torch.cuda.stream(side):
                        graphs[(step_no[0] + i) % len(graphs)].replay()
            torch.cuda.synchronize()
            agree = all(bool(torch.equal(forward(xs[i]), outs[i])) for i in range(len(graphs)))
        # ---- per-kernel pass: the same forward, eager, every C-ABI launch bracketed by HIP events
        rows, launch_order = [], []
        if kernel_pass and rank == 0:
            reps = min(steps, 20)
            _hip.PROFILE = []
            for _ in range(reps):
                forward()
            torch.cuda.synchronize()
            rows = kernel_table(_hip.PROFILE)
            per_fwd = len(_hip.PROFILE) // max(1, reps)
            launch_order = [[sym, list(ints)] for sym, ints, _, _ in _hip.PROFILE[:per_fwd]]
            _hip.PROFILE = None
    from pointasnl_amd.utils.nearest_neighbors.lib.python import nearest_neighbors as NN
    torch.cuda.synchronize()
    NN.check_deferred_flags(clear=True)  # (the reference's tie order is the default: the sticky tree-depth flag of every search so far)
    return {"B": B, "N": N, "elapsed": elapsed, "block_ms": block_ms, "per_rank_ms": per_rank_ms, "ms_per_step": elapsed / steps * 1e3, "enqueue_ms_per_step": enqueued / steps * 1e3, "clouds_per_s": world * B * steps / elapsed,
            "graph": bool(graphs), "pipeline": pipeline if graphs else "eager", "outputs_agree": agree, "gathered_ok": gathered_ok,
            "shards_differ": shards_differ, "rows": rows, "launch_order": launch_order, "pc": pc, "store": store,
            "mode_dev": mode_dev}


def ball_query_sweep(batches=(64, 256, 1024, 4096), iters=20):
    """The operator north_star puts a number on: query_ball_point at (B,1024) support / (B,512) queries, radius 0.2,
    nsample 32 (tf_grouping.py:78-88 scaled to the cls layer-1 shape), HIP-event median over `iters` launches."""
    import torch

    import pointasnl_amd as P
    from pointasnl_amd import _hip

    out, sources = [], set()
    for b in batches:
        x = torch.from_numpy(synth_clouds(4321 + b, min(b, 256), 1024)).cuda()
        if b > 256:
            x = x.repeat(b // 256, 1, 1)[torch.randperm(b, device="cuda")].contiguous()
        q = x[:, :512].contiguous()
        for _ in range(3):
            P.tf_grouping.query_ball_point(0.2, 32, x, q)
        torch.cuda.synchronize()
        _hip.PROFILE = []
        for _ in range(iters):
            P.tf_grouping.query_ball_point(0.2, 32, x, q)
        torch.cuda.synchronize()
        us = [e0.elapsed_time(e1) * 1e3 for _, _, e0, e1 in _hip.PROFILE]
        _hip.PROFILE = None
        med = float(np.median(us))
        by, fl, _ = algorithmic("pasnl_query_ball_point", (b, 1024, 512, 32))
        traffic, src = measured_traffic("pasnl_query_ball_point", [b, 1024, 512, 32])
        out.append({"B": b, "dims": [b, 1024, 512, 32], "radius": 0.2, "median_us": round(med, 2), "min_us": round(min(us), 2),
                    "alg_MB": round(by / 1e6, 2), "GB/s": round(by / med / 1e3, 1), "hbm_frac": round(by / med / 1e3 / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "valu_frac": measured_valu_frac("pasnl_query_ball_point", [b, 1024, 512, 32])})
        sources.add(src)
        del x, q
    return out, "; ".join(sorted(sources))


def precreate_streams():
    """The forward's streams, created in the order a plain N = 1 run creates them, BEFORE RCCL creates its own: HIP maps streams to
    a few hardware queues in creation order, and the side streams of a forward that share a queue serialise.  With the
    streams created after dist.init_process_group the one-rank RCCL path ran 1.75-1.79 ms per step against 1.32 of the plain
    path -- with NO collective in the step -- which would have read as a 25 % 'scaling loss' from N = 1 to N = 2."""
    import torch

    from pointasnl_amd.utils import pointasnl_util
    global _FORWARD_STREAM
    for slot in (1, 0, 2, 3, 4):  # the order of first use in a cls forward + the prefetch slots
        pointasnl_util._side_stream(slot)
    if _FORWARD_STREAM is None:
        _FORWARD_STREAM = torch.cuda.Stream(priority=-1)


def median_ms(r):
    """the median over the timed blocks of a run (auxiliary figures: other_configs, modes) -- a single block of ten steps met a
    stalled replay now and then (7.5 / 6.3 / 2.4-ms blocks among 1.4-ms ones, in the canonical-order runs as well), which the median drops"""
    return float(np.median(r["block_ms"]))


def input_stage_row(iters=10):
    """SURVEY 8(f) rank 4, the step BEFORE the path for configs[4]: voxel-grid subsampling of a raw scan at 0.06 m + the
    kNN crop of 8 x (10240 + buffer) points around 8 centres (semantic_kitti_dataset_grid.py:265-286), on the device, no
    host KD-tree.  Synthetic lidar-like scan (bench.synth_kitti's recipe before its crop), HIP-event medians."""
    import torch

    from pointasnl_amd.SemanticKITTI import semantic_kitti_dataset_grid as G
    from pointasnl_amd.utils.cpp_wrappers.cpp_subsampling import grid_subsampling

    rng = np.random.Generator(np.random.PCG64(4242))
    m = 120000                                            # a 64-beam sweep
    r = 2.0 + 38.0 * rng.random(m) ** 2
    th = rng.random(m) * 2 * np.pi
    raw = np.stack([r * np.cos(th), r * np.sin(th), rng.standard_normal(m) * 0.02], 1).astype(np.float32)
    raw_d = torch.from_numpy(raw).cuda()

    def ev(fn):
        us = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(us))

    sub = grid_subsampling.compute(raw, sampleDl=0.06)    # (host round trip: numpy in / out like the reference wrapper)
    t_sub = ev(lambda: grid_subsampling.compute(raw_d, sampleDl=0.06))
    scan = torch.from_numpy(sub).cuda()
    n = int(scan.shape[0])
    b, num_point, buf = 8, 10240, 2560
    centres = scan[torch.from_numpy(rng.integers(0, n, b)).cuda()].contiguous()
    ks = torch.from_numpy((num_point + buf + rng.integers(0, buf // 4, b)).astype(np.int32)).cuda()
    kcap = min(n, num_point + buf + buf // 4)
    G.select_batch(scan, centres, k=ks, kcap=kcap)
    t_crop = ev(lambda: G.select_batch(scan, centres, k=ks, kcap=kcap))
    by, _, _ = algorithmic("pasnl_knn_crop", (b, n, 0, kcap))
    return {"workload": "SemanticKITTI input stage on the device (SURVEY 8(f) rank 4): grid_subsampling 0.06 m + crop_pc's kNN search",
            "raw_points": m, "subsampled_points": n, "grid_subsample_us": round(t_sub, 1) if t_sub else None,
            "crops": b, "k": f"{num_point} + {buf}..{buf + buf // 4 - 1}", "knn_crop_us": round(t_crop, 1),
            "knn_crop_alg_MB": round(by / 1e6, 2), "knn_crop_GB/s": round(by / t_crop / 1e3, 1),
            "knn_crop_hbm_frac": round(by / t_crop / 1e3 / HBM_PEAK_GBS, 4),
            "note": "the crop is 8 launches of an exact radix selection over the scan (csrc/crop.hip): launch-bound at one scan; "
                    "the reference: a pickled sklearn KDTree per scan and ~4 ms per query on the host"}


def traffic_pass(args):
    """--traffic-pass (run under rocprofv3 --pmc by profiles/collect_traffic.sh): every workload's forward three times, eagerly
    and with every kernel on ONE stream in program order (no forks), then the operator sweep, with a marker kernel
    (torch.cuda._sleep) in front of every C-ABI launch, so that the counter rows between two markers belong to one launch
    whatever number of kernels it starts.  Prints the sequence of launches as one JSON line; the first forward of a workload
    (weights and workspaces are created) is marked "_unmeasured"."""
    import importlib

    import torch

    import pointasnl_amd as P
    from pointasnl_amd import _hip
    from pointasnl_amd.utils import pointasnl_util, tf_util

    _hip.lib()
    _hip.require_device()
    torch.cuda.set_device(0)
    pointasnl_util.OVERLAP = False
    seq = []
    _hip.MARK = seq
    with torch.no_grad():
        for ci, spec in WORKLOADS.items():
            model = importlib.import_module(f"pointasnl_amd.models.pointasnl_{spec['model']}")
            x = torch.from_numpy(make_input(ci, spec, 0)).cuda()
            tf_util.set_store(tf_util.VariableStore(seed=1234))
            for it in range(3):
                _hip.MARK_SKIP = it == 0
                if spec["model"] == "cls":
                    model.get_model(x, is_training=False, adaptive_sample=spec["AS"])
                else:
                    model.get_model(x, False, 20, feature_channel=spec.get("feature_channel", 0))
            torch.cuda.synchronize()
        for b in (64, 256, 1024, 4096):
            x = torch.from_numpy(synth_clouds(4321 + b, min(b, 256), 1024)).cuda()
            if b > 256:
                x = x.repeat(b // 256, 1, 1).contiguous()
            q = x[:, :512].contiguous()
            for it in range(3):
                _hip.MARK_SKIP = it == 0
                P.tf_grouping.query_ball_point(0.2, 32, x, q)
            torch.cuda.synchronize()
    _hip.MARK = None
    print(json.dumps({"launch_sequence": seq}), flush=True)


def main():
    if "--worker" not in sys.argv[1:] and os.environ.get("PASNL_BENCH_SUPERVISE", "1") != "0":
        sys.exit(supervise(sys.argv[1:]))
    beat("start")
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true", help="run the measurement in this process without a supervisor (the supervisor passes it; use it under profilers)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--proto-shape", choices=["cls", "sem_seg_res"], default="cls", help="--model none: the width of the stand-in logits")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps each: the first one is `value` (the contract's exactly-K-steps region), all of them go to ms_per_step_blocks")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="clouds per GPU (weak scaling); default: the BASELINE config's")
    ap.add_argument("--AS", action="store_true", help="time configs[2] (adaptive sampling on, noisy clouds) as the main workload")
    ap.add_argument("--noise", type=int, default=10)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl = RCCL over xGMI (the product path); gloo only with --model none")
    ap.add_argument("--model", default="cls", choices=["cls", "sem_seg", "sem_seg_res", "none"],
                    help="cls = the BASELINE metric (default).  sem_seg / sem_seg_res = configs[3] / configs[4] as the main "
                         "workload.  none = no model at all: only the launcher, the rendezvous, the per-step all-gather and the "
                         "timing protocol (CPU tests drive it with --backend gloo)")
    ap.add_argument("--points", type=int, default=0, help="points per cloud (default: 1024 cls, 8192 sem_seg, 10240 sem_seg_res)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--pipeline", default="prefetch", choices=["serial", "prefetch"],
                    help="prefetch (default): a serving loop -- the coordinate-only search prefix of batch k+1 (FPS + gather + kNN "
                         "of the input level, hand-written kernels only) runs on a side stream of batch k's graph, two input "
                         "buffers (see run_config); the line also carries the serial figure.  serial: one graph per forward, back "
                         "to back")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip `other_configs` and `ball_query_sweep` (they run at N=1 only)")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each entry of `other_configs`")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (RCCL init, per-step all-gather, barriers, max-over-ranks) even with "
                         "one rank: a 1-GPU box can then exercise everything but the wire")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.FLAG=VALUE",
                    help="A/B switch of the host mirror for measurements, e.g. --set tf_util.DENSE_ROWS=0 "
                         "--set pointasnl_util.SA_TAIL_FUSED=0 (modules of pointasnl_amd.utils, or _hip.LIB_PATH='<another build>'; "
                         "the line records them)")
    ap.add_argument("--launch-order", action="store_true", help="add the per-forward sequence of C-ABI launches to the JSON line")
    ap.add_argument("--traffic-pass", action="store_true", help="see traffic_pass(); used by profiles/collect_traffic.sh")
    args = ap.parse_args()

    if args.worker:  # do not outlive the supervisor (PR_SET_PDEATHSIG)
        import ctypes

        ctypes.CDLL(None).prctl(1, signal.SIGKILL)

    if os.environ.get("PASNL_BENCH_WATCHDOG"):  # diagnostics: dump every thread's Python stack and exit if the run stalls
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["PASNL_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or args.force_dist) and args.model != "none":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", MULTI_RANK_HW_QUEUES)  # (before torch loads the HIP runtime: the comment at the top of the file)

    import torch
    import torch.distributed as dist

    if world != args.gpus:  # never degrade silently: a line that says n_gpus=1 for a --gpus 8 request is a wrong measurement
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run", file=sys.stderr)
        sys.exit(4)
    if args.model == "none":
        args.batch = args.batch or 64
        return protocol_only(args, rank, world)
    if args.backend != "nccl":
        print("bench.py: the product path runs on RCCL (--backend nccl); gloo is for --model none", file=sys.stderr)
        sys.exit(4)
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs device {local_rank} but only {torch.cuda.device_count()} HIP device(s) are visible "
              f"(--gpus {args.gpus}): refusing to run", file=sys.stderr)
        sys.exit(4)
    if args.traffic_pass:
        return traffic_pass(args)

    from pointasnl_amd import _hip
    for item in args.set:  # A/B switches (module-level flags of the host mirror)
        import ast
        target, value = item.split("=", 1)
        try:
            m, flag = switch_target(target)
        except AttributeError as e:
            raise SystemExit(f"--set {item}: {e}")
        setattr(m, flag, ast.literal_eval(value))

    _hip.lib()
    _hip.require_device()
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    allreduce_check = None
    numa_node, cpus_bound = None, 0
    if multi:
        from pointasnl_amd import sharding as _sh

        if os.environ.get("PASNL_BENCH_NO_BIND") != "1":  # (tuning switch)
            numa_node, cpus_bound = _sh.bind_to_gpu_numa(local_rank)  # before RCCL starts its proxy thread (it inherits the mask)
        if os.environ.get("PASNL_BENCH_PRECREATE", "1") == "1":
            precreate_streams()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        t = torch.tensor([float(rank + 1)], device="cuda")
        dist.all_reduce(t)  # sanity: every rank took part in one RCCL all-reduce
        allreduce_check = float(t.item())
        assert allreduce_check == world * (world + 1) / 2, allreduce_check

    main_index = {"cls": 2 if args.AS else 1, "sem_seg": 3, "sem_seg_res": 4}[args.model]
    spec = dict(WORKLOADS[main_index])
    if args.AS:
        spec["noise"] = args.noise
    if args.batch:
        spec["batch"] = args.batch
    if args.points:
        spec["points"] = args.points
    res = run_config(main_index, spec, args.steps, args.warmup, rank=rank, world=world, multi=multi, graph=not args.no_graph,
                     pipeline=args.pipeline, extra_blocks=max(0, args.blocks - 1))
    serial = None
    if args.pipeline != "serial" and not args.no_graph:  # the same workload without the cross-batch overlap (every rank takes part)
        # (an auxiliary figure: the better of two short runs, so that one stalled replay -- a clock ramp, an allocator sync --
        # does not pass for the serial step time)
        r0 = min((run_config(main_index, spec, min(args.steps, 20), 3, rank=rank, world=world, multi=multi, graph=True,
                             kernel_pass=False, announce=False, pipeline="serial") for _ in range(2)), key=lambda r: r["ms_per_step"])
        serial = {"ms_per_step": round(r0["ms_per_step"], 4), "clouds_per_s": round(r0["clouds_per_s"], 2),
                  "outputs_agree": r0["outputs_agree"], "steps": min(args.steps, 20)}
        beat("post")
    rccl_ranks = dist.get_world_size() if multi else 1
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    rows = res["rows"]
    cpu = None
    if not args.no_cpu_baseline and args.model == "cls" and world == 1:  # rank 0 at N=1 only
        cpu = cpu_baseline(res["pc"], res["store"].export_numpy(), spec["AS"])
        beat("post")

    others, sweep, sweep_traffic_source, modes, input_stage = None, None, None, None, None
    if world == 1 and not args.no_others and not args.force_dist:
        others = []
        for ci, ospec in WORKLOADS.items():
            if ci == main_index:
                continue
            r = run_config(ci, ospec, args.other_steps, 4, graph=not args.no_graph, announce=False, pipeline=args.pipeline, extra_blocks=2)
            beat("post")
            rs = run_config(ci, ospec, args.other_steps, 3, graph=not args.no_graph, kernel_pass=False, announce=False,
                            pipeline="serial", extra_blocks=2) if args.pipeline != "serial" and not args.no_graph else None
            beat("post")
            others.append({"workload": ospec["name"] + f", batch={r['B']}", "steps": args.other_steps, "warmup": 4,
                           "pipeline": r["pipeline"], "serial_ms_per_step": round(median_ms(rs), 4) if rs else None,
                           "ms_per_step": round(median_ms(r), 4), "ms_per_step_blocks": [round(v, 4) for v in r["block_ms"]],
                           "clouds_per_s": round(r["B"] / median_ms(r) * 1e3, 2),
                           "points_per_s": round(r["B"] / median_ms(r) * 1e3 * r["N"], 1), "hip_graph": r["graph"],
                           "outputs_agree": r["outputs_agree"], "roofline": roofline_of(r["rows"]),
                           "handwritten_kernel_us_per_step": round(sum(k["avg_us"] * k["launches"] for k in r["rows"]) /
                                                                   max(1, min(args.other_steps, 20)), 1),
                           "kernels": sorted(r["rows"], key=lambda k: -k["avg_us"])[:8]})
        sweep, sweep_traffic_source = ball_query_sweep()
        beat("post")
        try:
            input_stage = input_stage_row()
        except Exception as e:  # (the input stage is not the path: its failure must not take the line down)
            input_stage = {"error": repr(e)}
        beat("post")
        modes = []
        for md in MODES:
            if args.no_graph:
                break
            mspec = dict(WORKLOADS[md["cfg"]], switches=md["switches"], lattice=md.get("lattice"))
            r = run_config(md["cfg"], mspec, args.other_steps, 4, graph=True, kernel_pass=False, announce=False, pipeline=args.pipeline,
                           extra_blocks=2)
            beat("post")
            base = res if md["cfg"] == main_index else None
            if base is None:
                base_ms = next((o["ms_per_step"] for ci, o in zip([c for c in WORKLOADS if c != main_index], others) if ci == md["cfg"]), None)
            else:
                base_ms = round(base["ms_per_step"], 4)
            modes.append({"mode": md["tag"], "workload": f"configs[{md['cfg']}]", "switches": {k: str(v) for k, v in md["switches"].items()},
                          "what": md["what"], "dtype": md["dtype"], "pipeline": r["pipeline"], "ms_per_step": round(median_ms(r), 4),
                          "ms_per_step_blocks": [round(v, 4) for v in r["block_ms"]],
                          "default_mode_ms_per_step": base_ms, "graph_equals_eager": r["outputs_agree"], **(r["mode_dev"] or {})})

    # The driver's record keeps the SCALAR values of `config` and the last 2 KB of this line: every figure the line is about is
    # repeated as a scalar in `config`, the long arrays come first and the compact summaries last.
    # The driver's record keeps the FIRST 24 keys of `config`: what the round is about comes first (the workload, the serial
    # figure, the ball-query fractions, every configuration's prefetch / serial ms); constants and the multi-rank checks follow.
    config = {"workload": spec["name"] + f", batch={res['B']}/GPU, seeded random weights",
              "global_batch": world * res["B"], "parallelism": f"batch-shard x{world}, RCCL all-gather of logits",
              "pipeline": res["pipeline"], "outputs_agree": res["outputs_agree"],
              "serial_ms_per_step": serial["ms_per_step"] if serial else None}
    if sweep is not None:
        for e in sweep:
            config[f"ball_hbm_frac_b{e['B']}"] = e["hbm_frac"]
        for e in sweep:
            config[f"ball_us_b{e['B']}"] = e["median_us"]
    summary = None
    if others is not None:
        summary = []
        tags = ("cfg2", "cfg3", "cfg4") if main_index == 1 else tuple(f"cfg{ci}" for ci in WORKLOADS if ci != main_index)
        for tag, o in zip(tags, others):
            config[f"{tag}_ms"] = o["ms_per_step"]
            config[f"{tag}_serial_ms"] = o["serial_ms_per_step"]
            r = o["roofline"] or {}
            summary.append({"cfg": tag, "ms": o["ms_per_step"], "serial_ms": o["serial_ms_per_step"], "clouds_per_s": o["clouds_per_s"],
                            "agree": o["outputs_agree"], "dominant": r.get("kernel"), "dims": r.get("dims"), "bound": r.get("bound"),
                            "frac": r.get("frac"), "avg_us": r.get("avg_us"), "traffic": r.get("traffic"), "alg_bytes": r.get("alg_bytes")})
        config["cfg_outputs_agree"] = all(bool(o["outputs_agree"]) for o in others)
    config.update({"serial_outputs_agree": serial["outputs_agree"] if serial else None,
                   "serial_clouds_per_s": serial["clouds_per_s"] if serial else None,
                   "enqueue_ms_per_step": round(res["enqueue_ms_per_step"], 4), "hip_graph": res["graph"],
                   "prefix_forked_at": os.environ.get("PASNL_BENCH_PREFETCH_AT", "start") if (args.model == "cls" and not args.AS and PREFIX_L2) else (FORK_AT_DEFAULT if args.model == "cls" else "head"),
                   "prefix_levels": 2 if (args.model == "cls" and not args.AS and PREFIX_L2) else 1, "settle_replays": SETTLE_REPLAYS,
                   "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")})
    for mo in modes or []:
        config[f"mode_{mo['mode']}_ms"] = mo["ms_per_step"]
    if multi:  # multi-rank checks (constants / nulls in a plain N = 1 run; --force-dist rehearses them with one rank)
        config.update({"rccl_ranks": rccl_ranks, "allreduce_check": allreduce_check, "gathered_rows_match_local": res["gathered_ok"],
                       "shards_differ": res["shards_differ"], "numa_node": numa_node, "cpus_bound": cpus_bound,
                       "per_rank_ms_per_step": [round(v, 4) for v in res["per_rank_ms"]]})
    # every tuning switch that was active in this process: an empty list on the driver's command line
    config["switches"] = sorted(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PASNL_BENCH_") and k not in SUPERVISOR_ENV) + \
        [f"--set {v}" for v in (args.set or [])]
    out = {
        "metric": "point-clouds/sec fwd (Bx1024 pts, ModelNet40 cls)" if args.model == "cls" else
                  f"point-clouds/sec fwd (Bx{res['N']} pts, pointasnl_{args.model})",
        "value": round(res["clouds_per_s"], 2),
        "unit": "point-clouds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(res["ms_per_step"], 4),
        "ms_per_step_blocks": {"blocks": [round(v, 4) for v in res["block_ms"]], "median": round(float(np.median(res["block_ms"])), 4),
                               "min": round(min(res["block_ms"]), 4), "max": round(max(res["block_ms"]), 4),
                               "note": "blocks[0] is the timed region `value` is computed from; the others follow it, each of `steps` steps between synchronisations"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": config,
        "roofline": roofline_of(rows),
        "cpu_baseline": cpu,
        "handwritten_kernel_us_per_step": round(sum(r["avg_us"] * r["launches"] for r in rows) / max(1, min(args.steps, 20)), 1),
        "kernels": rows,
        "other_configs": others,
        "modes": modes,
        "ball_traffic_source": sweep_traffic_source,
        "input_stage": input_stage,
        "serial": serial,
        "other_configs_summary": summary,
        "ball_query_sweep": sweep,
    }
    if args.launch_order:
        out["launch_order"] = res["launch_order"]
    if multi:
        dist.destroy_process_group()
    # the JSON line is the LAST line on stdout: RCCL leaves a version banner in the C stdio buffer, which would otherwise
    # be flushed after it when the process exits
    import ctypes

    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
