#!/usr/bin/env python
"""bench_ops.py -- per-kernel micro-benchmarks of the hand-written ops (not the driver's bench; see bench.py).

Times each C-ABI kernel alone with HIP events on the launch stream (median of --iters launches after warm-up),
at the shapes the three models use and over a batch sweep, and prints algorithmic GB/s and
TFLOP/s (SURVEY 8(d) formulas, bench.algorithmic) against the MI355X peaks.  Writes gpurun_out/bench_ops.json.

    python bench_ops.py [--only fps,knn,...] [--iters 30] [--sweep]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sweep", action="store_true", help="batch sweep for the roofline-vs-batch curves")
    ap.add_argument("--out", default="gpurun_out/bench_ops.json")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))

    import torch

    import pointasnl_amd as P
    from pointasnl_amd import _hip
    from pointasnl_amd.utils import pointasnl_util as U

    torch.cuda.set_device(0)
    rng = np.random.default_rng(0)

    def cloud(b, n):
        return torch.from_numpy(B.synth_clouds(int(rng.integers(1 << 30)), b, n)).cuda()

    def timeit(fn, label):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _hip.PROFILE = []
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        recs = _hip.PROFILE
        _hip.PROFILE = None
        out = []
        bysym = {}
        for sym, ints, e0, e1 in recs:
            bysym.setdefault((sym, ints), []).append(e0.elapsed_time(e1) * 1e3)
        for (sym, ints), us in bysym.items():
            med = float(np.median(us))
            by, fl, bound = B.algorithmic(sym, ints)
            row = {"label": label, "kernel": sym, "dims": list(ints), "median_us": round(med, 2), "min_us": round(min(us), 2),
                   "GB/s": round(by / med / 1e3, 1), "hbm_frac": round(by / med / 1e3 / B.HBM_PEAK_GBS, 4),
                   "TFLOP/s": round(fl / med / 1e6, 2), "alg_MB": round(by / 1e6, 2)}
            out.append(row)
            print(f"{label:34s} {sym[6:]:24s} {str(list(ints)):34s} {med:9.1f} us  {row['GB/s']:8.1f} GB/s ({row['hbm_frac']*100:5.2f}% HBM)"
                  f"  {row['TFLOP/s']:7.2f} TF/s", flush=True)
        return out

    rows = []

    def want(name):
        return not only or name in only

    batches = [64] + ([16, 256, 1024, 4096] if args.sweep else [])

    if want("fps"):
        for b in batches:
            x = cloud(b, 1024)
            rows += timeit(lambda: P.tf_sampling.farthest_point_sample(512, x), f"fps cls-L1 B={b}")
        x = cloud(64, 512)
        rows += timeit(lambda: P.tf_sampling.farthest_point_sample(128, x), "fps cls-L2 B=64")
        x = cloud(16, 8192)
        rows += timeit(lambda: P.tf_sampling.farthest_point_sample(1024, x), "fps scannet-L1 B=16")
        x = cloud(8, 10240)
        rows += timeit(lambda: P.tf_sampling.farthest_point_sample(1280, x), "fps kitti-L1 B=8")
    if want("knn"):
        for b in batches:
            x = cloud(b, 1024)
            q = x[:, :512].contiguous()
            rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, q, 32, dtype=torch.int32), f"knn cls-L1 B={b}")
        x = cloud(64, 512)
        q = x[:, :128].contiguous()
        rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, q, 64, dtype=torch.int32), "knn cls-L2 B=64")
        x = cloud(16, 8192)
        q = x[:, :1024].contiguous()
        rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, q, 32, dtype=torch.int32), "knn scannet-L1 B=16")
        rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, x, 16, dtype=torch.int32), "knn scannet-fa4 self B=16")
        x = cloud(8, 10240)
        rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, x, 32, dtype=torch.int32), "knn kitti-L0 B=8")
    if want("knn_tree"):
        # the reference's own order among equal distances (csrc/knn_tree.hip): an exactness mode, timed so that its cost is known
        for (b, n, m, k, name) in [(64, 1024, 512, 32, "cls-L1"), (16, 8192, 1024, 32, "scannet-L1"), (16, 8192, 8192, 16, "scannet-fa4 self"),
                                   (8, 10240, 10240, 32, "kitti-L0")]:
            x = cloud(b, n)
            q = x[:, :m].contiguous()
            rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, q, k, dtype=torch.int32, tie_order="nanoflann"),
                           f"knn nanoflann-order {name} B={b}")
            rows += timeit(lambda: P.nearest_neighbors.knn_batch(x, q, k, dtype=torch.int32), f"knn canonical {name} B={b}")
    if want("ball"):
        for b in batches:
            x = cloud(b, 1024)
            q = x[:, :512].contiguous()
            rows += timeit(lambda: P.tf_grouping.query_ball_point(0.2, 32, x, q), f"ball north-star B={b}")
        x = torch.rand((32, 512, 3), device="cuda")
        q = torch.rand((32, 128, 3), device="cuda")
        rows += timeit(lambda: P.tf_grouping.query_ball_point(0.1, 64, x, q), "ball ref-microbench B=32")
        x = cloud(16, 1024)
        rows += timeit(lambda: P.tf_grouping.query_ball_point(0.07, 20, x, x), "ball scannet-loss B=16")
    if want("group"):
        for b in batches:
            pts = torch.rand((b, 512, 128), device="cuda")
            idx = torch.randint(0, 512, (b, 128, 64), device="cuda", dtype=torch.int32)
            rows += timeit(lambda: P.tf_grouping.group_point(pts, idx), f"group cls-L2 C=128 B={b}")
        pts = torch.rand((64, 1024, 3), device="cuda")
        idx = torch.randint(0, 1024, (64, 512, 32), device="cuda", dtype=torch.int32)
        rows += timeit(lambda: P.tf_grouping.group_point(pts, idx), "group cls-L1 C=3 B=64")
    if want("interp"):
        x1, x2 = cloud(16, 8192), cloud(16, 1024)
        rows += timeit(lambda: P.tf_interpolate.three_nn(x1, x2), "three_nn scannet-fa4 B=16")
        for (b, n, m, name) in [(8, 10240, 1280, "kitti-fa4"), (8, 1280, 320, "kitti-fa3"), (16, 1024, 256, "scannet-fa3"), (16, 256, 64, "scannet-fa2")]:
            y1, y2 = cloud(b, n), cloud(b, m)
            rows += timeit(lambda: P.tf_interpolate.three_nn(y1, y2), f"three_nn {name} B={b}")
        d, i = P.tf_interpolate.three_nn(x1, x2)
        w = P.tf_interpolate.three_weights(d)
        pts = torch.rand((16, 1024, 128), device="cuda")
        rows += timeit(lambda: P.tf_interpolate.three_interpolate(pts, i, w), "interpolate scannet-fa4 c=128 B=16")
        rows += timeit(lambda: P.tf_interpolate.three_weights(d), "three_weights scannet-fa4 B=16")
    if want("nl"):
        for variant in (1, 2):
            for (b, p, n, cb, name) in [(64, 512, 1024, 32, "cls-L1"), (64, 128, 512, 64, "cls-L2"),
                                        (16, 1024, 8192, 32, "scannet-L1"), (8, 1280, 10240, 32, "kitti-L1_1")]:
                q = torch.randn((b, p, cb), device="cuda")
                kv = torch.randn((b, n, 2 * cb), device="cuda")
                rows += timeit(lambda: U.nl_attention(q, kv, variant=variant), f"nl {name} v{variant} B={b}")
    if want("as"):
        for (g, a, cb, name) in [(64 * 512, 12, 32, "cls-L1"), (64 * 128, 12, 65, "cls-L2"), (16 * 1024, 8, 32, "scannet-L1")]:
            q = torch.randn((g, a, cb), device="cuda")
            kv = torch.randn((g, a, 2 * cb), device="cuda")
            rows += timeit(lambda: U.as_attention(q, kv), f"as_attention {name}")
    if want("sacell"):
        from pointasnl_amd.utils import tf_util
        tf_util.set_store(tf_util.VariableStore(seed=5))
        for (b, n, c, m, k, c1, name) in [(64, 1024, 3, 512, 32, 64, "cls-L1"), (64, 512, 128, 128, 64, 128, "cls-L2"),
                                          (16, 8192, 3, 1024, 32, 32, "scannet-L1"), (16, 1024, 64, 256, 32, 64, "scannet-L2"),
                                          (8, 10240, 32, 1280, 32, 32, "kitti-L1")]:
            xyz = cloud(b, n)
            feat = torch.randn((b, n, c), device="cuda")
            idx = torch.randint(0, n, (b, m, k), device="cuda", dtype=torch.int32)
            nx = xyz[:, :m].contiguous()
            with tf_util.variable_scope(name):
                rows += timeit(lambda: U.sa_cell(xyz, feat, idx, nx, [c1, c1, 2 * c1], False, None, None, True), f"sa_cell {name} B={b}")
                if args.sweep:
                    def two():
                        npnt, _ = U.sa_group(xyz, feat, idx, nx)
                        U.sa_local_cell(npnt, [c1, c1, 2 * c1], False, None, None, True)
                    rows += timeit(two, f"sa_group+local_cell {name} B={b}")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
