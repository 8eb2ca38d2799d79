"""Randomised sweep of the default kNN order against the live reference library on tie-rich clouds (few listed queries per batch, so
that the tie paths take them): python tools/tie_path_fuzz.py [seconds] [seed] [plain]
plain: float clouds as they come (no quantisation), every point a query -- queries with DISTINCT distances, which keep the canonical
row: does the reference ever return something else there (its pruning bound's rounding)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pointasnl_amd as P
from oracle import ref

def run(budget=120.0, seed=1, plain=False, max_cases=None, save_failures=True):
    """-> (batches, queries, listed, mismatches)"""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    queries = 0
    cases = listed = left = bad = 0
    shapes = {}
    while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
        n = int(rng.choice([rng.integers(1, 12), rng.integers(11, 200), rng.integers(200, 2049), rng.integers(2049, 8193), rng.integers(8193, 12000)],
                           p=[0.05, 0.25, 0.4, 0.2, 0.1]))
        k = int(min(n, rng.choice([1, 2, 3, 8, 16, 32, 33, 64, 65, 100, 256], p=[.05, .05, .05, .1, .2, .25, .05, .1, .05, .05, .05])))
        b = int(rng.integers(1, 5))
        m = int(rng.integers(1, 1 + min(n, 24)))
        q = int(rng.integers(2, 14))
        kind = rng.integers(0, 4)
        if plain:
            n = int(rng.integers(64, 4000)); k = int(min(n, rng.choice([8, 16, 32, 64]))); m = n; q = 30; kind = int(rng.integers(0, 3))
        sup = rng.normal(size=(b, n, 3)).astype(np.float32)
        if kind == 1: sup[..., 2] = 0            # a plane
        if kind == 2: sup[..., 1:] *= 0.01       # a needle
        sup /= max(1e-6, np.abs(sup).max())
        sup = (np.round(sup * 2 ** q) / 2 ** q).astype(np.float32)
        if kind == 3 and n > 4:                  # duplicated points
            for _ in range(int(rng.integers(1, 4))):
                i, j = rng.integers(0, n, 2)
                sup[:, i] = sup[:, j]
        if plain or rng.random() < 0.5:
            qry = np.ascontiguousarray(sup[:, rng.permutation(n)[:m]])
        else:
            qry = ((np.round(rng.normal(size=(b, m, 3)) * 2 ** q) / 2 ** q) * 0.4).astype(np.float32)
        stats = []
        i64 = rng.random() < 0.3
        got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k,
                                            dtype=torch.int64 if i64 else torch.int32, stats=stats).cpu().numpy()
        want = ref.knn_batch(sup, qry, k)
        cases += 1
        queries += b * m
        listed += int(stats[0].sum()); left += int(stats[1].sum())
        key = "n<=2048,k<=64" if n <= 2048 and k <= 64 else ("n<=8192" if n <= 8192 else "n>8192")
        shapes[key] = shapes.get(key, 0) + int(stats[0].sum())
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH n", n, "k", k, "b", b, "m", m, "q", q, "kind", kind, "i64", i64, "listed", stats[0].tolist(), "left", stats[1].tolist(), flush=True)
            if save_failures:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{cases}.npz"), sup=sup, qry=qry, k=k)
    print(f"{cases} batches, {queries} queries, {listed} listed queries ({shapes}), {left} left to the builds by the standalone tie-path kernel, mismatches: {bad}")
    return cases, queries, listed, bad


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, len(sys.argv) > 3 and sys.argv[3] == "plain")
