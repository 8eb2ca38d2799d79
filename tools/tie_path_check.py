"""The tie-path kernel against the live reference library on batches it takes (at most 32 listed queries): clouds quantised to
2^-q so that runs of equal distances of every shape occur (inside the row, across its end, duplicates of the query).
python tools/tie_path_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pointasnl_amd as P
from oracle import ref

rng = np.random.default_rng(7)
bad = 0
tot = res = 0
for n, k, m in [(7, 3, 7), (40, 8, 20), (300, 16, 32), (1024, 32, 32), (1024, 1, 32), (2048, 64, 16), (2048, 100, 8), (4096, 32, 16), (8192, 32, 8), (8192, 200, 4)]:
    for q in (5, 7, 9, 11, 13):
        for b in (1, 3):
            sup = rng.normal(size=(b, n, 3)).astype(np.float32)
            sup /= np.maximum(1.0, np.abs(sup).max())
            sup = (np.round(sup * 2 ** q) / 2 ** q).astype(np.float32)
            if q >= 11:  # a few exact duplicates
                sup[:, 5] = sup[:, 3]
            mm = max(1, m // b)
            qry = np.ascontiguousarray(sup[:, :mm]) if q % 4 != 1 else (np.round(rng.normal(size=(b, mm, 3)) * 2 ** q) / 2 ** q).astype(np.float32) * 0.3
            stats = []
            got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, dtype=torch.int32, stats=stats)
            nflag, nwork = stats[0].cpu().numpy(), stats[1].cpu().numpy()
            want = ref.knn_batch(sup, qry, k)
            ok = np.array_equal(got.cpu().numpy(), want)
            tot += int(nflag.sum()); res += int(nflag.sum() - nwork.sum())
            print(f"n={n} k={k} m={mm} b={b} q={q}: listed {nflag.tolist()} left to the builds {nwork.tolist()} {'ok' if ok else 'MISMATCH'}", flush=True)
            if not ok:
                bad += 1
                w = np.argwhere((got.cpu().numpy() != want).any(-1))
                print("   first rows:", w[:4].tolist())
print("mismatches:", bad, " listed", tot, "resolved by the tie paths", res)
