"""Batches of MANY clouds (more listed clouds than the tie paths take: the hand-over to the builds) against the reference library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pointasnl_amd as P
from oracle import ref
rng = np.random.default_rng(5)
bad = 0
for b, n, m, k, q in [(300, 100, 20, 8, 4), (1000, 64, 8, 16, 3), (70, 2048, 4, 64, 6), (40, 3000, 2, 32, 6), (33, 9000, 1, 16, 7), (500, 300, 3, 100, 4), (65, 1024, 1024, 32, 30)]:
    sup = rng.normal(size=(b, n, 3)).astype(np.float32)
    sup /= np.abs(sup).max()
    if q < 30: sup = (np.round(sup * 2 ** q) / 2 ** q).astype(np.float32)
    qry = np.ascontiguousarray(sup[:, :m])
    stats = []
    got = P.nearest_neighbors.knn_batch(torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda(), k, dtype=torch.int32, stats=stats).cpu().numpy()
    want = ref.knn_batch(sup, qry, k)
    ok = np.array_equal(got, want)
    bad += not ok
    print(b, n, m, k, q, "listed", int(stats[0].sum()), "clouds listed", int((stats[0] > 0).sum()), "left", int(stats[1].sum()), "ok" if ok else "MISMATCH", flush=True)
print("bad", bad)
