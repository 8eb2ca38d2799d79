"""One saved batch of tools/tie_path_fuzz.py: default order vs every-query-through-the-tree vs the reference library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pointasnl_amd as P
from oracle import ref
d = np.load(sys.argv[1])
sup, qry, k = d["sup"], d["qry"], int(d["k"])
s, q = torch.from_numpy(sup).cuda(), torch.from_numpy(qry).cuda()
stats = []
got = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats).cpu().numpy()
full = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="nanoflann").cpu().numpy()
canon = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, tie_order="index").cpu().numpy()
want = ref.knn_batch(sup, qry, k)
print("listed", stats[0].tolist(), "left", stats[1].tolist(), stats[2].tolist())
print("default == ref", np.array_equal(got, want), "| tree == ref", np.array_equal(full, want), "| default == tree", np.array_equal(got, full))
for b in range(sup.shape[0]):
    for j in range(qry.shape[1]):
        if not np.array_equal(got[b, j], want[b, j]):
            dd = lambda idx: (((sup[b, idx].astype(np.float32) - qry[b, j]) ** 2)).astype(np.float32)
            dist = lambda idx: np.array([np.float32(np.float32(x[0] + x[1]) + x[2]) for x in dd(idx)])
            w = np.nonzero(got[b, j] != want[b, j])[0]
            print("cloud", b, "query", j, "differs at slots", w.tolist())
            print("  default ", got[b, j][w].tolist(), dist(got[b, j][w]).tolist())
            print("  ref     ", want[b, j][w].tolist(), dist(want[b, j][w]).tolist())
            print("  canon   ", canon[b, j][w].tolist())
            print("  coords default", sup[b, got[b, j][w]].tolist())
            print("  coords ref    ", sup[b, want[b, j][w]].tolist())
            print("  query", qry[b, j].tolist())
