"""Makes tests/golden/knn_prune_rounding_case.npz: one 7876-point cloud (a needle quantised to 2^-8, hundreds of duplicated points) and
one query for which the reference library itself does NOT return the 65 nearest points: nanoflann's pruning bound of one subtree
(mindistsq + cut_dist - dists[idx], nanoflann.hpp:1396-1404) rounds to one ulp ABOVE the distance of a point inside it while the
result set's worst distance equals that bound's neighbour -- the subtree is skipped and point 6152 (closer than the list's last
three entries) is missing from the reference's list.  Found by tools/tie_path_fuzz.py (batch 138807 of seed 1).
Expected output = the reference library's own (oracle/_ref/libref_knn.so, built from /root/reference by oracle/Makefile):
    python tests/golden/make_knn_prune_case.py tools/_cases/fuzz_fail_138807.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref

d = np.load(sys.argv[1])
sup, qry, k = d["sup"][3:4].copy(), d["qry"][3:4].copy(), int(d["k"])
want = ref.knn_batch(sup, qry, k)
dist = ((sup[0].astype(np.float64) - qry[0, 0]) ** 2).sum(1)
exact = np.argsort(dist, kind="stable")[:k]
assert set(exact.tolist()) != set(want[0, 0].tolist()), "the reference returns the exact set here: not the case this fixture is about"
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "knn_prune_rounding_case.npz"), sup=sup, qry=qry, k=k, reference=want)
print("reference misses", sorted(set(exact.tolist()) - set(want[0, 0].tolist())), "and holds instead", sorted(set(want[0, 0].tolist()) - set(exact.tolist())))
