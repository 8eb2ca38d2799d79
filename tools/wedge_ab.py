"""ADVICE r05 (ball_grid.hip:397): does the wedge test separate the library before / after the fix?
python tools/wedge_ab.py <lib.so>  -> number of mismatching (n, r) cases"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from pointasnl_amd import _hip
if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import pointasnl_amd as P
from oracle import ops as O
bad = 0
for n in (2048, 2047, 1500):
    for r in (0.24, 0.3, 0.12):
        rng = np.random.default_rng(n * 7 + int(r * 100))
        b, m, ns = 3, 400, 32
        t = rng.random((b, n, 1)) ** 0.5
        xyz1 = np.concatenate([10 * t, 1.5 * t * rng.random((b, n, 1)), 1.5 * t * rng.random((b, n, 1))], -1).astype(np.float32)
        xyz1[:, 0] = xyz1.max(1) - np.float32(0.05)
        xyz2 = (xyz1[:, :1] + (rng.random((b, m, 3)).astype(np.float32) - 0.5) * np.float32(2 * r)).astype(np.float32)
        xyz2[:, m // 2:] = xyz1[:, rng.integers(0, n, m - m // 2)][0][None]
        wi, wc = O.query_ball_point(r, ns, xyz1, xyz2)
        gi, gc = P.tf_grouping.query_ball_point(r, ns, torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda())
        ok = np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gc.cpu().numpy(), wc)
        bad += not ok
        print(n, r, "ok" if ok else f"MISMATCH rows={int((gi.cpu().numpy() != wi).any(-1).sum())} cnt={int((gc.cpu().numpy() != wc).sum())}")
print(os.path.basename(_hip.LIB_PATH), "mismatching cases:", bad)
