"""Are the tie flags of pasnl_knn_batch_ref deterministic and exactly the queries whose (K+1)-list holds equal distances among
its first K + 1 entries?  python tools/knn_flag_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import pointasnl_amd as P

for name, sup, m, k in [("scannet L1", B.synth_scannet(3, 16, 8192)[..., :3].copy(), 1024, 32), ("cls L1", B.synth_clouds(1, 64, 1024), 512, 32),
                        ("kitti L1", B.synth_kitti(4, 8, 10240), 1280, 32)]:
    s = torch.from_numpy(np.ascontiguousarray(sup)).cuda(); q = s[:, :m].contiguous()
    counts = []
    for it in range(6):
        stats = []
        out = P.nearest_neighbors.knn_batch(s, q, k, dtype=torch.int32, stats=stats)
        torch.cuda.synchronize()
        counts.append(stats[0].cpu().numpy().copy())
    same = all((c == counts[0]).all() for c in counts)
    # expected: the canonical (k+1)-list has two equal distances among entries 0..k
    idx = P.nearest_neighbors.knn_batch(s, q, k + 1, dtype=torch.int64, tie_order="index").cpu().numpy()
    d = ((sup[:, None, :m, :].transpose(0, 2, 1, 3) - np.take_along_axis(sup[:, None], idx[..., None], 2)) ** 2)
    qq = sup[:, :m]
    pts = np.take_along_axis(sup[:, None, :, :], idx[..., None], axis=2)
    dx = qq[:, :, None, :] - pts
    dist = ((dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2]).astype(np.float32)
    exp = (dist[..., 1:] == dist[..., :-1]).any(-1).sum(1)
    print(name, "deterministic" if same else "NOT deterministic", [int(c.sum()) for c in counts], "expected per cloud", exp.tolist(), "got", counts[0].tolist(), flush=True)
