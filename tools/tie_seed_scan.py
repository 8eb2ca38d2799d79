import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench as B
import pointasnl_amd as P
from pointasnl_amd import tf_sampling
for seed in range(1235, 1245):
    pc = B.synth_clouds(seed, 64, 1024); x = torch.from_numpy(pc).cuda()
    st = []; P.nearest_neighbors.knn_batch(x, x, 32, dtype=torch.int32, stats=st); a = int(st[0].sum())
    fi = tf_sampling.farthest_point_sample(512, x); x1 = tf_sampling.gather_point(x, fi)
    fi2 = tf_sampling.farthest_point_sample(128, x1); q2 = tf_sampling.gather_point(x1, fi2)
    st = []; P.nearest_neighbors.knn_batch(x1, q2, 64, dtype=torch.int32, stats=st); b2 = int(st[0].sum())
    print(seed, "L1 self flagged", a, "L2 flagged", b2, flush=True)
