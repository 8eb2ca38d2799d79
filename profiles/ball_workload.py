"""The ball-query launches the counter passes of profiles/collect_ball_counters.sh profile: north-star shape
(1024 support, 512 queries, r = 0.2, nsample = 32) at B = 64, 1024, 4096, 5 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import pointasnl_amd as P
for b in (64, 1024, 4096):
    x = torch.from_numpy(B.synth_clouds(1, b, 1024)).cuda()
    q = x[:, :512].contiguous()
    for _ in range(5):
        P.tf_grouping.query_ball_point(0.2, 32, x, q)
    torch.cuda.synchronize()
