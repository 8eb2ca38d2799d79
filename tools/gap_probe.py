"""Cost of a graph node by who launched it: chains of tiny kernels captured into a HIP graph, average time per node.
    python tools/gap_probe.py"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import pointasnl_amd  # noqa: E402
from pointasnl_amd import _hip  # noqa: E402
from pointasnl_amd.utils import pointnet_util as PU  # noqa: E402


def per_node(body, nodes, reps=50):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nodes)


x = torch.zeros(1024, device="cuda")
big = torch.zeros(64, 512, 128, device="cuda")
pool_in = torch.rand(4, 1, 64, 64, device="cuda")
pool_out = torch.empty(4, 64, device="cuda")
xyz = torch.rand(64, 1024, 3, device="cuda")
idx = torch.randint(0, 1024, (64, 512), device="cuda", dtype=torch.int32)


def torch_tiny():
    x.add_(1.0)


def pasnl_tiny():
    PU.max_pool_points(pool_in, out=pool_out)


def pasnl_gather():
    pointasnl_amd.tf_sampling.gather_point(xyz, idx)


def torch_big():
    big.add_(1.0)


N = 20
print(f"torch tiny x{N}:            {per_node(lambda: [torch_tiny() for _ in range(N)], N):6.2f} us/node")
print(f"pasnl tiny x{N}:            {per_node(lambda: [pasnl_tiny() for _ in range(N)], N):6.2f} us/node")
print(f"pasnl gather x{N}:          {per_node(lambda: [pasnl_gather() for _ in range(N)], N):6.2f} us/node")
print(f"alternating tiny x{N}:      {per_node(lambda: [(torch_tiny(), pasnl_tiny()) for _ in range(N // 2)], N):6.2f} us/node")
print(f"torch 16 MB rw x{N}:        {per_node(lambda: [torch_big() for _ in range(N)], N):6.2f} us/node")
print(f"torch 16 MB + pasnl tiny:   {per_node(lambda: [(torch_big(), pasnl_tiny()) for _ in range(N // 2)], N):6.2f} us/node")
print(f"torch 16 MB + torch tiny:   {per_node(lambda: [(torch_big(), torch_tiny()) for _ in range(N // 2)], N):6.2f} us/node")
