"""CPU, world_size 2, gloo: the batch-shard + all-gather path of pointasnl_amd/sharding.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointasnl_amd import sharding

    clouds = torch.arange(total * 5 * 3, dtype=torch.float32).reshape(total, 5, 3)

    def forward(x):  # stand-in for the per-cloud forward: any function that is independent per cloud
        return torch.stack([x.sum(dim=(1, 2)), x.amax(dim=(1, 2)), x[:, 0, 0]], dim=1)

    full = sharding.sharded_forward(forward, clouds, 3)
    lo, hi = sharding.shard_range(rank, world, total)
    g = sharding.LogitsGather(world, 4, 3, "cpu")
    eq = g.all_gather(torch.full((4, 3), float(rank))).clone()
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"e{rank}.npy"), eq.numpy())
    np.save(os.path.join(out_dir, f"s{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_sharded_forward_equals_unsharded(tmp_path, total):
    world = 2
    port = 29600 + (os.getpid() % 300) + total
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    clouds = torch.arange(total * 5 * 3, dtype=torch.float32).reshape(total, 5, 3)
    want = torch.stack([clouds.sum(dim=(1, 2)), clouds.amax(dim=(1, 2)), clouds[:, 0, 0]], dim=1).numpy()
    covered = []
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"r{r}.npy"), want)  # every rank holds the full result
        e = np.load(tmp_path / f"e{r}.npy")
        np.testing.assert_array_equal(e, np.repeat(np.arange(world, dtype=np.float32), 4)[:, None] * np.ones((1, 3)))
        covered.append(tuple(np.load(tmp_path / f"s{r}.npy")))
    assert covered[0][0] == 0 and covered[-1][1] == total and covered[0][1] == covered[1][0]


def test_shard_range_partitions():
    from pointasnl_amd.sharding import shard_range

    for total in (1, 7, 64, 65):
        for world in (1, 2, 4, 8):
            r = [shard_range(k, world, total) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_parse_cpulist_and_numa_binding(tmp_path):
    """The launcher pins a rank to its GPU's NUMA node: cpulist parsing, and a missing topology leaves the affinity alone."""
    import os

    from pointasnl_amd import sharding

    assert sharding.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert sharding.parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    assert sharding.bind_to_gpu_numa(0, sysfs=str(tmp_path)) == (None, 0)  # no sysfs topology (and no GPU here)
    assert os.sched_getaffinity(0) == before
