"""Batch sharding of the forward path over the GPUs of one node.

Every op and cell of the path is independent per cloud (inference batch norm uses running statistics), so the
path shards over the batch with NO data-path collective: rank r of W owns clouds [r*B/W, (r+1)*B/W).  The only
exchange is one all-gather of the per-shard logits per forward (cls: (B/W,40) fp32 = 10 KB per rank) -- on the
fully connected xGMI fabric a direct all-gather, latency- not bandwidth-bound.  One process per GPU; backend
"nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def shard_range(rank, world, total):
    """Clouds owned by `rank`: contiguous, sizes differ by at most one when world does not divide total."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class LogitsGather:
    """Pre-allocated all-gather of equally sized per-rank logits (weak scaling: same batch on every rank)."""

    def __init__(self, world, rows_per_rank, width, device, dtype=torch.float32, force=False):
        self.world = world
        self.force = force  # issue the collective even with one rank (exercises the RCCL path on a 1-GPU box)
        self.out = torch.empty((world * rows_per_rank, width), dtype=dtype, device=device)

    def all_gather(self, local):
        if self.world == 1 and not self.force:
            self.out.copy_(local)
        else:
            dist.all_gather_into_tensor(self.out, local.contiguous())
        return self.out

    # A serving loop over several output buffers: the collective of step k must not hold up step k + 1.  The blocking form above
    # makes the CALLER'S stream wait for the collective (c10d: work.wait() after the enqueue), so the next captured forward
    # queues behind the all-gather and its cross-stream hand-shakes -- measured with one rank on one MI355X: 1.87 ms per step
    # against 1.32 without the collective.  Here the collective is enqueued asynchronously (it still starts behind everything
    # the caller's stream holds at that moment) and the caller's stream waits for it only when the SAME buffer slot is about to
    # be produced again (`before_reuse(slot)`), i.e. one or more steps later.
    def all_gather_async(self, local, slot):
        if self.world == 1 and not self.force:
            self.out.copy_(local)
            return self.out
        if not hasattr(self, "_pending"):
            self._pending = {}
        self._pending[slot] = dist.all_gather_into_tensor(self.out, local.contiguous(), async_op=True)
        return self.out

    def before_reuse(self, slot):
        """call before the producer of buffer `slot` runs again: its last all-gather has to have read it"""
        w = getattr(self, "_pending", {}).pop(slot, None)
        if w is not None:
            w.wait()  # (a stream-level wait on the current stream, not a host synchronisation)

    def drain(self):
        for slot in list(getattr(self, "_pending", {})):
            self.before_reuse(slot)


def gather_ragged(local, total, width):
    """All-gather for shards of unequal size (strong-scaling split of a fixed batch): pad to the largest shard,
    gather, drop the padding.  Returns the (total, width) logits in cloud order on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(r, world, total)[1] - shard_range(r, world, total)[0] for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros((mx, width), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * mx, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def sharded_forward(forward, clouds, width):
    """Run `forward` on this rank's shard of `clouds` (B_total,...) and return the full (B_total,width) result."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(rank, world, clouds.shape[0])
    local = forward(clouds[lo:hi])
    return gather_ragged(local, clouds.shape[0], width)


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> sorted list of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of the host bridge a HIP device hangs off (its PCI function's sysfs `numa_node`), or None."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read())
        return node if node >= 0 else None
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def bind_to_gpu_numa(device_index, sysfs="/sys"):
    """One process per GPU: pin this rank's host threads to the CPUs of its GPU's NUMA node, so that the launch thread, the
    RCCL proxy thread and their queues do not cross the socket interconnect (eight ranks on a two-socket host otherwise
    share whatever cores the scheduler picks).  Returns (node, number of CPUs bound) or (None, 0) when the topology is not
    exposed (containers without sysfs NUMA files): the rank then keeps its inherited affinity."""
    node = gpu_numa_node(device_index, sysfs)
    if node is None:
        return None, 0
    try:
        cpus = parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return node, 0
        os.sched_setaffinity(0, allowed)
        return node, len(allowed)
    except OSError:
        return node, 0
