"""Multi-GPU plumbing for the `#.list` path: one sample (BAM) per GPU, then ONE sum of the
accumulating buffers (difference arrays + tile sums are contiguous int32, include/pandepth_amd.h
pd_device_buffer) over RCCL/xGMI, scan on the root.  Difference arrays are linear, so summing them
before the prefix sum equals the reference's sequential accumulation of every file into one array
(PD:2704-3014).  torch.distributed is used as the launcher/collective layer only."""
import torch
import torch.distributed as dist


class _DevBuf:
    """Zero-copy view of a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4",
                                         "data": (int(ptr), False), "version": 3}


def buffer_view(engine, device):
    """int32 torch tensor aliasing the engine's accumulating buffer (no copy)."""
    ptr, n_words, _ = engine.device_buffer()
    return torch.as_tensor(_DevBuf(ptr, n_words), device=device)


def assign_samples(n_samples, rank, world):
    """Round-robin assignment of list entries to ranks (sample k -> rank k % world)."""
    return [k for k in range(n_samples) if k % world == rank]


def sum_to_root(buf, root=0, group=None):
    """In-place int32 sum of every rank's buffer into `root`'s.  Returns True on the root."""
    if dist.is_initialized():
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM, group=group)
        return dist.get_rank(group) == root
    return True
