"""Multi-GPU plumbing: samples are split across ranks, the scene is replicated, and the float
accumulation buffers are summed onto rank 0 once per frame (RCCL reduce over xGMI; gloo on CPU).

Upstream is single-GPU (no collective anywhere), so this has no upstream counterpart.  The split is
by whole semantic batches so that every rank traces exactly the samples a single GPU would: the
random streams are keyed by (pixel, global sample index) and the jitter by global batch index,
hence the set of samples is the same for any rank count and only the float summation order differs.
"""


def partition_samples(n_samples, semantic_batch, world, rank):
    """(first_sample, count) of `rank`: contiguous runs of whole batches, remainders to the low ranks."""
    if n_samples < 0 or semantic_batch <= 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad partition arguments")
    n_batches = (n_samples + semantic_batch - 1) // semantic_batch
    base, extra = divmod(n_batches, world)
    first_batch = rank * base + min(rank, extra)
    my_batches = base + (1 if rank < extra else 0)
    first = first_batch * semantic_batch
    end = min(n_samples, (first_batch + my_batches) * semantic_batch)
    return first, max(0, end - first)


def reduce_to_root(buffer, root=0):
    """Sum `buffer` (a torch tensor, one per rank) onto `root` in place.  No-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(buffer, dst=root, op=dist.ReduceOp.SUM)
    return buffer
