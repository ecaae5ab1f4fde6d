"""Multi-GPU plumbing: samples are split across ranks, the scene is replicated, and the float
accumulation buffers are summed onto rank 0 once per frame (RCCL reduce over xGMI).

Upstream is single-GPU (no collective anywhere), so this has no upstream counterpart.  Both halves live in the
library (include/rtcore.h): rt_partition_samples cuts the frame into contiguous runs of whole semantic batches, so that
every rank traces exactly the samples a single GPU would (the random streams are keyed by (pixel, global sample
index) and the jitter by global batch index, hence the set of samples is the same for any rank count and only the
float summation order differs), and rt_reduce_to_root runs the one ncclReduce.  This module binds them for
one-process-per-GPU jobs started by torch.distributed.run, where torch.distributed only carries the communicator id.
"""
from . import capi


def partition_samples(n_samples, semantic_batch, world, rank):
    """(first_sample, count) of `rank`: contiguous runs of whole batches, remainders to the low ranks."""
    try:
        return capi.partition_samples(n_samples, semantic_batch, world, rank)
    except capi.RtError as e:
        raise ValueError(str(e))


def make_group(ctx, rank, world):
    """The rt_group of a one-process-per-GPU job: rank 0 makes the communicator id, torch.distributed hands it round."""
    import torch.distributed as dist
    box = [capi.group_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return capi.Group([ctx], first_rank=rank, world=world, unique_id=box[0])


def reduce_to_root(group, tensor, stream=0):
    """Sum `tensor` (float32, on this rank's GPU) of every rank onto rank 0, in place there, with one ncclReduce."""
    group.reduce_to_root([tensor.data_ptr()], tensor.numel(), [stream])
    return tensor
