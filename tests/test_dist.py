"""The multi-rank plumbing on CPU: rt_partition_samples (library, through the C ABI) and a world-size-2 gloo job.

What this covers: the partition (every sample once, whole batches), and that frames traced for the two partitions add up
to the one-rank frame within float-summation tolerance.  The frames here come from the CPU oracle, because this
container has no GPU; the kernels going through the same split and through rt_reduce_to_root (RCCL) are covered on the
GPU box by tests/test_gpu_multi.py."""
import os
import socket

import numpy as np
import pytest

from raytrace_amd.dist import partition_samples


@pytest.mark.parametrize("n,b,w", [(8192, 16, 8), (2048, 16, 3), (10, 4, 2), (7, 16, 4), (0, 4, 2), (100, 1, 7)])
def test_partition_covers_every_sample_once_in_whole_batches(n, b, w):
    seen = []
    for r in range(w):
        first, count = partition_samples(n, b, w, r)
        assert first % b == 0
        if count and first + count < n:
            assert count % b == 0            # only the run that ends the frame may end mid-batch
        seen += list(range(first, first + count))
    assert seen == list(range(n))


def test_partition_rejects_bad_arguments():
    with pytest.raises(ValueError):
        partition_samples(8, 0, 2, 0)
    with pytest.raises(ValueError):
        partition_samples(8, 4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from raytrace_amd import scene
    from raytrace_amd.dist import partition_samples
    dist.init_process_group("gloo", rank=rank, world_size=world)
    asm = scene.material_box(width=24, height=16, n_samples=16)
    objs = scene.finalize(asm, orc.OracleBackend())
    sc = orc.Scene(objs, asm.materials, asm.textures)
    p = orc.make_params(asm.config, 0x5EED0001, 4)
    first, count = partition_samples(16, 4, world, rank)
    acc, _ = sc.render(p, first, count)
    t = torch.from_numpy(acc)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put(t.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_split_matches_single_rank():
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from raytrace_amd import scene
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    summed = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    asm = scene.material_box(width=24, height=16, n_samples=16)
    objs = scene.finalize(asm, orc.OracleBackend())
    single, _ = orc.Scene(objs, asm.materials, asm.textures).render(orc.make_params(asm.config, 0x5EED0001, 4), 0, 16)
    # same samples, different float summation order: a few ulp of the pixel sum
    np.testing.assert_allclose(summed, single, rtol=1e-6, atol=1e-7)
    assert summed.sum() > 0
