"""Multi-GPU path on the one GPU available: several contexts, each tracing its partition of the samples with the HIP
kernels, combined by rt_reduce_to_root (the library's RCCL call) and compared with the one-context frame.

Contexts that share a device share an RCCL rank, so on a one-GPU box the communicator has one rank: the partition, the
per-context renders, the on-device sum, communicator creation (both ways: ncclCommInitAll and ncclCommInitRank with a
unique id) and the ncclReduce call itself all run; what needs more GPUs is only the traffic between ranks."""
import numpy as np
import pytest

from raytrace_amd import capi, scene
from tests.util import same_f32

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001


def _context(asm, mode):
    cfg = asm.config
    ctx = capi.Context(0)
    ctx.set_option(capi.OPT_MODE, mode)
    ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 16)
    ctx.upload_scene(scene.finalize(asm), asm.materials, asm.textures)
    ctx.init_randstate(SEED, cfg.width, cfg.height)
    ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    return ctx


@pytest.mark.parametrize("parts,with_id,mode", [(2, False, capi.MODE_WAVEFRONT), (3, True, capi.MODE_MEGAKERNEL), (1, True, capi.MODE_WAVEFRONT)])
def test_partitions_reduced_through_the_library_equal_the_whole_frame(parts, with_id, mode):
    import torch
    spp, batch = 40, 4                                    # 10 batches: 3 parts get 4 + 3 + 3
    asm = scene.material_box(width=72, height=48, n_samples=spp)
    cfg = asm.config
    npix = cfg.width * cfg.height
    ctxs = [_context(asm, mode) for _ in range(parts)]
    try:
        assert all(c.device() == 0 for c in ctxs)
        whole = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctxs[0].raytrace(npix, 0, spp, batch, cfg.max_depth, cfg.background, cfg.russian_roulette, whole.data_ptr())
        ctxs[0].synchronize()
        group = capi.Group(ctxs, first_rank=0, world=1, unique_id=capi.group_unique_id() if with_id else None)
        bufs = [torch.zeros(npix, 3, dtype=torch.float32, device="cuda") for _ in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        torch.cuda.synchronize()
        covered = 0
        for k, ctx in enumerate(ctxs):
            first, count = capi.partition_samples(spp, batch, parts, k)
            assert first == covered and first % batch == 0
            covered += count
            ctx.raytrace(npix, first, count, batch, cfg.max_depth, cfg.background, cfg.russian_roulette, bufs[k].data_ptr(), streams[k].cuda_stream)
        assert covered == spp
        group.reduce_to_root([b.data_ptr() for b in bufs], npix * 3, [s.cuda_stream for s in streams])
        for c in ctxs:
            c.synchronize()
        got = bufs[0].cpu().numpy()
        ref = whole.cpu().numpy()
        group.close()
    finally:
        for c in ctxs:
            c.close()
    assert ref.sum() > 0
    if parts == 1:
        assert same_f32(got, ref)                         # one rank, one context: the reduce must leave the buffer as it is
    else:
        # same samples, same batch totals, added in a different order: (b0+b1+b2+b3) + (b4+...) instead of one running sum.
        # Tolerance: 2 ulp of the pixel sum per partition (float32 eps = 1.2e-7).
        np.testing.assert_allclose(got, ref, rtol=parts * 2 * 1.2e-7, atol=1e-7)


def test_group_rejects_bad_worlds():
    asm = scene.material_box(width=8, height=8, n_samples=1)
    ctx = _context(asm, capi.MODE_MEGAKERNEL)
    try:
        with pytest.raises(capi.RtError, match="needs the id"):
            capi.Group([ctx], first_rank=0, world=2, unique_id=None)     # a world beyond this process without an id
        with pytest.raises(capi.RtError, match="more local devices"):
            capi.Group([ctx], first_rank=1, world=1, unique_id=capi.group_unique_id())
    finally:
        ctx.close()
