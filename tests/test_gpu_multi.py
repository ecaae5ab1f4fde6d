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


def test_two_devices_reduce_through_rccl():
    """Two RCCL ranks on two GPUs of one process (ncclCommInitAll): skipped on a one-GPU box, which is all this build has
    had -- until it runs, the cross-rank leg of rt_reduce_to_root is exercised with one rank only."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    spp, batch = 32, 4
    asm = scene.material_box(width=72, height=48, n_samples=spp)
    cfg = asm.config
    npix = cfg.width * cfg.height
    ctxs = []
    try:
        for dev in (0, 1):
            c = capi.Context(dev)
            c.upload_scene(scene.finalize(asm), asm.materials, asm.textures)
            c.init_randstate(SEED, cfg.width, cfg.height)
            c.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
            ctxs.append(c)
        whole = torch.zeros(npix, 3, dtype=torch.float32, device="cuda:0")
        ctxs[0].raytrace(npix, 0, spp, batch, cfg.max_depth, cfg.background, cfg.russian_roulette, whole.data_ptr())
        ctxs[0].synchronize()
        with pytest.raises(capi.RtError, match="first_rank must be 0"):
            capi.Group(ctxs, first_rank=1, world=3, unique_id=None)
        group = capi.Group(ctxs, first_rank=0, world=2, unique_id=None)
        bufs = [torch.zeros(npix, 3, dtype=torch.float32, device=f"cuda:{d}") for d in (0, 1)]
        for k, c in enumerate(ctxs):
            first, count = capi.partition_samples(spp, batch, 2, k)
            c.raytrace(npix, first, count, batch, cfg.max_depth, cfg.background, cfg.russian_roulette, bufs[k].data_ptr())
        group.reduce_to_root([b.data_ptr() for b in bufs], npix * 3, None)
        for c in ctxs:
            c.synchronize()
        np.testing.assert_allclose(bufs[0].cpu().numpy(), whole.cpu().numpy(), rtol=4 * 1.2e-7, atol=1e-7)
        group.close()
    finally:
        for c in ctxs:
            c.close()


def test_bench_rehearsal_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N = 2 -- python -m torch.distributed.run, one process per rank, rendezvous
    on 127.0.0.1 -- rehearsed on this box's one GPU: RT_BENCH_BACKEND=gloo lets the two ranks share the device and sum
    through host memory (RCCL takes a device once per communicator, so the nccl branch needs two GPUs).  Checked: the
    JSON line says two ranks and which reduce ran, the whole-job counts are those of a one-rank run over the same global
    samples, and the reduced picture equals the one-rank picture within one 8-bit level (float additions in another order)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from raytrace_amd import scene_files
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    common = ["--steps", "2", "--warmup", "1", "--width", "160", "--height", "96", "--detail", "0.05", "--no-cpu-baseline", "--no-device-tree",
              "--wf-log2", "18", "--share-primary", "1"]
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_INST_SPP="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--spp-per-step", "32",
                          "--save-image", str(tmp_path / "two.png")] + common,
                         capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--spp-per-step", "64",
                          "--save-image", str(tmp_path / "one.png")] + common,
                         capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-3000:]
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l1["n_gpus"] == 1 and l2["scaling"] == "weak"
    assert "gloo" in l2["config"]["reduce"] and "rehearsal" in l2["config"]["reduce"]
    assert "spp-split x2" in l2["config"]["partition"]
    assert l2["unit"] == "Mrays/s" and l2["value"] > 0 and l2["roofline"]["frac"] > 0
    # the same global samples either way: 2 ranks x 32 spp per step = 1 rank x 64 spp per step
    assert l2["seconds_per_frame"] > 0 and abs(l2["mpaths_per_s"] * l2["ms_per_step"] - l1["mpaths_per_s"] * l1["ms_per_step"]) < 1e-3 * l1["mpaths_per_s"] * l1["ms_per_step"]
    a = scene_files.read_png_rgb(str(tmp_path / "two.png")).astype(np.int32)
    b = scene_files.read_png_rgb(str(tmp_path / "one.png")).astype(np.int32)
    assert a.shape == b.shape == (96, 160, 3) and b.max() > 0
    assert np.abs(a - b).max() <= 1
