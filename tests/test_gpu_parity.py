"""GPU parity tests: the HIP path through the C ABI against the CPU oracle on the same inputs.

Bar: bit-exact for hit records (object, triangle, t, u, v), primary rays, per-pixel radiance sums and
device counters; float tolerance only where libm differs between host and device (pow in post_process).
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from raytrace_amd import capi, scene
from tests.util import bits, random_rays, same_f32, same_f32_or_both_nan

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _torch():
    import torch
    return torch


class Pair:
    """The same scene on both sides: product context on the GPU, oracle scene on the CPU."""

    def __init__(self, ctx, asm):
        self.ctx, self.asm, self.cfg = ctx, asm, asm.config
        self.objs = scene.finalize(asm)
        ctx.upload_scene(self.objs, asm.materials, asm.textures)
        self.oracle = orc.Scene(self.objs, asm.materials, asm.textures)
        ctx.init_randstate(SEED, self.cfg.width, self.cfg.height)
        ctx.setup_raytrace(self.cfg.width, self.cfg.height, self.cfg.camera_pos, self.cfg.camera_dir, self.cfg.camera_up, self.cfg.fov)

    def gpu_hits(self, rays):
        torch = _torch()
        n = len(rays)
        d_rays = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32)).cuda()
        d_obj = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        d_tri = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        d_t = torch.zeros(n, dtype=torch.float32, device="cuda")
        d_u = torch.zeros(n, dtype=torch.float32, device="cuda")
        d_v = torch.zeros(n, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        self.ctx.closest_hit(n, d_rays.data_ptr(), d_obj.data_ptr(), d_tri.data_ptr(), d_t.data_ptr(), d_u.data_ptr(), d_v.data_ptr())
        self.ctx.synchronize()
        return d_obj.cpu().numpy(), d_tri.cpu().numpy(), d_t.cpu().numpy(), d_u.cpu().numpy(), d_v.cpu().numpy()

    def gpu_primary_rays(self, batch):
        torch = _torch()
        d = torch.zeros(self.cfg.width * self.cfg.height, 6, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        self.ctx.primary_rays(batch, d.data_ptr())
        self.ctx.synchronize()
        return d.cpu().numpy()

    def gpu_render(self, first, n, batch, accum=None, counters=True):
        """Render through rt_raytrace in whatever mode the context is in (tests set it explicitly)."""
        torch = _torch()
        npix = self.cfg.width * self.cfg.height
        d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda") if accum is None else torch.from_numpy(accum).cuda()
        torch.cuda.synchronize()
        self.ctx.set_option(capi.OPT_COUNTERS, 1 if counters else 0)
        self.ctx.reset_counters()
        self.ctx.raytrace(npix, first, n, batch, self.cfg.max_depth, self.cfg.background, self.cfg.russian_roulette, d.data_ptr())
        cnt = self.ctx.counters()
        return d.cpu().numpy(), cnt

    def check_hits(self, rays):
        o, t, tt, u, v, _ = self.oracle.closest_hit(rays)
        go, gt, gtt, gu, gv = self.gpu_hits(rays)
        assert np.array_equal(go, o), f"{(go != o).sum()} object mismatches of {len(o)}"
        assert np.array_equal(gt, t), f"{(gt != t).sum()} triangle mismatches of {len(t)}"
        # a NaN ray's t, u, v are NaN on both sides; which NaN is the instruction scheduler's choice (tests/util.py)
        assert same_f32_or_both_nan(gtt, tt), f"{(bits(gtt) != bits(tt)).sum()} t mismatches"
        assert same_f32_or_both_nan(gu, u) and same_f32_or_both_nan(gv, v)
        return o


def test_primary_rays_bit_exact(gpu_ctx):
    p = Pair(gpu_ctx, scene.material_box(width=97, height=61, n_samples=4))     # odd sizes: partial tiles
    for batch in (0, 1, 7):
        assert same_f32(p.gpu_primary_rays(batch), orc.primary_rays(orc.make_params(p.cfg, SEED, 4), batch))


def test_closest_hit_bit_exact_small_scene(gpu_ctx):
    p = Pair(gpu_ctx, scene.material_box(width=96, height=64, n_samples=4))
    params = orc.make_params(p.cfg, SEED, 4)
    hit = p.check_hits(orc.primary_rays(params, 0))
    assert (hit >= 0).all()                                   # closed scene
    p.check_hits(random_rays(np.random.RandomState(5), 20000, -6.0, 6.0))


def test_closest_hit_edge_cases(gpu_ctx):
    p = Pair(gpu_ctx, scene.material_box(width=16, height=16, n_samples=1))
    rays = np.array([
        [0, 0, 5, 0, 0, -1],        # straight down an axis onto the floor (0 * inf in two slabs)
        [0, 0, 5, 0, 0, 1],         # straight up into the light
        [0.25, 0.25, 5, 0, 0, -1],  # onto the floor's shared diagonal: exact tie between the two triangles
        [0, 0, 5, 1, 0, 0], [0, 0, 5, 0, -1, 0],
        [0, 0, 5, 0, 0, 0],         # zero direction: every inverse is inf, every product NaN
        [0, 0, 5, np.nan, 0, 1],    # NaN direction
        [0, 0, 5, np.nan, np.nan, np.nan],   # all-NaN direction: every box entered, every triangle 'accepted', the last one tested wins
        [np.nan, 0, 5, np.nan, np.nan, np.nan],
        [1e30, 0, 0, -1, 0, 0],     # far outside
        [0, 0, 5, 1e-30, 1e-30, -1],
        [0, 0, 5, -0.0, 0.0, -1],   # signed zeros flip the slab order
    ], dtype=np.float32)
    p.check_hits(rays)


def test_closest_hit_golden_fixture(gpu_ctx):
    g = np.load(os.path.join(GOLDEN, "material_box_32x20.npz"))
    objs = [(g[f"tri{i}"].view(capi.TRIANGLE_DTYPE).reshape(-1), g[f"node{i}"].view(capi.NODE_DTYPE).reshape(-1)) for i in range(int(g["n_objects"]))]
    gpu_ctx.upload_scene(objs, g["materials"].view(capi.MATERIAL_DTYPE).reshape(-1), [g["texture0"]])
    cfg = scene.material_box(width=32, height=20, n_samples=8).config
    gpu_ctx.init_randstate(int(g["seed"]), cfg.width, cfg.height)
    gpu_ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    torch = _torch()
    rays = torch.from_numpy(g["rays"]).cuda()
    n = len(g["rays"])
    o = torch.zeros(n, dtype=torch.int32, device="cuda"); t = torch.zeros(n, dtype=torch.int32, device="cuda")
    tt = torch.zeros(n, dtype=torch.float32, device="cuda"); u = torch.zeros_like(tt); v = torch.zeros_like(tt)
    torch.cuda.synchronize()
    gpu_ctx.closest_hit(n, rays.data_ptr(), o.data_ptr(), t.data_ptr(), tt.data_ptr(), u.data_ptr(), v.data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(o.cpu().numpy(), g["hit_obj"]) and np.array_equal(t.cpu().numpy(), g["hit_tri"])
    assert same_f32(tt.cpu().numpy(), g["hit_t"]) and same_f32(u.cpu().numpy(), g["hit_u"]) and same_f32(v.cpu().numpy(), g["hit_v"])
    d = torch.zeros(cfg.width * cfg.height, 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.set_option(capi.OPT_COUNTERS, 1)
    gpu_ctx.reset_counters()
    gpu_ctx.raytrace(cfg.width * cfg.height, 0, 8, int(g["batch"]), cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
    cnt = gpu_ctx.counters()
    assert same_f32(d.cpu().numpy(), g["accum"])
    ref = [int(x) for x in g["counters"]]
    assert cnt[0] == ref[0] and cnt[2:5] == ref[2:5]


@pytest.mark.parametrize("name,kw", [
    ("luyu", dict(width=200, height=120, n_samples=4, statue_tris=60000)),
    ("formula1", dict(width=200, height=120, n_samples=4, detail=0.15, tex_size=128)),
    ("shuttle", dict(width=200, height=120, n_samples=4, detail=0.15, tex_size=128)),
    ("bike", dict(width=160, height=120, n_samples=4, detail=0.15, tex_size=64)),
    ("bunny", dict(width=120, height=120, n_samples=4, bunny_tris=40000)),
    ("house", dict(width=120, height=120, n_samples=4, detail=0.15, tex_size=64)),
    ("room", dict(width=150, height=100, n_samples=4, detail=0.5, tex_size=64)),
])
def test_closest_hit_bit_exact_benchmark_scenes(gpu_ctx, name, kw):
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    params = orc.make_params(p.cfg, SEED, 4)
    hit = p.check_hits(orc.primary_rays(params, 0))
    assert (hit >= 0).mean() > 0.3
    # degenerate rays in a many-object scene: zero direction (nothing can be hit) and all-NaN direction (the last
    # triangle upstream's walk tests wins) are answered without the walk and must still match the full walk
    o = np.array(p.cfg.camera_pos, dtype=np.float32)
    odd = np.array([[*o, 0, 0, 0], [*o, np.nan, np.nan, np.nan], [*o, -0.0, 0.0, -0.0]], dtype=np.float32)
    p.check_hits(odd)


@pytest.mark.parametrize("name,kw,spp,batch", [
    ("material_box", dict(width=64, height=40, n_samples=16), 16, 4),
    ("material_box", dict(width=33, height=17, n_samples=10), 10, 4),          # partial tiles, partial last batch
    ("luyu", dict(width=96, height=60, n_samples=8, statue_tris=20000), 8, 8),
    ("formula1", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=64), 8, 4),
    ("shuttle", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=64), 8, 2),
    ("bike", dict(width=80, height=60, n_samples=8, detail=0.05, tex_size=64), 8, 4),
    ("bunny", dict(width=64, height=64, n_samples=8, bunny_tris=8000), 8, 8),                 # rough and smooth metal
    ("cat", dict(width=64, height=64, n_samples=6, detail=0.05, tex_size=64), 6, 4),            # partial last batch
    ("room", dict(width=60, height=40, n_samples=8, detail=0.2, tex_size=64), 8, 4),            # ray depth 16, camera inside
])
def test_radiance_sums_and_counters_bit_exact(gpu_ctx, name, kw, spp, batch):
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), 0, spp)
    got, gcnt = p.gpu_render(0, spp, batch)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    # rays, triangle tests, shaded hits and texture fetches are properties of the visiting order and must match;
    # slab tests are fewer than upstream's because the object-level tree skips runs of objects
    assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2]
    assert 0 < gcnt[capi.CNT_BOX_TESTS] <= 2 * rcnt[1]
    assert gcnt[capi.CNT_SHADED] == rcnt[3] and gcnt[capi.CNT_TEX_FETCHES] == rcnt[4]
    assert gcnt[capi.CNT_PATHS] == p.cfg.width * p.cfg.height * spp


@pytest.mark.parametrize("name,kw,spp,batch,cap", [
    ("material_box", dict(width=64, height=40, n_samples=16), 16, 4, 11),      # pool of 2048 paths: many refills and compactions
    ("material_box", dict(width=33, height=17, n_samples=10), 10, 4, 10),      # partial tiles (lanes outside the image), pool of 1024
    ("luyu", dict(width=96, height=60, n_samples=8, statue_tris=20000), 8, 8, 14),
    ("formula1", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=64), 8, 4, 22),   # pool larger than the job
    ("shuttle", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=64), 8, 2, 12),
    ("room", dict(width=60, height=40, n_samples=8, detail=0.2, tex_size=64), 8, 4, 13),        # 16 bounces through a small pool
    ("house", dict(width=64, height=64, n_samples=8, detail=0.05, tex_size=64), 8, 8, 14),
])
def test_wavefront_variant_bit_exact(gpu_ctx, name, kw, spp, batch, cap):
    """The wavefront schedule (state in HBM, stage kernels, ballot compaction) traces the same paths: same image, same counts."""
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), 0, spp)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, cap)
    try:
        got, gcnt = p.gpu_render(0, spp, batch)
        # twice in a row on the same context: the control block and pools are reused
        again, _ = p.gpu_render(0, spp, batch)
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert same_f32(again, ref)
    assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2]
    assert gcnt[capi.CNT_SHADED] == rcnt[3] and gcnt[capi.CNT_TEX_FETCHES] == rcnt[4]
    assert gcnt[capi.CNT_PATHS] == p.cfg.width * p.cfg.height * spp


@pytest.mark.parametrize("first,spp,batch,cap", [
    (0, 33, 16, 12),        # groups of 16 samples, then one sample on its own; pool of 4096: refills cut through groups
    (0, 21, 6, 11),         # batch 6: groups of 2; an odd last group
    (12, 29, 12, 13),       # a later slice; batch 12: groups of 4; the slice ends inside a batch and inside a group
    (0, 15, 5, 12),         # batch 5: no power of two divides it, the run-after-run order
    (32, 64, 32, 18),       # batch 32: two groups of 16 share one primary ray
    (0, 48, 16, 10),        # pool of 1024 entries, smaller than one tile's group (4096 places)
])
def test_samples_of_a_batch_side_by_side_bit_exact(gpu_ctx, first, spp, batch, cap):
    """With one walk per sample (RT_OPT_SHARE_PRIMARY = 0) a tile's entries are laid out so that the samples of a semantic
    batch, which start with one and the same primary ray, sit in neighbouring lanes of the closest-hit kernel (groups of up
    to 16, a power of two that divides the batch).  Only the order of the entries changes: the image and every count are the
    oracle's, and those of the sample-after-sample order (RT_OPT_GROUP_SAMPLES = 0)."""
    p = Pair(gpu_ctx, scene.material_box(width=70, height=45, n_samples=first + spp))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), first, spp)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, cap)
    try:
        got, gcnt = p.gpu_render(first, spp, batch)
        gpu_ctx.set_option(capi.OPT_GROUP_SAMPLES, 0)
        plain, pcnt = p.gpu_render(first, spp, batch)
        gpu_ctx.set_option(capi.OPT_GROUP_SAMPLES, 1)
        gpu_ctx.set_option(capi.OPT_PASS, 9)                        # passes that cut through groups and batches
        passes, _ = p.gpu_render(first, spp, batch)
    finally:
        gpu_ctx.set_option(capi.OPT_GROUP_SAMPLES, 1)
        gpu_ctx.set_option(capi.OPT_PASS, 1024)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert same_f32(plain, ref) and same_f32(passes, ref)
    for c in (gcnt, pcnt):
        assert c[capi.CNT_RAYS] == rcnt[0] and c[capi.CNT_WALKS] == rcnt[0] and c[capi.CNT_TRI_TESTS] == rcnt[2]
        assert c[capi.CNT_SHADED] == rcnt[3] and c[capi.CNT_PATHS] == p.cfg.width * p.cfg.height * spp
    assert gcnt[capi.CNT_BOX_TESTS] == pcnt[capi.CNT_BOX_TESTS] and gcnt[capi.CNT_NODE_VISITS] == pcnt[capi.CNT_NODE_VISITS]


def test_started_call_returns_at_once_and_gives_the_same_buffer(gpu_ctx):
    """rt_raytrace_start hands the call to a thread of the library: it returns long before the frame is done (the blocking
    call cannot, in wavefront mode: the host drives the stages), a second start on the busy context is refused, and after
    rt_raytrace_wait the buffer is the blocking call's bit for bit.  A call that fails reports through the wait."""
    import time
    torch = _torch()
    p = Pair(gpu_ctx, scene.formula1(width=640, height=400, n_samples=128, detail=0.2, tex_size=64))
    npix, cfg = p.cfg.width * p.cfg.height, p.cfg
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    try:
        t0 = time.perf_counter()
        want, _ = p.gpu_render(0, 128, 16, counters=False)
        blocking = time.perf_counter() - t0
        d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu_ctx.raytrace_start(npix, 0, 128, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
        started = time.perf_counter() - t0
        with pytest.raises(capi.RtError, match="still driving"):
            gpu_ctx.raytrace_start(npix, 0, 128, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
        gpu_ctx.raytrace_wait()
        gpu_ctx.synchronize()
        got = d.cpu().numpy()
        assert same_f32(got, want)
        assert started < 0.25 * blocking, (started, blocking)
        gpu_ctx.raytrace_wait()                                     # nothing outstanding: fine
        # first_sample not a multiple of the batch: the started call's failure comes out of the wait, with its text
        gpu_ctx.raytrace_start(npix, 3, 16, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
        with pytest.raises(capi.RtError, match="multiple"):
            gpu_ctx.raytrace_wait()
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)


def test_wavefront_depth_zero_and_glass(gpu_ctx):
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    try:
        asm = scene.material_box(width=32, height=32, n_samples=4)
        asm.config.max_depth = 0
        p = Pair(gpu_ctx, asm)
        got, cnt = p.gpu_render(0, 4, 4)
        assert not got.any() and cnt[capi.CNT_RAYS] == 0 and cnt[capi.CNT_PATHS] == 32 * 32 * 4
        p = Pair(gpu_ctx, scene.formula1(width=96, height=60, n_samples=8, detail=0.05, glass=True))
        ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, 4), 0, 8)
        got, gcnt = p.gpu_render(0, 8, 4)
        assert same_f32(got, ref) and gcnt[capi.CNT_RAYS] == rcnt[0]
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)


def test_wavefront_scheduling_knobs_do_not_change_the_image(gpu_ctx):
    """Octant-ordered compaction, the LDS node cache, node burst, run length and workgroup size only reschedule work."""
    p = Pair(gpu_ctx, scene.formula1(width=96, height=60, n_samples=8, detail=0.05, tex_size=64))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, 4), 0, 8)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 13)
    try:
        for sort, hot, burst, run, vote, refill, block in ((0, 0, 1, 64, 0, 8, 64), (1, 0, 4, 1024, 16, 8, 256), (0, 64, 2, 100, 4, 1, 768),
                                                          (1, 512, 7, 4096, 64, 64, 1024), (1, 3, 4, 1024, 1024, 20, 512), (1, 140, 4, 256, 32, 16, 192)):
            gpu_ctx.set_option(capi.OPT_WF_VOTE, vote)
            gpu_ctx.set_option(capi.OPT_WF_REFILL, refill)
            gpu_ctx.set_option(capi.OPT_WF_SORT, sort)
            gpu_ctx.set_option(capi.OPT_HOT_NODES, hot)
            gpu_ctx.set_option(capi.OPT_WF_BURST, burst)
            gpu_ctx.set_option(capi.OPT_WF_RUN, run)
            gpu_ctx.set_option(capi.OPT_WF_BLOCK, block)
            gpu_ctx.set_option(capi.OPT_WF_LANES, 2 if block in (256, 1024) else 1)     # two pipelines on two streams: samples split between them
            got, gcnt = p.gpu_render(0, 8, 4)
            assert same_f32(got, ref), (sort, hot, burst, run, vote, refill, block)
            assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2], (sort, hot, burst, run, vote, refill, block)
            # the same schedule with the first walk of a batch shared: same image, same rays, fewer walks
            gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)
            gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 1)
            got, gcnt = p.gpu_render(0, 8, 4)
            gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
            gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
            assert same_f32(got, ref), ("shared", sort, hot, burst, run, vote, refill, block)
            assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_WALKS] < rcnt[0] and gcnt[capi.CNT_SHADED] == rcnt[3]
    finally:
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_WF_SORT, 1)
        gpu_ctx.set_option(capi.OPT_WF_VOTE, 32)
        gpu_ctx.set_option(capi.OPT_WF_REFILL, 24)
        gpu_ctx.set_option(capi.OPT_HOT_NODES, 100)
        gpu_ctx.set_option(capi.OPT_WF_BLOCK, 768)
        gpu_ctx.set_option(capi.OPT_WF_LANES, 1)
        gpu_ctx.set_option(capi.OPT_WF_BURST, 4)
        gpu_ctx.set_option(capi.OPT_WF_RUN, 256)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)


def test_glass_variant_bit_exact(gpu_ctx):
    # every car triangle refractive: long specular chains, the divergence stress case
    p = Pair(gpu_ctx, scene.formula1(width=96, height=60, n_samples=8, detail=0.05, glass=True))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, 4), 0, 8)
    got, gcnt = p.gpu_render(0, 8, 4)
    assert same_f32(got, ref)
    assert gcnt[capi.CNT_RAYS] == rcnt[0]


def test_launch_split_and_options_do_not_change_the_image(gpu_ctx):
    p = Pair(gpu_ctx, scene.material_box(width=64, height=40, n_samples=16))
    whole, _ = p.gpu_render(0, 16, 4)
    part, _ = p.gpu_render(0, 8, 4)
    part, _ = p.gpu_render(8, 8, 4, accum=part)
    assert same_f32(whole, part)
    gpu_ctx.set_option(capi.OPT_XCD_REMAP, 0)
    linear, _ = p.gpu_render(0, 16, 4, counters=False)
    gpu_ctx.set_option(capi.OPT_XCD_REMAP, 1)
    assert same_f32(whole, linear)
    # scheduling knobs: samples per lane and samples per buffered pass, including passes that cut batches
    for chunk, npass in ((1, 64), (3, 5), (16, 1), (5, 7), (64, 3)):
        gpu_ctx.set_option(capi.OPT_CHUNK, chunk)
        gpu_ctx.set_option(capi.OPT_PASS, npass)
        again, _ = p.gpu_render(0, 16, 4, counters=False)
        assert same_f32(whole, again), (chunk, npass)
    gpu_ctx.set_option(capi.OPT_CHUNK, 16)
    gpu_ctx.set_option(capi.OPT_PASS, 64)


def test_max_depth_zero_and_zero_samples(gpu_ctx):
    asm = scene.material_box(width=32, height=32, n_samples=4)
    asm.config.max_depth = 0
    p = Pair(gpu_ctx, asm)
    got, cnt = p.gpu_render(0, 4, 4)
    assert not got.any() and cnt[capi.CNT_RAYS] == 0 and cnt[capi.CNT_PATHS] == 32 * 32 * 4
    got, cnt = p.gpu_render(0, 0, 4)
    assert not got.any() and cnt[capi.CNT_PATHS] == 0


def test_post_process_within_tolerance(gpu_ctx):
    torch = _torch()
    rng = np.random.RandomState(2)
    buf = (rng.uniform(-0.5, 40.0, size=(5000, 3))).astype(np.float32)
    d = torch.from_numpy(buf).cuda()
    torch.cuda.synchronize()
    gpu_ctx.post_process(5000, 16, 1.8, d.data_ptr())
    gpu_ctx.synchronize()
    # powf differs between glibc and the device math library by a few ulp; everything before it is exact
    np.testing.assert_allclose(d.cpu().numpy(), orc.post_process(buf, 16, 1.8), rtol=4e-6, atol=1e-7)


def test_errors_are_reported(gpu_ctx):
    lib = capi.load()
    torch = _torch()
    fresh = capi.Context(0)
    d = torch.zeros(16, 3, device="cuda")
    with pytest.raises(capi.RtError, match="no scene"):
        fresh.raytrace(16, 0, 1, 1, 8, (0, 0, 0), 0.9, d.data_ptr())
    p = Pair(fresh, scene.material_box(width=4, height=4, n_samples=1))
    with pytest.raises(capi.RtError, match="n_pixel"):
        fresh.raytrace(15, 0, 1, 1, 8, (0, 0, 0), 0.9, d.data_ptr())
    with pytest.raises(capi.RtError, match="multiple"):
        fresh.raytrace(16, 3, 1, 4, 8, (0, 0, 0), 0.9, d.data_ptr())
    bad = p.asm.materials.copy()
    bad["texture"][0] = 5
    with pytest.raises(capi.RtError, match="texture index"):
        fresh.upload_scene(p.objs, bad, p.asm.textures)
    # a chain-shaped tree (every inner node = the chain so far + one more leaf) handed in through the ABI: refused before
    # the packer, which recurses once per level, walks it
    n = 700
    tris = np.zeros(n, dtype=capi.TRIANGLE_DTYPE)
    tris["v"]["position"][:, 1, 0] = 1.0
    tris["v"]["position"][:, 2, 1] = 1.0
    tris["v"]["position"][:, :, 2] = np.arange(n, dtype=np.float32)[:, None]
    nodes = np.zeros(2 * n - 1, dtype=capi.NODE_DTYPE)
    nodes["box_min"], nodes["box_max"] = (0, 0, 0), (1, 1, n)
    nodes[0]["is_leaf"], nodes[0]["left"], nodes[0]["right"] = 1, 0, -1
    for k in range(1, n):
        leaf, inner = 2 * k - 1, 2 * k
        nodes[leaf]["is_leaf"], nodes[leaf]["left"], nodes[leaf]["right"] = 1, k, -1
        nodes[inner]["is_leaf"], nodes[inner]["left"], nodes[inner]["right"] = 0, inner - 2 if k > 1 else 0, leaf
    with pytest.raises(capi.RtError, match="deeper than"):
        fresh.upload_scene([(tris, nodes)], p.asm.materials, p.asm.textures)
    fresh.close()
    assert lib.rt_last_error() != b""


def test_full_size_primary_hits_agree_with_oracle_sample(gpu_ctx):
    """Full benchmark geometry (513k triangles, 35 objects) at the full 2000x1200 frame on the GPU; the oracle
    checks a strided 1-in-97 subset of the pixels bit for bit."""
    p = Pair(gpu_ctx, scene.formula1(width=2000, height=1200, n_samples=1))
    rays = p.gpu_primary_rays(0)
    go, gt, gtt, gu, gv = p.gpu_hits(rays)
    sel = np.arange(0, len(rays), 97)
    o, t, tt, u, v, _ = p.oracle.closest_hit(rays[sel])
    assert np.array_equal(go[sel], o) and np.array_equal(gt[sel], t) and same_f32(gtt[sel], tt)
    assert same_f32(gu[sel], u) and same_f32(gv[sel], v)
    assert 0.3 < (go >= 0).mean() <= 1.0


def test_incoherent_secondary_rays_at_full_detail(gpu_ctx):
    """Rays the way bounces produce them: origins on the surfaces the camera sees, directions uniform on the sphere, on
    the full benchmark geometry.  Every one of 400k rays is compared with the oracle, not a sample."""
    p = Pair(gpu_ctx, scene.formula1(width=2000, height=1200, n_samples=1))
    rays = p.gpu_primary_rays(0)
    go, gt, gtt, gu, gv = p.gpu_hits(rays)
    rng = np.random.RandomState(17)
    pick = rng.choice(np.nonzero(go >= 0)[0], size=400_000, replace=False)
    origin = rays[pick, :3] + rays[pick, 3:] * gtt[pick, None]
    d = rng.standard_normal((len(pick), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.concatenate([origin, d], axis=1).astype(np.float32)
    hit = p.check_hits(sec)
    assert 0.5 < (hit >= 0).mean() < 1.0                       # most leave a surface into the scene, some escape


@pytest.mark.parametrize("name,kw,spp,batch", [
    ("formula1", dict(width=400, height=240, n_samples=16), 16, 16),                            # full detail: 513k triangles, 1024^2 textures
    ("formula1", dict(width=2000, height=1200, n_samples=2), 2, 2),                             # the benchmark's frame, every pixel
    ("formula1", dict(width=500, height=300, n_samples=8, glass=True), 8, 8),                   # all-refractive variant at full detail
    ("shuttle", dict(width=500, height=300, n_samples=8), 8, 4),
    ("stress", dict(width=320, height=180, n_samples=8, grid=(3, 3), tris_per_instance=50_000), 8, 4),
])
def test_full_detail_wavefront_render_bit_exact(gpu_ctx, name, kw, spp, batch):
    """The benchmark's kernel design and default options on benchmark-sized geometry, whole buffer against the oracle."""
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), 0, spp)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 19)
    try:
        got, gcnt = p.gpu_render(0, spp, batch)
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2] and gcnt[capi.CNT_SHADED] == rcnt[3]


def test_rmse_to_a_converged_image(gpu_ctx):
    """Statistical agreement, seed-independent: GPU frames with one seed against an oracle frame with another seed and
    many more samples.  The error must fall like 1/sqrt(spp) (tolerance: a 16x sample count must cut the RMSE by at
    least 2.8x; 4x is the expectation) and stay below 0.03 of the picture's mean at 1024 spp."""
    asm = scene.material_box(width=48, height=32, n_samples=2048)
    asm.config.gamma = 1.0
    p = Pair(gpu_ctx, asm)
    params = orc.make_params(p.cfg, 0xC0FFEE, 16)
    ref, _ = p.oracle.render(params, 0, 2048)
    ref = orc.post_process(ref, 2048, 1.0)                     # mean radiance per pixel, clamped to [0, 1]
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    try:
        rmse = {}
        for spp in (64, 1024):
            got, _ = p.gpu_render(0, spp, 16, counters=False)  # seed of the Pair, different from the oracle's
            got = orc.post_process(got, spp, 1.0)
            rmse[spp] = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
    assert rmse[1024] < rmse[64] / 2.8, rmse
    assert rmse[1024] < 0.03 * float(ref.mean()) + 0.01, (rmse, float(ref.mean()))


def test_full_size_sample_split_sums_to_the_whole_frame(gpu_ctx):
    """Size-independent property at the benchmark's full frame: tracing samples [0,16) and [16,32) into one buffer, in
    either kernel design, gives exactly the buffer of one 32-sample call (what the multi-GPU split relies on)."""
    p = Pair(gpu_ctx, scene.formula1(width=2000, height=1200, n_samples=32, detail=0.3, tex_size=256))
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 24)
    try:
        whole, cnt = p.gpu_render(0, 32, 16, counters=False)
        part, _ = p.gpu_render(0, 16, 16, counters=False)
        part, _ = p.gpu_render(16, 16, 16, accum=part, counters=False)
    finally:
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(whole, part)
    assert cnt[capi.CNT_PATHS] == 2000 * 1200 * 32 and np.isfinite(whole).all() and whole.mean() > 0


def test_full_size_shared_walk_equals_every_ray_walked(gpu_ctx):
    """Size-independent property at the benchmark's full frame and default pool size: with the first walk of a semantic
    batch shared (the library's default) the buffer and every per-sample count are those of one walk per sample; only
    the walks run are fewer, by exactly the primaries saved."""
    p = Pair(gpu_ctx, scene.formula1(width=2000, height=1200, n_samples=48, detail=0.3, tex_size=256))
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 27)
    try:
        every, c0 = p.gpu_render(0, 48, 16, counters=False)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)
        shared, c1 = p.gpu_render(0, 48, 16, counters=False)
        # a slice that ends inside a batch, added to an earlier slice: 32 + 16 samples in two calls
        part, _ = p.gpu_render(0, 32, 16, counters=False)
        part, _ = p.gpu_render(32, 16, 16, accum=part, counters=False)
    finally:
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(every, shared) and same_f32(every, part)
    npix = 2000 * 1200
    assert c0[capi.CNT_RAYS] == c1[capi.CNT_RAYS] and c0[capi.CNT_PATHS] == c1[capi.CNT_PATHS] == npix * 48
    assert c0[capi.CNT_WALKS] == c0[capi.CNT_RAYS]
    assert c1[capi.CNT_WALKS] == c1[capi.CNT_RAYS] - npix * 48 + npix * 3


def test_stress_scene_at_its_real_size(gpu_ctx):
    """The 30 M-triangle scene as benchmarked (8x8 instances of 470k triangles, 66 objects, 4.4 GB packed, deepest stacks,
    packed indices near 2^25) at the full 3840x2160 frame: every primary ray and 250k incoherent secondary rays are traced
    on the GPU; the oracle checks a strided sample of 60k primaries and 40k secondaries bit for bit (object, triangle,
    t, u, v).  Then a whole reduced frame (480x270, 4 spp) with the default wavefront options against the oracle's render."""
    asm = scene.stress(width=3840, height=2160, n_samples=4)
    p = Pair(gpu_ctx, asm)
    stats = gpu_ctx.scene_stats()
    assert stats["triangles"] > 30_000_000 and stats["objects"] == 66
    rays = p.gpu_primary_rays(0)
    assert len(rays) == 3840 * 2160
    go, gt, gtt, gu, gv = p.gpu_hits(rays)
    sel = np.arange(0, len(rays), 137)[:60_000]
    o, t, tt, u, v, _ = p.oracle.closest_hit(rays[sel])
    assert np.array_equal(go[sel], o), f"{(go[sel] != o).sum()} object mismatches"
    assert np.array_equal(gt[sel], t), f"{(gt[sel] != t).sum()} triangle mismatches"
    assert same_f32(gtt[sel], tt) and same_f32(gu[sel], u) and same_f32(gv[sel], v)
    assert (go >= 0).mean() > 0.5 and len(np.unique(go[sel])) > 30          # most of the instances are seen
    # bounce-like rays: origins on the surfaces the camera sees, directions uniform on the sphere
    rng = np.random.RandomState(23)
    pick = rng.choice(np.nonzero(go >= 0)[0], size=250_000, replace=False)
    origin = rays[pick, :3] + rays[pick, 3:] * gtt[pick, None]
    d = rng.standard_normal((len(pick), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.concatenate([origin, d], axis=1).astype(np.float32)
    so, st, stt, su, sv = p.gpu_hits(sec)
    sel = np.arange(0, len(sec), 6)[:40_000]
    o, t, tt, u, v, _ = p.oracle.closest_hit(sec[sel])
    assert np.array_equal(so[sel], o) and np.array_equal(st[sel], t)
    assert same_f32(stt[sel], tt) and same_f32(su[sel], u) and same_f32(sv[sel], v)
    # the whole path loop on this geometry, reduced frame
    cfg = p.cfg
    cfg.width, cfg.height = 480, 270
    gpu_ctx.init_randstate(SEED, cfg.width, cfg.height)
    gpu_ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    ref, rcnt = p.oracle.render(orc.make_params(cfg, SEED, 4), 0, 4)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 27)
    try:
        got, gcnt = p.gpu_render(0, 4, 4)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)                         # the library's default: one walk per pixel and batch
        shared, scnt = p.gpu_render(0, 4, 4)
    finally:
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2] and gcnt[capi.CNT_SHADED] == rcnt[3]
    assert gcnt[capi.CNT_MAX_STACK] > 11                                      # the private spill levels were in use
    assert same_f32(shared, ref) and scnt[capi.CNT_RAYS] == rcnt[0] and scnt[capi.CNT_SHADED] == rcnt[3]
    assert scnt[capi.CNT_WALKS] == rcnt[0] - 3 * cfg.width * cfg.height       # four samples, one batch: three camera walks saved per pixel


def test_hot_node_tuning_changes_numbers_not_results(gpu_ctx):
    """The calibration pass (one sample per pixel counted into a visit histogram, the 4096 records nearest the root
    renumbered by falling count -- RT_OPT_HOT_TUNE 1 -- and a copy of them numbered for every band of the picture -- 2)
    only moves records: same image, same counts, and more node visits served from LDS."""
    asm = scene.formula1(width=200, height=120, n_samples=64, detail=0.15, tex_size=64)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 18)
    try:
        out = {}
        for tune in (0, 1, 2):
            gpu_ctx.set_option(capi.OPT_HOT_TUNE, tune)
            p = Pair(gpu_ctx, asm)                                  # uploads again: the static numbering is back
            out[tune] = p.gpu_render(0, 64, 16)
            if tune:
                again = p.gpu_render(0, 64, 16)                     # second frame, same camera: no new calibration, same picture
                assert same_f32(again[0], out[tune][0])
                hits_tuned = p.check_hits(orc.primary_rays(orc.make_params(p.cfg, SEED, 16), 0))     # rt_closest_hit on the renumbered records
                assert (hits_tuned >= 0).any()
    finally:
        gpu_ctx.set_option(capi.OPT_HOT_TUNE, 2)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    (a, ca), (b, cb), (c, cc) = out[0], out[1], out[2]
    assert same_f32(a, b) and same_f32(a, c)
    for k in (capi.CNT_RAYS, capi.CNT_BOX_TESTS, capi.CNT_TRI_TESTS, capi.CNT_SHADED, capi.CNT_TEX_FETCHES, capi.CNT_PATHS, capi.CNT_NODE_VISITS):
        assert ca[k] == cb[k] == cc[k], k
    # one numbering for the view beats the static one; a frame this small has too few calibration paths for bands of its
    # own (2048 per band are asked for), so the third setting falls back to the second
    assert cc[capi.CNT_HOT_VISITS] == cb[capi.CNT_HOT_VISITS] > ca[capi.CNT_HOT_VISITS]


def test_band_copies_follow_the_camera_and_the_passes(gpu_ctx):
    """The per-band copies of the first node records (RT_OPT_HOT_TUNE 2) belong to one scene and one camera.  A tall frame
    (nine bands of the 100x200 calibration frame) in a pool that holds a fortieth of a pass, in passes of 64 samples: rendered, then the camera moves (the
    copies must be dropped and made again, or the launches would read records numbered for another view -- harmless to the
    bits, but the test also checks that the calibration ran again), then the setting goes 2 -> 1 -> 2 on the same scene.
    Every render against the oracle; LDS-served visits with the copies above those without."""
    import copy
    asm = scene.formula1(width=400, height=800, n_samples=128, detail=0.1, tex_size=32)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 19)
    gpu_ctx.set_option(capi.OPT_PASS, 64)
    try:
        p = Pair(gpu_ctx, asm)
        params = orc.make_params(p.cfg, SEED, 16)
        ref, rcnt = p.oracle.render(params, 0, 128)
        got, cnt2 = p.gpu_render(0, 128, 16)
        assert same_f32(got, ref) and cnt2[capi.CNT_RAYS] == rcnt[0] and cnt2[capi.CNT_TRI_TESTS] == rcnt[2]
        # another view of the same scene
        cfg2 = copy.copy(p.cfg)
        cfg2.camera_pos = (p.cfg.camera_pos[0] * 0.8, p.cfg.camera_pos[1] + 40.0, p.cfg.camera_pos[2] * 1.1)
        gpu_ctx.setup_raytrace(cfg2.width, cfg2.height, cfg2.camera_pos, cfg2.camera_dir, cfg2.camera_up, cfg2.fov)
        ref2, rcnt2 = p.oracle.render(orc.make_params(cfg2, SEED, 16), 0, 128)
        got2, cnt = p.gpu_render(0, 128, 16)
        assert same_f32(got2, ref2) and cnt[capi.CNT_RAYS] == rcnt2[0] and cnt[capi.CNT_TRI_TESTS] == rcnt2[2]
        hot = {}
        for tune in (1, 2):
            gpu_ctx.set_option(capi.OPT_HOT_TUNE, tune)
            again, c = p.gpu_render(0, 128, 16)
            assert same_f32(again, ref2), tune
            assert c[capi.CNT_NODE_VISITS] == cnt[capi.CNT_NODE_VISITS] and c[capi.CNT_BOX_TESTS] == cnt[capi.CNT_BOX_TESTS]
            hot[tune] = c[capi.CNT_HOT_VISITS]
        assert hot[2] > hot[1] and abs(hot[2] - cnt[capi.CNT_HOT_VISITS]) <= 0.01 * hot[2]       # made again, and the same again (up to ties)
    finally:
        gpu_ctx.set_option(capi.OPT_HOT_TUNE, 2)
        gpu_ctx.set_option(capi.OPT_PASS, 1024)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)


def test_library_defaults_all_at_once_against_the_oracle():
    """A context with NOTHING set -- wavefront pipeline, first walk of a batch shared, coinciding continued rays shared,
    LDS node records chosen by the calibration pass (64 spp is where it starts), default pool size -- on the
    benchmark scene at full detail, 400x240, 64 spp in batches of 16: the whole buffer and every per-sample count against
    the oracle's render.  Every knob has a test of its own; this is their conjunction, as a user of the library gets it."""
    torch = _torch()
    asm = scene.formula1(width=400, height=240, n_samples=64)
    cfg = asm.config
    objs = scene.finalize(asm)
    ref, rcnt = orc.Scene(objs, asm.materials, asm.textures).render(orc.make_params(cfg, SEED, 16), 0, 64)
    ctx = capi.Context(0)
    try:
        ctx.upload_scene(objs, asm.materials, asm.textures)
        ctx.init_randstate(SEED, cfg.width, cfg.height)
        ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
        npix = cfg.width * cfg.height
        d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.raytrace(npix, 0, 64, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
        cnt = ctx.counters()                                        # the counters every build keeps: rays, paths, walks
        got = d.cpu().numpy()
        # the same call again with the counting build (the only option touched), into a fresh buffer
        ctx.set_option(capi.OPT_COUNTERS, 1)
        ctx.reset_counters()
        d.zero_()
        torch.cuda.synchronize()
        ctx.raytrace(npix, 0, 64, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
        ccnt = ctx.counters()
        counted = d.cpu().numpy()
    finally:
        ctx.close()
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert same_f32(counted, ref)
    assert cnt[capi.CNT_RAYS] == rcnt[0] and cnt[capi.CNT_PATHS] == npix * 64
    assert ccnt[capi.CNT_RAYS] == rcnt[0] and ccnt[capi.CNT_SHADED] == rcnt[3] and ccnt[capi.CNT_TEX_FETCHES] == rcnt[4]
    # four batches of 16: at least the fifteen repeated first walks of every batch are saved, coinciding continued rays on top
    assert cnt[capi.CNT_WALKS] <= rcnt[0] - npix * 64 + npix * 4 and abs(cnt[capi.CNT_WALKS] - ccnt[capi.CNT_WALKS]) <= 1e-3 * cnt[capi.CNT_WALKS]
    assert ccnt[capi.CNT_HOT_VISITS] > 0


@pytest.mark.parametrize("name,kw,spp,batch,ring_mb,cap,shared", [
    ("material_box", dict(width=200, height=120, n_samples=32), 32, 8, 1, 14, 0),       # 104 tiles, a ring of 10: the planner holds new paths back
    ("material_box", dict(width=200, height=120, n_samples=32), 32, 8, 1, 14, 1),       # the same with the first walk shared
    ("material_box", dict(width=200, height=120, n_samples=20), 20, 8, 0, 18, 1),       # a ring of ONE tile; the pass ends inside a batch
    ("room", dict(width=120, height=80, n_samples=16, detail=0.2, tex_size=64), 16, 4, 1, 14, 1),   # depth 16: paths live twice as long
    ("formula1", dict(width=400, height=240, n_samples=32, detail=0.1, tex_size=64), 32, 16, 2, 16, 1),
])
def test_slot_ring_shorter_than_the_frame_bit_exact(gpu_ctx, name, kw, spp, batch, ring_mb, cap, shared):
    """The wavefront pass sums a 16x16 tile's slots as soon as the tile's last path has ended and hands the tile's place in
    the slot ring to a later tile (RT_OPT_SLOT_RING_MB).  With a ring far shorter than the frame the planner has to hold
    new paths back until places come free: the image and the counts must be those of the oracle all the same, in one call
    and in two calls that split the samples."""
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), 0, spp)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, cap)
    gpu_ctx.set_option(capi.OPT_SLOT_RING_MB, ring_mb)
    gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, shared)
    gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, shared)
    try:
        got, gcnt = p.gpu_render(0, spp, batch)
        half = (spp // 2) // batch * batch
        part, _ = p.gpu_render(0, half, batch)
        part, _ = p.gpu_render(half, spp - half, batch, accum=part)
        gpu_ctx.set_option(capi.OPT_PASS, 8)                         # several passes per call, each with its own ring
        passes, _ = p.gpu_render(0, spp, batch)
    finally:
        gpu_ctx.set_option(capi.OPT_PASS, 1024)
        gpu_ctx.set_option(capi.OPT_SLOT_RING_MB, 10240)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert same_f32(part, ref) and same_f32(passes, ref)
    assert gcnt[capi.CNT_RAYS] == rcnt[0] and gcnt[capi.CNT_SHADED] == rcnt[3] and gcnt[capi.CNT_PATHS] == p.cfg.width * p.cfg.height * spp


def _runs(first, n, batch):
    """Semantic batches the samples [first, first + n) touch: one shared walk each."""
    return (first + n - 1) // batch - first // batch + 1 if n > 0 else 0


@pytest.mark.parametrize("name,kw,first,n,batch,cap,lanes,shared", [
    ("material_box", dict(width=64, height=40, n_samples=16), 0, 16, 4, 11, 1, True),     # pool of 2048 entries: many refills
    ("material_box", dict(width=33, height=17, n_samples=10), 0, 10, 4, 11, 1, True),     # partial tiles, partial last batch
    ("material_box", dict(width=33, height=17, n_samples=16), 4, 7, 4, 12, 1, True),      # a later slice that ends inside a batch
    ("material_box", dict(width=64, height=40, n_samples=64), 0, 64, 64, 11, 1, False),   # a run longer than an eighth of a region: not shared
    ("material_box", dict(width=64, height=40, n_samples=8), 0, 8, 64, 18, 1, True),      # batch longer than the call: one run of 8 per pixel
    ("luyu", dict(width=96, height=60, n_samples=8, statue_tris=20000), 0, 8, 8, 14, 1, True),
    ("formula1", dict(width=96, height=60, n_samples=16, detail=0.05, tex_size=64), 0, 16, 16, 22, 1, True),   # pool larger than the job
    ("formula1", dict(width=96, height=60, n_samples=16, detail=0.05, tex_size=64), 0, 16, 4, 14, 2, True),    # two pipelines split the samples
    ("room", dict(width=60, height=40, n_samples=8, detail=0.2, tex_size=64), 0, 8, 4, 13, 1, True),          # 16 bounces, camera inside
    ("shuttle", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=64), 0, 8, 1, 12, 1, False),      # batch of one: nothing to share
    ("bunny", dict(width=64, height=64, n_samples=16, bunny_tris=8000), 0, 16, 16, 16, 1, True),               # rough and smooth metal: the smooth one reflects all samples alike
    ("house", dict(width=64, height=64, n_samples=8, detail=0.05, tex_size=64), 0, 8, 8, 14, 1, True),
    ("bike", dict(width=80, height=60, n_samples=8, detail=0.05, tex_size=64), 0, 8, 4, 15, 2, True),
])
def test_shared_primary_walk_bit_exact(gpu_ctx, name, kw, first, n, batch, cap, lanes, shared):
    """RT_OPT_SHARE_PRIMARY: the samples of a semantic batch start with one and the same primary ray
    (cuda/raytrace.cu:210), so one closest-hit walk serves them all.  The image and every count upstream's execution
    defines per sample stay what they were; the walks run are exactly the primaries saved fewer."""
    p = Pair(gpu_ctx, getattr(scene, name)(**kw))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, batch), first, n)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, cap)
    gpu_ctx.set_option(capi.OPT_WF_LANES, lanes)
    gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)
    try:
        got, gcnt = p.gpu_render(first, n, batch)
        again, _ = p.gpu_render(first, n, batch)
        # ... and with the walks of continued paths shared where their rays coincide (mirrors, glass)
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 1)
        both, bcnt = p.gpu_render(first, n, batch)
    finally:
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_WF_LANES, 1)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(got, ref), f"{(bits(got) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ"
    assert same_f32(again, ref)
    assert same_f32(both, ref), f"{(bits(both) != bits(ref)).any(axis=1).sum()} of {len(ref)} pixels differ with equal rays shared"
    assert bcnt[capi.CNT_RAYS] == rcnt[0] and bcnt[capi.CNT_SHADED] == rcnt[3] and bcnt[capi.CNT_TEX_FETCHES] == rcnt[4]
    assert bcnt[capi.CNT_WALKS] <= gcnt[capi.CNT_WALKS]
    npix = p.cfg.width * p.cfg.height
    assert gcnt[capi.CNT_RAYS] == rcnt[0]
    assert gcnt[capi.CNT_SHADED] == rcnt[3] and gcnt[capi.CNT_TEX_FETCHES] == rcnt[4]
    assert gcnt[capi.CNT_PATHS] == npix * n
    if shared and lanes == 1:
        assert gcnt[capi.CNT_WALKS] == rcnt[0] - npix * n + npix * _runs(first, n, batch)
        assert gcnt[capi.CNT_TRI_TESTS] < rcnt[2]
    elif shared:
        assert gcnt[capi.CNT_WALKS] < rcnt[0]            # how the pipelines cut the samples is theirs to choose
    else:
        assert gcnt[capi.CNT_WALKS] == rcnt[0] and gcnt[capi.CNT_TRI_TESTS] == rcnt[2]


def test_equal_rays_share_a_walk_in_a_glass_scene(gpu_ctx):
    """RT_OPT_SHARE_EQUAL on the all-glass car: after the shared first hit most samples of a batch refract the same way and
    are one ray again, bounce after bounce.  The image and the per-sample counts are the oracle's, the walks far fewer."""
    p = Pair(gpu_ctx, scene.formula1(width=128, height=80, n_samples=32, detail=0.05, glass=True))
    ref, rcnt = p.oracle.render(orc.make_params(p.cfg, SEED, 16), 0, 32)
    gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_WAVEFRONT)
    gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 17)
    gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 1)
    try:
        first_only, c1 = p.gpu_render(0, 32, 16)
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 1)
        got, c2 = p.gpu_render(0, 32, 16)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)             # equal rays can also be shared without the shared first walk
        alone, c3 = p.gpu_render(0, 32, 16)
    finally:
        gpu_ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
        gpu_ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
        gpu_ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
        gpu_ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    assert same_f32(first_only, ref) and same_f32(got, ref) and same_f32(alone, ref)
    for c in (c1, c2, c3):
        assert c[capi.CNT_RAYS] == rcnt[0] and c[capi.CNT_SHADED] == rcnt[3] and c[capi.CNT_PATHS] == 128 * 80 * 32
    print(f"walks per upstream ray: first walk shared {c1[capi.CNT_WALKS] / rcnt[0]:.3f}, equal rays too {c2[capi.CNT_WALKS] / rcnt[0]:.3f}, equal rays only {c3[capi.CNT_WALKS] / rcnt[0]:.3f}")
    assert c2[capi.CNT_WALKS] < 0.8 * c1[capi.CNT_WALKS] and c3[capi.CNT_WALKS] < rcnt[0]
