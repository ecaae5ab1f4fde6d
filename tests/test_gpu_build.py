"""The scene tree built on the GPU (RT_OPT_GPU_BVH): a different tree than upstream's, so its hit records are checked
against an exhaustive search over all triangles (the oracle's brute-force nearest t), not against upstream's walk.

What may differ from the exhaustive search, and does in well under 1 % of rays: a box whose slab test rejects a ray that
the Moller-Trumbore test of a triangle inside would accept (both tests are upstream's arithmetic; they disagree for rays
that graze a box face or a triangle edge).  Such a ray's hit is then a farther triangle or a miss -- never a nearer one,
and never a t that no triangle gives.  Upstream's own walk has the same property on its own boxes."""
import numpy as np
import pytest

from oracle import oracle as orc
from raytrace_amd import capi, scene
from tests.util import bits, random_rays, same_f32

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001


def _hits(ctx, rays):
    import torch
    n = len(rays)
    d_rays = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32)).cuda()
    d_obj = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    d_tri = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    d_t, d_u, d_v = (torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(3))
    torch.cuda.synchronize()
    ctx.closest_hit(n, d_rays.data_ptr(), d_obj.data_ptr(), d_tri.data_ptr(), d_t.data_ptr(), d_u.data_ptr(), d_v.data_ptr())
    ctx.synchronize()
    return d_obj.cpu().numpy(), d_tri.cpu().numpy(), d_t.cpu().numpy(), d_u.cpu().numpy(), d_v.cpu().numpy()


def _two_contexts(asm):
    """The same scene twice: upstream's trees re-packed (default) and the tree built on the device."""
    cfg = asm.config
    objs = scene.finalize(asm)
    ref = capi.Context(0)
    ref.upload_scene(objs, asm.materials, asm.textures)
    own = capi.Context(0)
    own.set_option(capi.OPT_GPU_BVH, 1)
    own.upload_scene([(t, None) for t, _ in objs], asm.materials, asm.textures)        # no node arrays at all
    for c in (ref, own):
        c.init_randstate(SEED, cfg.width, cfg.height)
        c.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
    return objs, ref, own


@pytest.mark.parametrize("name,kw,box", [
    ("material_box", dict(width=96, height=64, n_samples=4), 6.0),
    ("formula1", dict(width=200, height=120, n_samples=4, detail=0.15, tex_size=64), 150.0),
    ("luyu", dict(width=200, height=120, n_samples=4, statue_tris=60000), 200.0),
])
def test_device_built_tree_agrees_with_exhaustive_search(name, kw, box):
    asm = getattr(scene, name)(**kw)
    objs, ref, own = _two_contexts(asm)
    try:
        st = own.scene_stats()
        assert st["triangles"] == sum(len(t) for t, _ in objs) and 0 < st["stack_slots"] <= 60
        oracle = orc.Scene(objs, asm.materials, asm.textures)
        rays = np.concatenate([orc.primary_rays(orc.make_params(asm.config, SEED, 4), 0)[::3],
                               random_rays(np.random.RandomState(11), 20000, -box, box)])
        brute = oracle.closest_hit_brute(rays)
        o, tr, t, u, v = _hits(own, rays)
        ro, rtr, rt, ru, rv = _hits(ref, rays)
        finite = np.isfinite(rays).all(axis=1)
        # never nearer than the exhaustive search, never a t no triangle gives
        assert (t[finite] >= brute[finite]).all()
        same = bits(t) == bits(brute)
        missed = ~same & finite
        print(f"{name}: {missed.sum()} of {len(rays)} rays differ from the exhaustive search (slab test rejects a box whose triangle test accepts); "
              f"upstream's tree: {(bits(rt) != bits(brute)).sum()}")
        assert same.mean() > 0.995
        assert ((o >= 0) == np.isfinite(t)).all() and (u[o >= 0] >= 0).all() and (v[o >= 0] >= 0).all() and ((u + v)[o >= 0] <= 1).all()
        # where both trees see the same t, they name the same triangle except at exact ties (shared edges), where the
        # last triangle visited wins and the visiting orders differ
        both = bits(t) == bits(rt)
        assert both.mean() > 0.995
        ties = both & ((o != ro) | (tr != rtr))
        print(f"{name}: {ties.sum()} exact ties resolved differently")
        assert ties.mean() < 0.02
        assert same_f32(u[both & ~ties], ru[both & ~ties])
    finally:
        ref.close()
        own.close()


def test_device_built_tree_renders_the_same_picture_up_to_tie_rays():
    import torch
    asm = scene.formula1(width=160, height=96, n_samples=16, detail=0.1, tex_size=64)
    cfg = asm.config
    _, ref, own = _two_contexts(asm)
    try:
        npix = cfg.width * cfg.height
        out = []
        for c in (ref, own):
            d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            c.raytrace(npix, 0, 16, 4, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
            c.synchronize()
            out.append(d.cpu().numpy())
        a, b = out
        identical = (bits(a) == bits(b)).all(axis=1).mean()
        print(f"{identical:.4f} of the pixels bit-identical between the two trees")
        assert identical > 0.9                       # same seed, same draws: paths differ only where a tie or a grazing ray falls
        assert abs(float(a.mean()) - float(b.mean())) < 0.02 * float(a.mean())
    finally:
        ref.close()
        own.close()


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5])
def test_device_built_tree_of_a_few_triangles(n_tris):
    """One leaf (no records at all), one record with two leaves, odd counts: the edge shapes of the hierarchy."""
    tris = np.zeros(n_tris, dtype=capi.TRIANGLE_DTYPE)
    for k in range(n_tris):
        tris["v"]["position"][k] = [[0, 0, k], [1, 0, k], [0, 1, k]]
        tris["v"]["normal"][k] = [[0, 0, 1]] * 3
    mats = np.zeros(1, dtype=capi.MATERIAL_DTYPE)
    mats["type"] = capi.MAT_LAMBERTIAN
    mats["texture"] = -1
    ctx = capi.Context(0)
    try:
        ctx.set_option(capi.OPT_GPU_BVH, 1)
        ctx.upload_scene([(tris, None)], mats, [])
        rays = np.array([[0.25, 0.25, 10, 0, 0, -1], [0.25, 0.25, -10, 0, 0, 1], [5, 5, 10, 0, 0, -1], [0.25, 0.25, 1.5, 0, 0, -1]], dtype=np.float32)
        o, tr, t, u, v = _hits(ctx, rays)
        assert list(tr[:3]) == [n_tris - 1, 0, -1] and list(o[:3]) == [0, 0, -1]
        assert t[0] == 10 - (n_tris - 1) and t[1] == 10 and np.isinf(t[2])
        assert tr[3] == (1 if n_tris > 1 else 0) and u[0] == 0.25 and v[0] == 0.25
    finally:
        ctx.close()


def test_device_build_of_the_30m_triangle_scene_is_fast_and_agrees_with_upstreams_walk():
    asm = scene.stress(width=960, height=540, n_samples=1)
    cfg = asm.config
    objs, ref, own = _two_contexts(asm)
    try:
        st = own.scene_stats()
        print(f"30 M triangles: device build {st['gpu_build_ms']:.1f} ms, copy of the triangle arrays {st['gpu_copy_ms']:.1f} ms, "
              f"{st['inner_nodes']} records, stack need {st['stack_slots']}, binary levels {st['gpu_binary_levels']}")
        assert st["triangles"] > 30_000_000
        assert st["gpu_build_ms"] < 500.0
        import torch
        d = torch.zeros(cfg.width * cfg.height, 6, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        own.primary_rays(0, d.data_ptr())
        own.synchronize()
        rays = d.cpu().numpy()
        o, tr, t, u, v = _hits(own, rays)
        ro, rtr, rt, ru, rv = _hits(ref, rays)
        agree = bits(t) == bits(rt)
        print(f"{(~agree).sum()} of {len(rays)} primary rays see another t than upstream's walk")
        assert agree.mean() > 0.995 and (o >= 0).mean() > 0.5
    finally:
        ref.close()
        own.close()


def test_device_build_is_repeatable_and_the_code_bit_hierarchy_still_works(monkeypatch):
    """The agglomeration numbers its inner nodes in the order the hardware makes them; nothing a ray sees may depend on
    that: two builds of one scene render the same buffer bit for bit and count the same tests.  The older hierarchy (the
    one the Morton code bits give, RT_BUILD_TREE=0) stays selectable and agrees with the exhaustive search as well."""
    import torch
    asm = scene.formula1(width=160, height=96, n_samples=8, detail=0.1, tex_size=64)
    cfg = asm.config
    objs = scene.finalize(asm)
    npix = cfg.width * cfg.height

    def render():
        ctx = capi.Context(0)
        try:
            ctx.set_option(capi.OPT_GPU_BVH, 1)
            ctx.set_option(capi.OPT_COUNTERS, 1)
            # which coinciding continued rays share a walk depends on the order the waves' compaction atomics arrive in, so the
            # walks run (and the tests inside them) may differ by a handful between two renders; the test counts compared below
            # are to tell trees apart, so every ray is walked
            ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
            ctx.upload_scene([(t, None) for t, _ in objs], asm.materials, asm.textures)
            ctx.init_randstate(SEED, cfg.width, cfg.height)
            ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
            d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            ctx.raytrace(npix, 0, 8, 4, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
            c = ctx.counters()
            rays = np.concatenate([orc.primary_rays(orc.make_params(cfg, SEED, 4), 0)[::5], random_rays(np.random.RandomState(11), 4000, -150.0, 150.0)])
            return d.cpu().numpy(), c, _hits(ctx, rays), rays
        finally:
            ctx.close()

    a, ca, ha, rays = render()
    b, cb, hb, _ = render()
    assert same_f32(a, b)
    for k in (capi.CNT_RAYS, capi.CNT_BOX_TESTS, capi.CNT_TRI_TESTS, capi.CNT_SHADED):
        assert ca[k] == cb[k], k
    assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ha, hb))
    monkeypatch.setenv("RT_BUILD_TREE", "0")
    c, cc, hc, _ = render()
    oracle = orc.Scene(objs, asm.materials, asm.textures)
    brute = oracle.closest_hit_brute(rays)
    for name, h in (("agglomerated", ha), ("code bits", hc)):
        t = h[2]
        assert (t >= brute).all(), name                        # never nearer than the nearest triangle
        assert (bits(t) == bits(brute)).mean() > 0.995, name
    assert cc[capi.CNT_TRI_TESTS] > ca[capi.CNT_TRI_TESTS]       # what the better leaves are for


def test_device_build_falls_back_when_the_agglomeration_crawls(capfd):
    """Triangles along a line whose gaps shrink steadily: every cluster prefers its right-hand neighbour, only the last two
    choose each other, and the agglomeration joins one pair a round.  The builder gives up after a fixed number of
    rounds and builds the hierarchy the code bits give; the hits are still those of the exhaustive search."""
    n = 1500
    gaps = 0.995 ** np.arange(n)
    x = np.concatenate([[0.0], np.cumsum(gaps)[:-1]])
    tris = np.zeros(n, dtype=capi.TRIANGLE_DTYPE)
    for k in range(n):
        tris["v"]["position"][k] = [[x[k], 0, 0], [x[k], 1, 0], [x[k], 0, 1]]
        tris["v"]["normal"][k] = [[1, 0, 0]] * 3
    mats = np.zeros(1, dtype=capi.MATERIAL_DTYPE)
    mats["type"] = capi.MAT_LAMBERTIAN
    mats["texture"] = -1
    ctx = capi.Context(0)
    try:
        ctx.set_option(capi.OPT_GPU_BVH, 1)
        ctx.upload_scene([(tris, None)], mats, [])
        assert ctx.scene_stats()["triangles"] == n
        assert "building the hierarchy of the code bits instead" in capfd.readouterr().err
        rng = np.random.RandomState(5)
        o = np.stack([rng.uniform(-5, x[-1] + 5, 3000), rng.uniform(0.05, 0.4, 3000), rng.uniform(0.05, 0.4, 3000)], axis=1)
        d = np.stack([rng.choice([-1.0, 1.0], 3000), rng.uniform(-0.01, 0.01, 3000), rng.uniform(-0.01, 0.01, 3000)], axis=1)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d], axis=1).astype(np.float32)
        _, tr, t, _, _ = _hits(ctx, rays)
        bvh = orc.build_bvh(tris)
        brute = orc.Scene([(tris, bvh)], mats, []).closest_hit_brute(rays)
        assert (bits(t) == bits(brute)).mean() > 0.995 and (t >= brute).all() and np.isfinite(t).mean() > 0.5
    finally:
        ctx.close()


def test_hot_node_tuning_on_a_device_built_tree_moves_records_not_results():
    """RT_OPT_GPU_BVH with RT_OPT_HOT_TUNE 0 and 2 (the default: one numbering for the view, one copy per band) at 64 spp (the calibration only runs from 64 samples per call on): the
    renumbering of the LDS candidates must leave a device-built tree's results alone -- same buffer bit for bit, same
    counts, the same closest-hit records before and after the renumbering -- and its premise is checked on the records
    themselves: parents carry lower numbers than their children, so no record behind the candidates refers into them."""
    import torch
    asm = scene.formula1(width=200, height=120, n_samples=64, detail=0.15, tex_size=64)
    cfg = asm.config
    objs = scene.finalize(asm)
    rays = np.concatenate([orc.primary_rays(orc.make_params(cfg, SEED, 16), 0)[::2], random_rays(np.random.RandomState(5), 8000, -150.0, 150.0)])
    npix = cfg.width * cfg.height
    out = {}
    for tune in (0, 2):
        ctx = capi.Context(0)
        try:
            ctx.set_option(capi.OPT_GPU_BVH, 1)
            ctx.set_option(capi.OPT_HOT_TUNE, tune)
            ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 18)
            ctx.upload_scene([(t, None) for t, _ in objs], asm.materials, asm.textures)
            ctx.init_randstate(SEED, cfg.width, cfg.height)
            ctx.setup_raytrace(cfg.width, cfg.height, cfg.camera_pos, cfg.camera_dir, cfg.camera_up, cfg.fov)
            order = ctx.check_node_numbering()
            assert order["records"] > order["candidates"] == 4096, order          # the tree is larger than the candidate set: the premise matters
            assert order["refs_into_candidates_from_behind"] == 0 and order["refs_out_of_range"] == 0 and order["refs_not_ascending"] == 0, order
            before = _hits(ctx, rays)
            d = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            ctx.set_option(capi.OPT_COUNTERS, 1)
            ctx.reset_counters()
            ctx.raytrace(npix, 0, 64, 16, cfg.max_depth, cfg.background, cfg.russian_roulette, d.data_ptr())
            cnt = ctx.counters()
            after = _hits(ctx, rays)                                              # on the renumbered records when tune != 0
            for a, b in zip(before, after):
                assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
            order = ctx.check_node_numbering()
            assert order["refs_into_candidates_from_behind"] == 0 and order["refs_out_of_range"] == 0, order
            out[tune] = (d.cpu().numpy(), cnt, before)
        finally:
            ctx.close()
    (a, ca, ha), (b, cb, hb) = out[0], out[2]
    assert same_f32(a, b), f"{(bits(a) != bits(b)).any(axis=1).sum()} pixels differ between tuned and untuned numbering"
    for k in (capi.CNT_RAYS, capi.CNT_SHADED, capi.CNT_TEX_FETCHES, capi.CNT_PATHS):
        assert ca[k] == cb[k], k
    # Which continued rays are found to coincide (RT_OPT_SHARE_EQUAL looks at the 15 lanes before a survivor) depends on
    # how the waves' compaction atomics happened to be ordered, so the walks run -- and the tests inside them -- may
    # differ by a handful from one render to the next; the rays they answer do not.
    for k in (capi.CNT_WALKS, capi.CNT_BOX_TESTS, capi.CNT_TRI_TESTS, capi.CNT_NODE_VISITS):
        assert abs(ca[k] - cb[k]) <= 1e-3 * ca[k], k
    assert cb[capi.CNT_HOT_VISITS] > 1.02 * ca[capi.CNT_HOT_VISITS]               # the calibration did run and did move records
    for x, y in zip(ha, hb):                                                      # two builds of the same scene answer alike (numbers are addresses)
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
