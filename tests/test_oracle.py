"""CPU tests of the oracle itself: known-answer vectors, structural invariants, self-consistency.

The upstream repository has no tests or golden vectors for this path, so these pin the oracle against
published third-party vectors (Philox), analytic cases and its own committed golden outputs.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from raytrace_amd import procgen, scene
from tests.util import bits, random_rays, same_f32

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_struct_sizes_match_upstream_layout():
    # sizes the upstream structs have under CUDA's alignment rules (float2 is 8-aligned)
    lib = orc.load()
    assert lib.orc_sizeof(0) == 104   # Triangle
    assert lib.orc_sizeof(1) == 36    # BVHNode
    assert lib.orc_sizeof(3) == 24    # Ray
    assert orc.TRIANGLE_DTYPE.itemsize == 104 and orc.NODE_DTYPE.itemsize == 36


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert list(orc.philox([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(orc.philox([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_mapping_and_draw_addressing():
    # every draw has a counter address (pixel, index, slot, stream); uint32 -> (0, 1] is cuRAND's mapping
    c1, c2 = np.float32(2.3283064365386963e-10), np.float32(1.1641532182693481e-10)
    for (pixel, index, slot, stream) in [(7, 3, 0, 0), (7, 3, 65536 + 1, 0), (123456, 9, 0, 1)]:
        raw = orc.philox([pixel, index, slot, stream], [0x5EED0001, 0])
        expect = (raw.astype(np.float32) * c1 + c2).astype(np.float32)
        assert same_f32(orc.uniforms(0x5EED0001, pixel, index, slot, stream), expect)
    # the extremes of the mapping: 0 -> 2^-33 (> 0), 2^32-1 -> 1.0 exactly
    assert (np.float32(0) * c1 + c2) > 0 and (np.float32(4294967295) * c1 + c2) == np.float32(1.0)
    u = np.concatenate([orc.uniforms(1, p, 0, 0, 0) for p in range(5000)])
    assert u.min() > 0.0 and u.max() <= 1.0 and abs(float(u.mean()) - 0.5) < 1e-2


def _plane_object():
    asm = scene.assemble(scene.RenderConfig(8, 8, 1), [scene.ObjectBlock(procgen.plane1(), "lambertian", albedo=(1, 1, 1))])
    return asm, scene.finalize(asm, orc.OracleBackend())


def test_plane_hit_known_answer():
    # a ray dropped straight onto the untransformed unit plane from y = 3 hits its first triangle at t = 3
    asm, objs = _plane_object()
    sc = orc.Scene(objs, asm.materials)
    obj, tri, t, u, v, _ = sc.closest_hit(np.array([[0.5, 3.0, -0.5, 0.0, -1.0, 0.0]], dtype=np.float32))
    assert obj[0] == 0 and tri[0] == 0 and t[0] == np.float32(3.0)
    # the shared diagonal belongs to both triangles; the leaf tests left then right and `t > rec.t` rejects only
    # strictly greater t, so the second triangle wins the tie (cuda/raytrace.cu:38, 57-62)
    obj, tri, t, u, v, _ = sc.closest_hit(np.array([[0.25, 3.0, 0.25, 0.0, -1.0, 0.0]], dtype=np.float32))
    assert tri[0] == 1 and t[0] == np.float32(3.0)
    # miss
    obj, tri, t, u, v, _ = sc.closest_hit(np.array([[2.5, 3.0, 0.0, 0.0, -1.0, 0.0]], dtype=np.float32))
    assert obj[0] == -1 and tri[0] == -1 and np.isinf(t[0])


@pytest.mark.parametrize("n_tris", [1, 2, 3, 4, 5, 7, 64, 1000])
def test_bvh_structure(n_tris):
    rng = np.random.RandomState(n_tris)
    tris = np.zeros(n_tris, dtype=orc.TRIANGLE_DTYPE)
    c = rng.uniform(-10, 10, size=(n_tris, 1, 3))
    tris["v"]["position"] = (c + rng.uniform(-1, 1, size=(n_tris, 3, 3))).astype(np.float32)
    nodes = orc.build_bvh(tris)
    leaves = nodes[nodes["is_leaf"] != 0]
    inner = nodes[nodes["is_leaf"] == 0]
    # <= 2 triangles per leaf, every triangle exactly once
    ids = np.concatenate([leaves["left"], leaves["right"][leaves["right"] >= 0]])
    assert sorted(ids.tolist()) == list(range(n_tris))
    assert (n_tris + 1) // 2 <= len(leaves) <= n_tris and len(inner) == len(leaves) - 1
    # post-order: children precede parents, root is last
    idx = np.nonzero(nodes["is_leaf"] == 0)[0]
    assert (inner["left"] < idx).all() and (inner["right"] < idx).all()
    if len(inner):
        assert nodes[-1]["is_leaf"] == 0
    # boxes contain their children / triangles
    for i, n in enumerate(nodes):
        if n["is_leaf"]:
            for k in (n["left"], n["right"]):
                if k >= 0:
                    p = tris[k]["v"]["position"]
                    assert (p >= n["box_min"]).all() and (p <= n["box_max"]).all()
        else:
            for k in (n["left"], n["right"]):
                assert (nodes[k]["box_min"] >= n["box_min"]).all() and (nodes[k]["box_max"] <= n["box_max"]).all()


def test_traversal_agrees_with_exhaustive_search():
    asm = scene.material_box()
    objs = scene.finalize(asm, orc.OracleBackend())
    sc = orc.Scene(objs, asm.materials, asm.textures)
    rays = random_rays(np.random.RandomState(3), 4000, -6.0, 6.0)
    obj, tri, t, u, v, _ = sc.closest_hit(rays)
    tb = sc.closest_hit_brute(rays)
    # identical except where a slab test rejects a box whose triangle Moller-Trumbore would accept
    mism = bits(t) != bits(tb)
    assert mism.mean() < 2e-3
    assert (t[mism] >= tb[mism]).all()


def test_render_is_independent_of_threads_and_launch_split():
    asm = scene.material_box(width=24, height=16, n_samples=8)
    objs = scene.finalize(asm, orc.OracleBackend())
    sc = orc.Scene(objs, asm.materials, asm.textures)
    p = orc.make_params(asm.config, 0x5EED0001, 4)
    full, cnt_full = sc.render(p, 0, 8)
    part, c1 = sc.render(p, 0, 4)
    part, c2 = sc.render(p, 4, 4, part)
    assert same_f32(full, part)
    assert (cnt_full == c1 + c2).all()
    # a different semantic batch changes the jitter sharing, hence the image
    p2 = orc.make_params(asm.config, 0x5EED0001, 8)
    other, _ = sc.render(p2, 0, 8)
    assert not same_f32(full, other)


def test_depth_exhausted_paths_add_nothing():
    # two facing mirrors and no light: with max_depth 0 nothing is traced at all
    asm = scene.material_box(width=8, height=8, n_samples=4)
    objs = scene.finalize(asm, orc.OracleBackend())
    sc = orc.Scene(objs, asm.materials, asm.textures)
    cfg = asm.config
    cfg.max_depth = 0
    acc, cnt = sc.render(orc.make_params(cfg, 1, 4), 0, 4)
    assert cnt[0] == 0 and not acc.any()


def test_post_process_known_answers():
    buf = np.array([[0.0, 8.0, 64.0], [-4.0, 2.0, 1e9]], dtype=np.float32)
    out = orc.post_process(buf, 16, 2.0)
    np.testing.assert_allclose(out, [[0.0, np.sqrt(0.5), 1.0], [0.0, np.sqrt(0.125), 1.0]], rtol=1e-6)


def test_texture_fetch_spec():
    rgba = np.zeros((2, 2, 4), dtype=np.uint8)
    rgba[0, 0] = (255, 0, 0, 255)
    rgba[0, 1] = (0, 255, 0, 255)
    rgba[1, 0] = (0, 0, 255, 255)
    rgba[1, 1] = (255, 255, 255, 255)
    # texel centres return the texel; the midpoint is the average; far outside is the (zero) border
    out = orc.tex2d(rgba, [[0.25, 0.25], [0.75, 0.25], [0.5, 0.5], [3.0, 3.0], [0.0, 0.0]])
    np.testing.assert_allclose(out[0], [1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(out[1], [0, 1, 0], atol=1e-6)
    np.testing.assert_allclose(out[2], [0.5, 0.5, 0.5], atol=1e-6)
    np.testing.assert_allclose(out[3], [0, 0, 0], atol=0)
    np.testing.assert_allclose(out[4], [0.25, 0, 0], atol=1e-6)   # corner: three of four taps are border


def test_golden_fixture():
    """The committed fixture (tests/golden/make_golden.py) pins the oracle's outputs bit for bit."""
    g = np.load(os.path.join(GOLDEN, "material_box_32x20.npz"))
    objs = [(g[f"tri{i}"].view(orc.TRIANGLE_DTYPE).reshape(-1), g[f"node{i}"].view(orc.NODE_DTYPE).reshape(-1)) for i in range(int(g["n_objects"]))]
    mats = g["materials"].view(orc.MATERIAL_DTYPE).reshape(-1)
    sc = orc.Scene(objs, mats, [g["texture0"]])
    for i, (t, n) in enumerate(objs):
        again = orc.build_bvh(t)
        assert np.array_equal(again["left"], n["left"]) and np.array_equal(again["right"], n["right"]) and same_f32(again["box_min"], n["box_min"])
    obj, tri, t, u, v, _ = sc.closest_hit(g["rays"])
    assert np.array_equal(obj, g["hit_obj"]) and np.array_equal(tri, g["hit_tri"]) and same_f32(t, g["hit_t"])
    assert same_f32(u, g["hit_u"]) and same_f32(v, g["hit_v"])
    cfg = scene.material_box(width=32, height=20, n_samples=8).config
    p = orc.make_params(cfg, int(g["seed"]), int(g["batch"]))
    assert same_f32(orc.primary_rays(p, 1), g["primary_rays_b1"])
    acc, cnt = sc.render(p, 0, 8)
    assert same_f32(acc, g["accum"])
    assert np.array_equal(cnt, g["counters"])
