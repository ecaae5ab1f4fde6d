"""The reference's own input fixtures, read where they lie: its eight scene files (assets/scene/scene_*.txt) and its two
meshes (assets/obj/simple/box.obj, plane1.obj).  They are the only data the reference tree holds for this path (it has no
tests and no golden outputs), so they are what can turn "numbers verbatim from the scene files" in raytrace_amd/scene.py
and "the two meshes upstream ships" in raytrace_amd/procgen.py from claims into checks.

Nothing is copied: the files are opened under /root/reference, which exists only in the build container, so the whole
module is skipped elsewhere (the GPU box never sees it).  Three readers meet on every file:
  * the product's C++ SceneLoader / OBJ reader (bin/scene_dump, bin/obj_dump) on the reference file itself,
  * a twenty-line tokeniser in this file, written from the grammar of csrc/sceneloader.cpp:57-96, 269-321,
  * raytrace_amd/scene.py / procgen.py, the Python restatement that the GPU tests and the benchmark are fed from.
The models the scene files name are not in the reference tree (downloaded by hand upstream); the procedural stand-in is
written to the path the scene file names, inside a scratch directory that plays the repository root.
"""
import os
import subprocess

import numpy as np
import pytest

from raytrace_amd import procgen, scene, scene_files
from tests.test_host_obj import dump as obj_dump, parse as parse_obj_dump
from tests.test_host_scene import SCENE_DUMP, assert_loader_output_matches

REF = "/root/reference"
REF_SCENES = os.path.join(REF, "assets", "scene")
REF_SIMPLE = os.path.join(REF, "assets", "obj", "simple")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF_SCENES) and os.path.exists(SCENE_DUMP) and os.path.exists(os.path.join(ROOT, "bin", "obj_dump"))),
                                reason="the reference tree is only present in the build container (and the host tools must be built)")

# scene file -> the restatement in raytrace_amd/scene.py, with arguments that keep its stand-in model small
SCENES = {
    "scene_luyu": ("luyu", dict(statue_tris=3000)),
    "scene_formula1": ("formula1", dict(detail=0.05, tex_size=32)),
    "scene_shuttle": ("shuttle", dict(detail=0.05, tex_size=32)),
    "scene_bike": ("bike", dict(detail=0.03, tex_size=16)),
    "scene_bunny": ("bunny", dict(bunny_tris=2000)),
    "scene_cat": ("cat", dict(detail=0.03, tex_size=16)),
    "scene_house": ("house", dict(detail=0.03, tex_size=16)),
    "scene_room": ("room", dict(detail=0.1, tex_size=16)),
}
# scene.py keeps these three at the benchmark's frame (2000x1200, 8192 spp; 400x240, 16 spp) instead of the file's
BENCHMARK_SIZED = {"scene_luyu", "scene_formula1", "scene_shuttle"}


def tokenise(path):
    """The scene grammar of csrc/sceneloader.cpp:269-321 and :57-96, independent of the product's loader: blank-separated
    words; a word `#` drops the rest of its line; `path` and `mtldir` take the rest of the line; the top-level `end`
    ends the file.  Returns (config dict of word lists, list of object dicts)."""
    config, objects, block = {}, [], None
    arity = {"max_ray_depth": 1, "max_surface": 1, "size": 2, "sample_per_pixel": 1, "fov": 1, "gamma": 1, "background": 3, "camera_pos": 3,
             "camera_dir": 3, "camera_up": 3, "russian_roulette": 1, "position": 3, "rotation": 3, "scale": 3, "albedo": 3, "optical_density": 1,
             "metal_fuzz": 1, "type": 1, "disable_normal": 0, "simple_material": 0}
    with open(path) as f:
        lines = f.read().split("\n")
    words = []                                     # (word, number of its line, column behind it)
    for number, ln in enumerate(lines):
        at = 0
        while True:
            while at < len(ln) and ln[at] in " \t\r":
                at += 1
            if at >= len(ln):
                break
            end = at
            while end < len(ln) and ln[end] not in " \t\r":
                end += 1
            words.append((ln[at:end], number, end))
            at = end
    assert words[0][0] == "scene_config"
    i = 1
    while i < len(words):
        w, number, behind = words[i]
        i += 1
        if w in ("#", "path", "mtldir"):
            if w != "#":
                block[w] = lines[number][behind:].strip()
            while i < len(words) and words[i][1] == number:
                i += 1                              # the remaining words of this line belong to the comment / the file name
            continue
        if w == "object":
            block = {}
            continue
        if w == "end":
            if block is None:
                break                               # the top-level end: nothing behind it is read
            objects.append(block)
            block = None
            continue
        n = arity[w]
        (block if block is not None else config)[w] = [words[i + k][0] for k in range(n)]
        i += n
    return config, objects


def floats(ws):
    return tuple(float(np.float32(float(w))) for w in ws)


def f32tuple(values):
    return tuple(float(np.float32(v)) for v in values)


@pytest.mark.parametrize("mesh", ["plane1", "box"])
def test_upstreams_two_meshes_are_what_procgen_restates(mesh):
    """bin/obj_dump (the product's OBJ reader) on the reference's own file against procgen's arrays: same triangles in the
    same order, corner for corner -- positions, normals, texture coordinates -- one shape, no material."""
    rc, text = obj_dump(os.path.join(REF_SIMPLE, mesh + ".obj"))
    assert rc == 0
    got = parse_obj_dump(text)
    model = getattr(procgen, mesh)()
    assert len(got["shapes"]) == len(model.shapes) == 1 and got["materials"] == []
    shp = model.shapes[0]
    tris = got["shapes"][0][1]
    assert len(tris) == len(shp)
    for k, (corners, material) in enumerate(tris):
        assert material == -1 == int(shp.mtl[k])
        for c, (v, vt, vn) in enumerate(corners):
            assert np.array_equal(np.array(got["vertices"][v], dtype=np.float32), shp.pos[k, c]), (k, c)
            assert np.array_equal(np.array(got["normals"][vn], dtype=np.float32), shp.nrm[k, c]), (k, c)
            assert np.array_equal(np.array(got["texcoords"][vt], dtype=np.float32), shp.uv[k, c]), (k, c)


@pytest.mark.parametrize("name", sorted(SCENES))
def test_scene_file_values_are_what_scene_py_restates(name):
    """Every config value and every object block of the reference's scene file, word for word, against the RenderConfig and
    the ObjectBlocks raytrace_amd/scene.py builds for that scene."""
    fn, kw = SCENES[name]
    config, objects = tokenise(os.path.join(REF_SCENES, name + ".txt"))
    asm = getattr(scene, fn)(**kw)
    cfg = asm.config
    assert int(config["max_ray_depth"][0]) == cfg.max_depth
    assert floats(config["fov"]) == f32tuple([cfg.fov_deg]) and floats(config["gamma"]) == f32tuple([cfg.gamma])
    assert floats(config["russian_roulette"]) == f32tuple([cfg.russian_roulette])
    assert floats(config["background"]) == f32tuple(cfg.background)
    assert floats(config["camera_pos"]) == f32tuple(cfg.camera_pos) and floats(config["camera_dir"]) == f32tuple(cfg.camera_dir)
    assert floats(config["camera_up"]) == f32tuple(cfg.camera_up)
    if name not in BENCHMARK_SIZED:
        assert (int(config["size"][0]), int(config["size"][1]), int(config["sample_per_pixel"][0])) == (cfg.width, cfg.height, cfg.n_samples)
    assert len(objects) == len(asm.blocks)
    for k, (obj, blk) in enumerate(zip(objects, asm.blocks)):
        # keys a block leaves out keep the loader's initial values (csrc/sceneloader.cpp:41-55)
        assert floats(obj.get("position", ["0", "0", "0"])) == f32tuple(blk.position), (k, "position")
        assert floats(obj.get("rotation", ["0", "0", "0"])) == f32tuple(blk.rotation), (k, "rotation")
        assert floats(obj.get("scale", ["1", "1", "1"])) == f32tuple(blk.scale), (k, "scale")
        assert floats(obj.get("albedo", ["0", "0", "0"])) == f32tuple(blk.albedo), (k, "albedo")
        assert obj["type"][0] == blk.type, (k, "type")
        assert ("simple_material" in obj) == blk.simple_material and ("disable_normal" in obj) == blk.disable_normal, k
        if "metal_fuzz" in obj:
            assert floats(obj["metal_fuzz"]) == f32tuple([blk.metal_fuzz]), (k, "metal_fuzz")
        if "optical_density" in obj:
            assert floats(obj["optical_density"]) == f32tuple([blk.optical_density]), (k, "optical_density")
        simple = obj["path"].startswith("assets/obj/simple/") and os.path.basename(obj["path"]) in ("plane1.obj", "box.obj")
        if simple:        # the block uses one of the two meshes upstream ships: the restatement must use the same one
            expect = procgen.plane1() if obj["path"].endswith("plane1.obj") else procgen.box()
            assert len(blk.model.shapes) == 1 and np.array_equal(blk.model.shapes[0].pos, expect.shapes[0].pos), k


@pytest.mark.parametrize("name", sorted(SCENES))
def test_product_loader_reads_the_reference_scene_file_in_place(tmp_path, name):
    """bin/scene_dump (the product's SceneLoader: scene text, OBJ/MTL, textures, transform, BVH build) is started on the
    reference's scene file itself, from a scratch directory laid out like the repository root: assets/obj/simple/plane1.obj
    and box.obj are links to the reference's files, the model the scene names (absent upstream) is the stand-in written
    to the very path the file gives.  What the loader built must equal, array for array, what raytrace_amd/scene.py
    assembles -- triangles after the transform, BVH nodes, material table, textures, camera and render settings."""
    fn, kw = SCENES[name]
    scene_path = os.path.join(REF_SCENES, name + ".txt")
    config, objects = tokenise(scene_path)
    asm = getattr(scene, fn)(**kw)
    root = tmp_path / "root"
    simple = root / "assets" / "obj" / "simple"
    simple.mkdir(parents=True)
    for mesh in ("plane1.obj", "box.obj"):
        os.symlink(os.path.join(REF_SIMPLE, mesh), str(simple / mesh))
    for obj, blk in zip(objects, asm.blocks):
        target = root / obj["path"]
        if target.exists() or target.is_symlink():
            continue
        target.parent.mkdir(parents=True, exist_ok=True)
        # the material file gets a name without blanks: an OBJ's `mtllib` line is split at blanks (third/tiny_obj_loader.h:2035-2052)
        scene_files.write_obj(blk.model, str(target), mtl_name="standin.mtl")
        mtldir = obj.get("mtldir")
        if mtldir is not None:                      # the MTL file is written beside the OBJ: the scene's mtldir must be that directory
            assert os.path.normpath(str(root / mtldir)) == os.path.normpath(str(target.parent)), (mtldir, obj["path"])
    out = tmp_path / "dump"
    out.mkdir()
    r = subprocess.run([SCENE_DUMP, scene_path, str(out)], capture_output=True, text=True, timeout=300, cwd=str(root))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    size = (int(config["size"][0]), int(config["size"][1]), int(config["sample_per_pixel"][0]))
    assert_loader_output_matches(asm, r.stdout, out, size_and_spp=size)
