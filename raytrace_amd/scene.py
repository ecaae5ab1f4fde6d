"""Scene assembly for tests and the benchmark: the host-side logic of upstream's
SceneLoader (csrc/sceneloader.cpp:39-221, 254-321) applied to in-memory models.

`assemble` turns object blocks (model + transform + base material) into
upstream-layout triangle arrays and a material table; `finalize` runs the
vertex transform and the BVH build through a backend -- the product library
(raytrace_amd.capi) or, in tests only, the CPU oracle -- so that both sides can
be fed bit-identical inputs.
"""
from dataclasses import dataclass, field

import numpy as np

from . import procgen
from .capi import (MAT_LAMBERTIAN, MAT_LIGHT, MAT_METAL, MAT_REFLECTIVE, MAT_REFRACTIVE, MATERIAL_DTYPE, NODE_DTYPE,
                   TRIANGLE_DTYPE)

TYPE_BY_NAME = {"light": MAT_LIGHT, "lambertian": MAT_LAMBERTIAN, "metal": MAT_METAL, "refractive": MAT_REFRACTIVE,
                "reflective": MAT_REFLECTIVE}


@dataclass
class ObjectBlock:
    """One `object ... end` block of a scene file (csrc/sceneloader.cpp:57-96)."""
    model: procgen.Model
    type: str
    position: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (0.0, 0.0, 0.0)
    scale: tuple = (1.0, 1.0, 1.0)
    albedo: tuple = (0.0, 0.0, 0.0)
    optical_density: float = 1.0
    metal_fuzz: float = 0.0
    disable_normal: bool = False
    simple_material: bool = False


@dataclass
class RenderConfig:
    """include/common.cuh:19-33; fov is kept in degrees here and converted as the loader does."""
    width: int
    height: int
    n_samples: int
    max_depth: int = 8
    fov_deg: float = 60.0
    gamma: float = 1.0
    russian_roulette: float = 0.9
    background: tuple = (0.01, 0.01, 0.01)
    camera_pos: tuple = (0.0, 0.0, 0.0)
    camera_dir: tuple = (1.0, 0.0, 0.0)
    camera_up: tuple = (0.0, 0.0, 1.0)

    @property
    def fov(self):
        # csrc/sceneloader.cpp:287: float * double(M_PI) / 180.0f, stored back into a float
        return float(np.float32(np.float64(np.float32(self.fov_deg)) * np.pi / np.float64(np.float32(180.0))))


@dataclass
class Assembled:
    config: RenderConfig
    triangles: list = field(default_factory=list)    # per upstream object (= OBJ shape), pre-transform
    transforms: list = field(default_factory=list)   # (scale, translation, rotation_deg) per object
    materials: np.ndarray = None
    textures: list = field(default_factory=list)     # uint8 (h, w, 4), rows bottom-up
    names: list = field(default_factory=list)
    blocks: list = field(default_factory=list)       # the object blocks this was assembled from (raytrace_amd.scene_files writes them out)

    @property
    def n_triangles(self):
        return sum(len(t) for t in self.triangles)


def _face_normals(pos):
    """normalize(cross(e1, e2)) in float32, operand order of csrc/sceneloader.cpp:158-160 / common.cuh:75-94."""
    f = np.float32
    e1 = (pos[:, 1] - pos[:, 0]).astype(f)
    e2 = (pos[:, 2] - pos[:, 0]).astype(f)
    c = np.empty_like(e1)
    c[:, 0] = (e1[:, 1] * e2[:, 2]).astype(f) - (e1[:, 2] * e2[:, 1]).astype(f)
    c[:, 1] = (e1[:, 2] * e2[:, 0]).astype(f) - (e1[:, 0] * e2[:, 2]).astype(f)
    c[:, 2] = (e1[:, 0] * e2[:, 1]).astype(f) - (e1[:, 1] * e2[:, 0]).astype(f)
    n = np.sqrt((((c[:, 0] * c[:, 0]).astype(f) + (c[:, 1] * c[:, 1]).astype(f)).astype(f) + (c[:, 2] * c[:, 2]).astype(f)).astype(f)).astype(f)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (c / n[:, None]).astype(f)


def assemble(config, blocks):
    """SceneLoader::load_object for each block: one upstream object per non-empty shape; materials =
    base material, then one per MTL record unless simple_material (sceneloader.cpp:132-219)."""
    out = Assembled(config=config, blocks=list(blocks))
    mats = []
    for blk in blocks:
        base_type = TYPE_BY_NAME[blk.type]
        material_offset = len(mats)
        for shp in blk.model.shapes:
            n = len(shp)
            if n == 0:
                continue                                           # sceneloader.cpp:136-139
            tri = np.zeros(n, dtype=TRIANGLE_DTYPE)
            tri["v"]["position"] = shp.pos
            if blk.disable_normal or not shp.has_normals:
                tri["v"]["normal"] = _face_normals(shp.pos)[:, None, :]
            else:
                tri["v"]["normal"] = shp.nrm
            if shp.has_uvs:
                tri["v"]["texcoord"] = shp.uv                      # upstream leaves these uninitialised otherwise; zero here
            mid = np.full(n, material_offset, dtype=np.int32)
            if not blk.simple_material:
                has = shp.mtl >= 0
                mid[has] += shp.mtl[has] + 1                       # sceneloader.cpp:180-183
            tri["material_id"] = mid
            out.triangles.append(tri)
            out.transforms.append((tuple(blk.scale), tuple(blk.position), tuple(blk.rotation)))
            out.names.append(shp.name)
        # sceneloader.cpp:188-219.  `type` and `optical_density` are locals that persist across the MTL loop.
        mats.append((base_type, blk.albedo, blk.optical_density, blk.metal_fuzz, -1))
        if not blk.simple_material and blk.model.mtls:
            optical_density = blk.optical_density
            for m in blk.model.mtls:
                mtype = {1: MAT_LAMBERTIAN, 2: MAT_METAL, 3: MAT_LIGHT, 4: MAT_REFRACTIVE, 5: MAT_REFLECTIVE}.get(m["illum"], MAT_LAMBERTIAN)
                if mtype == MAT_LIGHT:
                    albedo = m["Ke"]
                else:
                    albedo = m["Kd"]
                    if mtype == MAT_REFRACTIVE:
                        optical_density = m["Ni"]
                        f = np.float32
                        dsv = f(m["d"])
                        albedo = tuple(float(f(f(f(1.0) - dsv) * f(1.0)) + f(dsv * f(a))) for a in albedo)
                tex = -1
                if m["map_Kd"] is not None:
                    tex = len(out.textures)
                    out.textures.append(np.ascontiguousarray(m["map_Kd"], dtype=np.uint8))
                mats.append((mtype, albedo, optical_density, blk.metal_fuzz, tex))
    arr = np.zeros(len(mats), dtype=MATERIAL_DTYPE)
    for i, (t, a, od, fz, tx) in enumerate(mats):
        arr[i]["type"] = t
        arr[i]["albedo"] = a
        arr[i]["optical_density"] = od
        arr[i]["metal_fuzz"] = fz
        arr[i]["texture"] = tx
    out.materials = arr
    return out


class ProductBackend:
    """Vertex transform + BVH build through librtcore.so."""
    name = "rtcore"

    def apply_transform(self, tris, scale, translation, rotation):
        from . import capi
        return capi.apply_transform(tris, scale, translation, rotation)

    def build_bvh(self, tris):
        from . import capi
        return capi.build_bvh(tris)


def finalize(asm, backend=None, progress=None):
    """Transform every object's triangles and build its BVH (sceneloader.cpp:336-375).
    Returns [(triangles, nodes)] in upstream layout.  `progress(i, n)` is called after each object."""
    backend = backend or ProductBackend()
    objs = []
    for i, (tri, (scale, pos, rot)) in enumerate(zip(asm.triangles, asm.transforms)):
        t = tri.copy()
        backend.apply_transform(t, scale, pos, rot)
        nodes = backend.build_bvh(t)
        assert nodes.dtype == NODE_DTYPE
        objs.append((t, nodes))
        if progress:
            progress(i + 1, len(asm.triangles))
    return objs


# ---------------------------------------------------------------------------
# the benchmark / test scenes: upstream scene files' numbers + procedural models
# ---------------------------------------------------------------------------
def luyu(width=400, height=240, n_samples=16, statue_tris=200_000):
    """assets/scene/scene_luyu.txt:6-58 with the Lu Yu scan replaced by procgen.statue."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.0, russian_roulette=0.9,
                       background=(0.05, 0.05, 0.05), camera_pos=(-280, 0, 190), camera_dir=(1, 0, -0.26), camera_up=(0, 0, 1))
    blocks = [
        ObjectBlock(procgen.statue(statue_tris), "metal", albedo=(0.6, 0.6, 0.6), metal_fuzz=0.05, optical_density=3, simple_material=True),
        ObjectBlock(procgen.plane1(), "light", position=(0, 0, 400), rotation=(90, 0, 0), scale=(2e2, 0, 2e2), albedo=(1, 1, 1)),
        ObjectBlock(procgen.box(), "metal", position=(16.73, 83.83, 132), scale=(20, 20, 20), albedo=(1, 0, 0), metal_fuzz=0.1,
                    optical_density=3, simple_material=True),
        ObjectBlock(procgen.plane1(), "lambertian", rotation=(90, 0, 0), scale=(1e6, 1e6, 1e6), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def formula1(width=2000, height=1200, n_samples=8192, glass=False, detail=1.0, tex_size=1024):
    """assets/scene/scene_formula1.txt:6-73 (the block after the top-level `end` is never read) with the car replaced
    by procgen.formula1.  glass=True is the file's `simple_material` variant: every car triangle refractive, density 3."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.8, russian_roulette=0.9,
                       background=(0.01, 0.01, 0.01), camera_pos=(240, -190, 70), camera_dir=(-1, 0.49, -0.2), camera_up=(0, 0, 1))
    car = procgen.formula1(detail=detail, with_textures=not glass, tex_size=tex_size)
    blocks = [
        ObjectBlock(car, "refractive", rotation=(90, 100, 0), albedo=(0.8, 0.8, 0.8), metal_fuzz=0.1, optical_density=3, simple_material=glass),
        ObjectBlock(procgen.plane1(), "lambertian", rotation=(90, 0, 0), scale=(1e3, 1e3, 1e3), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(0, -30, 220), rotation=(90, 0, 0), scale=(40, 0, 80), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(240, 40, 100), rotation=(90, 0, -80), scale=(40, 0, 80), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(60, -260, 90), rotation=(-20, 0, 20), scale=(75, 0, 75), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def shuttle(width=2000, height=1200, n_samples=8192, detail=1.0, tex_size=1024):
    """assets/scene/scene_shuttle.txt:6-66 with the shuttle replaced by procgen.shuttle."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.2, russian_roulette=0.9,
                       background=(0.01, 0.01, 0.01), camera_pos=(-4, 7, 2.5), camera_dir=(0.43, -1, -0.2), camera_up=(0, 0, 1))
    blocks = [
        ObjectBlock(procgen.shuttle(detail=detail, tex_size=tex_size), "lambertian", rotation=(90, 0, 0), albedo=(1, 1, 1), metal_fuzz=0.2, optical_density=3),
        ObjectBlock(procgen.plane1(), "light", position=(0, 0, 8), rotation=(90, 0, 0), scale=(8, 0, 3), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "lambertian", position=(0, 0, -100), rotation=(90, 0, 0), scale=(1e6, 1e6, 1e6), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(0, 8, 3), rotation=(0, 0, 0), scale=(4, 0, 2), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(-12, 0, 3), rotation=(0, 0, 90), scale=(2, 0, 2), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def bike(width=1600, height=1200, n_samples=2048, detail=1.0, tex_size=1024):
    """assets/scene/scene_bike.txt:6-52 with the motorbike replaced by the formula-1 stand-in fitted into the box a
    model standing on that scene's ground occupies (y up before the file's `rotation 90 0 0`)."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.6, russian_roulette=0.9,
                       background=(0.05, 0.05, 0.05), camera_pos=(-1100, 0, 1000), camera_dir=(1, 0, -0.22), camera_up=(0, 0, 1))
    model = procgen.fitted(procgen.formula1(detail=detail, tex_size=tex_size), (-260, 700, 0), (260, 1150, 1000))
    blocks = [
        ObjectBlock(model, "metal", position=(0, 500, -700), rotation=(90, 0, 0), albedo=(1, 1, 1), metal_fuzz=0.2, optical_density=3),
        ObjectBlock(procgen.plane1(), "light", position=(-200, 0, 1400), rotation=(90, 0, 0), scale=(2e2, 0, 1e3), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(-1101, 0, 400), rotation=(0, 0, 90), scale=(1e3, 0, 1e3), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "lambertian", rotation=(90, 0, 0), scale=(1e4, 0, 1e4), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def bunny(width=1200, height=1200, n_samples=256, bunny_tris=70_000):
    """assets/scene/scene_bunny.txt:6-65 with bunny.obj (a single shape, no MTL) replaced by the statue generator
    fitted into the Stanford bunny's box.  Two `simple_material` metals of different fuzz, two lights."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.2, russian_roulette=0.9,
                       background=(0.01, 0.0, 0.01), camera_pos=(0, -9, 4.5), camera_dir=(0, 1, -0.1), camera_up=(0, 0, 1))
    model = procgen.fitted(procgen.statue(bunny_tris), (-0.095, 0.033, -0.062), (0.061, 0.187, 0.059))
    blocks = [
        ObjectBlock(model, "metal", position=(1, 2, -2), rotation=(90, 0, 0), scale=(50, 50, 50), albedo=(0.1, 0.1, 1.0), metal_fuzz=0.05,
                    optical_density=3, simple_material=True),
        ObjectBlock(procgen.box(), "metal", position=(1.2, -1.5, 0.5), rotation=(0, 0, 15), albedo=(1, 0.1, 0.1), metal_fuzz=0.5,
                    optical_density=3, simple_material=True),
        ObjectBlock(procgen.plane1(), "light", position=(0, -10, 0), scale=(5, 0, 5), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(0, 0, 10), rotation=(90, 0, 0), scale=(5, 0, 5), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "lambertian", rotation=(90, 0, 0), scale=(10, 0, 10), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def cat(width=600, height=600, n_samples=32, detail=1.0, tex_size=512):
    """assets/scene/scene_cat.txt:6-38 with the cat replaced by the shuttle stand-in fitted into a box the camera sees.
    The only light is a plane scaled by 1e4 on every axis, tilted 60 degrees about z."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=45, gamma=1.0, russian_roulette=0.9,
                       background=(0.1, 0.1, 0.1), camera_pos=(3, 0, 3), camera_dir=(-1, 0, 0), camera_up=(0, 0, 1))
    model = procgen.fitted(procgen.shuttle(detail=detail, tex_size=tex_size), (-14, -14, 32), (14, 14, 66))
    blocks = [
        ObjectBlock(model, "lambertian", position=(-3, 0.0, -2), rotation=(0, 0, 45), scale=(0.1, 0.1, 0.1), albedo=(1, 1, 1), metal_fuzz=0.01,
                    optical_density=3),
        ObjectBlock(procgen.plane1(), "light", position=(5, 4, 0), rotation=(0, 0, -60), scale=(1e4, 1e4, 1e4), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def house(width=600, height=600, n_samples=32, detail=1.0, tex_size=512):
    """assets/scene/scene_house.txt:6-46 with the bamboo house replaced by the shuttle stand-in (y up, as the file's
    `rotation 90 36 0` expects) between a light plane 100 above and a white plane 100 below."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=45, gamma=1.0, russian_roulette=0.9,
                       background=(0.05, 0.05, 0.05), camera_pos=(10, 0, 0), camera_dir=(-1, 0, 0.1), camera_up=(0, 0, 1))
    model = procgen.fitted(procgen.shuttle(detail=detail, tex_size=tex_size), (-6, 0, -6), (6, 7, 6))
    blocks = [
        ObjectBlock(model, "lambertian", position=(0, -2, -0.2), rotation=(90, 36, 0), scale=(0.4, 0.4, 0.4), albedo=(1, 1, 1), metal_fuzz=0.2,
                    optical_density=3),
        ObjectBlock(procgen.plane1(), "light", position=(0, 0, 100), rotation=(90, 0, 0), scale=(1e6, 1e6, 1e6), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "lambertian", position=(0, 0, -100), rotation=(90, 0, 0), scale=(1e6, 1e6, 1e6), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def room(width=1200, height=800, n_samples=512, detail=1.0, tex_size=256):
    """assets/scene/scene_room.txt:6-38: a y-up interior at ray depth 16, the camera inside the model, one ceiling
    light; the room model replaced by procgen.room."""
    cfg = RenderConfig(width, height, n_samples, max_depth=16, fov_deg=60, gamma=1.0, russian_roulette=0.9,
                       background=(0.05, 0.05, 0.05), camera_pos=(600, 100, -900), camera_dir=(1, 0, 1), camera_up=(0, 1, 0))
    blocks = [
        ObjectBlock(procgen.room(detail=detail, tex_size=tex_size), "lambertian", albedo=(1, 1, 1), metal_fuzz=0.2, optical_density=3),
        ObjectBlock(procgen.plane1(), "light", position=(693, 290, -820), scale=(164, 0, 160), albedo=(1, 1, 1)),
    ]
    return assemble(cfg, blocks)


def stress(width=3840, height=2160, n_samples=1024, grid=(8, 8), tris_per_instance=470_000):
    """Many rotated/translated instances of a bumpy blob on a ground plane under one large light."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=55, gamma=1.8, russian_roulette=0.9,
                       background=(0.02, 0.02, 0.03), camera_pos=(-60, -75, 55), camera_dir=(0.6, 0.75, -0.45), camera_up=(0, 0, 1))
    types = ["lambertian", "metal", "refractive", "reflective"]
    blocks = []
    k = 0
    models = {}              # seven distinct meshes, instanced
    for gy in range(grid[1]):
        for gx in range(grid[0]):
            seed = 1 + (k % 7)
            if seed not in models:
                models[seed] = procgen.blob(tris_per_instance, seed=seed)
            m = models[seed]
            t = types[k % 4]
            blocks.append(ObjectBlock(m, t, position=((gx - grid[0] / 2 + 0.5) * 16.0, (gy - grid[1] / 2 + 0.5) * 16.0, 6.5),
                                      rotation=(17.0 * k % 360, 31.0 * k % 360, 7.0 * k % 360), scale=(6, 6, 6),
                                      albedo=(0.35 + 0.6 * ((k * 37) % 10) / 10, 0.35 + 0.6 * ((k * 53) % 10) / 10, 0.35 + 0.6 * ((k * 71) % 10) / 10),
                                      metal_fuzz=0.1, optical_density=1.5, simple_material=True))
            k += 1
    blocks.append(ObjectBlock(procgen.plane1(), "lambertian", rotation=(90, 0, 0), scale=(1e4, 1e4, 1e4), albedo=(0.8, 0.8, 0.8)))
    blocks.append(ObjectBlock(procgen.plane1(), "light", position=(0, 0, 120), rotation=(90, 0, 0), scale=(90, 0, 90), albedo=(1, 1, 1)))
    return assemble(cfg, blocks)


def material_box(width=64, height=40, n_samples=16, tex_size=64):
    """Small closed test scene that exercises every material type, a texture, a light and the background."""
    cfg = RenderConfig(width, height, n_samples, max_depth=8, fov_deg=60, gamma=1.8, russian_roulette=0.9,
                       background=(0.05, 0.06, 0.08), camera_pos=(-9, 0.5, 3.2), camera_dir=(1, -0.05, -0.2), camera_up=(0, 0, 1))
    tex_model = procgen.plane1()
    tex_model.mtls = [procgen.mtl("floor", 1, Kd=(1, 1, 1), map_Kd=procgen.texture("checker", tex_size, 5))]
    tex_model.shapes[0].mtl[:] = 0
    ball = procgen.Model([procgen._grid_surface("ball", procgen._ellipsoid((0, 0, 0), (1, 1, 1)), 24, 16, -1)])
    bumpy = procgen.Model([procgen._grid_surface("bumpy", procgen._ellipsoid((0, 0, 0), (1, 1, 1), bumps=0.1), 28, 20, -1)])
    blocks = [
        ObjectBlock(tex_model, "lambertian", rotation=(90, 0, 0), scale=(12, 12, 12), albedo=(1, 1, 1)),
        ObjectBlock(procgen.plane1(), "light", position=(0, 0, 8), rotation=(90, 0, 0), scale=(4, 0, 4), albedo=(1, 1, 1)),
        ObjectBlock(ball, "refractive", position=(0, -2.6, 1.5), scale=(1.5, 1.5, 1.5), albedo=(0.95, 0.95, 1.0), optical_density=1.5, simple_material=True),
        ObjectBlock(bumpy, "metal", position=(1, 1.2, 1.6), scale=(1.5, 1.5, 1.5), albedo=(0.9, 0.6, 0.2), metal_fuzz=0.15, simple_material=True),
        ObjectBlock(procgen.box(), "reflective", position=(3.5, -0.5, 1.5), rotation=(0, 0, 30), scale=(1.2, 4, 3), albedo=(0.9, 0.9, 0.9), simple_material=True),
        ObjectBlock(procgen.box(), "lambertian", position=(-1, 3.6, 1.0), rotation=(0, 0, -20), scale=(2, 2, 2), albedo=(0.2, 0.7, 0.3), simple_material=True),
        ObjectBlock(procgen.box(), "lambertian", position=(2, 0, 4.5), scale=(30, 30, 18), albedo=(0.7, 0.7, 0.75), simple_material=True),
    ]
    return assemble(cfg, blocks)
