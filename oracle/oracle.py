"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  PARITY UNPINNED:
see the header of rt_oracle.cpp.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

# same host layouts as the product ABI (they are upstream's layouts)
VERTEX_DTYPE = np.dtype({"names": ["position", "normal", "texcoord"], "formats": [("<f4", 3), ("<f4", 3), ("<f4", 2)],
                         "offsets": [0, 12, 24], "itemsize": 32})
TRIANGLE_DTYPE = np.dtype({"names": ["material_id", "v"], "formats": ["<i4", (VERTEX_DTYPE, 3)], "offsets": [0, 8], "itemsize": 104})
NODE_DTYPE = np.dtype({"names": ["is_leaf", "box_min", "box_max", "left", "right"],
                       "formats": ["u1", ("<f4", 3), ("<f4", 3), "<i4", "<i4"], "offsets": [0, 4, 16, 28, 32], "itemsize": 36})
MATERIAL_DTYPE = np.dtype({"names": ["type", "albedo", "optical_density", "metal_fuzz", "texture", "pad"],
                           "formats": ["<i4", ("<f4", 3), "<f4", "<f4", "<i4", "<i4"], "offsets": [0, 4, 16, 20, 24, 28], "itemsize": 32})


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32), ("semantic_batch", C.c_int32),
                ("fov", C.c_float), ("russian_roulette", C.c_float), ("background", C.c_float * 3),
                ("camera_pos", C.c_float * 3), ("camera_dir", C.c_float * 3), ("camera_up", C.c_float * 3),
                ("seed", C.c_uint64)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is not built: run `make -C oracle`")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.orc_sizeof.argtypes = [C.c_int]
    lib.orc_build_bvh.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.orc_apply_transform.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.orc_apply_transform.restype = None
    lib.orc_philox4x32_10.argtypes = [vp, vp, vp]
    lib.orc_philox4x32_10.restype = None
    lib.orc_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    lib.orc_uniforms.restype = None
    lib.orc_scene_create.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    lib.orc_scene_create.restype = vp
    lib.orc_scene_destroy.argtypes = [vp]
    lib.orc_scene_destroy.restype = None
    lib.orc_camera_setup.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_float, vp]
    lib.orc_camera_setup.restype = None
    lib.orc_primary_rays.argtypes = [C.POINTER(RenderParams), C.c_uint32, vp]
    lib.orc_primary_rays.restype = None
    lib.orc_closest_hit.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.orc_closest_hit.restype = None
    lib.orc_closest_hit_brute.argtypes = [vp, C.c_int, vp, vp]
    lib.orc_closest_hit_brute.restype = None
    lib.orc_render.argtypes = [vp, C.POINTER(RenderParams), C.c_int, C.c_int, vp, vp]
    lib.orc_render.restype = None
    lib.orc_post_process.argtypes = [C.c_int, C.c_int, C.c_float, vp]
    lib.orc_post_process.restype = None
    lib.orc_tex2d.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.orc_tex2d.restype = None
    lib.orc_scatter.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    lib.orc_num_threads.restype = C.c_int
    assert lib.orc_sizeof(0) == 104 and lib.orc_sizeof(1) == 36 and lib.orc_sizeof(2) == 32 and lib.orc_sizeof(3) == 24
    assert lib.orc_sizeof(4) == C.sizeof(RenderParams)
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build_bvh(tris):
    lib = load()
    tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE)
    nodes = np.zeros(2 * len(tris) + 1, dtype=NODE_DTYPE)
    n = lib.orc_build_bvh(tris.ctypes.data, len(tris), nodes.ctypes.data, len(nodes))
    assert n > 0
    return nodes[:n].copy()


def apply_transform(tris, scale, translation, rotation_deg):
    lib = load()
    assert tris.dtype == TRIANGLE_DTYPE and tris.flags.c_contiguous
    s, t, r = _f32(scale), _f32(translation), _f32(rotation_deg)
    lib.orc_apply_transform(s.ctypes.data, t.ctypes.data, r.ctypes.data, tris.ctypes.data, len(tris))
    return tris


class OracleBackend:
    name = "oracle"
    apply_transform = staticmethod(apply_transform)
    build_bvh = staticmethod(build_bvh)


def philox(ctr, key):
    lib = load()
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib.orc_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def uniforms(seed, pixel, index, slot, stream):
    """The four (0,1] draws at one counter address."""
    out = np.zeros(4, dtype=np.float32)
    load().orc_uniforms(seed, pixel, index, slot, stream, out.ctypes.data)
    return out


def make_params(cfg, seed, semantic_batch):
    """cfg: raytrace_amd.scene.RenderConfig (duck-typed)."""
    p = RenderParams()
    p.width, p.height, p.max_depth, p.semantic_batch = cfg.width, cfg.height, cfg.max_depth, semantic_batch
    p.fov, p.russian_roulette = cfg.fov, cfg.russian_roulette
    p.background = (C.c_float * 3)(*cfg.background)
    p.camera_pos = (C.c_float * 3)(*cfg.camera_pos)
    p.camera_dir = (C.c_float * 3)(*cfg.camera_dir)
    p.camera_up = (C.c_float * 3)(*cfg.camera_up)
    p.seed = seed
    return p


class Scene:
    """Oracle scene over upstream-layout arrays: objects = [(triangles, nodes)], materials, textures."""

    def __init__(self, objects, materials, textures=()):
        self.lib = load()
        self._keep = []
        n = len(objects)
        tp = (C.c_void_p * n)()
        npn = (C.c_void_p * n)()
        nt = (C.c_int * n)()
        nn = (C.c_int * n)()
        for i, (t, nd) in enumerate(objects):
            t = np.ascontiguousarray(t, dtype=TRIANGLE_DTYPE)
            nd = np.ascontiguousarray(nd, dtype=NODE_DTYPE)
            self._keep += [t, nd]
            tp[i], npn[i], nt[i], nn[i] = t.ctypes.data, nd.ctypes.data, len(t), len(nd)
        mats = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self._keep.append(mats)
        k = max(1, len(textures))
        xp = (C.c_void_p * k)()
        xw = (C.c_int * k)()
        xh = (C.c_int * k)()
        for i, tx in enumerate(textures):
            tx = np.ascontiguousarray(tx, dtype=np.uint8)
            self._keep.append(tx)
            xp[i], xw[i], xh[i] = tx.ctypes.data, tx.shape[1], tx.shape[0]
        self.h = self.lib.orc_scene_create(n, tp, nt, npn, nn, len(mats), mats.ctypes.data, len(textures), xp, xw, xh)

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def closest_hit(self, rays):
        rays = _f32(rays).reshape(-1, 6)
        n = len(rays)
        obj = np.zeros(n, dtype=np.int32)
        tri = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        u = np.zeros(n, dtype=np.float32)
        v = np.zeros(n, dtype=np.float32)
        cnt = np.zeros(5, dtype=np.uint64)
        self.lib.orc_closest_hit(self.h, n, rays.ctypes.data, obj.ctypes.data, tri.ctypes.data, t.ctypes.data, u.ctypes.data,
                                 v.ctypes.data, cnt.ctypes.data)
        return obj, tri, t, u, v, cnt

    def closest_hit_brute(self, rays):
        rays = _f32(rays).reshape(-1, 6)
        t = np.zeros(len(rays), dtype=np.float32)
        self.lib.orc_closest_hit_brute(self.h, len(rays), rays.ctypes.data, t.ctypes.data)
        return t

    def render(self, params, first_sample, n_samples, accum=None):
        n = params.width * params.height
        if accum is None:
            accum = np.zeros((n, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == 3 * n
        cnt = np.zeros(5, dtype=np.uint64)
        self.lib.orc_render(self.h, C.byref(params), first_sample, n_samples, accum.ctypes.data, cnt.ctypes.data)
        return accum, cnt


def primary_rays(params, batch_index):
    rays = np.zeros((params.width * params.height, 6), dtype=np.float32)
    load().orc_primary_rays(C.byref(params), batch_index, rays.ctypes.data)
    return rays


def camera_setup(width, height, pos, direction, up, fov):
    out = np.zeros(12, dtype=np.float32)
    p, d, u = _f32(pos), _f32(direction), _f32(up)
    load().orc_camera_setup(width, height, p.ctypes.data, d.ctypes.data, u.ctypes.data, fov, out.ctypes.data)
    return out


def post_process(buf, spp, gamma):
    buf = np.ascontiguousarray(buf, dtype=np.float32).copy()
    load().orc_post_process(buf.size // 3, spp, gamma, buf.ctypes.data)
    return buf


def tex2d(rgba, uv):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    uv = _f32(uv).reshape(-1, 2)
    out = np.zeros((len(uv), 3), dtype=np.float32)
    load().orc_tex2d(rgba.ctypes.data, rgba.shape[1], rgba.shape[0], len(uv), uv.ctypes.data, out.ctypes.data)
    return out


def num_threads():
    return load().orc_num_threads()
