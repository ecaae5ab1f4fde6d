"""ctypes binding of librtcore.so (the C ABI declared in include/rtcore.h).

This module is plumbing for tests and the benchmark: it loads the in-tree
shared library and exposes each entry point with its argument types.  There is
no fallback: if the library is missing, loading raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "librtcore.so")     # RT_LIB: another build of the same library (A/B runs)

# numpy views of the upstream host layouts (include/rtcore.h)
VERTEX_DTYPE = np.dtype(
    {"names": ["position", "normal", "texcoord"],
     "formats": [("<f4", 3), ("<f4", 3), ("<f4", 2)],
     "offsets": [0, 12, 24], "itemsize": 32})
TRIANGLE_DTYPE = np.dtype(
    {"names": ["material_id", "v"], "formats": ["<i4", (VERTEX_DTYPE, 3)],
     "offsets": [0, 8], "itemsize": 104})
NODE_DTYPE = np.dtype(
    {"names": ["is_leaf", "box_min", "box_max", "left", "right"],
     "formats": ["u1", ("<f4", 3), ("<f4", 3), "<i4", "<i4"],
     "offsets": [0, 4, 16, 28, 32], "itemsize": 36})
MATERIAL_DTYPE = np.dtype(
    {"names": ["type", "albedo", "optical_density", "metal_fuzz", "texture", "pad"],
     "formats": ["<i4", ("<f4", 3), "<f4", "<f4", "<i4", "<i4"],
     "offsets": [0, 4, 16, 20, 24, 28], "itemsize": 32})

MAT_NONE, MAT_LIGHT, MAT_LAMBERTIAN, MAT_METAL, MAT_REFRACTIVE, MAT_REFLECTIVE = range(6)

CNT_RAYS, CNT_BOX_TESTS, CNT_TRI_TESTS, CNT_SHADED, CNT_TEX_FETCHES, CNT_PATHS, CNT_MAX_STACK, CNT_WAVE_NODE_TRIPS, CNT_WAVE_LEAF_TRIPS, CNT_NODE_VISITS, CNT_HOT_VISITS, CNT_CYC_REFILL, CNT_CYC_NODE, CNT_CYC_LEAF, CNT_CYC_WAVE, CNT_WAVE_SPAN_TICKS, CNT_WAVES, CNT_MAX_RAY_VISITS, CNT_NONFINITE_RAYS, CNT_LAUNCH_SLOT_TICKS, CNT_TAIL_IDLE_TICKS, CNT_CYC_CLAIM, CNT_CLAIMS, CNT_WALKS = range(24)
CNT_COUNT = 24
OPT_COUNTERS, OPT_XCD_REMAP, OPT_CHUNK, OPT_PASS, OPT_LDS_SLOTS, OPT_MODE, OPT_WF_CAPACITY_LOG2, OPT_WF_RUN, OPT_WF_BURST = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_TIMING = 11
OPT_WF_SORT = 10
OPT_HOT_NODES = 12
OPT_WF_VOTE = 13
OPT_WF_REFILL = 14
OPT_WF_BLOCK = 15
OPT_WF_LANES = 16
OPT_GPU_BVH = 17
OPT_HOT_TUNE = 18
OPT_SHARE_PRIMARY = 19
OPT_SHARE_EQUAL = 20
OPT_SLOT_RING_MB = 21
OPT_GROUP_SAMPLES = 22
MODE_MEGAKERNEL, MODE_WAVEFRONT = 0, 1
STATIC_SCRATCH_BYTES = 64 << 20
GROUP_ID_BYTES = 128


class RtObject(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("bvh", C.c_void_p), ("n_triangles", C.c_int32), ("n_bvh_nodes", C.c_int32)]


class RtTexture(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class RtTransform(C.Structure):
    _fields_ = [("scale", C.c_float * 3), ("translation", C.c_float * 3), ("rotation", C.c_float * 3)]


EXPORTS = [
    "rt_last_error", "rt_abi_version", "rt_build_bvh", "rt_apply_transform", "rt_create", "rt_destroy", "rt_set_option",
    "rt_upload_scene", "rt_init_randstate", "rt_setup_raytrace", "rt_raytrace", "rt_raytrace_start", "rt_raytrace_wait", "rt_raytrace_static", "rt_post_process", "rt_primary_rays",
    "rt_closest_hit", "rt_get_counters", "rt_reset_counters", "rt_get_timing", "rt_scene_stats", "rt_check_node_numbering", "rt_device_alloc", "rt_device_free",
    "rt_device_zero", "rt_copy_to_host", "rt_copy_to_device", "rt_synchronize",
    "rt_get_device", "rt_partition_samples", "rt_group_unique_id", "rt_group_create", "rt_group_destroy", "rt_reduce_to_root",
]

_lib = None


def load():
    """Load librtcore.so once and declare prototypes.  Raises OSError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is not built: run `make` (or __graft_entry__.build()) first")
    # PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  A process that will also use torch
    # (for device memory, streams, RCCL) must bind ONE HIP runtime: import torch first so that librtcore's
    # NEEDED libamdhip64.so.7 resolves to the copy torch already loaded.  Loading in the other order leaves the
    # process with a runtime that reports no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    fp = C.POINTER(C.c_float)
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = []
    lib.rt_abi_version.restype = i32
    protos = {
        "rt_build_bvh": [vp, i32, vp, i32],
        "rt_apply_transform": [C.POINTER(RtTransform), vp, i32],
        "rt_create": [i32, C.POINTER(vp)],
        "rt_destroy": [vp],
        "rt_set_option": [vp, i32, i32],
        "rt_upload_scene": [vp, C.POINTER(RtObject), i32, vp, i32, C.POINTER(RtTexture), i32],
        "rt_init_randstate": [vp, u64, i32, i32],
        "rt_setup_raytrace": [vp, i32, i32, fp, fp, fp, f32],
        "rt_raytrace": [vp, i32, i32, i32, i32, i32, fp, f32, vp, vp],
        "rt_raytrace_static": [vp, i32, i32, i32, i32, i32, fp, f32, vp, vp],
        "rt_raytrace_start": [vp, i32, i32, i32, i32, i32, fp, f32, vp, vp],
        "rt_raytrace_wait": [vp],
        "rt_post_process": [vp, i32, i32, f32, vp, vp],
        "rt_primary_rays": [vp, u32, vp, vp],
        "rt_closest_hit": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
        "rt_get_counters": [vp, C.POINTER(u64)],
        "rt_reset_counters": [vp],
        "rt_get_timing": [vp, C.POINTER(C.c_double)],
        "rt_scene_stats": [vp, C.POINTER(u64)],
        "rt_check_node_numbering": [vp, C.POINTER(u64)],
        "rt_device_alloc": [vp, u64, C.POINTER(vp)],
        "rt_device_free": [vp, vp],
        "rt_device_zero": [vp, vp, u64, vp],
        "rt_copy_to_host": [vp, vp, vp, u64],
        "rt_copy_to_device": [vp, vp, vp, u64],
        "rt_synchronize": [vp],
        "rt_get_device": [vp, C.POINTER(i32)],
        "rt_partition_samples": [i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)],
        "rt_group_unique_id": [vp],
        "rt_group_create": [C.POINTER(vp), i32, i32, i32, vp, C.POINTER(vp)],
        "rt_group_destroy": [vp],
        "rt_reduce_to_root": [vp, C.POINTER(vp), u64, C.POINTER(vp)],
    }
    for name, args in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = lib
    return lib


class RtError(RuntimeError):
    pass


def check(rc):
    if rc < 0:
        raise RtError(f"rtcore error {rc}: {load().rt_last_error().decode()}")
    return rc


def f3(values):
    return (C.c_float * 3)(*[float(v) for v in values])


def build_bvh(tris):
    """rt_build_bvh on a TRIANGLE_DTYPE array -> NODE_DTYPE array (post-order, root last)."""
    lib = load()
    tris = np.ascontiguousarray(tris, dtype=TRIANGLE_DTYPE)
    nodes = np.zeros(2 * len(tris) + 1, dtype=NODE_DTYPE)
    n = check(lib.rt_build_bvh(tris.ctypes.data, len(tris), nodes.ctypes.data, len(nodes)))
    return nodes[:n].copy()


def apply_transform(tris, scale, translation, rotation_deg):
    """rt_apply_transform in place on a TRIANGLE_DTYPE array."""
    lib = load()
    assert tris.dtype == TRIANGLE_DTYPE and tris.flags.c_contiguous
    xf = RtTransform(f3(scale), f3(translation), f3(rotation_deg))
    check(lib.rt_apply_transform(C.byref(xf), tris.ctypes.data, len(tris)))
    return tris


class Context:
    """One rt_context on one device.  Device buffers are passed as raw addresses (ints)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.rt_create(device, C.byref(h)))
        self.h = h
        self._keep = None

    def close(self):
        if self.h:
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        check(self.lib.rt_set_option(self.h, opt, value))

    def device(self):
        d = C.c_int()
        check(self.lib.rt_get_device(self.h, C.byref(d)))
        return d.value

    def upload_scene(self, objects, materials, textures=()):
        """objects: list of (triangles TRIANGLE_DTYPE, nodes NODE_DTYPE); materials: MATERIAL_DTYPE array;
        textures: list of uint8 arrays of shape (h, w, 4)."""
        objs = (RtObject * len(objects))()
        keep = []
        for i, (t, n) in enumerate(objects):
            t = np.ascontiguousarray(t, dtype=TRIANGLE_DTYPE)
            if n is None:                                   # only with OPT_GPU_BVH: the tree is built on the device
                keep.append(t)
                objs[i] = RtObject(t.ctypes.data, None, len(t), 0)
                continue
            n = np.ascontiguousarray(n, dtype=NODE_DTYPE)
            keep += [t, n]
            objs[i] = RtObject(t.ctypes.data, n.ctypes.data, len(t), len(n))
        mats = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        texs = (RtTexture * max(1, len(textures)))()
        for i, tx in enumerate(textures):
            tx = np.ascontiguousarray(tx, dtype=np.uint8)
            assert tx.ndim == 3 and tx.shape[2] == 4
            keep.append(tx)
            texs[i] = RtTexture(tx.ctypes.data, tx.shape[1], tx.shape[0])
        check(self.lib.rt_upload_scene(self.h, objs, len(objects), mats.ctypes.data, len(mats), texs, len(textures)))

    def init_randstate(self, seed, width, height):
        check(self.lib.rt_init_randstate(self.h, seed, width, height))

    def setup_raytrace(self, width, height, pos, direction, up, fov):
        check(self.lib.rt_setup_raytrace(self.h, width, height, f3(pos), f3(direction), f3(up), fov))

    def raytrace(self, n_pixel, first_sample, spp, semantic_batch, max_depth, ambient, rr, d_out, stream=0):
        check(self.lib.rt_raytrace(self.h, n_pixel, first_sample, spp, semantic_batch, max_depth, f3(ambient), rr, d_out, stream))

    def raytrace_start(self, n_pixel, first_sample, spp, semantic_batch, max_depth, ambient, rr, d_out, stream=0):
        """rt_raytrace handed to a thread of the library; returns at once.  The context belongs to the call until raytrace_wait."""
        check(self.lib.rt_raytrace_start(self.h, n_pixel, first_sample, spp, semantic_batch, max_depth, f3(ambient), rr, d_out, stream))

    def raytrace_wait(self):
        check(self.lib.rt_raytrace_wait(self.h))

    def raytrace_static(self, n_pixel, first_sample, spp, semantic_batch, max_depth, ambient, rr, d_out, stream=0):
        """rt_raytrace without any allocation: sample slots in the library's static device array."""
        check(self.lib.rt_raytrace_static(self.h, n_pixel, first_sample, spp, semantic_batch, max_depth, f3(ambient), rr, d_out, stream))

    def post_process(self, n_pixels, spp, gamma, d_out, stream=0):
        check(self.lib.rt_post_process(self.h, n_pixels, spp, gamma, d_out, stream))

    def primary_rays(self, batch_index, d_rays, stream=0):
        check(self.lib.rt_primary_rays(self.h, batch_index, d_rays, stream))

    def closest_hit(self, n_rays, d_rays, d_obj, d_tri, d_t, d_u=0, d_v=0, stream=0):
        check(self.lib.rt_closest_hit(self.h, n_rays, d_rays, d_obj, d_tri, d_t, d_u, d_v, stream))

    def counters(self):
        out = (C.c_uint64 * CNT_COUNT)()
        check(self.lib.rt_get_counters(self.h, out))
        return list(out)

    def timing(self):
        out = (C.c_double * 8)()
        check(self.lib.rt_get_timing(self.h, out))
        return {"extend_ms": out[0], "extend_launches": int(out[1]), "shade_ms": out[2], "shade_launches": int(out[3]),
                "path_ms": out[4], "path_launches": int(out[5])}

    def reset_counters(self):
        check(self.lib.rt_reset_counters(self.h))

    def scene_stats(self):
        out = (C.c_uint64 * 8)()
        check(self.lib.rt_scene_stats(self.h, out))
        return {"triangles": out[0], "inner_nodes": out[1], "stack_slots": out[2], "device_bytes": out[3], "objects": out[4],
                "gpu_build_ms": out[5] / 1000.0, "gpu_copy_ms": out[6] / 1000.0, "gpu_binary_levels": out[7]}

    def check_node_numbering(self):
        out = (C.c_uint64 * 8)()
        check(self.lib.rt_check_node_numbering(self.h, out))
        return {"records": out[0], "candidates": out[1], "refs_into_candidates_from_behind": out[2], "refs_out_of_range": out[3],
                "refs_not_ascending": out[4]}

    def synchronize(self):
        check(self.lib.rt_synchronize(self.h))

    # raw device-memory helpers (for callers without torch)
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.rt_device_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self.lib.rt_device_free(self.h, ptr))

    def zero(self, ptr, nbytes, stream=0):
        check(self.lib.rt_device_zero(self.h, ptr, nbytes, stream))

    def to_host(self, ptr, array):
        check(self.lib.rt_copy_to_host(self.h, array.ctypes.data, ptr, array.nbytes))
        return array

    def to_device(self, ptr, array):
        array = np.ascontiguousarray(array)
        check(self.lib.rt_copy_to_device(self.h, ptr, array.ctypes.data, array.nbytes))


def partition_samples(n_samples, semantic_batch, parts, part):
    """rt_partition_samples -> (first_sample, count) of `part`."""
    first, count = C.c_int(), C.c_int()
    check(load().rt_partition_samples(n_samples, semantic_batch, parts, part, C.byref(first), C.byref(count)))
    return first.value, count.value


def group_unique_id():
    """rt_group_unique_id -> 128 bytes to hand to every process of the world."""
    buf = (C.c_uint8 * GROUP_ID_BYTES)()
    check(load().rt_group_unique_id(buf))
    return bytes(buf)


class Group:
    """rt_group over the contexts this process drives (RCCL communicator over their distinct devices)."""

    def __init__(self, contexts, first_rank=0, world=None, unique_id=None):
        self.lib = load()
        self.n = len(contexts)
        arr = (C.c_void_p * self.n)(*[c.h for c in contexts])
        if world is None:
            world = len({c.device() for c in contexts})
        idbuf = None
        if unique_id is not None:
            assert len(unique_id) == GROUP_ID_BYTES
            idbuf = (C.c_uint8 * GROUP_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(self.lib.rt_group_create(arr, self.n, first_rank, world, idbuf, C.byref(h)))
        self.h = h

    def reduce_to_root(self, d_buffers, count, streams=None):
        bufs = (C.c_void_p * self.n)(*d_buffers)
        st = (C.c_void_p * self.n)(*streams) if streams is not None else None
        check(self.lib.rt_reduce_to_root(self.h, bufs, count, st))

    def close(self):
        if self.h:
            self.lib.rt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
