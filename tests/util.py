"""Shared helpers for the parity tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_f32(a, b):
    """Bit-exact float32 equality (NaN payloads and signed zeros included)."""
    return np.array_equal(bits(a), bits(b))


def same_f32_or_both_nan(a, b):
    """Bit-exact float32 equality, except that any NaN equals any NaN.  Sign and payload of a NaN that comes out of
    arithmetic on NaN operands follow the operand order of the machine instruction, which a compiler is free to
    swap for commutative operations -- upstream's compiler included -- so they are not part of any contract."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def same_triangles(a, b):
    return (np.array_equal(a["material_id"], b["material_id"])
            and all(same_f32(a["v"][f], b["v"][f]) for f in ("position", "normal", "texcoord")))


def same_nodes(a, b):
    """Node-for-node equality.  Box bounds are compared as numbers: fminf/fmaxf may return either zero
    for (+0, -0), which upstream leaves to its host compiler and which no slab test can tell apart."""
    return (len(a) == len(b) and np.array_equal(a["is_leaf"] != 0, b["is_leaf"] != 0) and np.array_equal(a["left"], b["left"])
            and np.array_equal(a["right"], b["right"]) and np.array_equal(a["box_min"], b["box_min"])
            and np.array_equal(a["box_max"], b["box_max"]))


def random_rays(rng, n, lo, hi, axis_aligned_fraction=0.1):
    """Rays with origins in a box and directions toward points in the box; some exactly axis-aligned."""
    o = rng.uniform(lo, hi, size=(n, 3))
    tgt = rng.uniform(lo, hi, size=(n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = int(n * axis_aligned_fraction)
    for i in range(k):
        a = i % 3
        d[i] = 0.0
        d[i, a] = 1.0 if (i // 3) % 2 == 0 else -1.0
    return np.concatenate([o, d], axis=1).astype(np.float32)
