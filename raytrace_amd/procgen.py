"""Deterministic procedural stand-ins for the OBJ models the upstream scene files name.

The upstream scenes download their models (formula 1, Lu Yu, shuttle) from URLs
in comments; the models are not in the upstream tree and there is no network.
These generators build multi-shape meshes with the same structure an OBJ+MTL
pair has (shapes, per-face material ids, MTL records with illum/Kd/Ke/Ni/d/
map_Kd), sized like the real models, so the loader logic and the kernels see
the same kind of input.  Everything is evaluated in float64 and rounded once to
float32.
"""
import numpy as np


class Shape:
    """One OBJ shape: per-triangle vertex positions/normals/uvs and MTL ids."""

    def __init__(self, name, pos, nrm, uv, mtl):
        self.name = name
        self.pos = np.ascontiguousarray(pos, dtype=np.float32)   # (n, 3, 3)
        self.nrm = np.ascontiguousarray(nrm, dtype=np.float32)   # (n, 3, 3)
        self.uv = np.ascontiguousarray(uv, dtype=np.float32)     # (n, 3, 2)
        self.mtl = np.ascontiguousarray(mtl, dtype=np.int32)     # (n,)  index into Model.mtls, -1 = none
        self.has_normals = True
        self.has_uvs = True

    def __len__(self):
        return len(self.pos)


class Model:
    """One OBJ file: shapes + MTL records (dicts: name, illum, Kd, Ke, Ni, d, map_Kd)."""

    def __init__(self, shapes, mtls=()):
        self.shapes = list(shapes)
        self.mtls = list(mtls)

    @property
    def n_triangles(self):
        return sum(len(s) for s in self.shapes)


def mtl(name, illum, Kd=(0.8, 0.8, 0.8), Ke=(0, 0, 0), Ni=1.0, d=1.0, map_Kd=None):
    return {"name": name, "illum": illum, "Kd": tuple(Kd), "Ke": tuple(Ke), "Ni": float(Ni), "d": float(d), "map_Kd": map_Kd}


# ---------------------------------------------------------------------------
# the two meshes upstream does ship (assets/obj/simple): a unit plane and a unit box
# ---------------------------------------------------------------------------
def plane1():
    v = np.array([[1, 0, 1], [1, 0, -1], [-1, 0, -1], [-1, 0, 1]], dtype=np.float64)
    vt = np.array([[1, 1], [1, 0], [0, 0], [0, 1]], dtype=np.float64)
    faces = [(0, 1, 2), (0, 2, 3)]
    pos = np.array([[v[i] for i in f] for f in faces])
    uv = np.array([[vt[i] for i in f] for f in faces])
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), (2, 3, 1))
    return Model([Shape("plane", pos, nrm, uv, [-1, -1])])


def box():
    v = np.array([[-.5, -.5, -.5], [-.5, -.5, .5], [-.5, .5, -.5], [-.5, .5, .5],
                  [.5, -.5, -.5], [.5, -.5, .5], [.5, .5, -.5], [.5, .5, .5]], dtype=np.float64)
    vt = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], dtype=np.float64)
    vn = np.array([[0, 0, 1], [0, 0, -1], [0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0]], dtype=np.float64)
    # (vertex, texcoord, normal) triples, 1-based as in an OBJ face line
    f = [((1, 1, 6), (2, 2, 6), (3, 3, 6)), ((2, 2, 6), (4, 4, 6), (3, 3, 6)),
         ((8, 4, 1), (2, 2, 1), (6, 2, 1)), ((8, 4, 1), (4, 4, 1), (2, 2, 1)),
         ((6, 2, 5), (7, 3, 5), (8, 4, 5)), ((6, 2, 5), (5, 1, 5), (7, 3, 5)),
         ((1, 1, 2), (3, 3, 2), (7, 3, 2)), ((1, 1, 2), (7, 3, 2), (5, 1, 2)),
         ((1, 1, 4), (5, 1, 4), (2, 2, 4)), ((2, 2, 4), (5, 1, 4), (6, 2, 4)),
         ((3, 3, 3), (4, 4, 3), (7, 3, 3)), ((4, 4, 3), (8, 4, 3), (7, 3, 3))]
    pos = np.array([[v[a - 1] for a, _, _ in tri] for tri in f])
    uv = np.array([[vt[b - 1] for _, b, _ in tri] for tri in f])
    nrm = np.array([[vn[c - 1] for _, _, c in tri] for tri in f])
    return Model([Shape("box", pos, nrm, uv, [-1] * 12)])


# ---------------------------------------------------------------------------
# parametric surfaces
# ---------------------------------------------------------------------------
def _grid_surface(name, fn, nu, nv, mtl_id, flip=False):
    """Tessellate fn(u, v) -> (..., 3) over [0,1]^2 into 2*nu*nv triangles with smooth normals."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = fn(U, V)
    eps = 1e-5
    du = fn(np.clip(U + eps, 0, 1), V) - fn(np.clip(U - eps, 0, 1), V)
    dv = fn(U, np.clip(V + eps, 0, 1)) - fn(U, np.clip(V - eps, 0, 1))
    N = np.cross(du, dv)
    if flip:
        N = -N
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(ln > 1e-30, N / np.maximum(ln, 1e-30), np.array([0.0, 1.0, 0.0]))
    T = np.stack([U, V], axis=-1)

    def quad(A):
        a, b, c, d = A[:-1, :-1], A[1:, :-1], A[1:, 1:], A[:-1, 1:]
        t1 = np.stack([a, b, c], axis=2)
        t2 = np.stack([a, c, d], axis=2)
        return np.concatenate([t1.reshape(-1, 3, A.shape[-1]), t2.reshape(-1, 3, A.shape[-1])], axis=0)

    pos, nrm, uv = quad(P), quad(N), quad(T)
    # drop zero-area triangles (poles) so every triangle has a usable normal and determinant
    e1 = pos[:, 1] - pos[:, 0]
    e2 = pos[:, 2] - pos[:, 0]
    area = np.linalg.norm(np.cross(e1, e2), axis=-1)
    keep = area > 1e-12 * max(1.0, float(np.abs(pos).max()) ** 2)
    pos, nrm, uv = pos[keep], nrm[keep], uv[keep]
    return Shape(name, pos, nrm, uv, np.full(len(pos), mtl_id, dtype=np.int32))


def _ellipsoid(center, radii, bumps=0.0, freq=(7, 5), phase=0.0):
    cx, cy, cz = center
    rx, ry, rz = radii

    def fn(U, V):
        th = U * 2.0 * np.pi
        ph = V * np.pi
        r = 1.0 + bumps * (np.sin(freq[0] * th + phase) * np.sin(freq[1] * ph) + 0.5 * np.sin((2 * freq[0] + 1) * th) * np.sin((2 * freq[1] + 1) * ph + phase))
        return np.stack([cx + rx * r * np.sin(ph) * np.cos(th), cy + ry * r * np.cos(ph), cz + rz * r * np.sin(ph) * np.sin(th)], axis=-1)
    return fn


def _torus(center, axis, R, r):
    """Torus around `axis` (0=x, 1=y, 2=z)."""
    c = np.asarray(center, dtype=np.float64)

    def fn(U, V):
        a = U * 2.0 * np.pi
        b = V * 2.0 * np.pi
        ring = R + r * np.cos(b)
        p0 = r * np.sin(b)
        p1 = ring * np.cos(a)
        p2 = ring * np.sin(a)
        comps = [None, None, None]
        comps[axis] = p0
        comps[(axis + 1) % 3] = p1
        comps[(axis + 2) % 3] = p2
        return np.stack(comps, axis=-1) + c
    return fn


def _loft(z0, z1, half_w, half_h, mid_y, power=4.0):
    """Superellipse cross-section swept along z; half_w/half_h/mid_y are functions of s in [0,1]."""

    def fn(U, V):
        s = V
        z = z0 + (z1 - z0) * s
        a = U * 2.0 * np.pi
        ca, sa = np.cos(a), np.sin(a)
        e = 2.0 / power
        x = half_w(s) * np.sign(ca) * np.abs(ca) ** e
        y = mid_y(s) + half_h(s) * np.sign(sa) * np.abs(sa) ** e
        return np.stack([x, y, z], axis=-1)
    return fn


def _slab(center, size, tilt=0.0):
    """A rounded slab (thin superellipsoid) -- wings and planks."""
    cx, cy, cz = center
    sx, sy, sz = size

    def fn(U, V):
        th = U * 2.0 * np.pi
        ph = V * np.pi
        e = 0.35
        cth, sth, cph, sph = np.cos(th), np.sin(th), np.cos(ph), np.sin(ph)
        f = lambda t: np.sign(t) * np.abs(t) ** e
        x = sx * f(sph) * f(cth)
        y = sy * f(cph)
        z = sz * f(sph) * f(sth)
        ct, st = np.cos(tilt), np.sin(tilt)
        return np.stack([cx + x, cy + y * ct - z * st, cz + y * st + z * ct], axis=-1)
    return fn


# ---------------------------------------------------------------------------
# procedural RGBA8 textures
# ---------------------------------------------------------------------------
def texture(kind, size=1024, seed=1):
    i = np.arange(size, dtype=np.int64)
    X, Y = np.meshgrid(i, i, indexing="xy")
    rgba = np.zeros((size, size, 4), dtype=np.uint8)
    rgba[..., 3] = 255
    if kind == "stripes":
        band = ((X * 8) // size + (Y * 3) // size) % 2
        rgba[..., 0] = np.where(band == 0, 220, 250)
        rgba[..., 1] = np.where(band == 0, 30, 240)
        rgba[..., 2] = np.where(band == 0, 25, 235)
    elif kind == "checker":
        c = ((X * 16) // size + (Y * 16) // size) % 2
        rgba[..., 0] = np.where(c == 0, 40, 230)
        rgba[..., 1] = np.where(c == 0, 40, 230)
        rgba[..., 2] = np.where(c == 0, 45, 60)
    elif kind == "carbon":
        w = ((X // 4) + (Y // 4)) % 2
        h = (X * 2654435761 + Y * 40503 + seed * 97) % 64
        base = np.where(w == 0, 35, 60) + h // 4
        rgba[..., 0] = base
        rgba[..., 1] = base
        rgba[..., 2] = base + 8
    else:   # "noise": hashed value noise, smooth at 32-texel scale
        g = 32
        hx, hy = X // g, Y // g
        fx, fy = (X % g) / g, (Y % g) / g

        def hsh(a, b, c):
            v = (a * 374761393 + b * 668265263 + c * 2147483647 + seed * 144665) & 0xFFFFFFFF
            v = ((v ^ (v >> 13)) * 1274126177) & 0xFFFFFFFF
            return ((v ^ (v >> 16)) & 0xFF).astype(np.float64)
        for ch in range(3):
            n00, n10 = hsh(hx, hy, ch), hsh(hx + 1, hy, ch)
            n01, n11 = hsh(hx, hy + 1, ch), hsh(hx + 1, hy + 1, ch)
            val = (n00 * (1 - fx) + n10 * fx) * (1 - fy) + (n01 * (1 - fx) + n11 * fx) * fy
            rgba[..., ch] = np.clip(val * 0.6 + 90, 0, 255).astype(np.uint8)
    return rgba


# ---------------------------------------------------------------------------
# model stand-ins
# ---------------------------------------------------------------------------
def statue(n_tris=200_000):
    """Lu Yu stand-in: one bumpy single-shape figure, z up, ~260 units tall (scene_luyu applies no rotation)."""
    n = max(8, int(round((n_tris / 2.0) ** 0.5)))

    def fn(U, V):
        th = U * 2.0 * np.pi
        ph = V * np.pi
        # a body that narrows to a neck and a head: radius profile along the axis
        s = 1.0 - V
        prof = 0.55 + 0.45 * np.sin(np.pi * np.clip(s * 1.15, 0, 1)) - 0.25 * np.exp(-((s - 0.78) / 0.05) ** 2) + 0.18 * np.exp(-((s - 0.9) / 0.07) ** 2)
        r = prof * (1.0 + 0.06 * np.sin(9 * th) * np.sin(7 * ph) + 0.03 * np.sin(23 * th + 1.3) * np.sin(17 * ph))
        x = 62.0 * r * np.sin(ph) * np.cos(th)
        y = 55.0 * r * np.sin(ph) * np.sin(th)
        z = 130.0 + 130.0 * np.cos(ph)
        return np.stack([x, y, z], axis=-1)
    return Model([_grid_surface("statue", fn, n, n, -1)])


def formula1(detail=1.0, with_textures=True, tex_size=1024):
    """Formula-1 stand-in: ~25 shapes, 10 MTL materials (illum 1/2/3/4/5, four with map_Kd), ~600k triangles at detail=1.
    Model space is y up, length along z, as the upstream scene's `rotation 90 100 0` expects."""
    def d(n):
        return max(4, int(round(n * detail ** 0.5)))
    T = (lambda k, s: texture(k, tex_size, s)) if with_textures else (lambda k, s: None)
    mtls = [
        mtl("paint", 2, Kd=(0.85, 0.08, 0.06), map_Kd=T("stripes", 1)),
        mtl("carbon", 1, Kd=(0.08, 0.08, 0.09), map_Kd=T("carbon", 2)),
        mtl("tyre", 1, Kd=(0.04, 0.04, 0.04)),
        mtl("rim", 2, Kd=(0.75, 0.75, 0.78)),
        mtl("glass", 4, Kd=(0.9, 0.95, 1.0), Ni=1.5, d=0.25),
        mtl("chrome", 5, Kd=(0.95, 0.95, 0.95)),
        mtl("helmet", 1, Kd=(0.9, 0.8, 0.1), map_Kd=T("checker", 3)),
        mtl("wing", 2, Kd=(0.9, 0.9, 0.9), map_Kd=T("noise", 4)),
        mtl("rainlight", 3, Ke=(1.0, 0.15, 0.1)),
        mtl("plank", 1, Kd=(0.35, 0.25, 0.15)),
    ]
    PAINT, CARBON, TYRE, RIM, GLASS, CHROME, HELMET, WING, RAIN, PLANK = range(10)
    shapes = []
    # monocoque: superellipse loft
    hw = lambda s: 18.0 + 42.0 * np.sin(np.pi * np.clip(s, 0, 1)) ** 1.5
    hh = lambda s: 14.0 + 22.0 * np.sin(np.pi * np.clip(s * 0.9 + 0.05, 0, 1))
    my = lambda s: 34.0 + 10.0 * np.sin(np.pi * s)
    shapes.append(_grid_surface("monocoque", _loft(-190.0, 140.0, hw, hh, my), d(360), d(300), PAINT))
    # nose
    shapes.append(_grid_surface("nose", _loft(138.0, 235.0, lambda s: 20.0 * (1 - s) ** 0.8 + 3.0, lambda s: 15.0 * (1 - s) + 4.0, lambda s: 34.0 - 12.0 * s), d(120), d(90), PAINT))
    # cockpit canopy
    shapes.append(_grid_surface("canopy", _ellipsoid((0, 72, 5), (24, 20, 52)), d(110), d(70), GLASS))
    shapes.append(_grid_surface("helmet", _ellipsoid((0, 70, 0), (11, 12, 12)), d(60), d(40), HELMET))
    # wheels
    for k, (sx, sz) in enumerate([(-1, 1), (1, 1), (-1, -1), (1, -1)]):
        c = (sx * 82.0, 33.0, 150.0 if sz > 0 else -150.0)
        shapes.append(_grid_surface(f"tyre{k}", _torus(c, 0, 21.0, 12.0), d(150), d(70), TYRE))
        shapes.append(_grid_surface(f"rim{k}", _ellipsoid(c, (9.0, 17.0, 17.0), bumps=0.04, freq=(5, 3)), d(70), d(40), RIM))
        shapes.append(_grid_surface(f"arm{k}", _slab((sx * 50.0, 36.0, c[2]), (34.0, 1.6, 5.0)), d(40), d(24), CARBON))
    # front wing (3 elements) and rear wing (3 elements)
    for k in range(3):
        shapes.append(_grid_surface(f"fwing{k}", _slab((0, 12.0 + 6.0 * k, 222.0 - 9.0 * k), (96.0, 1.4, 9.0), tilt=0.12 * (k + 1)), d(110), d(50), WING))
        shapes.append(_grid_surface(f"rwing{k}", _slab((0, 92.0 + 7.0 * k, -205.0 - 6.0 * k), (64.0, 1.6, 11.0), tilt=0.2 * (k + 1)), d(100), d(50), WING))
    for k, sx in enumerate((-1, 1)):
        shapes.append(_grid_surface(f"endplate{k}", _slab((sx * 66.0, 84.0, -208.0), (1.6, 26.0, 22.0)), d(50), d(40), CARBON))
        shapes.append(_grid_surface(f"sidepod{k}", _ellipsoid((sx * 52.0, 36.0, -40.0), (26.0, 20.0, 78.0), bumps=0.03), d(130), d(90), PAINT))
        shapes.append(_grid_surface(f"mirror{k}", _ellipsoid((sx * 40.0, 74.0, 40.0), (7.0, 4.0, 3.0)), d(30), d(20), CHROME))
    shapes.append(_grid_surface("airbox", _ellipsoid((0, 86.0, -70.0), (16.0, 18.0, 60.0)), d(100), d(70), CHROME))
    shapes.append(_grid_surface("plank", _slab((0, 10.0, -20.0), (40.0, 1.5, 170.0)), d(80), d(60), PLANK))
    shapes.append(_grid_surface("rainlight", _slab((0, 40.0, -196.0), (6.0, 4.0, 1.5)), d(16), d(12), RAIN))
    return Model(shapes, mtls)


def shuttle(detail=1.0, tex_size=1024):
    """Transport-shuttle stand-in (~10 units long, y up): fuselage, wings, engines, windows; ~450k triangles at detail=1."""
    def d(n):
        return max(4, int(round(n * detail ** 0.5)))
    mtls = [
        mtl("hull", 1, Kd=(0.85, 0.86, 0.9), map_Kd=texture("noise", tex_size, 7)),
        mtl("trim", 2, Kd=(0.2, 0.3, 0.7)),
        mtl("window", 4, Kd=(0.85, 0.95, 1.0), Ni=1.45, d=0.2),
        mtl("engine", 2, Kd=(0.3, 0.3, 0.32)),
        mtl("glow", 3, Ke=(0.4, 0.7, 1.0)),
        mtl("mirror", 5, Kd=(0.9, 0.9, 0.9)),
        mtl("deck", 1, Kd=(0.5, 0.45, 0.4), map_Kd=texture("checker", tex_size, 8)),
    ]
    HULL, TRIM, WINDOW, ENGINE, GLOW, MIRROR, DECK = range(7)
    shapes = []
    shapes.append(_grid_surface("fuselage", _loft(-3.2, 3.4, lambda s: 0.5 + 0.9 * np.sin(np.pi * np.clip(s, 0, 1)) ** 0.8, lambda s: 0.45 + 0.6 * np.sin(np.pi * np.clip(s, 0, 1)), lambda s: 1.3 + 0.2 * s, power=3.0), d(380), d(300), HULL))
    shapes.append(_grid_surface("cockpit", _ellipsoid((0, 1.9, 2.0), (0.7, 0.45, 1.1)), d(100), d(70), WINDOW))
    for k, sx in enumerate((-1, 1)):
        shapes.append(_grid_surface(f"wing{k}", _slab((sx * 2.1, 1.2, -0.6), (1.5, 0.08, 1.3), tilt=0.05), d(140), d(80), HULL))
        shapes.append(_grid_surface(f"nacelle{k}", _ellipsoid((sx * 2.3, 1.25, -1.4), (0.42, 0.42, 1.3), bumps=0.03), d(130), d(90), ENGINE))
        shapes.append(_grid_surface(f"nozzle{k}", _ellipsoid((sx * 2.3, 1.25, -2.75), (0.3, 0.3, 0.12)), d(40), d(30), GLOW))
        shapes.append(_grid_surface(f"fin{k}", _slab((sx * 0.9, 2.4, -2.6), (0.06, 0.7, 0.6), tilt=-0.3), d(60), d(40), TRIM))
        shapes.append(_grid_surface(f"skid{k}", _slab((sx * 0.8, 0.25, 0.4), (0.12, 0.1, 2.2)), d(50), d(30), MIRROR))
        for w in range(3):
            shapes.append(_grid_surface(f"port{k}{w}", _ellipsoid((sx * 1.22, 1.65, -0.9 + 0.9 * w), (0.08, 0.22, 0.3)), d(30), d(20), WINDOW))
    shapes.append(_grid_surface("ramp", _slab((0, 0.55, -3.0), (0.9, 0.05, 0.9), tilt=0.5), d(70), d(50), DECK))
    shapes.append(_grid_surface("dish", _torus((0, 2.6, 0.2), 1, 0.5, 0.07), d(90), d(30), TRIM))
    return Model(shapes, mtls)


def blob(n_tris, seed=1):
    """One bumpy unit-scale ellipsoid; the stress scene instantiates many of these."""
    n = max(8, int(round((n_tris / 2.0) ** 0.5)))
    rs = np.random.RandomState(seed)
    f = (int(rs.randint(3, 9)), int(rs.randint(3, 9)))
    return Model([_grid_surface(f"blob{seed}", _ellipsoid((0, 0, 0), (1.0, 0.8, 0.9), bumps=0.12, freq=f, phase=float(rs.rand())), n, n, -1)])


def fitted(model, lo, hi):
    """The model scaled and moved per axis so that its bounding box becomes [lo, hi]; normals and uvs stay as they are.
    Lets one generator stand in for models of very different sizes (the upstream scene files carry the transforms of
    the real models, so a stand-in has to occupy about the same model-space box)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    allpos = np.concatenate([sh.pos.reshape(-1, 3) for sh in model.shapes]).astype(np.float64)
    a, b = allpos.min(axis=0), allpos.max(axis=0)
    k = (hi - lo) / np.where(b > a, b - a, 1.0)
    shapes = []
    for sh in model.shapes:
        out = Shape(sh.name, (sh.pos.astype(np.float64) - a) * k + lo, sh.nrm, sh.uv, sh.mtl)
        out.has_normals, out.has_uvs = sh.has_normals, sh.has_uvs
        shapes.append(out)
    return Model(shapes, model.mtls)


def room(detail=1.0, tex_size=256):
    """Dining-room stand-in, y up, about 1400 x 300 x 1700 units with the corner at the origin and z running negative:
    walls seen from inside, a table, seats, a mirror, a glass vase and a small lamp; 8 MTL records, two textured."""
    def d(n):
        return max(4, int(round(n * detail ** 0.5)))
    mtls = [
        mtl("plaster", 1, Kd=(0.82, 0.8, 0.76)),
        mtl("floor", 1, Kd=(0.6, 0.5, 0.4), map_Kd=texture("checker", tex_size, 11)),
        mtl("wood", 1, Kd=(0.45, 0.3, 0.18), map_Kd=texture("noise", tex_size, 12)),
        mtl("fabric", 1, Kd=(0.3, 0.35, 0.6)),
        mtl("steel", 2, Kd=(0.8, 0.8, 0.82)),
        mtl("mirror", 5, Kd=(0.95, 0.95, 0.95)),
        mtl("vase", 4, Kd=(0.8, 1.0, 0.9), Ni=1.5, d=0.3),
        mtl("bulb", 3, Ke=(1.0, 0.9, 0.7)),
    ]
    PLASTER, FLOOR, WOOD, FABRIC, STEEL, MIRROR, VASE, BULB = range(8)
    shapes = []
    # the six faces of the room as thin slabs (each its own shape, as a modelling tool would export them)
    cx, cy, cz = 700.0, 150.0, -850.0
    shapes.append(_grid_surface("floor", _slab((cx, -2.0, cz), (700.0, 2.0, 850.0)), d(60), d(60), FLOOR))
    shapes.append(_grid_surface("ceiling", _slab((cx, 302.0, cz), (700.0, 2.0, 850.0)), d(40), d(40), PLASTER))
    shapes.append(_grid_surface("wall_x0", _slab((-2.0, cy, cz), (2.0, 150.0, 850.0)), d(40), d(40), PLASTER))
    shapes.append(_grid_surface("wall_x1", _slab((1402.0, cy, cz), (2.0, 150.0, 850.0)), d(40), d(40), PLASTER))
    shapes.append(_grid_surface("wall_z0", _slab((cx, cy, 2.0), (700.0, 150.0, 2.0)), d(40), d(40), PLASTER))
    shapes.append(_grid_surface("wall_z1", _slab((cx, cy, -1702.0), (700.0, 150.0, 2.0)), d(40), d(40), PLASTER))
    shapes.append(_grid_surface("table_top", _slab((900.0, 78.0, -600.0), (160.0, 4.0, 90.0)), d(80), d(60), WOOD))
    for k, (sx, sz) in enumerate([(-1, -1), (1, -1), (-1, 1), (1, 1)]):
        shapes.append(_grid_surface(f"table_leg{k}", _slab((900.0 + sx * 140.0, 37.0, -600.0 + sz * 75.0), (5.0, 37.0, 5.0)), d(16), d(24), STEEL))
        shapes.append(_grid_surface(f"seat{k}", _ellipsoid((900.0 + sx * 230.0, 45.0, -600.0 + sz * 40.0), (42.0, 45.0, 42.0), bumps=0.05), d(90), d(60), FABRIC))
    shapes.append(_grid_surface("mirror", _slab((1398.0, 160.0, -700.0), (1.0, 90.0, 140.0)), d(20), d(20), MIRROR))
    shapes.append(_grid_surface("vase", _ellipsoid((900.0, 112.0, -600.0), (18.0, 30.0, 18.0), bumps=0.08, freq=(6, 4)), d(120), d(90), VASE))
    shapes.append(_grid_surface("lamp", _ellipsoid((1150.0, 200.0, -350.0), (14.0, 14.0, 14.0)), d(30), d(20), BULB))
    shapes.append(_grid_surface("ring", _torus((1150.0, 222.0, -350.0), 1, 30.0, 4.0), d(80), d(24), STEEL))
    return Model(shapes, mtls)
