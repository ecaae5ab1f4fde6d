"""Host side above the C ABI, part 2: image files and the scene loader (raytrace_amd/host), CPU only.

The scene loader is checked against raytrace_amd.scene.assemble/finalize -- the Python plumbing that feeds the
same library entry points -- on scenes written out as scene text + OBJ + MTL + PNG, array for array.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from raytrace_amd import capi, scene, scene_files
from tests.util import same_f32, same_nodes, same_triangles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE_DUMP = os.path.join(ROOT, "bin", "scene_dump")
PNG_TOOL = os.path.join(ROOT, "bin", "png_tool")

pytestmark = pytest.mark.skipif(not (os.path.exists(SCENE_DUMP) and os.path.exists(PNG_TOOL)), reason="host tools not built (make host)")


def run(*cmd):
    return subprocess.run(list(cmd), capture_output=True, text=True, timeout=300)


# ---------------------------------------------------------------------------------------------------------------
# image files
# ---------------------------------------------------------------------------------------------------------------
def filtered(rows, bpp, filters=None):
    """Scanlines with a filter type byte in front of each, the way one image (or one interlace pass) is stored."""
    if not rows:
        return b""
    raw, prev = b"", bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = filters[y % len(filters)] if filters else 0
        out = bytearray(len(row))
        for i in range(len(row)):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (row[i] - pred) & 255
        raw += bytes([f]) + bytes(out)
        prev = row
    return raw


def png_file(width, height, depth, colour, raw, palette=None, interlace=0):
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, colour, 0, 0, interlace))
    if palette is not None:
        body += chunk(b"PLTE", bytes(palette))
    z = zlib.compress(raw)
    body += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:])       # split stream: two IDAT chunks
    return b"\x89PNG\r\n\x1a\n" + body + chunk(b"IEND", b"")


def png_bytes(width, height, depth, colour, rows, palette=None, filters=None):
    """A PNG from raw sample rows (bytes per row already packed), with the given filter type per row."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    return png_file(width, height, depth, colour, filtered(rows, max(1, channels * depth // 8), filters), palette)


def pack_row(samples, depth):
    """One scanline from a flat list of samples: big-endian 16-bit, bytes, or sub-byte samples packed high bit first."""
    if depth == 16:
        return b"".join(struct.pack(">H", int(v)) for v in samples)
    if depth == 8:
        return bytes(int(v) for v in samples)
    per = 8 // depth
    vals = [int(v) for v in samples] + [0] * (-len(samples) % per)
    return bytes(sum(vals[i + k] << (8 - depth * (k + 1)) for k in range(per)) for i in range(0, len(vals), per))


def png_adam7(samples, depth, colour, palette=None, filters=None):
    """An interlaced PNG from samples[h, w, channels]: seven sub-images, each with its own scanlines and filters."""
    h, w, channels = samples.shape
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += filtered([pack_row(sub[y].reshape(-1), depth) for y in range(sub.shape[0])], max(1, channels * depth // 8), filters)
    return png_file(w, h, depth, colour, raw, palette, interlace=1)


def decode(path, tmp_path):
    out = str(tmp_path / "decoded.rgb")
    r = run(PNG_TOOL, "decode", path, out)
    if r.returncode != 0:
        return None, r.stderr
    w, h = (int(x) for x in r.stdout.split())
    return np.fromfile(out, dtype=np.uint8).reshape(h, w, 3), ""


def test_png_decoder_colour_types_depths_and_filters(tmp_path):
    rng = np.random.RandomState(3)
    w, h = 13, 9
    # 8-bit RGB, all five filters in turn
    rgb = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    p = str(tmp_path / "rgb.png")
    open(p, "wb").write(png_bytes(w, h, 8, 2, [rgb[y].tobytes() for y in range(h)], filters=[0, 1, 2, 3, 4]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, rgb)
    # RGBA: alpha dropped (cv::imread IMREAD_COLOR)
    rgba = rng.randint(0, 256, size=(h, w, 4)).astype(np.uint8)
    open(p, "wb").write(png_bytes(w, h, 8, 6, [rgba[y].tobytes() for y in range(h)], filters=[4, 3]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, rgba[:, :, :3])
    # grey 8 and grey+alpha: replicated to three channels
    g = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
    open(p, "wb").write(png_bytes(w, h, 8, 0, [g[y].tobytes() for y in range(h)], filters=[1]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, np.repeat(g[:, :, None], 3, axis=2))
    ga = rng.randint(0, 256, size=(h, w, 2)).astype(np.uint8)
    open(p, "wb").write(png_bytes(w, h, 8, 4, [ga[y].tobytes() for y in range(h)], filters=[2]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, np.repeat(ga[:, :, :1], 3, axis=2))
    # 16-bit RGB: the high byte of each sample is kept
    rgb16 = rng.randint(0, 65536, size=(h, w, 3)).astype(">u2")
    open(p, "wb").write(png_bytes(w, h, 16, 2, [rgb16[y].tobytes() for y in range(h)], filters=[4]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, (rgb16.astype(np.uint16) >> 8).astype(np.uint8))
    # palette, 4 bits per index; grey, 1 and 2 bits per pixel (scaled to 0..255)
    pal = rng.randint(0, 256, size=(16, 3)).astype(np.uint8)
    idx = rng.randint(0, 16, size=(h, w))
    rows = []
    for y in range(h):
        v = list(idx[y]) + [0] * (w % 2)
        rows.append(bytes((v[i] << 4) | v[i + 1] for i in range(0, len(v), 2)))
    open(p, "wb").write(png_bytes(w, h, 4, 3, rows, palette=pal.tobytes()))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, pal[idx])
    for depth in (1, 2):
        v = rng.randint(0, 1 << depth, size=(h, w))
        rows = []
        per = 8 // depth
        for y in range(h):
            vals = list(v[y]) + [0] * (-w % per)
            rows.append(bytes(sum(vals[i + k] << (8 - depth * (k + 1)) for k in range(per)) for i in range(0, len(vals), per)))
        open(p, "wb").write(png_bytes(w, h, depth, 0, rows))
        got, _ = decode(p, tmp_path)
        assert np.array_equal(got[:, :, 0], (v * 255 // ((1 << depth) - 1)).astype(np.uint8))


def test_png_decoder_adam7(tmp_path):
    """Interlaced files: every size from 1x1 to beyond one 8x8 tile, so that passes come out empty, partial and full."""
    rng = np.random.RandomState(6)
    p = str(tmp_path / "i.png")
    for w, h in [(1, 1), (2, 1), (1, 3), (3, 2), (4, 4), (5, 7), (8, 8), (9, 9), (13, 17), (33, 5)]:
        rgb = rng.randint(0, 256, size=(h, w, 3))
        open(p, "wb").write(png_adam7(rgb, 8, 2, filters=[0, 1, 2, 3, 4]))
        got, err = decode(p, tmp_path)
        assert got is not None, err
        assert np.array_equal(got, rgb.astype(np.uint8)), (w, h)
    w, h = 21, 11
    rgba16 = rng.randint(0, 65536, size=(h, w, 4))
    open(p, "wb").write(png_adam7(rgba16, 16, 6, filters=[4]))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, (rgba16[:, :, :3] >> 8).astype(np.uint8))
    for depth in (1, 2, 4):                                   # sub-byte samples: each pass row is padded to a byte on its own
        v = rng.randint(0, 1 << depth, size=(h, w, 1))
        open(p, "wb").write(png_adam7(v, depth, 0, filters=[1, 2]))
        got, _ = decode(p, tmp_path)
        assert np.array_equal(got[:, :, 0], (v[:, :, 0] * 255 // ((1 << depth) - 1)).astype(np.uint8)), depth
    pal = rng.randint(0, 256, size=(4, 3)).astype(np.uint8)
    idx = rng.randint(0, 4, size=(h, w, 1))
    open(p, "wb").write(png_adam7(idx, 2, 3, palette=pal.tobytes()))
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, pal[idx[:, :, 0]])


def test_png_decoder_rejects_what_it_cannot_read(tmp_path):
    rgb = np.zeros((4, 4, 3), dtype=np.uint8)
    good = png_bytes(4, 4, 8, 2, [rgb[y].tobytes() for y in range(4)])
    p = str(tmp_path / "bad.png")
    open(p, "wb").write(good[:40] + bytes([good[40] ^ 1]) + good[41:])              # flipped bit -> CRC mismatch
    assert decode(p, tmp_path)[0] is None
    open(p, "wb").write(good[:len(good) // 2])                                      # truncated
    assert decode(p, tmp_path)[0] is None
    open(p, "wb").write(b"\xff\xd8\xff\xe0" + b"\0" * 64)                           # JPEG signature
    got, err = decode(p, tmp_path)
    assert got is None and "JPEG" in err
    assert decode(str(tmp_path / "missing.png"), tmp_path)[0] is None


def test_ppm_and_pgm(tmp_path):
    rng = np.random.RandomState(4)
    rgb = rng.randint(0, 256, size=(5, 7, 3)).astype(np.uint8)
    p = str(tmp_path / "a.ppm")
    open(p, "wb").write(b"P6\n# comment\n7 5\n255\n" + rgb.tobytes())
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, rgb)
    open(p, "wb").write(b"P5 7 5 255\n" + rgb[:, :, 0].tobytes())
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, np.repeat(rgb[:, :, :1], 3, axis=2))


def test_png_writer_round_trip(tmp_path):
    rng = np.random.RandomState(5)
    rgb = rng.randint(0, 256, size=(33, 47, 3)).astype(np.uint8)
    src, p = str(tmp_path / "in.rgb"), str(tmp_path / "out.png")
    rgb.tofile(src)
    assert run(PNG_TOOL, "encode", src, "47", "33", p).returncode == 0
    assert np.array_equal(scene_files.read_png_rgb(p), rgb)                         # an independent decoder reads it
    got, _ = decode(p, tmp_path)
    assert np.array_equal(got, rgb)


def test_frame_conversion_matches_convertTo_semantics(tmp_path):
    """csrc/worker.cpp:25-27: flip rows, v*255 in binary32, round half to even, saturate; NaN -> 0."""
    vals = np.array([0.0, 1.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, -0.3, 1.7, np.nan, np.inf, -np.inf, 0.49999 / 255,
                     0.2, 0.7, 1e-9], dtype=np.float32)
    frame = np.zeros((2, len(vals), 3), dtype=np.float32)
    frame[0, :, 0] = vals          # first row of the buffer = bottom row of the picture
    frame[1, :, 1] = vals
    src, p = str(tmp_path / "f.f32"), str(tmp_path / "f.png")
    frame.tofile(src)
    assert run(PNG_TOOL, "frame", src, str(len(vals)), "2", p).returncode == 0
    img = scene_files.read_png_rgb(p)
    s = (vals * np.float32(255.0)).astype(np.float32)
    with np.errstate(invalid="ignore"):
        expect = np.where(np.isnan(s), 0, np.clip(np.rint(s.astype(np.float64)), 0, 255)).astype(np.uint8)   # rint = ties to even
    assert list(expect[:6]) == [0, 255, 0, 2, 2, 254]
    assert np.array_equal(img[1, :, 0], expect) and np.array_equal(img[0, :, 1], expect)
    assert not img[0, :, 0].any() and not img[1, :, 1].any()


# ---------------------------------------------------------------------------------------------------------------
# scene loader
# ---------------------------------------------------------------------------------------------------------------
def load_with_cpp(scene_path, tmp_path):
    out = tmp_path / "dump"
    out.mkdir(exist_ok=True)
    r = run(SCENE_DUMP, scene_path, str(out))
    return r, out


def summary(stdout):
    rows = {}
    for ln in stdout.splitlines():
        if ln.startswith(("config ", "background ", "objects ")):
            rows[ln.split()[0]] = ln.split()[1:]
    return rows


def assert_loader_output_matches(asm, stdout, out, size_and_spp=None):
    """What bin/scene_dump printed and wrote for a scene file against the Python assembly of the same scene, array for
    array.  `size_and_spp` = (width, height, n_samples) the file states, when they are not the assembly's."""
    s = summary(stdout)
    cfg = asm.config
    width, height, n_samples = size_and_spp or (cfg.width, cfg.height, cfg.n_samples)
    assert [int(s["config"][i]) for i in (1, 3, 5, 6, 8)] == [cfg.max_depth, 1000000, width, height, n_samples]
    assert float.fromhex(s["config"][10]) == cfg.fov                                           # degrees -> radians as sceneloader.cpp:287
    assert same_f32(np.array([float.fromhex(s["config"][12]), float.fromhex(s["config"][14])], dtype=np.float32),
                    np.array([cfg.gamma, cfg.russian_roulette], dtype=np.float32))
    got_cam = np.array([float.fromhex(x) for x in s["background"] if x.startswith(("0x", "-0x"))], dtype=np.float32)
    assert same_f32(got_cam, np.array(list(cfg.background) + list(cfg.camera_pos) + list(cfg.camera_dir) + list(cfg.camera_up), dtype=np.float32))
    objs = scene.finalize(asm)                                                                  # transform + BVH through the same library calls
    assert int(s["objects"][0]) == len(objs)
    for i, (tri, bvh) in enumerate(objs):
        t = np.fromfile(str(out / f"tri{i}.bin"), dtype=capi.TRIANGLE_DTYPE)
        b = np.fromfile(str(out / f"bvh{i}.bin"), dtype=capi.NODE_DTYPE)
        assert len(t) == len(tri) and same_triangles(t, tri), f"object {i}: triangles differ"
        assert same_nodes(b, bvh), f"object {i}: BVH differs"
    m = np.fromfile(str(out / "materials.bin"), dtype=capi.MATERIAL_DTYPE)
    assert len(m) == len(asm.materials)
    for f in ("type", "texture"):
        assert np.array_equal(m[f], asm.materials[f])
    for f in ("albedo", "optical_density", "metal_fuzz"):
        assert same_f32(m[f], asm.materials[f]), f
    assert int(s["objects"][4]) == len(asm.textures)
    for i, tex in enumerate(asm.textures):
        got = np.fromfile(str(out / f"tex{i}.bin"), dtype=np.uint8).reshape(tex.shape)
        assert np.array_equal(got[:, :, :3], tex[:, :, :3]) and (got[:, :, 3] == 255).all()


@pytest.mark.parametrize("name,kw", [
    ("material_box", dict(width=32, height=20, n_samples=8)),                                   # textured MTL record, every material type
    ("formula1", dict(width=96, height=60, n_samples=8, detail=0.05, tex_size=32)),             # 31 shapes, 10 MTL records, 4 textures
    ("formula1", dict(width=96, height=60, n_samples=8, detail=0.05, glass=True)),              # simple_material
    ("shuttle", dict(width=64, height=40, n_samples=4, detail=0.05, tex_size=32)),
    ("luyu", dict(width=64, height=40, n_samples=4, statue_tris=3000)),
    ("bike", dict(width=64, height=48, n_samples=4, detail=0.03, tex_size=16)),                 # fitted model far from the origin
    ("bunny", dict(width=48, height=48, n_samples=4, bunny_tris=2000)),                         # scale 50 on a 0.15-unit mesh
    ("cat", dict(width=48, height=48, n_samples=4, detail=0.03, tex_size=16)),
    ("house", dict(width=48, height=48, n_samples=4, detail=0.03, tex_size=16)),                # planes scaled by 1e6
    ("room", dict(width=48, height=32, n_samples=4, detail=0.1, tex_size=16)),                  # y-up camera, depth 16
])
def test_scene_loader_matches_python_assembly(tmp_path, name, kw):
    asm = getattr(scene, name)(**kw)
    path = scene_files.write_scene(asm, str(tmp_path / "files"))
    r, out = load_with_cpp(path, tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert_loader_output_matches(asm, r.stdout, out)


BASE = """scene_config
# a comment line
max_ray_depth 4
size 8 6
sample_per_pixel 2
fov 90
gamma 2.2
background 0.1 0.2 0.3
camera_pos 0 -3 1 camera_dir 0 1 0
camera_up 0 0 1
russian_roulette 0.8
"""

TRI_OBJ = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"


def write_case(tmp_path, text, obj_name="a tri.obj"):
    d = tmp_path / "case"
    d.mkdir(exist_ok=True)
    (d / obj_name).write_text(TRI_OBJ)
    p = d / "scene.txt"
    p.write_text(text.replace("@OBJ@", str(d / obj_name)))
    return str(p)


def test_scene_grammar_quirks(tmp_path):
    """csrc/sceneloader.cpp:57-96, 264-321: keys in any order and several per line; `#` is a word of its own and eats the
    rest of its line; `path` takes the rest of the line (blanks allowed); the top-level `end` stops reading."""
    text = BASE + "object\n  type lambertian # rest ignored position 9 9 9\n  path @OBJ@\n  albedo 0.5 0.25 1\n  scale 2 2 2 position 1 0 0\nend\nend\nobject\n  path /nonexistent.obj\n  type light\nend\n"
    r, out = load_with_cpp(write_case(tmp_path, text), tmp_path)
    assert r.returncode == 0, r.stderr
    s = summary(r.stdout)
    assert s["objects"][0] == "1" and s["objects"][2] == "1"                   # the block after `end` was never read
    tri = np.fromfile(str(out / "tri0.bin"), dtype=capi.TRIANGLE_DTYPE)
    assert np.array_equal(tri["v"]["position"][0], np.array([[1, 0, 0], [3, 0, 0], [1, 2, 0]], dtype=np.float32))   # scale, then translate
    assert np.array_equal(tri["v"]["normal"][0], np.tile(np.array([0, 0, 1], dtype=np.float32), (3, 1)))            # face normal: the file has no vn
    assert not tri["v"]["texcoord"].any()                                       # no vt: zeros here (uninitialised upstream)
    m = np.fromfile(str(out / "materials.bin"), dtype=capi.MATERIAL_DTYPE)
    assert m["type"][0] == capi.MAT_LAMBERTIAN and list(m["albedo"][0]) == [0.5, 0.25, 1.0]


@pytest.mark.parametrize("text,needle", [
    ("scene_conf\n", "Invalid configuration file"),
    (BASE + "end\n", "No object loaded"),
    (BASE + "shininess 3\nend\n", "Unknown key shininess"),
    (BASE + "object\n path @OBJ@\nend\nend\n", "No material type specified"),
    (BASE + "object\n type lambertian\nend\nend\n", "No object file specified"),
    (BASE + "object\n path @OBJ@\n type plastic\nend\nend\n", "Unknown material type plastic"),
    (BASE + "object\n path @OBJ@\n type light\n colour 1 1 1\nend\nend\n", "Unknown key in object colour"),
    (BASE + "object\n path @OBJ@\n type light\nend\n", "No object file specified"),          # no closing top-level `end`: phantom object
    (BASE + "object\n path /nonexistent/x.obj\n type light\nend\nend\n", "Cannot open file"),
    (BASE + "object\n path @OBJ@\n type light\n scale 1 x 1\nend\nend\n", "bad number"),
])
def test_scene_errors(tmp_path, text, needle):
    r, _ = load_with_cpp(write_case(tmp_path, text), tmp_path)
    assert r.returncode == 1 and "load_scene -1" in r.stdout
    assert needle in r.stderr


def test_missing_scene_file(tmp_path):
    r, _ = load_with_cpp(str(tmp_path / "none.txt"), tmp_path)
    assert r.returncode == 1 and "Cannot open file" in r.stderr


def test_tracked_benchmark_scene_files_are_what_the_generator_writes(tmp_path):
    """benchmarks/scenes/*.txt are the text files tools/make_bench_scenes.py writes for the five BASELINE configurations
    (the model files beside them are generated on demand and not tracked)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_bench_scenes", os.path.join(root, "tools", "make_bench_scenes.py"))
    mbs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mbs)
    assert sorted(mbs.CONFIGS) == ["formula1", "formula1_glass", "luyu", "shuttle", "stress"]
    for name in ("luyu", "formula1_glass"):                         # two of the five; the others differ only in size
        path, _ = mbs.write(name, False, root=str(tmp_path))
        tracked = open(os.path.join(root, "benchmarks", "scenes", f"scene_{name}.txt")).read()
        assert open(path).read() == tracked
        assert "size 400 240" in tracked if name == "luyu" else "simple_material" in tracked


def test_material_table_and_ids_by_hand(tmp_path):
    """Expectations written out by hand from upstream's text (csrc/sceneloader.cpp:132-219), independent of
    raytrace_amd.scene: one render object per shape; the block's own material first, then one per MTL record in file
    order; a face's id = block offset + MTL index + 1, or the block's own material when the face has none; illum 1..5 ->
    lambertian, metal, light, refractive, reflective (anything else lambertian); a light takes Ke, the others Kd; a
    refractive record takes Ni as optical density -- which then stays in force for the records after it -- and the albedo
    (1-d) + d*Kd; metal_fuzz is the block's for every record; `simple_material` gives every face the block's own."""
    d = tmp_path / "hand"
    d.mkdir()
    (d / "m.mtl").write_text(
        "newmtl red\nillum 1\nKd 1 0 0\n\n"
        "newmtl lamp\nillum 3\nKd 0.5 0.5 0.5\nKe 4 5 6\n\n"
        "newmtl glass\nillum 4\nKd 0.5 1 0\nNi 1.5\nd 0.25\n\n"
        "newmtl mirror\nillum 5\nKd 0.9 0.8 0.7\nNi 9\n\n"
        "newmtl odd\nillum 7\nKd 0.1 0.2 0.3\n")
    (d / "two.obj").write_text(
        "mtllib m.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n"
        "o first\nf 1 2 3\nusemtl glass\nf 1 2 4\nusemtl red\nf 1 3 4\n"
        "o second\nusemtl mirror\nf 2 3 4\n")
    block = ("object\n path {0}/two.obj\n mtldir {0}\n type metal\n albedo 0.3 0.3 0.3\n optical_density 2\n metal_fuzz 0.125\n{1}end\n")
    text = BASE + block.format(d, "") + block.format(d, " simple_material\n") + "end\n"
    p = d / "scene.txt"
    p.write_text(text)
    r, out = load_with_cpp(str(p), tmp_path)
    assert r.returncode == 0, r.stderr
    s = summary(r.stdout)
    assert s["objects"][0] == "4" and s["objects"][2] == "7"          # 2 shapes per block; 1 + 5 materials, then 1 (simple_material adds no MTL records)
    m = np.fromfile(str(out / "materials.bin"), dtype=capi.MATERIAL_DTYPE)
    L, ME, LI, RA, RE = capi.MAT_LAMBERTIAN, capi.MAT_METAL, capi.MAT_LIGHT, capi.MAT_REFRACTIVE, capi.MAT_REFLECTIVE
    assert list(m["type"]) == [ME, L, LI, RA, RE, L, ME]
    f = np.float32
    assert same_f32(m["albedo"], np.array([[0.3, 0.3, 0.3], [1, 0, 0], [4, 5, 6],
                                           [f(0.75) + f(0.25) * f(0.5), f(0.75) + f(0.25) * f(1), f(0.75) + f(0.25) * f(0)],
                                           [0.9, 0.8, 0.7], [0.1, 0.2, 0.3], [0.3, 0.3, 0.3]], dtype=np.float32))
    assert list(m["optical_density"]) == [2, 2, 2, 1.5, 1.5, 1.5, 2]   # Ni of the glass record carries over to mirror and odd, not Ni 9
    assert (m["metal_fuzz"] == f(0.125)).all() and (m["texture"] == -1).all()
    ids = [list(np.fromfile(str(out / f"tri{i}.bin"), dtype=capi.TRIANGLE_DTYPE)["material_id"]) for i in range(4)]
    assert ids == [[0, 3, 1], [4], [6, 6, 6], [6]]                      # block 1: own, glass (0+2+1), red (0+0+1); mirror (0+3+1); block 2: all its own (offset 6)
