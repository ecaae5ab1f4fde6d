"""JPEG textures (raytrace_amd/host/jpeg_decode.cpp), CPU only.

Upstream decodes textures with cv::imread (csrc/sceneloader.cpp:224): libjpeg-turbo with its defaults.  The files under
tests/golden/jpeg were encoded by Pillow's libjpeg-turbo and each carries the RGB bytes that library's decoder returned
(make_fixtures.py, committed beside them); the decoder here must reproduce them byte for byte.  Not covered by a
fixture, because Pillow cannot write them: 4:4:0 and 4:1:1 chroma layouts (the vertical triangle filter and sample
replication), which are therefore parity unpinned.
"""
import glob
import os
import subprocess

import numpy as np
import pytest

from raytrace_amd import capi, scene_files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "jpeg")
PNG_TOOL = os.path.join(ROOT, "bin", "png_tool")
SCENE_DUMP = os.path.join(ROOT, "bin", "scene_dump")

pytestmark = pytest.mark.skipif(not (os.path.exists(SCENE_DUMP) and os.path.exists(PNG_TOOL)), reason="host tools not built (make host)")

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.jpg")) if os.path.exists(p[:-4] + ".expect.png"))


def decode(path, tmp_path):
    out = str(tmp_path / "decoded.rgb")
    r = subprocess.run([PNG_TOOL, "decode", path, out], capture_output=True, text=True, timeout=120)
    if r.returncode != 0:
        return None, r.stderr
    w, h = (int(x) for x in r.stdout.split())
    return np.fromfile(out, dtype=np.uint8).reshape(h, w, 3), ""


def test_fixture_set_is_complete():
    assert len(CASES) >= 18
    for need in ("base_444", "base_422", "base_420", "grey", "prog_420", "restart_420", "restart_prog", "rgb_adobe", "photo_420"):
        assert need in CASES


@pytest.mark.parametrize("name", CASES)
def test_decodes_to_the_bytes_libjpeg_turbo_returns(tmp_path, name):
    got, err = decode(os.path.join(GOLDEN, name + ".jpg"), tmp_path)
    assert got is not None, err
    expect = scene_files.read_png_rgb(os.path.join(GOLDEN, name + ".expect.png"))
    assert got.shape == expect.shape
    assert np.array_equal(got, expect), "%d samples differ, largest by %d" % (
        (got != expect).sum(), np.abs(got.astype(int) - expect.astype(int)).max())


def test_refused_and_damaged_files(tmp_path):
    got, err = decode(os.path.join(GOLDEN, "cmyk.jpg"), tmp_path)                  # four components
    assert got is None and "JPEG" in err and "component" in err
    data = open(os.path.join(GOLDEN, "base_420.jpg"), "rb").read()
    p = str(tmp_path / "cut.jpg")
    # entropy data cut short: the rest of the picture decodes from zero bits, as the library does (it only warns)
    open(p, "wb").write(data[:len(data) - 200])
    got, err = decode(p, tmp_path)
    expect = scene_files.read_png_rgb(os.path.join(GOLDEN, "base_420.expect.png"))
    assert got is not None and got.shape == expect.shape
    assert np.array_equal(got[:8], expect[:8])                                     # the first MCU row was complete
    # cut inside the headers: nothing to decode
    sos = data.find(b"\xff\xda")
    open(p, "wb").write(data[:sos])
    got, err = decode(p, tmp_path)
    assert got is None and "JPEG" in err
    # arithmetic-coded frame marker
    sof = data.find(b"\xff\xc0")
    open(p, "wb").write(data[:sof] + b"\xff\xc9" + data[sof + 2:])
    got, err = decode(p, tmp_path)
    assert got is None and "arithmetic" in err
    # 12-bit samples
    open(p, "wb").write(data[:sof + 4] + b"\x0c" + data[sof + 5:])
    got, err = decode(p, tmp_path)
    assert got is None and "8-bit" in err


def test_scene_with_a_jpeg_texture(tmp_path):
    """map_Kd naming a .jpg: the loader's texture holds the decoded picture, flipped top to bottom, alpha 255
    (csrc/sceneloader.cpp:223-252)."""
    d = tmp_path / "case"
    d.mkdir()
    jpg = open(os.path.join(GOLDEN, "photo_420.jpg"), "rb").read()
    (d / "skin.jpg").write_bytes(jpg)
    (d / "m.mtl").write_text("newmtl skin\nKd 1 1 1\nmap_Kd skin.jpg\n")
    (d / "t.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl skin\nf 1/1 2/2 3/3\n")
    (d / "scene.txt").write_text(
        "scene_config\nmax_ray_depth 2\nsize 8 6\nsample_per_pixel 1\nfov 60\ngamma 2.2\nbackground 0 0 0\n"
        "camera_pos 0 -3 0 camera_dir 0 1 0 camera_up 0 0 1\nrussian_roulette 0.9\n"
        "object\n path %s\n mtldir %s\n type lambertian\nend\nend\n" % (d / "t.obj", d))
    out = tmp_path / "dump"
    out.mkdir()
    r = subprocess.run([SCENE_DUMP, str(d / "scene.txt"), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    expect = scene_files.read_png_rgb(os.path.join(GOLDEN, "photo_420.expect.png"))
    tex = np.fromfile(str(out / "tex0.bin"), dtype=np.uint8).reshape(expect.shape[0], expect.shape[1], 4)
    assert np.array_equal(tex[:, :, :3], expect[::-1]) and (tex[:, :, 3] == 255).all()
    m = np.fromfile(str(out / "materials.bin"), dtype=capi.MATERIAL_DTYPE)
    assert (m["texture"] == 0).any()
