"""GPU tests of the host side: bin/main and bin/server drive the same C ABI as the Python plumbing and must
produce the same picture, byte for byte, for the same seed."""
import os
import socket
import subprocess
import time
import urllib.error
import urllib.request

import numpy as np
import pytest

from raytrace_amd import capi, scene, scene_files

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "bin", "main")
SERVER = os.path.join(ROOT, "bin", "server")
SEED = 0x5EED0002


def expected_png_pixels(asm, batch, seed, camera=None):
    """The frame through the Python plumbing: rt_raytrace + rt_post_process, then the worker's conversion
    (csrc/worker.cpp:25-27: flip, x255 in binary32, round half to even, saturate)."""
    import torch
    cfg = asm.config
    ctx = capi.Context(0)
    try:
        ctx.upload_scene(scene.finalize(asm), asm.materials, asm.textures)
        ctx.init_randstate(seed, cfg.width, cfg.height)
        pos, d, up = camera if camera else (cfg.camera_pos, cfg.camera_dir, cfg.camera_up)
        ctx.setup_raytrace(cfg.width, cfg.height, pos, d, up, cfg.fov)
        npix = cfg.width * cfg.height
        buf = torch.zeros(npix, 3, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        if batch <= 0 or batch > cfg.n_samples:
            batch = cfg.n_samples
        ctx.raytrace(npix, 0, cfg.n_samples, batch, cfg.max_depth, cfg.background, cfg.russian_roulette, buf.data_ptr())
        ctx.post_process(npix, cfg.n_samples, cfg.gamma, buf.data_ptr())
        ctx.synchronize()
        frame = buf.cpu().numpy().reshape(cfg.height, cfg.width, 3)
    finally:
        ctx.close()
    s = (frame[::-1] * np.float32(255.0)).astype(np.float32)
    return np.where(np.isnan(s), 0, np.clip(np.rint(s.astype(np.float64)), 0, 255)).astype(np.uint8)


@pytest.mark.parametrize("batch", [4, 0, 5])
def test_bin_main_renders_the_same_picture(tmp_path, batch):
    asm = scene.material_box(width=64, height=40, n_samples=16)
    path = scene_files.write_scene(asm, str(tmp_path / "files"), name="boxes")
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, RT_SEED=str(SEED))
    r = subprocess.run([MAIN, path, str(out), str(batch)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "---> Raytracing Start <---" in r.stdout and "Image saved to" in r.stdout and "encoder: cleaning up" in r.stdout
    png = out / "boxes-output.png"                          # <output_dir>/<scene name>-output.png (csrc/worker.cpp:36)
    assert png.exists()
    got = scene_files.read_png_rgb(str(png))
    want = expected_png_pixels(asm, batch, SEED)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{(got != want).any(axis=2).sum()} of {got.shape[0] * got.shape[1]} pixels differ"
    assert want.std() > 10                                  # a picture, not a constant


def test_bin_main_with_a_jpeg_texture(tmp_path):
    """The textured floor's image file replaced by a JPEG (cv::imread goes by content, not by name; so does the loader
    here): bin/main's picture equals the Python route's with the pixels libjpeg-turbo decodes from that file."""
    import glob
    golden = os.path.join(ROOT, "tests", "golden", "jpeg")
    pixels = scene_files.read_png_rgb(os.path.join(golden, "photo_420.expect.png"))
    tex = np.concatenate([pixels[::-1], np.full(pixels.shape[:2] + (1,), 255, dtype=np.uint8)], axis=2)   # stored bottom-up, RGBA
    asm = scene.material_box(width=64, height=40, n_samples=16)
    assert len(asm.textures) == 1
    asm.textures[0] = np.ascontiguousarray(tex)
    asm.blocks[0].model.mtls[0]["map_Kd"] = asm.textures[0]
    path = scene_files.write_scene(asm, str(tmp_path / "files"), name="jpegfloor")
    files = glob.glob(str(tmp_path / "files" / "tex_*.png"))
    assert len(files) == 1
    with open(files[0], "wb") as f:
        f.write(open(os.path.join(golden, "photo_420.jpg"), "rb").read())
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MAIN, path, str(out), "4"], capture_output=True, text=True, timeout=120, env=dict(os.environ, RT_SEED=str(SEED)))
    assert r.returncode == 0, r.stderr[-2000:]
    got = scene_files.read_png_rgb(str(out / "jpegfloor-output.png"))
    want = expected_png_pixels(asm, 4, SEED)
    assert np.array_equal(got, want), f"{(got != want).any(axis=2).sum()} pixels differ"


def test_bin_main_usage_and_bad_scene(tmp_path):
    r = subprocess.run([MAIN], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "Usage:" in r.stderr       # csrc/main.cpp:10-13
    bad = tmp_path / "bad.txt"
    bad.write_text("not a scene\n")
    r = subprocess.run([MAIN, str(bad), str(tmp_path), "1"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "Failed to load scene file" in r.stderr     # csrc/worker.cpp:123-126 exits with 0


def http(method, url, body=None, timeout=10):
    req = urllib.request.Request(url, data=body, method=method)
    try:
        with urllib.request.urlopen(req, timeout=timeout) as resp:
            return resp.status, resp.read(), resp.headers.get("Content-Type")
    except urllib.error.HTTPError as e:
        return e.code, e.read(), e.headers.get("Content-Type")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def wait_for(predicate, seconds=60.0, step=0.05):
    end = time.time() + seconds
    while time.time() < end:
        v = predicate()
        if v:
            return v
        time.sleep(step)
    return None


def test_bin_server_routes(tmp_path):
    asm = scene.material_box(width=48, height=32, n_samples=8)
    path = scene_files.write_scene(asm, str(tmp_path / "files"), name="served")
    static = tmp_path / "static"
    static.mkdir()
    port = free_port()
    env = dict(os.environ, RT_SEED=str(SEED), RT_STATIC_DIR=str(static))
    log = open(str(tmp_path / "server.log"), "w")
    proc = subprocess.Popen([SERVER, path, "4", "127.0.0.1", str(port)], stdout=log, stderr=subprocess.STDOUT, env=env)
    base = f"http://127.0.0.1:{port}"
    try:
        def first_image():
            try:
                st, body, ctype = http("GET", base + "/image")
            except (urllib.error.URLError, ConnectionError, OSError):
                return None
            return (body, ctype) if st == 200 else None
        got = wait_for(first_image, 90)
        assert got, "no frame within 90 s: " + open(str(tmp_path / "server.log")).read()[-1500:]
        body, ctype = got
        assert ctype == "image/png" and body[:8] == b"\x89PNG\r\n\x1a\n"
        (tmp_path / "first.png").write_bytes(body)
        first = scene_files.read_png_rgb(str(tmp_path / "first.png"))
        assert np.array_equal(first, expected_png_pixels(asm, 4, SEED))
        # the frame is also a static file below the base directory (csrc/server.cpp:53)
        st, body2, ctype = http("GET", base + "/served-output.png")
        assert st == 200 and body2 == body and ctype == "image/png"
        # GET / serves index.html from the static directory, 404 while it is missing (csrc/server.cpp:40-50)
        st, b, _ = http("GET", base + "/")
        assert st == 404 and b == b"File not found"
        (static / "index.html").write_text("<html>hello</html>")
        st, b, ctype = http("GET", base + "/")
        assert st == 200 and b == b"<html>hello</html>" and ctype == "text/html"
        cfg = asm.config
        st, b, _ = http("GET", base + "/camera-pos")
        assert st == 200 and b.decode() == "%f,%f,%f" % tuple(float(np.float32(x)) for x in cfg.camera_pos)
        # update-camera: the position is discarded, the direction comes from the angles (csrc/worker.cpp:185-199)
        st, b, _ = http("POST", base + "/update-camera", b"100,200,300,10,75,0")
        assert st == 200 and b == b"updated"
        st, b, _ = http("GET", base + "/camera-pos")
        assert b.decode() == "%f,%f,%f" % tuple(float(np.float32(x)) for x in cfg.camera_pos)
        a, bt = np.float32(10) * np.float32(0.0174532925), np.float32(75) * np.float32(0.0174532925)
        cam = (cfg.camera_pos, (float(np.cos(a)), float(np.sin(a)), float(-np.cos(bt))), (0.0, 0.0, 1.0))

        def moved_image():
            st, body, _ = http("GET", base + "/image")
            if st != 200 or body == got[0]:
                return None
            return body
        moved = wait_for(moved_image, 60)
        assert moved, "the camera update produced no new frame"
        (tmp_path / "moved.png").write_bytes(moved)
        pix = scene_files.read_png_rgb(str(tmp_path / "moved.png"))
        want = expected_png_pixels(asm, 4, SEED, camera=cam)
        # cosf/sinf of the host library vs numpy: allow the last bit of the direction to differ -> compare loosely
        assert (np.abs(pix.astype(int) - want.astype(int)) > 2).mean() < 0.02
        st, b, _ = http("POST", base + "/reset-camera")
        assert st == 200 and b == b"Camera reset"

        def back_image():
            st, body, _ = http("GET", base + "/image")
            return body if st == 200 and body == got[0] else None
        assert wait_for(back_image, 60), "reset-camera did not bring the first frame back"
        assert http("GET", base + "/nothing-here")[0] == 404
        assert http("GET", base + "/../server.log")[0] == 404
        st, b, _ = http("GET", base + "/stop")
        assert st == 200 and b == b"Server stopped"
        assert proc.wait(timeout=60) == 0
    finally:
        if proc.poll() is None:
            proc.kill()
            proc.wait(timeout=10)
        log.close()


def test_bin_main_on_several_contexts(tmp_path):
    """RT_GPU_LIST=0,0,0: three contexts (here on one GPU) each trace their partition of the batches on a host thread of
    their own, the library adds the buffers up (rt_reduce_to_root) and the first context post-processes with the total
    sample count.  Same samples as the one-context frame, batch totals added in another order: a pixel may move by one
    8-bit level where the sum sits on a rounding boundary."""
    asm = scene.material_box(width=64, height=40, n_samples=24)
    path = scene_files.write_scene(asm, str(tmp_path / "files"), name="split")
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, RT_SEED=str(SEED), RT_GPU_LIST="0,0,0")
    r = subprocess.run([MAIN, path, str(out), "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "3 contexts on 1 GPU(s)" in r.stdout and "Image saved to" in r.stdout
    got = scene_files.read_png_rgb(str(out / "split-output.png")).astype(int)
    want = expected_png_pixels(asm, 4, SEED).astype(int)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01


def test_bin_main_on_a_benchmark_scene_file(tmp_path):
    """The benchmark's route through the reference's entry point: a scene file as tools/make_bench_scenes.py writes it
    (model paths relative to the directory bin/main is started in, as in upstream's assets/scene/*.txt), here the formula-1
    configuration at a reduced frame, detail and sample count.  The PNG equals the Python route's byte for byte."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_bench_scenes", os.path.join(ROOT, "tools", "make_bench_scenes.py"))
    mbs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mbs)
    path, asm = mbs.write("formula1", True, root=str(tmp_path), assembler=lambda: scene.formula1(200, 120, 8, detail=0.1, tex_size=64))
    assert os.path.relpath(path, str(tmp_path)) == "benchmarks/scenes/scene_formula1.txt"
    assert "path benchmarks/assets/formula1/formula1_obj0.obj" in open(path).read()
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MAIN, "benchmarks/scenes/scene_formula1.txt", "out", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RT_SEED=str(SEED)), cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "---> done in " in r.stdout
    got = scene_files.read_png_rgb(str(out / "scene_formula1-output.png"))
    want = expected_png_pixels(asm, 4, SEED)
    assert np.array_equal(got, want), f"{(got != want).any(axis=2).sum()} pixels differ"


def test_bin_main_with_the_tree_built_on_the_device(tmp_path):
    """RT_GPU_BVH=1: the loader builds no trees, the library builds one on the device.  Same seed, same draws; the picture
    is the default one except where a ray meets an exact tie or a grazing disagreement (another tree, another visiting
    order), which this closed scene's shared edges do produce in a few pixels."""
    asm = scene.material_box(width=64, height=40, n_samples=16)
    path = scene_files.write_scene(asm, str(tmp_path / "files"), name="boxes")
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MAIN, path, str(out), "4"], capture_output=True, text=True, timeout=120, env=dict(os.environ, RT_SEED=str(SEED), RT_GPU_BVH="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BVH tree: left to the device" in r.stdout and "Image saved to" in r.stdout
    got = scene_files.read_png_rgb(str(out / "boxes-output.png")).astype(int)
    want = expected_png_pixels(asm, 4, SEED).astype(int)
    assert (np.abs(got - want).max(axis=2) > 8).mean() < 0.05 and abs(got.mean() - want.mean()) < 1.0
