"""The readers of untrusted text -- scene file, OBJ, MTL -- under a mutation fuzzer built with AddressSanitizer and
UBSan (tools/host/fuzz_readers.cpp, bin/fuzz_readers).  A reader may reject a damaged file; it may not read out of
bounds, overflow a signed integer or end the process.  The runs here are short (seconds); the same command with 30 000
iterations per seed was run while the readers were hardened (one finding: a directory named by a damaged `mtllib` line
opened as a stream and threw from inside the first read)."""
import os
import subprocess

import pytest

from raytrace_amd import scene, scene_files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "bin", "fuzz_readers")
OBJ_DUMP = os.path.join(ROOT, "bin", "obj_dump")
SCENE_DUMP = os.path.join(ROOT, "bin", "scene_dump")

pytestmark = pytest.mark.skipif(not os.path.exists(FUZZ), reason="bin/fuzz_readers not built (make host)")


def fuzz(kind, seed, work, iterations, rng):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([FUZZ, kind, seed, str(work), str(iterations), str(rng)], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "no sanitizer report" in r.stderr, r.stderr[-2000:]
    return r.stderr


@pytest.mark.parametrize("seed", ["mixed", "tri_usemtl_last", "empty_mtl"])
def test_obj_and_mtl_readers_survive_damaged_files(tmp_path, seed):
    out = fuzz("obj", os.path.join(ROOT, "tests", "golden", "obj", seed + ".obj"), tmp_path / "w", 1500, 17)
    assert " read, " in out


def test_scene_loader_survives_damaged_files(tmp_path):
    asm = scene.material_box(width=16, height=10, n_samples=2, tex_size=8)
    path = scene_files.write_scene(asm, str(tmp_path / "seed"), name="seed")
    fuzz("scene", path, tmp_path / "w", 400, 23)
    # the textured multi-shape model of that scene as an OBJ/MTL seed as well
    fuzz("obj", str(tmp_path / "seed" / "seed_obj0.obj"), tmp_path / "w2", 1500, 29)


def test_a_directory_where_a_file_is_expected_is_rejected_not_read(tmp_path):
    d = tmp_path / "case"
    d.mkdir()
    (d / "sub").mkdir()
    obj = d / "a.obj"
    obj.write_text("mtllib sub\nmtllib .\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    r = subprocess.run([OBJ_DUMP, str(obj), str(d)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "shapes 1" in r.stdout and "materials 0" in r.stdout
    r = subprocess.run([OBJ_DUMP, str(d / "sub")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stdout.strip() == "error"
    r = subprocess.run([SCENE_DUMP, str(d / "sub")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "load_scene -1" in r.stdout
