"""Host side above the C ABI, part 1: the OBJ/MTL reader (raytrace_amd/host/obj_model.cpp) through bin/obj_dump.

Parity unpinned by an executable reference: upstream reads models with tinyobjloader, which could not be run in
this build; the expectations below are derived by hand from its published behaviour (third/tiny_obj_loader.h,
lines cited per test) and the frozen dumps only guard against regressions.
"""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tests", "golden", "obj")
TOOL = os.path.join(ROOT, "bin", "obj_dump")

pytestmark = pytest.mark.skipif(not os.path.exists(TOOL), reason="bin/obj_dump not built (make host)")


def dump(path, *extra):
    r = subprocess.run([TOOL, path, *extra], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout


def f32(hexword):
    return struct.unpack(">f", bytes.fromhex(hexword))[0]


def parse(text):
    """dump text -> dict(vertices, shapes [(name, [(corners, material)])], materials [dict])"""
    lines = text.splitlines()
    out = {"vertices": [], "normals": [], "texcoords": [], "shapes": [], "materials": []}
    i = 0
    for key in ("vertices", "normals", "texcoords"):
        n = int(lines[i].split()[1]); i += 1
        out[key] = [tuple(f32(w) for w in lines[i + k].split()) for k in range(n)]
        i += n
    ns = int(lines[i].split()[1]); i += 1
    for _ in range(ns):
        head = lines[i]; i += 1
        name = head[head.index("[") + 1:head.rindex("]")]
        n = int(head.split()[-1])
        tris = []
        for k in range(n):
            parts = lines[i + k].split()
            tris.append((tuple(tuple(int(x) for x in c.split("/")) for c in parts[:3]), int(parts[4])))
        i += n
        out["shapes"].append((name, tris))
    nm = int(lines[i].split()[1]); i += 1
    for k in range(nm):
        ln = lines[i + k]
        name = ln[ln.index("[") + 1:ln.index("]")]
        rest = ln[ln.index("]") + 1:].split()
        tex = ln[ln.rindex("[") + 1:ln.rindex("]")]
        out["materials"].append({"name": name, "illum": int(rest[1]), "Kd": tuple(f32(w) for w in rest[3:6]), "Ke": tuple(f32(w) for w in rest[7:10]),
                                 "Ni": f32(rest[11]), "d": f32(rest[13]), "map_Kd": tex})
    return out


@pytest.mark.parametrize("name", ["mixed", "tri_usemtl_last", "empty_mtl"])
def test_frozen_dumps(name):
    rc, text = dump(os.path.join(OBJ, name + ".obj"))
    assert rc == 0
    assert text == open(os.path.join(OBJ, name + ".dump")).read()


def test_number_grammar():
    """tryParseDouble, tiny_obj_loader.h:897-1028: leading dot, explicit plus, bare trailing dot, exponent forms;
    a word that is not a number leaves the default 0; digits followed by junk keep the digits."""
    m = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])
    assert m["vertices"][7] == (0.5, -5.0, 2.0)                       # ".5 -.5e1 +2" (the three colour values are ignored)
    assert m["vertices"][8] == (1.0, 150.0, f32("3a83126f"))          # "1. 1.5E+2 1e-3" -> 0.001f
    assert m["vertices"][9] == (0.0, 12.0, 7.0)                       # "abc 12abc 7"
    assert m["texcoords"][1] == (1.0, 0.0)                            # "vt 1": v defaults to 0 (:1053-1058)
    assert m["texcoords"][2] == (1.0, 1.0)                            # third component dropped


def test_quads_split_along_the_shorter_diagonal():
    """exportGroupsToShape, tiny_obj_loader.h:1510-1605: d02 < d13 -> (0,1,2),(0,2,3), otherwise (0,1,3),(1,2,3)."""
    tris = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])["shapes"][0][1]
    assert [t[0][0][0] for t in tris[:2]] == [0, 1] and tris[0][0] == ((0, -1, -1), (1, -1, -1), (3, -1, -1))   # square: equal diagonals
    assert tris[1][0] == ((1, -1, -1), (2, -1, -1), (3, -1, -1))
    assert tris[2][0] == ((0, 0, -1), (1, 1, -1), (4, 2, -1)) and tris[3][0] == ((0, 0, -1), (4, 2, -1), (5, 3, -1))  # d02 4.25 < d13 10.25


def test_index_forms_and_relative_indices():
    """parseTriple/fixIndex, tiny_obj_loader.h:819-852, 1188-1239: v, v/vt, v//vn, v/vt/vn, negative = relative to the
    count so far, zero allowed (as -1) for vt and vn only."""
    shapes = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])["shapes"]
    tris = shapes[0][1]
    assert tris[4][0] == ((0, -1, 0), (1, -1, 0), (2, -1, 0))          # 1//1 2//1 3//1
    assert tris[5][0] == ((0, 0, 0), (1, 1, 1), (2, 2, 2))             # 1/1/1 2/2/2 3/3/3
    assert {c[0] for t in tris[6:8] for c in t[0]} == {18, 19, 20, 21}  # f -4 -3 -2 -1 with 22 vertices read
    assert shapes[2][1][0][0] == ((0, -1, -1), (1, -1, -1), (2, -1, -1))   # 1/0 2/0 3/0
    for bad in ("bad_zero_index", "bad_relative_index"):
        rc, text = dump(os.path.join(OBJ, bad + ".obj"))
        assert rc == 1 and text.strip() == "error"


def test_polygon_ear_clipping_by_hand():
    """The pentagon (3,0) (4,0) (4,1) (3.5,0.4) (3,1) -- vertex 3 is reflex and lies inside the first candidate ear:
    ears come out as (1,2,3), (3,4,0), (0,1,3) in the order of tiny_obj_loader.h:1799-1962 (worked by hand)."""
    tris = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])["shapes"][1][1]
    assert [tuple(c[0] for c in t[0]) for t in tris[:3]] == [(11, 12, 13), (13, 14, 10), (10, 11, 13)]
    # every polygon yields n - 2 triangles: pentagon 3, hexagon 4, L-shaped hexagon 4
    assert len(tris) == 3 + 4 + 4 + 1


def test_shapes_groups_and_materials():
    """LoadObj, tiny_obj_loader.h:2849-3011, 3130-3139: `g a b  c` joins names with one blank; a bare `g` line is not a
    group statement; `o` keeps the rest of the line verbatim; usemtl splits faces by material inside one shape; an
    unknown material name gives -1; a shape of lines only is kept by `o`; a trailing usemtl does not drop the shape."""
    m = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])
    names = [s[0] for s in m["shapes"]]
    assert names == ["", "first second third", " object with blanks  ", "onlylines", "tail"]
    assert [t[1] for t in m["shapes"][0][1]] == [-1, -1, -1, -1, 0, 0, 0, 0]
    assert [t[1] for t in m["shapes"][1][1]] == [0, 0, 0, 1, 1, 1, 1, -1, -1, -1, -1, -1]
    assert len(m["shapes"][3][1]) == 0 and [t[1] for t in m["shapes"][4][1]] == [2]
    one = parse(dump(os.path.join(OBJ, "tri_usemtl_last.obj"))[1])
    assert len(one["shapes"]) == 1 and len(one["shapes"][0][1]) == 1 and one["materials"] == []


def test_mtl_rules():
    """LoadMtl, tiny_obj_loader.h:2069-2467: map_Kd without any Kd so far gives Kd 0.6 (and only then: the flag is never
    reset); `d` wins over `Tr`; Tr alone means d = 1 - Tr; texture options are skipped and the file name runs to the
    end of the line; a later record with a used name is stored but lookups keep the first; `newmtl` without a name is
    not a statement; an empty MTL file yields one default record."""
    m = parse(dump(os.path.join(OBJ, "mixed.obj"))[1])["materials"]
    assert [k["name"] for k in m] == ["red", "glass", "emit", "mirror", "red"]
    assert m[0]["Kd"] == (f32("3f19999a"),) * 3 and m[0]["map_Kd"] == "tex one.png" and m[0]["illum"] == 2
    assert m[1]["d"] == 0.25 and m[1]["Ni"] == f32("3fb9999a") and m[1]["map_Kd"] == "glass.png" and m[1]["illum"] == 4
    assert m[2]["Ke"] == (10.0, 9.5, 8.0) and m[2]["d"] == f32("3f333333") and m[2]["Kd"] == (0.0, 0.0, 0.0) and m[2]["map_Kd"] == "later.png"
    assert m[3]["illum"] == 5 and m[3]["Kd"] == (1.0, 1.0, 1.0)
    assert m[4]["illum"] == 7 and m[4]["Kd"] == (0.5, 0.5, 0.5)          # the nameless `newmtl` line did not start a record
    e = parse(dump(os.path.join(OBJ, "empty_mtl.obj"))[1])
    assert len(e["materials"]) == 1 and e["materials"][0]["name"] == "" and e["materials"][0]["d"] == 1.0
    assert [t[1] for t in e["shapes"][0][1]] == [0, 0]                   # `usemtl` with an empty name finds that record


def test_missing_file_is_an_error(tmp_path):
    rc, text = dump(str(tmp_path / "nope.obj"))
    assert rc == 1 and text.strip() == "error"
