"""Writes an assembled scene out as the files upstream's loader reads: a scene text file
(grammar of csrc/sceneloader.cpp:254-321 and :57-96), one OBJ (+ MTL, + PNG textures) per object block.

Used to drive the C++ host side (bin/main, bin/server, bin/scene_dump) with the same procedural scenes the
Python plumbing feeds to the library directly, so both routes can be compared array for array.
"""
import os
import struct
import zlib

import numpy as np


def _g(x):
    """Shortest decimal that round-trips a binary32 value."""
    return "%.9g" % float(np.float32(x))


def write_png_rgb(path, rgb):
    """8-bit RGB PNG, rows top to bottom (rgb: uint8 array (h, w, 3))."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape

    def chunk(kind, data):
        body = kind + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) +
                chunk(b"IEND", b""))


def read_png_rgb(path):
    """Decoder for the PNGs this package writes (8-bit RGB, any row filter, not interlaced) -> uint8 (h, w, 3)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) & 0xFFFFFFFF == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0], "bad CRC"
        if kind == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, interlace) == (8, 2, 0), "only 8-bit RGB, not interlaced"
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), dtype=np.uint8)
    for y in range(h):
        f, row = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(3 * w, dtype=np.int32)
        if f == 0:
            cur = row
        elif f == 2:
            cur = (row + up) & 255
        else:
            cur = np.zeros(3 * w, dtype=np.int32)
            for i in range(3 * w):
                a = cur[i - 3] if i >= 3 else 0
                c = up[i - 3] if i >= 3 else 0
                b = up[i]
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (row[i] + pred) & 255
        out[y] = cur.astype(np.uint8)
    return out.reshape(h, w, 3)


def write_obj(model, obj_path, mtl_name=None, tex_prefix="tex"):
    """One OBJ per model: every triangle corner gets its own v / vt / vn line (no sharing), one `o` per shape,
    `usemtl` where the face material changes.  Returns the MTL file name written next to it (or None)."""
    directory = os.path.dirname(obj_path)
    lines = []
    if model.mtls:
        mtl_name = mtl_name or os.path.splitext(os.path.basename(obj_path))[0] + ".mtl"
        lines.append("mtllib %s" % mtl_name)
        with open(os.path.join(directory, mtl_name), "w") as f:
            for k, m in enumerate(model.mtls):
                f.write("newmtl %s\nillum %d\nKd %s %s %s\nKe %s %s %s\nNi %s\nd %s\n" % (
                    m["name"], m["illum"], _g(m["Kd"][0]), _g(m["Kd"][1]), _g(m["Kd"][2]), _g(m["Ke"][0]), _g(m["Ke"][1]), _g(m["Ke"][2]),
                    _g(m["Ni"]), _g(m["d"])))
                if m["map_Kd"] is not None:
                    tex_file = "%s_%s_%d.png" % (tex_prefix, os.path.splitext(mtl_name)[0], k)
                    tex = np.asarray(m["map_Kd"], dtype=np.uint8)
                    write_png_rgb(os.path.join(directory, tex_file), tex[::-1, :, :3])      # stored bottom-up, files are top-down
                    f.write("map_Kd %s\n" % tex_file)
                f.write("\n")
    # Vertex data goes out through numpy ("%.9g" is the shortest text that round-trips every binary32 value), faces in
    # runs of one material; the text is the same as a line-by-line writer would produce.
    nv = nt = nn = 0
    current = None          # material name in force
    with open(obj_path, "w") as f:
        f.write("".join(line + "\n" for line in lines))
        for shp in model.shapes:
            f.write("o %s\n" % shp.name)
            n = len(shp)
            np.savetxt(f, shp.pos.reshape(-1, 3).astype(np.float64), fmt="v %.9g %.9g %.9g")
            if shp.has_uvs:
                np.savetxt(f, shp.uv.reshape(-1, 2).astype(np.float64), fmt="vt %.9g %.9g")
            if shp.has_normals:
                np.savetxt(f, shp.nrm.reshape(-1, 3).astype(np.float64), fmt="vn %.9g %.9g %.9g")
            corner = np.arange(3 * n, dtype=np.int64).reshape(n, 3) + 1
            cols, fmt = [nv + corner], "%d"
            if shp.has_uvs and shp.has_normals:
                cols, fmt = [nv + corner, nt + corner, nn + corner], "%d/%d/%d"
            elif shp.has_uvs:
                cols, fmt = [nv + corner, nt + corner], "%d/%d"
            elif shp.has_normals:
                cols, fmt = [nv + corner, nn + corner], "%d//%d"
            table = np.stack(cols, axis=2).reshape(n, -1)              # per face: corner 0's indices, corner 1's, corner 2's
            face_fmt = "f " + " ".join([fmt] * 3)
            mtl_ids = np.asarray(shp.mtl)
            starts = [0] + [int(i) for i in np.nonzero(mtl_ids[1:] != mtl_ids[:-1])[0] + 1] + [n]
            for lo, hi in zip(starts[:-1], starts[1:]):
                if hi <= lo:
                    continue
                m = int(mtl_ids[lo])
                name = model.mtls[m]["name"] if m >= 0 else None
                if name != current and (name is not None or current is not None):
                    f.write("usemtl %s\n" % (name if name is not None else "__none__"))     # an unknown name means "no material"
                    current = name
                np.savetxt(f, table[lo:hi], fmt=face_fmt)
            nv += 3 * n
            nt += 3 * n if shp.has_uvs else 0
            nn += 3 * n if shp.has_normals else 0
    return mtl_name if model.mtls else None


def write_scene(asm, directory, name="scene", relative_to=None, scene_path=None, write_models=True):
    """Writes <directory>/<name>.txt plus the model files; returns the scene file's path.  Paths inside the scene
    file are absolute, so the file can be loaded from any working directory -- unless `relative_to` names a directory:
    then they are relative to it (upstream's scene files name their models relative to the repository root, where
    bin/main is started).  `scene_path` puts the text file somewhere else than the model files.  Blocks that share one
    model object share one OBJ file.  write_models=False writes the text file only."""
    os.makedirs(directory, exist_ok=True)
    directory = os.path.abspath(directory)

    def shown(path):
        return os.path.relpath(path, os.path.abspath(relative_to)) if relative_to else path

    cfg = asm.config
    out = ["scene_config", "", "# written by raytrace_amd.scene_files", "",
           "max_ray_depth %d" % cfg.max_depth, "max_surface 1000000",
           "camera_pos %s %s %s" % tuple(_g(x) for x in cfg.camera_pos),
           "camera_dir %s %s %s" % tuple(_g(x) for x in cfg.camera_dir),
           "camera_up %s %s %s" % tuple(_g(x) for x in cfg.camera_up),
           "fov %s" % _g(cfg.fov_deg), "gamma %s" % _g(cfg.gamma), "size %d %d" % (cfg.width, cfg.height),
           "sample_per_pixel %d" % cfg.n_samples, "background %s %s %s" % tuple(_g(x) for x in cfg.background),
           "russian_roulette %s" % _g(cfg.russian_roulette), ""]
    written = {}
    for k, blk in enumerate(asm.blocks):
        obj_file = written.get(id(blk.model))
        if obj_file is None:
            obj_file = os.path.join(directory, "%s_obj%d.obj" % (name, k))
            if write_models:
                write_obj(blk.model, obj_file)
            written[id(blk.model)] = obj_file
        out += ["object", "    path %s" % shown(obj_file), "    mtldir %s" % (shown(directory) + ("/" if relative_to else "")),
                "    position %s %s %s" % tuple(_g(x) for x in blk.position),
                "    rotation %s %s %s" % tuple(_g(x) for x in blk.rotation),
                "    scale %s %s %s" % tuple(_g(x) for x in blk.scale),
                "    albedo %s %s %s" % tuple(_g(x) for x in blk.albedo),
                "    type %s" % blk.type, "    metal_fuzz %s" % _g(blk.metal_fuzz), "    optical_density %s" % _g(blk.optical_density)]
        if blk.disable_normal:
            out.append("    disable_normal")
        if blk.simple_material:
            out += ["    # every face takes the object's own material", "    simple_material"]
        out += ["end", ""]
    out += ["end", ""]
    path = scene_path or os.path.join(directory, name + ".txt")
    with open(path, "w") as f:
        f.write("\n".join(out))
    return path
