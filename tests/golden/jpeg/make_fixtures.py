"""Writes the JPEG fixtures of this directory: files encoded by Pillow (libjpeg-turbo) and, beside each, the RGB bytes
that Pillow's decoder (libjpeg-turbo with its defaults: slow-integer inverse DCT, fancy upsampling -- the settings
cv::imread uses upstream, csrc/sceneloader.cpp:224) returns for it, stored as PNG so that they stay small.

    python tests/golden/jpeg/make_fixtures.py

Needs Pillow with JPEG support; the tests read only the files written here.
"""
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
from raytrace_amd.scene_files import write_png_rgb  # noqa: E402


def picture(w, h, seed):
    """Smooth gradients, a few hard edges and some noise: exercises every coefficient range and the chroma filters."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 127 + 120 * np.sin(x / 5.0 + seed) * np.cos(y / 7.0)
    g = 255 * ((x + y) % 32 < 16) * (x / max(w - 1, 1))
    b = 255 * (y / max(h - 1, 1)) ** 2 + 30 * rng.standard_normal((h, w))
    img = np.stack([r, g, b], axis=-1)
    img[h // 3: h // 3 + 3, :, :] = (255, 0, 0)
    img[:, w // 2: w // 2 + 2, :] = (0, 0, 255)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [
    # name, width, height, mode, save options
    ("base_444", 40, 24, "RGB", dict(quality=90, subsampling=0)),
    ("base_422", 37, 19, "RGB", dict(quality=85, subsampling=1)),
    ("base_420", 45, 27, "RGB", dict(quality=75, subsampling=2)),
    ("base_420_16", 16, 16, "RGB", dict(quality=95, subsampling=2)),
    ("base_420_tiny", 3, 5, "RGB", dict(quality=90, subsampling=2)),
    ("base_422_narrow", 4, 9, "RGB", dict(quality=90, subsampling=1)),
    ("base_420_opt", 64, 48, "RGB", dict(quality=60, subsampling=2, optimize=True)),
    ("base_q10", 33, 33, "RGB", dict(quality=10, subsampling=2)),
    ("base_q100", 24, 24, "RGB", dict(quality=100, subsampling=0)),
    ("grey", 29, 31, "L", dict(quality=80)),
    ("prog_420", 50, 34, "RGB", dict(quality=80, subsampling=2, progressive=True)),
    ("prog_444", 23, 41, "RGB", dict(quality=92, subsampling=0, progressive=True)),
    ("prog_grey", 35, 18, "L", dict(quality=70, progressive=True)),
    ("restart_420", 70, 40, "RGB", dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    ("restart_prog", 48, 48, "RGB", dict(quality=80, subsampling=1, progressive=True, restart_marker_rows=1)),
    ("rgb_adobe", 30, 20, "RGB", dict(quality=90, keep_rgb=True)),
    ("one_pixel", 1, 1, "RGB", dict(quality=90, subsampling=2)),
    ("photo_420", 200, 152, "RGB", dict(quality=85, subsampling=2)),
    ("photo_prog_422", 131, 77, "RGB", dict(quality=50, subsampling=1, progressive=True, optimize=True)),
    ("cmyk", 16, 16, "CMYK", dict(quality=90)),           # refused by the decoder: no expectation file
]


def main():
    for name, w, h, mode, opts in CASES:
        src = picture(w, h, len(name))
        im = Image.fromarray(src, "RGB").convert(mode)
        buf = io.BytesIO()
        im.save(buf, "JPEG", **opts)
        data = buf.getvalue()
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        if mode == "CMYK":
            continue
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert dec.shape == (h, w, 3)
        write_png_rgb(os.path.join(HERE, name + ".expect.png"), dec)
        print(name, len(data), "bytes")


if __name__ == "__main__":
    main()
