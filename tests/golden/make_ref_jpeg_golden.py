#!/usr/bin/env python3
"""Writes tests/golden/ref_jpeg.npz: JPEG files (written here with Pillow, which exists in the build container only) and
what THE REFERENCE'S vendored stb_image decodes from them (stbi_load(..., 4) through oracle/_ref, `make -C oracle ref`).
Data only: file bytes and rgba8 arrays.  tests/test_ref_pin_cpu.py holds the library's JPEG decoder to them bit for bit."""
import ctypes as C
import io
import os
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def picture(rng, w, h, chans):
    """smooth gradients + texture + a few hard edges: exercises DC prediction, long zero runs and saturating pixels"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, chans))
    for c in range(chans):
        out[..., c] = 128 + 100 * np.sin(x / (3.0 + 2 * c) + c) * np.cos(y / (4.0 + c)) + rng.normal(0, 12, (h, w))
    out[h // 3:h // 3 + max(1, h // 6), :, :] = 255
    out[:, w // 2:w // 2 + max(1, w // 8), 0] = 0
    return np.clip(out, 0, 255).astype(np.uint8)


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfovpt_ref.so"))
    rng = np.random.default_rng(23)
    files = {}

    def add(name, arr, mode, **kw):
        buf = io.BytesIO()
        Image.fromarray(arr if arr.shape[2] > 1 else arr[..., 0], mode).save(buf, "JPEG", **kw)
        files[name] = buf.getvalue()
    sizes = [(1, 1), (7, 5), (16, 16), (33, 17), (50, 41), (8, 24)]
    for (w, h) in sizes:
        for sub in (0, 1, 2):
            add("base_%dx%d_s%d.jpg" % (w, h, sub), picture(rng, w, h, 3), "RGB", quality=int(rng.choice([35, 75, 92])), subsampling=sub)
    for (w, h) in [(9, 7), (40, 24)]:
        add("gray_%dx%d.jpg" % (w, h), picture(rng, w, h, 1), "L", quality=80)
        add("gray_prog_%dx%d.jpg" % (w, h), picture(rng, w, h, 1), "L", quality=60, progressive=True)
        for sub in (0, 2):
            add("prog_%dx%d_s%d.jpg" % (w, h, sub), picture(rng, w, h, 3), "RGB", quality=70, subsampling=sub, progressive=True)
        add("opt_%dx%d.jpg" % (w, h), picture(rng, w, h, 3), "RGB", quality=85, subsampling=1, optimize=True)
        add("cmyk_%dx%d.jpg" % (w, h), picture(rng, w, h, 4), "CMYK", quality=80)
    add("q5_23x19.jpg", picture(rng, 23, 19, 3), "RGB", quality=5, subsampling=2)
    add("q100_23x19.jpg", picture(rng, 23, 19, 3), "RGB", quality=100, subsampling=0)
    for kw, tag in (({"restart_marker_blocks": 3}, "rstb"), ({"restart_marker_rows": 1}, "rstr")):
        try:
            add("%s_37x29.jpg" % tag, picture(rng, 37, 29, 3), "RGB", quality=75, subsampling=2, **kw)
            add("%s_prog_37x29.jpg" % tag, picture(rng, 37, 29, 3), "RGB", quality=75, subsampling=1, progressive=True, **kw)
        except Exception as e:
            print("skipped", tag, e)
    for sub in ("4:1:1", "4:4:0"):
        try:
            add("sub%s_34x18.jpg" % sub.replace(":", ""), picture(rng, 34, 18, 3), "RGB", quality=75, subsampling=sub)
        except Exception as e:
            print("skipped subsampling", sub, e)
    try:
        add("keeprgb_21x13.jpg", picture(rng, 21, 13, 3), "RGB", quality=90, keep_rgb=True)
    except Exception as e:
        print("skipped keep_rgb", e)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, data in files.items():
            path = os.path.join(d, name)
            with open(path, "wb") as f:
                f.write(data)
            w, hh = C.c_int(0), C.c_int(0)
            assert L.ref_stbi_load(path.encode(), C.byref(w), C.byref(hh), None, C.c_size_t(0)), name
            px = np.zeros((hh.value, w.value, 4), np.uint8)
            L.ref_stbi_load(path.encode(), C.byref(w), C.byref(hh), px.ctypes.data_as(C.c_void_p), C.c_size_t(px.size))
            out["file:" + name] = np.frombuffer(data, np.uint8)
            out["stbi_load:" + name] = px
    np.savez_compressed(os.path.join(HERE, "ref_jpeg.npz"), **out)
    print("ref_jpeg.npz: %d files, %d bytes of JPEG" % (len(files), sum(len(v) for v in files.values())))


if __name__ == "__main__":
    main()
