#!/usr/bin/env python3
"""Writes tests/golden/ref_vectors.npz and tests/golden/ref_loaders.npz: inputs and the outputs THE REFERENCE'S OWN
CODE gives for them, through oracle/_ref/libfovpt_ref.so (= oracle/ref_shim.cpp over the reference's headers,
Model.cpp, sutil/Camera.cpp and its vendored tinyobjloader / stb_image, compiled where they lie under
/root/reference by `make -C oracle ref`).

Run in the build container (needs /root/reference):   python tests/golden/make_ref_golden.py
The .npz files are data (seeded inputs, expected outputs, small input files as byte arrays); no reference text.
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_cases  # noqa: E402
from common import encode_hdr_rle, encode_png, encode_tga, tga_rle  # noqa: E402

OBJ = """# three shapes, three materials, quads, negative indices, all corner syntaxes, a texture used by two shapes,
# a corner shared by two materials of one shape, a (shape, material) whose corners are all known already (its mesh is
# dropped, its texture is loaded all the same and takes an id)
mtllib scene.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 0.25 0.5 2
v 1.5 0.5 2
v 1.5 1.75 2
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vt 0.25 0.75
vn 0 0 1
vn 0 1 0
o wall
usemtl red
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl lamp
f 5//1 6//1 -5//2
usemtl red
f 7/5/2 8/2/2 9/3/2
g floor
usemtl red
f 1 2 6 5
f -1 -2 -6
usemtl tiles
f 1/1 2/2 6/3
f 1/1 6/3 5/4
g shared
usemtl red
f 1 2 3
f 2 3 4
usemtl lamp
f 5 6 7
f 6 7 8
f 3 4 5
usemtl tiles
f 1 2 3
g after
usemtl tiles
f 1/1 2/2 6/3
"""
MTL = """newmtl red
Kd 0.8 0.1 0.2
map_Kd -s 1 1 1 tex.ppm
newmtl lamp
Kd 1 1 1
Ke 5 4 3
newmtl tiles
Kd 0.5 0.5 0.5
map_Kd tiles.png
"""


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


dump_model = ref_cases.dump_model


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfovpt_ref.so"))

    # ---- unit-level vectors (tests/ref_cases.py) ----
    inp, out = ref_cases.run_all(L, "ref_")
    blob = {"in:" + k: v for k, v in inp.items()}
    blob.update({"out:" + k: v for k, v in out.items()})
    # layouts and constructor defaults (a2/a3)
    mat = np.zeros(104, np.uint8)
    assert L.ref_material_default(_p(mat), 104) == 104
    offs = np.zeros(32, np.int32)
    n = L.ref_material_offsets(_p(offs), 32)
    pst = np.zeros(9, np.int32)
    L.ref_probe_struct(_p(pst))
    blob["layout:material_default_bytes"] = mat
    blob["layout:material_offsets"] = offs[:n]
    blob["layout:probe_sizeof_and_offsets"] = pst
    blob["layout:trianglemesh_default_texture_id"] = np.int32(L.ref_trianglemesh_default_texture_id())
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **blob)
    print("ref_vectors.npz: %d arrays" % len(blob))

    # ---- loaders: loadOBJ, addBox, stbi_load, stbi_loadf ----
    rng = np.random.default_rng(11)
    files = {"scene.obj": OBJ.encode(), "scene.mtl": MTL.encode()}
    px = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8)
    files["tex.ppm"] = b"P6\n2 2\n255\n" + px.tobytes()
    files["tiles.png"] = encode_png(rng.integers(0, 256, (5, 7, 4)), 8, 6)
    pngs = {
        "gray8.png": encode_png(rng.integers(0, 256, (6, 9, 1)), 8, 0),
        "gray1.png": encode_png(rng.integers(0, 2, (7, 11, 1)), 1, 0),
        "gray16.png": encode_png(rng.integers(0, 65536, (4, 5, 1)), 16, 0),
        "rgb8.png": encode_png(rng.integers(0, 256, (6, 5, 3)), 8, 2),
        "rgb16_trns.png": encode_png(np.concatenate([np.full((1, 4, 3), 513), rng.integers(0, 65536, (3, 4, 3))]), 16, 2,
                                     trns=bytes([2, 1, 2, 1, 2, 1])),
        "pal4.png": encode_png(rng.integers(0, 16, (5, 9, 1)), 4, 3, plte=bytes(rng.integers(0, 256, 48).tolist()),
                               trns=bytes(rng.integers(0, 256, 7).tolist())),
        "ga8.png": encode_png(rng.integers(0, 256, (3, 4, 2)), 8, 4),
        "rgba16_adam7.png": encode_png(rng.integers(0, 65536, (9, 10, 4)), 16, 6, interlace=True),
        "rgb8_adam7.png": encode_png(rng.integers(0, 256, (5, 3, 3)), 8, 2, interlace=True),
    }
    files.update(pngs)
    # Truevision TGA, the texture format of the reference's default scene (crytek_sponza, main.cpp:196): true colour 24 / 32
    # bits bottom-up and top-down, run-length packets, gray, gray + alpha, 5-5-5, colour-mapped with indices past the palette
    tw, th = 7, 5
    t24, t32 = rng.integers(0, 256, (tw * th, 3), dtype=np.uint8), rng.integers(0, 256, (tw * th, 4), dtype=np.uint8)
    r24, r32 = rng.integers(0, 256, (tw * th, 3), dtype=np.uint8), rng.integers(0, 256, (tw * th, 4), dtype=np.uint8)
    g8, ga16 = rng.integers(0, 256, (tw * th, 1), dtype=np.uint8), rng.integers(0, 256, (tw * th, 2), dtype=np.uint8)
    v555 = rng.integers(0, 1 << 16, tw * th).astype("<u2")
    pal24 = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    pal_idx = rng.integers(0, 20, (tw * th, 1), dtype=np.uint8)
    tgas = {
        "tc24.tga": encode_tga(tw, th, 2, 24, t24.tobytes()),
        "tc32_top.tga": encode_tga(tw, th, 2, 32, t32.tobytes(), desc=0x28),
        "rle24.tga": encode_tga(tw, th, 10, 24, tga_rle(r24)),
        "rle32_top.tga": encode_tga(tw, th, 10, 32, tga_rle(r32), desc=0x20, idfield=b""),
        "gray8_rle.tga": encode_tga(tw, th, 11, 8, tga_rle(g8), desc=0x20),
        "gray_alpha16.tga": encode_tga(tw, th, 3, 16, ga16.tobytes()),
        "rgb555.tga": encode_tga(tw, th, 2, 16, v555.tobytes(), desc=0x20),
        "pal8.tga": encode_tga(tw, th, 1, 8, pal_idx.tobytes(), cmap=pal24.tobytes(), cm_len=16, cm_bits=24),
    }
    files.update(tgas)
    rgbe = rng.integers(0, 256, (6, 16, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(118, 142, (6, 16))
    rgbe[2, :, 0] = 7
    rgbe[4, 3:9] = 0
    files["probe_rle.hdr"] = encode_hdr_rle(rgbe)
    flat = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5\n" + rng.integers(1, 256, (3, 5, 4), dtype=np.uint8).tobytes()
    files["probe_flat.hdr"] = flat

    L.ref_load_obj.restype = C.c_void_p
    L.ref_box_model.restype = C.c_void_p
    L.ref_model_free.argtypes = [C.c_void_p]
    lb = {}
    with tempfile.TemporaryDirectory() as d:
        for name, data in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(data)
            lb["file:" + name] = np.frombuffer(data, np.uint8)
        h = L.ref_load_obj(os.path.join(d, "scene.obj").encode())
        assert h, "reference loadOBJ failed"
        lb.update({"obj:" + k: v for k, v in dump_model(L, h).items()})
        L.ref_model_free(h)
        for name in list(pngs) + list(tgas) + ["tex.ppm", "tiles.png"]:
            w, hh = C.c_int(0), C.c_int(0)
            assert L.ref_stbi_load(os.path.join(d, name).encode(), C.byref(w), C.byref(hh), None, C.c_size_t(0)), name
            out8 = np.zeros((hh.value, w.value, 4), np.uint8)
            L.ref_stbi_load(os.path.join(d, name).encode(), C.byref(w), C.byref(hh), _p(out8), C.c_size_t(out8.size))
            lb["stbi_load:" + name] = out8
        for name in ("probe_rle.hdr", "probe_flat.hdr", "rgb8.png", "rgba16_adam7.png", "tex.ppm"):
            w, hh = C.c_int(0), C.c_int(0)
            assert L.ref_stbi_loadf(os.path.join(d, name).encode(), C.byref(w), C.byref(hh), None, C.c_size_t(0)), name
            outf = np.zeros((hh.value, w.value, 4), np.float32)
            L.ref_stbi_loadf(os.path.join(d, name).encode(), C.byref(w), C.byref(hh), _p(outf), C.c_size_t(outf.size))
            lb["stbi_loadf:" + name] = outf
    center, half = np.float32([1.5, -2.0, 0.25]), np.float32([0.5, 1.25, 3.0])
    h = L.ref_box_model(_p(center), _p(half))
    lb["box:center"], lb["box:half"] = center, half
    lb.update({"box:" + k: v for k, v in dump_model(L, h).items()})
    L.ref_model_free(h)
    np.savez_compressed(os.path.join(HERE, "ref_loaders.npz"), **lb)
    print("ref_loaders.npz: %d arrays" % len(lb))


if __name__ == "__main__":
    main()
