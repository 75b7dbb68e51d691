"""Scene ingestion on the caller's side of the path: Wavefront OBJ (+MTL, diffuse texture) -> Model,
with the semantics of the reference's loadOBJ (PT_sv5_/Model.cpp:138-217), which drives the
vendored tinyobjloader (support/tinyobjloader, LoadObj with triangulate = true) and stb_image.

What is reproduced:
  * one TriangleMesh per (shape, material id), material ids visited in ascending order (:170-176);
    a shape is what tinyobj calls a shape: the faces between `o` / `g` statements
  * vertices de-duplicated on the (position, normal, texcoord) index triple, in first-use order,
    with addVertex's normal/texcoord fill rules (:50-83)
  * Material() constructor defaults with only `color` (Kd) and `emission` (Ke) overridden (:190-191)
  * the diffuse texture (map_Kd) as RGBA8, mirrored along y as the reference does after stbi_load
    (:117-126); a texture that cannot be loaded gives id -1 (:129-131)
Deliberate differences (each only where the reference misbehaves):
  * the de-duplication map is per mesh, not per shape: the reference shares one map between the
    meshes of a shape, so a corner reused under a second material indexes into the wrong mesh
  * faces without `usemtl` keep the Material() defaults instead of reading materials[-1]
  * textures are cached per model, not per shape (no duplicate uploads)
  * polygons with more than 3 corners are fan-triangulated from their first corner, which is what
    tinyobj's ear clipping yields for convex polygons; concave polygons may be cut differently
"""
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .abi import Material
from .scenes import Model, TriangleMesh


def _parse_mtl(path: str) -> Dict[str, dict]:
    mats: Dict[str, dict] = {}
    cur = None
    if not os.path.exists(path):
        return mats
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split("#", 1)[0].split()
            if not t:
                continue
            k = t[0]
            if k == "newmtl":
                cur = {"Kd": None, "Ke": (0.0, 0.0, 0.0), "map_Kd": ""}
                mats[" ".join(t[1:])] = cur
            elif cur is None:
                continue
            elif k == "Kd" and len(t) >= 4:
                cur["Kd"] = tuple(float(x) for x in t[1:4])
            elif k == "Ke" and len(t) >= 4:
                cur["Ke"] = tuple(float(x) for x in t[1:4])
            elif k == "map_Kd" and len(t) >= 2:
                cur["map_Kd"] = t[-1]                      # options (-s, -o, ...) precede the file name
    for m in mats.values():                                # tinyobj initialises diffuse to 0 when Kd is absent
        if m["Kd"] is None:
            m["Kd"] = (0.0, 0.0, 0.0)
    return mats


_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))   # x0, y0, dx, dy


def _png_unfilter(raw: memoryview, pos: int, rows: int, rowbytes: int, bpp: int) -> Tuple[np.ndarray, int]:
    """Undo the per-scanline filters (PNG 1.2 section 6) of `rows` scanlines starting at raw[pos]."""
    out = np.zeros((rows, rowbytes), np.uint8)
    prev = np.zeros(rowbytes, np.uint8)
    for y in range(rows):
        ft = raw[pos]
        line = np.frombuffer(raw[pos + 1:pos + 1 + rowbytes], np.uint8)
        if len(line) != rowbytes:
            raise ValueError("png: truncated image data")
        pos += 1 + rowbytes
        if ft == 0:
            cur = line.copy()
        elif ft == 1:                                            # Sub: a running sum per channel
            cur = line.copy()
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp], dtype=np.uint64).astype(np.uint8)
        elif ft == 2:                                            # Up
            cur = line + prev
        elif ft in (3, 4):                                       # Average, Paeth: sequential along the row
            cur_l = [0] * rowbytes
            ln, pv = line.tolist(), prev.tolist()
            if ft == 3:
                for i in range(rowbytes):
                    a = cur_l[i - bpp] if i >= bpp else 0
                    cur_l[i] = (ln[i] + ((a + pv[i]) >> 1)) & 255
            else:
                for i in range(rowbytes):
                    a = cur_l[i - bpp] if i >= bpp else 0
                    b = pv[i]
                    c = pv[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur_l[i] = (ln[i] + pr) & 255
            cur = np.array(cur_l, np.uint8)
        else:
            raise ValueError("png: bad filter type")
        out[y] = cur
        prev = cur
    return out, pos


def decode_png(data: bytes) -> np.ndarray:
    """PNG -> (H, W, 4) uint8 as stbi_load(..., STBI_rgb_alpha) returns it (support/stb_image via
    Model.cpp:106-107): all colour types, bit depths 1-16 (16-bit samples keep their high byte, gray
    of 1/2/4 bits is scaled to 0..255), palette and tRNS transparency (a colour key gives alpha 0),
    Adam7 interlacing; gamma and colour-profile chunks are ignored."""
    import struct
    import zlib
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("png: bad signature")
    pos, idat, plte, trns, hdr = 8, [], None, None, None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None or not idat:
        raise ValueError("png: no header or no image data")
    w, h, depth, ctype, comp, filt, interlace = hdr
    if w == 0 or h == 0 or comp or filt or interlace > 1 or ctype not in (0, 2, 3, 4, 6) or depth not in (1, 2, 4, 8, 16):
        raise ValueError("png: unsupported header")
    if (ctype == 3 and depth == 16) or (ctype in (2, 4, 6) and depth < 8) or (ctype == 3 and plte is None):
        raise ValueError("png: bad colour type / depth")
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = memoryview(zlib.decompress(b"".join(idat)))
    bpp = max(1, chans * depth // 8)

    def samples(rows: np.ndarray, pw: int) -> np.ndarray:       # (ph, rowbytes) bytes -> (ph, pw, chans) uint16 samples
        if depth == 8:
            return rows.reshape(rows.shape[0], pw, chans).astype(np.uint16)
        if depth == 16:
            v = rows.reshape(rows.shape[0], pw, chans, 2).astype(np.uint16)
            return (v[..., 0] << 8) | v[..., 1]
        bits = np.unpackbits(rows, axis=1)[:, :pw * depth].reshape(rows.shape[0], pw, depth).astype(np.uint16)
        return (bits << np.arange(depth - 1, -1, -1, dtype=np.uint16)).sum(axis=2, dtype=np.uint16)[..., None]

    img = np.zeros((h, w, chans), np.uint16)
    off = 0
    if not interlace:
        rows, off = _png_unfilter(raw, 0, h, (w * chans * depth + 7) // 8, bpp)
        img[:] = samples(rows, w)
    else:
        for x0, y0, dx, dy in _ADAM7:
            pw, ph = (w - x0 + dx - 1) // dx, (h - y0 + dy - 1) // dy
            if pw <= 0 or ph <= 0:
                continue
            rows, off = _png_unfilter(raw, off, ph, (pw * chans * depth + 7) // 8, bpp)
            img[y0::dy, x0::dx] = samples(rows, pw)

    rgba = np.empty((h, w, 4), np.uint8)
    if ctype == 3:
        idx = img[..., 0]
        plte = np.concatenate([plte[:256], np.zeros((256 - min(len(plte), 256), 3), np.uint8)])   # an index past the palette reads black
        alpha = np.full(256, 255, np.uint8)
        if trns is not None:
            alpha[:min(len(trns), 256)] = np.frombuffer(trns, np.uint8)[:256]
        rgba[..., :3] = plte[idx]
        rgba[..., 3] = alpha[idx]
        return rgba
    keyed = None
    if trns is not None and ctype in (0, 2):                    # colour key, compared at the file's precision
        key = np.frombuffer(trns[:len(trns) & ~1], ">u2")[:chans].astype(np.uint16)
        if depth <= 8:
            key &= 255                                           # stb keeps the low byte of an 8-bit key
        if len(key) == chans:
            keyed = np.all(img == key, axis=2)
    if depth == 16:
        v = (img >> 8).astype(np.uint8)
    elif depth == 8:
        v = img.astype(np.uint8)
    else:
        v = (img * {1: 255, 2: 85, 4: 17}[depth]).astype(np.uint8)
    if ctype in (0, 4):
        rgba[..., 0] = rgba[..., 1] = rgba[..., 2] = v[..., 0]
        rgba[..., 3] = v[..., 1] if ctype == 4 else 255
    else:
        rgba[..., :3] = v[..., :3]
        rgba[..., 3] = v[..., 3] if ctype == 6 else 255
    if keyed is not None:
        rgba[..., 3] = np.where(keyed, 0, 255)
    return rgba


def decode_tga(data: bytes) -> np.ndarray:
    """Truevision TGA -> (H, W, 4) uint8 the way stb_image reads it: image types 1/2/3 and their RLE forms
    9/10/11; 8 (gray or index), 15/16 (5-5-5), 24 and 32 bits; rows bottom-up unless descriptor bit 5 is set."""
    import struct
    if len(data) < 18:
        raise ValueError("tga: truncated header")
    idlen, cmap_type, itype, cm_first, cm_len, cm_bits, _x0, _y0, w, h, bits, desc = struct.unpack("<BBBHHBHHHHBB", data[:18])
    rle = itype >= 8
    base = itype & 7
    if base not in (1, 2, 3) or w == 0 or h == 0 or cmap_type > 1 or (base == 1) != (cmap_type == 1):
        raise ValueError("tga: unsupported image type")
    if base == 1 and bits not in (8, 16) or base == 3 and bits not in (8, 16) or base == 2 and bits not in (15, 16, 24, 32):
        raise ValueError("tga: unsupported pixel depth")
    pos = 18 + idlen

    def expand(px: np.ndarray, nbits: int, gray: bool) -> np.ndarray:   # (n, bytes) -> (n, 4) RGBA
        n = px.shape[0]
        out = np.full((n, 4), 255, np.uint8)
        if nbits == 8:
            out[:, 0] = out[:, 1] = out[:, 2] = px[:, 0]
        elif nbits in (15, 16) and gray:                       # gray + alpha
            out[:, 0] = out[:, 1] = out[:, 2] = px[:, 0]
            out[:, 3] = px[:, 1]
        elif nbits in (15, 16):                                # 5-5-5, top bit ignored
            v = px[:, 0].astype(np.uint32) | (px[:, 1].astype(np.uint32) << 8)
            for c, sh in ((0, 10), (1, 5), (2, 0)):
                out[:, c] = (((v >> sh) & 31) * 255 // 31).astype(np.uint8)
        else:
            out[:, 0], out[:, 1], out[:, 2] = px[:, 2], px[:, 1], px[:, 0]
            if nbits == 32:
                out[:, 3] = px[:, 3]
        return out

    palette = None
    if cmap_type:
        if cm_bits not in (8, 15, 16, 24, 32):
            raise ValueError("tga: unsupported palette depth")
        eb = (cm_bits + 7) // 8
        pos += cm_first                                         # stb skips "first entry index" BYTES, then reads cm_len entries
        pal = np.frombuffer(data[pos:pos + cm_len * eb], np.uint8)
        if len(pal) != cm_len * eb:
            raise ValueError("tga: truncated palette")
        pos += cm_len * eb
        if base == 1:
            palette = expand(pal.reshape(cm_len, eb), cm_bits, False)
    nb = (bits + 7) // 8
    n = w * h
    if not rle:
        px = np.frombuffer(data[pos:pos + n * nb], np.uint8)
        if len(px) != n * nb:
            raise ValueError("tga: truncated image data")
        px = px.reshape(n, nb)
    else:
        chunks, got = [], 0
        while got < n:
            if pos >= len(data):
                raise ValueError("tga: truncated RLE data")
            c = data[pos]
            cnt = (c & 127) + 1
            pos += 1
            if c & 128:
                one = np.frombuffer(data[pos:pos + nb], np.uint8)
                if len(one) != nb:
                    raise ValueError("tga: truncated RLE data")
                chunks.append(np.tile(one, (cnt, 1)))
                pos += nb
            else:
                lit = np.frombuffer(data[pos:pos + cnt * nb], np.uint8)
                if len(lit) != cnt * nb:
                    raise ValueError("tga: truncated RLE data")
                chunks.append(lit.reshape(cnt, nb))
                pos += cnt * nb
            got += cnt
        px = np.concatenate(chunks)[:n]
    if base == 1:
        idx = px[:, 0].astype(np.int64) if nb == 1 else (px[:, 0].astype(np.int64) | (px[:, 1].astype(np.int64) << 8))
        idx = np.where(idx >= cm_len, 0, idx)                   # stb: an index outside the palette reads entry 0
        rgba = palette[idx]
    else:
        rgba = expand(px, bits, base == 3)
    rgba = rgba.reshape(h, w, 4)
    if not (desc >> 5) & 1:                                     # bottom-left origin: stb hands rows back top-down
        rgba = rgba[::-1]
    return np.ascontiguousarray(rgba)


def _decode_ppm(data: bytes) -> np.ndarray:
    toks: List[bytes] = []
    pos = 0
    while len(toks) < 4:
        end = data.find(b"\n", pos)
        if end < 0:
            raise ValueError("ppm: truncated header")
        toks += data[pos:end].split(b"#", 1)[0].split()
        pos = end + 1
    w, h, mx = int(toks[1]), int(toks[2]), int(toks[3])
    if toks[0] != b"P6" or mx != 255:
        raise ValueError("ppm: only binary 8-bit P6")
    rgb = np.frombuffer(data[pos:pos + w * h * 3], np.uint8).reshape(h, w, 3)
    return np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)


def decode_jpeg_native(path: str) -> np.ndarray:
    """(H, W, 4) uint8 as stbi_load(path, ..., 4) returns it, through the library's host-side decoder (no GPU needed)."""
    import ctypes as C
    from . import lib
    L = lib.load_loader()
    w, h, px = C.c_int32(0), C.c_int32(0), C.c_void_p()
    lib.check(None, L.fovpt_image_load_rgba8(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)), L)
    try:
        words = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint32)), (h.value, w.value)).copy()
    finally:
        L.fovpt_image_free_rgba8(px)
    return np.stack([(words >> s) & 255 for s in (0, 8, 16, 24)], -1).astype(np.uint8)


def _load_texture(path: str) -> Optional[np.ndarray]:
    """RGBA8 as (H, W) uint32, mirrored along y (Model.cpp:117-126).  PNG, TGA and binary PPM are decoded
    here; JPEG by the library's host-side decoder (fovpt_image_load_rgba8: stb_image's inverse DCT, chroma
    interpolation and colour conversion, bit for bit -- Pillow's libjpeg would give other pixels; the decoder is also
    built as the host-only csrc/libfovpt_loader.so, so this needs no HIP runtime); anything else counts as "could not
    load" (texture id -1, Model.cpp:129-131).  Of the formats the reference's stb_image reads, BMP, GIF, PSD, PIC and PNM
    other than binary P6 are NOT decoded here or in the library (no pinned decoder exists for them; the reference's scenes use
    TGA, PNG and JPEG): such a texture is id -1, which a caller sees as an untextured mesh."""
    if not os.path.exists(path):
        return None
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    return _decode_texture_bytes(data, path)


def _decode_texture_bytes(data: bytes, path: str = "") -> Optional[np.ndarray]:
    """The bytes of an image file (or of an image embedded in a glTF buffer / data URI: `path` empty) -> RGBA8 as (H, W)
    uint32 mirrored along y, or None."""
    try:
        if data[:8] == b"\x89PNG\r\n\x1a\n":
            rgba = decode_png(data)
        elif data[:2] == b"P6":
            rgba = _decode_ppm(data)
        elif path.lower().endswith(".tga"):
            rgba = decode_tga(data)
        elif data[:3] == b"\xff\xd8\xff":
            if path and os.path.exists(path):
                rgba = decode_jpeg_native(path)
            else:
                import tempfile
                with tempfile.NamedTemporaryFile(suffix=".jpg") as tmp:
                    tmp.write(data)
                    tmp.flush()
                    rgba = decode_jpeg_native(tmp.name)
        else:
            return None
    except Exception:
        return None
    rgba = rgba[::-1]                                       # mirror along y
    px = rgba.astype(np.uint32)
    return np.ascontiguousarray(px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | (px[..., 3] << 24))


def _corner(tok: str, nv: int, nt: int, nn: int) -> Tuple[int, int, int]:
    """'v', 'v/vt', 'v//vn', 'v/vt/vn' -> zero-based (v, vn, vt) with -1 for absent; negative = relative."""
    parts = tok.split("/")

    def fix(s, n):
        if s == "":
            return -1
        i = int(s)
        return i - 1 if i > 0 else n + i
    v = fix(parts[0], nv)
    vt = fix(parts[1], nt) if len(parts) > 1 else -1
    vn = fix(parts[2], nn) if len(parts) > 2 else -1
    return v, vn, vt


def _triangulate(cs: list, pos: list) -> list:
    """Polygon -> triangles the way the reference's OBJ reader does it (tinyobjloader 2.0.0-rc, vendored under
    support/tinyobjloader and called with triangulate = true, Model.cpp:150-157): ear clipping in binary32 on the
    projection onto the two axes chosen from the first non-degenerate corner; a convex planar quad comes out as the
    fan (0,1,2) (0,2,3), concave and non-planar polygons do not.  Corners are (v, vn, vt) index triples."""
    n = len(cs)
    if n < 3:
        return []
    if n == 3:
        return [tuple(cs)]
    f32 = np.float32
    eps = f32(np.finfo(np.float32).eps)
    P = [tuple(f32(x) for x in pos[c[0]]) if 0 <= c[0] < len(pos) else None for c in cs]
    axes = [1, 2]
    for k in range(n):
        p0, p1, p2 = P[k], P[(k + 1) % n], P[(k + 2) % n]
        if p0 is None or p1 is None or p2 is None:
            continue
        e0 = [p1[a] - p0[a] for a in range(3)]
        e1 = [p2[a] - p1[a] for a in range(3)]
        cx = abs(e0[1] * e1[2] - e0[2] * e1[1])
        cy = abs(e0[2] * e1[0] - e0[0] * e1[2])
        cz = abs(e0[0] * e1[1] - e0[1] * e1[0])
        if cx > eps or cy > eps or cz > eps:
            if not (cx > cy and cx > cz):
                axes[0] = 0
                if cz > cx and cz > cy:
                    axes[1] = 1
            break
    area = f32(0)
    for k in range(n):
        p0, p1 = P[k], P[(k + 1) % n]
        if p0 is None or p1 is None:
            continue
        area = area + (p0[axes[0]] * p1[axes[1]] - p0[axes[1]] * p1[axes[0]]) * f32(0.5)
    rem = list(range(n))                                    # positions in cs of the remaining polygon
    out = []
    guess = 0
    remaining_iter = n
    prev_n = n
    while len(rem) > 3 and remaining_iter > 0:
        m = len(rem)
        if guess >= m:
            guess -= m
        if prev_n != m:
            prev_n = m
            remaining_iter = m
        else:
            remaining_iter -= 1
        ind = [rem[(guess + k) % m] for k in range(3)]
        vx = [P[i][axes[0]] if P[i] is not None else f32(0) for i in ind]
        vy = [P[i][axes[1]] if P[i] is not None else f32(0) for i in ind]
        cross = (vx[1] - vx[0]) * (vy[2] - vy[1]) - (vy[1] - vy[0]) * (vx[2] - vx[1])
        if cross * area < 0:
            guess += 1
            continue
        overlap = False
        for other in range(3, m):
            q = P[rem[(guess + other) % m]]
            if q is None:
                continue
            tx, ty = q[axes[0]], q[axes[1]]
            c = False
            j = 2
            for i in range(3):                             # pnpoly on the candidate ear
                if (vy[i] > ty) != (vy[j] > ty):
                    with np.errstate(all="ignore"):
                        if tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i]:
                            c = not c
                j = i
            if c:
                overlap = True
                break
        if overlap:
            guess += 1
            continue
        out.append((cs[ind[0]], cs[ind[1]], cs[ind[2]]))
        del rem[(guess + 1) % m]
    if len(rem) == 3:
        out.append((cs[rem[0]], cs[rem[1]], cs[rem[2]]))
    return out


def load_obj(obj_file: str) -> Model:
    model_dir = os.path.dirname(obj_file)
    pos: List[Tuple[float, float, float]] = []
    nrm: List[Tuple[float, float, float]] = []
    tex: List[Tuple[float, float]] = []
    materials: Dict[str, dict] = {}
    mat_ids: Dict[str, int] = {}
    shapes: List[dict] = []
    cur = {"faces": [], "mats": []}
    cur_mat = -1

    def flush():
        nonlocal cur
        if cur["faces"]:
            shapes.append(cur)
        cur = {"faces": [], "mats": []}

    with open(obj_file, "r", errors="replace") as f:
        for line in f:
            t = line.split("#", 1)[0].split()
            if not t:
                continue
            k = t[0]
            if k == "v":
                pos.append((float(t[1]), float(t[2]), float(t[3])))
            elif k == "vn":
                nrm.append((float(t[1]), float(t[2]), float(t[3])))
            elif k == "vt":
                tex.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif k == "f":
                cs = [_corner(x, len(pos), len(tex), len(nrm)) for x in t[1:]]
                for tri in _triangulate(cs, pos):
                    cur["faces"].append(tri)
                    cur["mats"].append(cur_mat)
            elif k in ("o", "g"):
                flush()
            elif k == "usemtl":
                name = " ".join(t[1:])
                cur_mat = mat_ids.get(name, -1)
            elif k == "mtllib":
                for name in t[1:]:
                    for mname, m in _parse_mtl(os.path.join(model_dir, name)).items():
                        if mname not in mat_ids:
                            mat_ids[mname] = len(mat_ids)
                            materials[mname] = m
    flush()
    if not pos:
        raise RuntimeError("Could not read OBJ model from " + obj_file)          # Model.cpp:160-162
    mat_list = sorted(mat_ids, key=lambda n: mat_ids[n])
    P = np.asarray(pos, np.float32).reshape(-1, 3)
    N = np.asarray(nrm, np.float32).reshape(-1, 3)
    T = np.asarray(tex, np.float32).reshape(-1, 2)

    model = Model()
    for shape in shapes:
        # Both maps live per SHAPE in the reference (Model.cpp:174-175), not per mesh and not per model:
        #  * a texture named by two shapes is loaded twice and gets two ids;
        #  * a corner that an earlier material of the same shape already added keeps the id it got in THAT mesh --
        #    the later mesh then indexes its own vertex array with it (a quirk of the reference, reproduced; if the
        #    id is out of range for the later mesh the reference reads out of bounds and fovpt_set_scene rejects it).
        known: Dict[Tuple[int, int, int], int] = {}
        known_textures: Dict[str, int] = {}
        for mid in sorted(set(shape["mats"])):                  # std::set<int>: ascending
            vtx: List[np.ndarray] = []
            nrms: List[np.ndarray] = []
            tcs: List[np.ndarray] = []
            idx: List[Tuple[int, int, int]] = []

            def add_vertex(c):
                if c in known:
                    return known[c]
                new_id = len(vtx)
                known[c] = new_id
                vtx.append(P[c[0]])
                if c[1] >= 0:
                    while len(nrms) < len(vtx):
                        nrms.append(N[c[1]])
                if c[2] >= 0:
                    while len(tcs) < len(vtx):
                        tcs.append(T[c[2]])
                if tcs:
                    while len(tcs) < len(vtx):
                        tcs.append(np.zeros(2, np.float32))     # vector::resize pads with zeros
                if nrms:
                    while len(nrms) < len(vtx):
                        nrms.append(np.zeros(3, np.float32))
                return new_id

            for face, fm in zip(shape["faces"], shape["mats"]):
                if fm != mid:
                    continue
                idx.append(tuple(add_vertex(c) for c in face))
            # (the reference sets material and texture inside the face loop, :190-201: the texture of a mesh that ends up
            # empty -- every corner already known from an earlier material of the shape -- is still loaded and keeps its id)
            mat = Material.reference_default()
            tex_id = -1
            if mid >= 0:
                m = materials[mat_list[mid]]
                mat.color.set(m["Kd"])
                mat.emission.set(m["Ke"])
                name = m["map_Kd"]
                if name:
                    if name not in known_textures:
                        px = _load_texture(os.path.join(model_dir, name.replace("\\", "/")))       # :100-102
                        if px is None:
                            known_textures[name] = -1
                        else:
                            known_textures[name] = len(model.textures)
                            model.textures.append(px)
                    tex_id = known_textures[name]
            if not vtx:
                continue                                        # :204-205
            model.meshes.append(TriangleMesh(
                np.asarray(vtx, np.float32).reshape(-1, 3), np.asarray(idx, np.uint32).reshape(-1, 3), mat,
                np.asarray(tcs, np.float32).reshape(-1, 2) if tcs else None, tex_id))
    return model


def load_gltf_native(gltf_file: str) -> Model:
    """load_gltf through the library's host-side loader (fovpt_model_load_gltf, csrc/model_loader.cpp) -- what a C++ caller gets
    from `loadGLTF` in include/Model.h."""
    return load_obj_native(gltf_file, entry="fovpt_model_load_gltf")


def load_obj_native(obj_file: str, entry: str = "fovpt_model_load_obj") -> Model:
    """The same model through the library's host-side loader (fovpt_model_load_obj, csrc/model_loader.cpp) -- what a C++
    caller gets from `loadOBJ` in include/Model.h.  Each mesh also carries `.normal` ((N,3) float32 or None)."""
    import ctypes as C
    from . import abi, lib
    L = lib.load_loader()
    h = C.c_void_p()
    lib.check(None, getattr(L, entry)(os.fsencode(obj_file), C.byref(h)), L)
    try:
        nm, nt = C.c_int(0), C.c_int(0)
        lib.check(None, L.fovpt_model_counts(h, C.byref(nm), C.byref(nt)), L)
        model = Model()

        def arr(ptr, n, cols, dtype):
            if not ptr or n == 0:
                return None
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), (n * cols,)).view(dtype).reshape(n, cols).copy()
        for k in range(nm.value):
            d = abi.ModelMesh()
            lib.check(None, L.fovpt_model_get_mesh(h, k, C.byref(d)), L)
            mat = Material()
            C.memmove(C.byref(mat), C.byref(d.material), C.sizeof(Material))
            mesh = TriangleMesh(arr(d.vertex, d.num_vertices, 3, np.float32), arr(d.index, d.num_triangles, 3, np.uint32), mat,
                                arr(d.texcoord, d.num_texcoords, 2, np.float32), int(d.diffuse_texture_id))
            mesh.normal = arr(d.normal, d.num_normals, 3, np.float32)
            model.meshes.append(mesh)
        for k in range(nt.value):
            px, w, hh = C.c_void_p(), C.c_int(0), C.c_int(0)
            lib.check(None, L.fovpt_model_get_texture(h, k, C.byref(px), C.byref(w), C.byref(hh)), L)
            model.textures.append(np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint32)), (hh.value, w.value)).copy())
        return model
    finally:
        L.fovpt_model_destroy(h)


# ------------------------------------------------------------------------------------------
# Radiance .hdr environment maps -> float4 texels, as the reference's loadProbe gets them from
# stbi_loadf(file, &w, &h, &channels, 4) (PT_sv5_/main.cpp:160-171; stb_image v2.x is vendored under
# support/stb).  What is reproduced of stb's HDR reader:
#   * header: first token "#?RADIANCE" or "#?RGBE"; lines up to the first empty one, among them
#     "FORMAT=32-bit_rle_rgbe"; then "-Y <height> +X <width>" (the only orientation it accepts)
#   * scanlines: new-style RLE (2, 2, width hi, width lo; four channel planes; count > 128 = run of
#     count-128, else literal bytes) when 8 <= width < 32768, flat RGBE quadruples otherwise, and flat
#     data also when the FIRST scanline does not start with the RLE marker
#   * texel: (r, g, b) * 2^(e-136) as binary32 (exact: 8-bit mantissa times a power of two), alpha 1;
#     e == 0 gives (0, 0, 0, 1).  No gamma, no scale, no vertical flip (linear data stays as it is).
# ------------------------------------------------------------------------------------------
def _rgbe_to_float4(rgbe: np.ndarray) -> np.ndarray:
    """rgbe: (..., 4) uint8 -> (..., 4) float32."""
    e = rgbe[..., 3].astype(np.int32)
    f1 = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = np.empty(rgbe.shape[:-1] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f1[..., None]
    out[..., :3][e == 0] = 0.0
    out[..., 3] = 1.0
    return out


def load_hdr(path: str) -> np.ndarray:
    """Decode a Radiance RGBE file into an (H, W, 4) float32 array (row 0 = first scanline in the file)."""
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def token():
        nonlocal pos
        end = raw.find(b"\n", pos)
        if end < 0:
            end = len(raw)
        line = raw[pos:end]
        pos = min(len(raw), end + 1)
        return line.decode("latin-1")

    first = token()
    if first not in ("#?RADIANCE", "#?RGBE"):
        raise ValueError("%s: not a Radiance HDR file" % path)
    valid = False
    while True:
        line = token()
        if line == "":
            break
        if line == "FORMAT=32-bit_rle_rgbe":
            valid = True
        if pos >= len(raw):
            break
    if not valid:
        raise ValueError("%s: unsupported HDR format (FORMAT=32-bit_rle_rgbe expected)" % path)
    res = token()
    parts = res.split()
    if len(parts) != 4 or parts[0] != "-Y" or parts[2] != "+X":
        raise ValueError("%s: unsupported HDR data layout %r (-Y h +X w expected)" % (path, res))
    height, width = int(parts[1]), int(parts[3])
    if width <= 0 or height <= 0:
        raise ValueError("%s: bad HDR size" % path)
    data = np.frombuffer(raw, np.uint8, offset=pos)

    def flat(buf):
        need = width * height * 4
        if buf.size < need:
            raise ValueError("%s: truncated HDR data" % path)
        return _rgbe_to_float4(buf[:need].reshape(height, width, 4))

    if width < 8 or width >= 32768:
        return flat(data)
    if data.size >= 4 and not (data[0] == 2 and data[1] == 2 and not (data[2] & 0x80)):
        return flat(data)                      # old-style file: the bytes are plain pixels
    rows = np.empty((height, width, 4), np.uint8)
    p = 0
    for j in range(height):
        if p + 4 > data.size:
            raise ValueError("%s: truncated HDR data" % path)
        if data[p] != 2 or data[p + 1] != 2 or (data[p + 2] & 0x80):
            raise ValueError("%s: scanline %d is not run-length encoded" % (path, j))
        if ((int(data[p + 2]) << 8) | int(data[p + 3])) != width:
            raise ValueError("%s: invalid decoded scanline length" % path)
        p += 4
        for k in range(4):
            i = 0
            while i < width:
                if p >= data.size:
                    raise ValueError("%s: truncated HDR data" % path)
                count = int(data[p]); p += 1
                if count > 128:
                    count -= 128
                    if count > width - i or p >= data.size:
                        raise ValueError("%s: bad RLE data in HDR" % path)
                    rows[j, i:i + count, k] = data[p]; p += 1
                else:
                    if count > width - i or p + count > data.size:
                        raise ValueError("%s: bad RLE data in HDR" % path)
                    if count == 0:
                        raise ValueError("%s: bad RLE data in HDR (empty dump)" % path)
                    rows[j, i:i + count, k] = data[p:p + count]; p += count
                i += count
    return _rgbe_to_float4(rows)


def ldr_to_float4(rgba8: np.ndarray) -> np.ndarray:
    """What stbi_loadf(file, ..., 4) makes of an 8-bit image (stb_image's stbi__ldr_to_hdr with its default gamma 2.2f and
    scale 1): colour channels (float)pow(c / 255.0f, 2.2f) -- the binary32 quotient raised in binary64 by libm's pow --
    alpha c / 255.0f."""
    import math
    c = rgba8.astype(np.float32) / np.float32(255.0)
    g = float(np.float32(2.2))
    lut = np.array([math.pow(float(np.float32(k) / np.float32(255.0)), g) for k in range(256)], np.float64).astype(np.float32)   # libm pow, as stb
    out = np.empty(rgba8.shape[:2] + (4,), np.float32)
    out[..., :3] = lut[rgba8[..., :3]]
    out[..., 3] = c[..., 3]
    return out


def load_probe_texels(path: str) -> np.ndarray:
    """The float4 texels loadProbe hands to BuildCDF: stbi_loadf reads Radiance .hdr files as they are and any 8-bit
    format it knows through stbi__ldr_to_hdr.  (The comment at main.cpp:220 says .exr is readable too; stb_image has no
    EXR reader, stbi_loadf returns NULL for one.)"""
    with open(path, "rb") as f:
        head = f.read(16)
    if head.startswith(b"#?RADIANCE") or head.startswith(b"#?RGBE"):
        return load_hdr(path)
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        rgba = decode_png(data)
    elif data[:2] == b"P6":
        rgba = _decode_ppm(data)
    elif path.lower().endswith(".tga"):
        rgba = decode_tga(data)
    else:
        raise ValueError("%s: not an image stbi_loadf can read here (.hdr, .png, .ppm, .tga)" % path)
    return ldr_to_float4(rgba)


def load_probe_texels_native(path: str) -> np.ndarray:
    """The same texels through the library (fovpt_image_load_float4, csrc/model_loader.cpp): .hdr, .png, binary .ppm."""
    import ctypes as C
    from . import lib
    L = lib.load_loader()
    w, h, px = C.c_int(0), C.c_int(0), C.c_void_p()
    lib.check(None, L.fovpt_image_load_float4(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)), L)
    try:
        return np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), (h.value, w.value, 4)).copy()
    finally:
        L.fovpt_image_free(px)


def load_probe(hdr_file: str):
    """loadProbe (PT_sv5_/main.cpp:160-171): texels from the file, then ProbeData::BuildCDF."""
    from .renderer import ProbeData
    return ProbeData(load_probe_texels(hdr_file)).BuildCDF()


# ------------------------------------------------------------------------------------------
# glTF 2.0 (.gltf, JSON + external or data-URI buffers) -> Model.  In the reference tinygltf feeds only the
# SDK's sutil::Scene (sutil/Scene.cpp:109-442), never `Model`; this is the glue BASELINE.json's Bistro
# configuration needs, with sutil's traversal rules:
#   * root nodes are the nodes that are nobody's child (`scenes` is ignored, :425-437)
#   * node transform = parent * matrix * T * R * S in binary32 (:148); `matrix` is column-major in the file
#   * a camera node is skipped; a node WITH a mesh contributes its primitives and its children are NOT
#     visited; only a node without camera and mesh descends (:150-251)
#   * only TRIANGLES primitives (mode 4, the default); anything else is skipped (:181-185)
#   * materials: pbrMetallicRoughness baseColorFactor / roughnessFactor / metallicFactor / baseColorTexture
#     (:333-418)
# and the conventions of the OBJ path for what `Model` needs beyond that: one TriangleMesh per primitive with
# vertices in world space, Material() constructor defaults with color / roughness / metallic from the file and
# emission = emissiveFactor (0 when absent, like a missing Ke), texture id -1 when the image cannot be decoded.
# ------------------------------------------------------------------------------------------
_GLTF_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _m4(rows):
    return np.array(rows, np.float32).reshape(4, 4)


def _matmul4(a, b):
    """sutil::Matrix<4,4>::operator* (sutil/Matrix.h:339-355): sum = 0; sum += a[i][k] * b[k][j] for k = 0..3, in binary32 (no
    BLAS, no fused multiply-add: the library's C++ loader states the same operations in the same order)."""
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            acc = np.float32(0.0)
            for k in range(4):
                acc = np.float32(acc + np.float32(a[i, k] * b[k, j]))
            out[i, j] = acc
    return out


def _quat_matrix(x, y, z, w):
    """sutil::Quaternion(w, x, y, z).rotationMatrix() (sutil/Quaternion.h:239-269): binary32, the quaternion
    is used as given (not normalised), same operation order."""
    qw, qx, qy, qz = (np.float32(v) for v in (w, x, y, z))
    one, two = np.float32(1), np.float32(2)
    return _m4([[one - two * qy * qy - two * qz * qz, two * qx * qy - two * qz * qw, two * qx * qz + two * qy * qw, 0],
                [two * qx * qy + two * qz * qw, one - two * qx * qx - two * qz * qz, two * qy * qz - two * qx * qw, 0],
                [two * qx * qz - two * qy * qw, two * qy * qz + two * qx * qw, one - two * qx * qx - two * qy * qy, 0],
                [0, 0, 0, 1]])


def load_gltf(gltf_file: str) -> Model:
    import base64
    import json
    base = os.path.dirname(os.path.abspath(gltf_file))
    with open(gltf_file, "rb") as f:
        raw = f.read()
    glb_bin = None
    if raw[:4] == b"glTF":
        # binary container (.glb; tinygltf's LoadBinaryFromFile -- the reference only ever calls LoadASCIIFromFile,
        # sutil/Scene.cpp:265, so this is glue beyond it): 12-byte header {magic, version 2, length}, then chunks
        # {length, type, data}: the first is JSON, an optional second one BIN, which serves a buffer without `uri`
        import struct
        magic, version, total = struct.unpack_from("<4sII", raw, 0)
        if version != 2 or total > len(raw):
            raise ValueError("glb: unsupported version %d or truncated file" % version)
        pos, chunks = 12, []
        while pos + 8 <= total:
            clen, ctype = struct.unpack_from("<II", raw, pos)
            chunks.append((ctype, raw[pos + 8:pos + 8 + clen]))
            pos += 8 + clen + ((-clen) % 4)
        if not chunks or chunks[0][0] != 0x4E4F534A:
            raise ValueError("glb: the first chunk is not JSON")
        g = json.loads(chunks[0][1].decode("utf-8"))
        for ctype, data in chunks[1:]:
            if ctype == 0x004E4942:
                glb_bin = data
                break
    else:
        g = json.loads(raw.decode("utf-8"))

    def read_uri(uri: str) -> bytes:
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(base, uri), "rb") as fh:
            return fh.read()

    buffers = []
    for k, b in enumerate(g.get("buffers", [])):
        if "uri" in b:
            buffers.append(read_uri(b["uri"]))
        elif k == 0 and glb_bin is not None:
            buffers.append(glb_bin)
        else:
            raise ValueError("gltf: buffer %d has no uri" % k)

    def accessor(idx: int) -> np.ndarray:
        a = g["accessors"][idx]
        dt, nc, count = np.dtype(_GLTF_COMP[a["componentType"]]), _GLTF_NCOMP[a["type"]], a["count"]
        if "bufferView" not in a:
            return np.zeros((count, nc), dt)
        bv = g["bufferViews"][a["bufferView"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        elem = dt.itemsize * nc
        stride = bv.get("byteStride", 0) or elem
        raw = np.frombuffer(buffers[bv["buffer"]], np.uint8, count=(count - 1) * stride + elem if count else 0, offset=off)
        if stride == elem:
            out = raw.view(dt).reshape(count, nc)
        else:
            out = np.stack([raw[i * stride:i * stride + elem].view(dt) for i in range(count)]).reshape(count, nc)
        if a.get("normalized") and dt.kind in "ui":
            scale = np.float32(np.iinfo(dt).max)
            out = np.maximum(out.astype(np.float32) / scale, np.float32(-1.0))
        return out

    model = Model()
    tex_cache: Dict[int, int] = {}

    def texture_id(tex_index: Optional[int]) -> int:
        if tex_index is None:
            return -1
        if tex_index in tex_cache:
            return tex_cache[tex_index]
        tid = -1
        try:
            img = g["images"][g["textures"][tex_index]["source"]]
            px = None
            if "uri" in img:                         # a file beside the scene, or a data: URI
                if not img["uri"].startswith("data:"):
                    px = _load_texture(os.path.join(base, img["uri"]))
                elif "," in img["uri"]:
                    px = _decode_texture_bytes(base64.b64decode(img["uri"].split(",", 1)[1]))
            elif "bufferView" in img:                # a range of a buffer: the usual case in a .glb
                bv = g["bufferViews"][img["bufferView"]]
                off, n, blob = bv.get("byteOffset", 0), bv.get("byteLength", -1), buffers[bv["buffer"]]
                if 0 <= off <= len(blob) and 0 < n <= len(blob) - off:
                    px = _decode_texture_bytes(bytes(blob[off:off + n]))
            if px is not None:
                model.textures.append(px)
                tid = len(model.textures) - 1
        except (KeyError, IndexError, TypeError):
            tid = -1
        tex_cache[tex_index] = tid
        return tid

    def material(idx: Optional[int]):
        m = Material.reference_default()
        m.emission.set((0.0, 0.0, 0.0))
        tid = -1
        if idx is not None and 0 <= idx < len(g.get("materials", [])):
            gm = g["materials"][idx]
            pbr = gm.get("pbrMetallicRoughness", {})
            c = pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])
            m.color.set((float(c[0]), float(c[1]), float(c[2])))
            m.roughness = float(pbr.get("roughnessFactor", 1.0))
            m.metallic = float(pbr.get("metallicFactor", 1.0))
            e = gm.get("emissiveFactor", [0.0, 0.0, 0.0])
            m.emission.set((float(e[0]), float(e[1]), float(e[2])))
            if "baseColorTexture" in pbr:
                tid = texture_id(pbr["baseColorTexture"].get("index"))
        return m, tid

    nodes = g.get("nodes", [])
    is_root = [True] * len(nodes)
    for n in nodes:
        for ch in n.get("children", []):
            is_root[ch] = False

    def visit(node: dict, parent: np.ndarray):
        t = node.get("translation")
        T = _m4([[1, 0, 0, t[0]], [0, 1, 0, t[1]], [0, 0, 1, t[2]], [0, 0, 0, 1]]) if t else np.eye(4, dtype=np.float32)
        r = node.get("rotation")
        R = _quat_matrix(*r) if r else np.eye(4, dtype=np.float32)
        s = node.get("scale")
        S = _m4([[s[0], 0, 0, 0], [0, s[1], 0, 0], [0, 0, s[2], 0], [0, 0, 0, 1]]) if s else np.eye(4, dtype=np.float32)
        mm = node.get("matrix")
        M = np.array(mm, np.float32).reshape(4, 4).T if mm else np.eye(4, dtype=np.float32)     # column-major in the file
        xf = _matmul4(_matmul4(_matmul4(_matmul4(parent, M), T), R), S)
        if "camera" in node:
            return
        if "mesh" in node:
            for prim in g["meshes"][node["mesh"]].get("primitives", []):
                if prim.get("mode", 4) != 4:
                    continue
                pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
                # Matrix<4,4> * float4 with w = 1 (Matrix.h:467-485): ((m0 x + m1 y) + m2 z) + m3, element-wise binary32
                x, y, z = pos[:, 0], pos[:, 1], pos[:, 2]
                world = np.stack([((xf[i, 0] * x + xf[i, 1] * y) + xf[i, 2] * z) + xf[i, 3] for i in range(3)], 1).astype(np.float32)
                if "indices" in prim:
                    idx = accessor(prim["indices"]).astype(np.uint32).reshape(-1)
                else:
                    idx = np.arange(pos.shape[0], dtype=np.uint32)
                idx = idx[:idx.size // 3 * 3].reshape(-1, 3)
                tc = None
                if "TEXCOORD_0" in prim["attributes"]:
                    tc = np.ascontiguousarray(accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32)[:, :2])
                    if tc.shape[0] != pos.shape[0]:        # one texcoord per vertex, or none (as the library's loader)
                        tc = None
                mat, tid = material(prim.get("material"))
                model.meshes.append(TriangleMesh(vertex=np.ascontiguousarray(world), index=np.ascontiguousarray(idx), material=mat,
                                                 texcoord=tc, texture_id=tid if tc is not None else -1))
            return                                  # sutil does not descend below a mesh node
        for ch in node.get("children", []):
            visit(nodes[ch], xf)

    for i, n in enumerate(nodes):
        if is_root[i]:
            visit(n, np.eye(4, dtype=np.float32))
    return model
