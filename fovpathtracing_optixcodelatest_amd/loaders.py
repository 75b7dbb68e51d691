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
                cur = {"Kd": (0.8, 0.8, 0.8) if False else None, "Ke": (0.0, 0.0, 0.0), "map_Kd": ""}
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


def _load_texture(path: str) -> Optional[np.ndarray]:
    """RGBA8 as (H, W) uint32, mirrored along y (Model.cpp:117-126).  Needs Pillow for PNG/JPG/TGA;
    binary PPM (P6) is read natively."""
    if not os.path.exists(path):
        return None
    try:
        with open(path, "rb") as f:
            head = f.read(2)
        if head == b"P6":
            with open(path, "rb") as f:
                toks: List[bytes] = []
                while len(toks) < 4:
                    line = f.readline()
                    if not line:
                        return None
                    toks += line.split(b"#", 1)[0].split()
                w, h, mx = int(toks[1]), int(toks[2]), int(toks[3])
                rgb = np.frombuffer(f.read(w * h * 3), np.uint8).reshape(h, w, 3)
            rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
        else:
            from PIL import Image                          # optional dependency
            rgba = np.asarray(Image.open(path).convert("RGBA"), np.uint8)
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
                for i in range(1, len(cs) - 1):                 # fan from the first corner
                    cur["faces"].append((cs[0], cs[i], cs[i + 1]))
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
    known_textures: Dict[str, int] = {}
    for shape in shapes:
        for mid in sorted(set(shape["mats"])):                  # std::set<int>: ascending
            known: Dict[Tuple[int, int, int], int] = {}
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
            if not vtx:
                continue
            mat = Material.reference_default()
            tex_id = -1
            if mid >= 0:
                m = materials[mat_list[mid]]
                mat.color.set(m["Kd"])
                mat.emission.set(m["Ke"])
                name = m["map_Kd"]
                if name:
                    if name not in known_textures:
                        px = _load_texture(os.path.join(model_dir, name.replace("\\\\", "/")))
                        if px is None:
                            known_textures[name] = -1
                        else:
                            known_textures[name] = len(model.textures)
                            model.textures.append(px)
                    tex_id = known_textures[name]
            model.meshes.append(TriangleMesh(
                np.asarray(vtx, np.float32).reshape(-1, 3), np.asarray(idx, np.uint32).reshape(-1, 3), mat,
                np.asarray(tcs, np.float32).reshape(-1, 2) if tcs else None, tex_id))
    return model
