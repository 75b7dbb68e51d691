"""Procedural scenes in the reference's Model layout (PT_sv5_/Model.h:10-43).

No scene assets exist offline (the reference loads Sponza/San Miguel from absolute Windows
paths, PT_sv5_/main.cpp:196-207), so the benchmark configs of BASELINE.json use seeded
synthetic geometry of the same triangle budget and "architectural" depth complexity:

  cornell_box()      32 triangles, the C1 parity anchor
  atrium(n)          Sponza-class two-storey colonnaded hall, ~n triangles (C2/C3: 262,144)
  atrium(3.8e6)      Bistro-class budget (C4/C5)

A mesh is what Model.cpp's loadOBJ produces per material (vertex/index/texcoord arrays, one
Material, one diffuse texture id); TriangleMesh::normal is not generated because the path
never reads vertex normals (deviceProgram.cu:632-634).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .abi import Material


@dataclass
class TriangleMesh:
    vertex: np.ndarray                     # (N,3) float32
    index: np.ndarray                      # (M,3) uint32
    material: Material
    texcoord: Optional[np.ndarray] = None  # (N,2) float32
    texture_id: int = -1                   # Model.h:19 defaults to 1; every loader overwrites it


@dataclass
class Model:
    meshes: List[TriangleMesh] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)   # (H,W) uint32 RGBA8

    @property
    def num_triangles(self):
        return int(sum(m.index.shape[0] for m in self.meshes))


# ----------------------------------------------------------------------------------------
def _mat(color, emission=(0, 0, 0), **kw):
    m = Material.reference_default()
    m.color.set(color)
    m.emission.set(emission)
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def matte(color, emission=(0, 0, 0)):
    """The C1/C2 'diffuse' preset of SURVEY 8(d)."""
    return _mat(color, emission, transmission=0.0, metallic=0.0, specular=0.5, specularTint=0.0, roughness=0.5)


def diffuse_only(color):
    return _mat(color, (0, 0, 0), transmission=0.0, metallic=0.0, specular=0.0, clearcoat=0.0,
                subsurface=0.0, roughness=1.0)


def app_default(color, emission=(0, 0, 0)):
    """What loadOBJ leaves in place: Material() ctor defaults, only color/emission overridden
    (Model.cpp:190-191)."""
    return _mat(color, emission)


def _quad(p0, p1, p2, p3):
    v = np.array([p0, p1, p2, p3], dtype=np.float32)
    i = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    return v, i


def _merge(parts):
    vs, is_, base = [], [], 0
    for v, i in parts:
        vs.append(v)
        is_.append(i + base)
        base += v.shape[0]
    return np.concatenate(vs).astype(np.float32), np.concatenate(is_).astype(np.uint32)


def cornell_box() -> Model:
    """Classic Cornell box, 32 triangles: 5 walls (10), short block (10), tall block (10), light (2)."""
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    M = Model()
    floor = _quad((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2))
    ceil_ = _quad((556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0))
    back = _quad((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2))
    v, i = _merge([floor, ceil_, back])
    M.meshes.append(TriangleMesh(v, i, matte(white)))
    v, i = _quad((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2))
    M.meshes.append(TriangleMesh(v, i, matte(green)))
    v, i = _quad((552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0))
    M.meshes.append(TriangleMesh(v, i, matte(red)))
    short = [
        _quad((130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)),
        _quad((290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)),
        _quad((130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)),
        _quad((82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)),
        _quad((240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)),
    ]
    v, i = _merge(short)
    M.meshes.append(TriangleMesh(v, i, matte(white)))
    tall = [
        _quad((423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)),
        _quad((423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)),
        _quad((472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)),
        _quad((314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)),
        _quad((265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)),
    ]
    v, i = _merge(tall)
    M.meshes.append(TriangleMesh(v, i, matte(white)))
    v, i = _quad((343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227))
    M.meshes.append(TriangleMesh(v, i, matte((0.78, 0.78, 0.78), emission=(15, 15, 5))))
    assert M.num_triangles == 32
    return M


CORNELL_CAMERA = dict(eye=(278.0, 273.0, -800.0), lookat=(278.0, 273.0, 0.0), up=(0.0, 1.0, 0.0), fovy=40.0)
# PT_sv5_/main.cpp:240-251 (CRYTEK_SPONZA camera)
ATRIUM_CAMERA = dict(eye=(-1293.07, 154.681, 0.0), lookat=(200.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovy=45.0)


# ----------------------------------------------------------------------------------------
def _grid(nu, nv, fn, flip=False):
    """Tessellated parametric patch: fn(u,v) -> (x,y,z) arrays for u,v in [0,1]; 2*nu*nv triangles."""
    u = np.linspace(0.0, 1.0, nu + 1, dtype=np.float64)
    v = np.linspace(0.0, 1.0, nv + 1, dtype=np.float64)
    U, V = np.meshgrid(u, v, indexing="xy")
    x, y, z = fn(U, V)
    vert = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    tc = np.stack([U, V], axis=-1).reshape(-1, 2).astype(np.float32)
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).ravel()
    b = a + 1
    c = a + (nu + 1)
    d = c + 1
    if flip:
        tri = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)])
    else:
        tri = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    return vert, tri.astype(np.uint32), tc


def _value_noise(U, V, rng, cells=8):
    g = rng.random((cells + 1, cells + 1))
    x, y = U * cells, V * cells
    x0, y0 = np.minimum(x.astype(int), cells - 1), np.minimum(y.astype(int), cells - 1)
    fx, fy = x - x0, y - y0
    fx, fy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    return (g[y0, x0] * (1 - fx) * (1 - fy) + g[y0, x0 + 1] * fx * (1 - fy)
            + g[y0 + 1, x0] * (1 - fx) * fy + g[y0 + 1, x0 + 1] * fx * fy)


def _texture(kind, rng, size=256):
    """Procedural RGBA8 texture, (size,size) uint32 (little-endian R,G,B,A)."""
    y, x = np.mgrid[0:size, 0:size]
    u, v = x / size, y / size
    base = rng.random(3) * 0.5 + 0.4
    if kind == 0:      # checker
        t = ((x // (size // 8) + y // (size // 8)) % 2).astype(np.float64)
        t = 0.55 + 0.45 * t
    elif kind == 1:    # bricks
        row = y // (size // 8)
        xx = (x + (row % 2) * (size // 8)) % (size // 4)
        t = np.where((xx < 3) | (y % (size // 8) < 3), 0.45, 1.0)
    elif kind == 2:    # stripes
        t = 0.7 + 0.3 * np.sin(u * 2 * np.pi * 6)
    else:              # noise
        t = 0.5 + 0.5 * _value_noise(u, v, rng, 16)
    rgb = np.clip(base[None, None, :] * t[..., None], 0, 1)
    rgb8 = (rgb * 255.0 + 0.5).astype(np.uint32)
    return (rgb8[..., 0] | (rgb8[..., 1] << 8) | (rgb8[..., 2] << 16) | (255 << 24)).astype(np.uint32)


def atrium(target_triangles: int = 262144, seed: int = 1234, material="app", tolerance=0.01) -> Model:
    """Sponza-class hall: 3700 x 1500 x 1800 units, open roof slot, two colonnades with arches,
    an upper gallery, hanging curtains, displaced floor.  The tessellation factor is solved so the
    triangle count lands within `tolerance` of target_triangles."""
    mk = {"app": app_default, "diffuse": diffuse_only, "matte": matte}[material]

    def build(s):
        rng = np.random.default_rng(seed)
        M = Model()
        for k in range(8):
            M.textures.append(_texture(k % 4, rng))
        colors = rng.random((25, 3)) * 0.6 + 0.3
        mats = [mk(tuple(c)) for c in colors]
        X0, X1, Y1, Z0, Z1 = -1900.0, 1800.0, 1500.0, -900.0, 900.0
        n = lambda base: max(1, int(round(base * s)))

        def add(vtc, mat_id, tex=-1):
            v, i, tc = vtc
            M.meshes.append(TriangleMesh(v, i, mats[mat_id % 25], tc, tex))

        # floor: displaced grid (normal up)
        nf = rng.random()
        add(_grid(n(96), n(48), lambda U, V: (X0 + (X1 - X0) * U,
                                               6.0 * np.sin(U * 40 + nf) * np.cos(V * 23) - 3.0,
                                               Z0 + (Z1 - Z0) * V), flip=True), 0, 0)
        # walls (facing inward)
        add(_grid(n(64), n(32), lambda U, V: (X0 + (X1 - X0) * U, Y1 * V, Z0 + 4 * np.sin(U * 60) * np.sin(V * 30))), 1, 1)
        add(_grid(n(64), n(32), lambda U, V: (X0 + (X1 - X0) * U, Y1 * V, Z1 + 4 * np.sin(U * 60) * np.sin(V * 30)), flip=True), 1, 1)
        add(_grid(n(32), n(32), lambda U, V: (X0 + 0 * U, Y1 * V, Z0 + (Z1 - Z0) * U), flip=True), 2, 1)
        add(_grid(n(32), n(32), lambda U, V: (X1 + 0 * U, Y1 * V, Z0 + (Z1 - Z0) * U)), 2, 1)
        # roof: two slabs leaving a central slot open to the sky (normal down)
        add(_grid(n(64), n(12), lambda U, V: (X0 + (X1 - X0) * U, Y1 + 0 * U, Z0 + 500 * V)), 3)
        add(_grid(n(64), n(12), lambda U, V: (X0 + (X1 - X0) * U, Y1 + 0 * U, Z1 - 500 + 500 * V)), 3)
        # upper gallery slabs along both sides (top + bottom faces)
        for zc, mid in ((Z0 + 250, 4), (Z1 - 250, 4)):
            add(_grid(n(64), n(8), lambda U, V, zc=zc: (X0 + (X1 - X0) * U, 700 + 0 * U, zc - 250 + 500 * V), flip=True), mid, 2)
            add(_grid(n(64), n(8), lambda U, V, zc=zc: (X0 + (X1 - X0) * U, 660 + 0 * U, zc - 250 + 500 * V)), mid, 2)
        # colonnades: 2 rows x 12 columns on each storey, fluted cylinders
        ncol = 12
        for row, zc in enumerate((Z0 + 520, Z1 - 520)):
            for k in range(ncol):
                xc = X0 + 200 + (X1 - X0 - 400) * k / (ncol - 1)
                for (y0, y1, rad, mid) in ((0.0, 660.0, 55.0, 5), (700.0, 1300.0, 40.0, 6)):
                    add(_grid(n(24), n(40), lambda U, V, xc=xc, zc=zc, y0=y0, y1=y1, rad=rad: (
                        xc + (rad + 3 * np.cos(U * 2 * np.pi * 12)) * np.cos(U * 2 * np.pi) * (1 + 0.15 * np.exp(-40 * V) + 0.15 * np.exp(-40 * (1 - V))),
                        y0 + (y1 - y0) * V,
                        zc + (rad + 3 * np.cos(U * 2 * np.pi * 12)) * np.sin(U * 2 * np.pi) * (1 + 0.15 * np.exp(-40 * V) + 0.15 * np.exp(-40 * (1 - V)))),
                        flip=True), mid + row, 3)
            # arches between neighbouring columns (half tori)
            for k in range(ncol - 1):
                xa = X0 + 200 + (X1 - X0 - 400) * k / (ncol - 1)
                xb = X0 + 200 + (X1 - X0 - 400) * (k + 1) / (ncol - 1)
                xm, R = 0.5 * (xa + xb), 0.5 * (xb - xa)
                add(_grid(n(20), n(10), lambda U, V, xm=xm, R=R, zc=zc: (
                    xm + (R - 18 * np.cos(V * 2 * np.pi)) * np.cos(np.pi * U),
                    560 + (R - 18 * np.cos(V * 2 * np.pi)) * np.sin(np.pi * U) * 0.6,
                    zc + 30 * np.sin(V * 2 * np.pi)), flip=True), 8 + row, 3)
        # hanging curtains: heightfields with folds + value noise
        for k in range(8):
            xc = X0 + 450 + (X1 - X0 - 900) * (k % 4) / 3.0
            zc = (Z0 + 330) if k < 4 else (Z1 - 330)
            ph = rng.random() * 6.28
            crng = np.random.default_rng(seed + 100 + k)
            add(_grid(n(72), n(72), lambda U, V, xc=xc, zc=zc, ph=ph, crng=crng: (
                xc - 170 + 340 * U,
                1250 - 520 * V,
                zc + 28 * np.sin(U * 2 * np.pi * 5 + ph) * (0.3 + V) + 18 * (_value_noise(U, V, crng) - 0.5))),
                10 + k, 4 + (k % 4))
            # curtains are thin: add the back side so both facings occlude
            add(_grid(n(72), n(72), lambda U, V, xc=xc, zc=zc, ph=ph, crng=crng: (
                xc - 170 + 340 * U,
                1250 - 520 * V,
                zc + 1.5 + 28 * np.sin(U * 2 * np.pi * 5 + ph) * (0.3 + V) + 18 * (_value_noise(U, V, crng) - 0.5)),
                flip=True), 10 + k, 4 + (k % 4))
        # a few emissive lanterns (small spheres) under the gallery
        for k in range(6):
            xc = X0 + 500 + (X1 - X0 - 1000) * k / 5.0
            zc = Z0 + 250 if k % 2 == 0 else Z1 - 250
            v, i, tc = _grid(n(10), n(8), lambda U, V, xc=xc, zc=zc: (
                xc + 25 * np.sin(np.pi * V) * np.cos(2 * np.pi * U),
                560 + 25 * np.cos(np.pi * V),
                zc + 25 * np.sin(np.pi * V) * np.sin(2 * np.pi * U)))
            M.meshes.append(TriangleMesh(v, i, mk((1.0, 0.9, 0.7)) if material != "app" else app_default((1.0, 0.9, 0.7), (4.0, 3.5, 2.5)), tc, -1))
        return M

    s = np.sqrt(target_triangles / 262144.0)
    best, best_err = None, None
    for _ in range(16):
        model = build(s)
        cnt = model.num_triangles
        err = abs(cnt - target_triangles)
        if best is None or err < best_err:
            best, best_err = model, err
        if err <= tolerance * target_triangles:
            break
        # damped correction: integer rounding of the per-part resolutions makes the count jumpy
        s *= (target_triangles / cnt) ** 0.35
    return best


STREET_CAMERA = dict(eye=(-2300.0, 170.0, 40.0), lookat=(0.0, 330.0, 0.0), up=(0.0, 1.0, 0.0), fovy=55.0)


def street(target_triangles: int = 3800000, seed: int = 4321, material="app", tolerance=0.01) -> Model:
    """Bistro-class exterior: a street 5200 units long between two rows of 20 facade modules each
    (relief walls, recessed windows with sills and glass, balconies with baluster rows, cornices, awnings), a cobbled
    road with kerbs, lamp posts with emissive lanterns, planters, and 14 trees whose crowns are thousands of opaque
    foliage cards (no alpha: the reference has no cut-outs either).  Unlike the atrium hall it is open to the sky, seen
    along its length (long rays, many candidate layers in depth: cards, balusters, posts), and most triangles are small
    relative to the boxes around them.  The tessellation factor is solved for the triangle count like atrium()."""
    mk = {"app": app_default, "diffuse": diffuse_only, "matte": matte}[material]

    def build(s):
        rng = np.random.default_rng(seed)
        M = Model()
        for k in range(8):
            M.textures.append(_texture(k % 4, rng))
        colors = rng.random((32, 3)) * 0.55 + 0.3
        mats = [mk(tuple(c)) for c in colors]
        leaf_mats = [mk((0.15 + 0.2 * rng.random(), 0.45 + 0.4 * rng.random(), 0.1 + 0.15 * rng.random())) for _ in range(4)]
        n = lambda base: max(1, int(round(base * s)))

        def add(vtc, mat, tex=-1):
            v, i, tc = vtc
            M.meshes.append(TriangleMesh(v, i, mat if isinstance(mat, Material) else mats[mat % 32], tc, tex))

        def box(x0, x1, y0, y1, z0, z1, mat, res=(1, 1, 1), tex=-1):
            """six tessellated faces, outward normals"""
            rx, ry, rz = res
            add(_grid(rx, ry, lambda U, V: (x0 + (x1 - x0) * U, y0 + (y1 - y0) * V, z1 + 0 * U)), mat, tex)                 # +z
            add(_grid(rx, ry, lambda U, V: (x0 + (x1 - x0) * U, y0 + (y1 - y0) * V, z0 + 0 * U), flip=True), mat, tex)      # -z
            add(_grid(rz, ry, lambda U, V: (x1 + 0 * U, y0 + (y1 - y0) * V, z0 + (z1 - z0) * U), flip=True), mat, tex)      # +x
            add(_grid(rz, ry, lambda U, V: (x0 + 0 * U, y0 + (y1 - y0) * V, z0 + (z1 - z0) * U)), mat, tex)                 # -x
            add(_grid(rx, rz, lambda U, V: (x0 + (x1 - x0) * U, y1 + 0 * U, z0 + (z1 - z0) * V), flip=True), mat, tex)      # +y
            add(_grid(rx, rz, lambda U, V: (x0 + (x1 - x0) * U, y0 + 0 * U, z0 + (z1 - z0) * V)), mat, tex)                 # -y

        def cylinder(xc, zc, y0, y1, rad, mat, nu, nv, bulge=0.0):
            add(_grid(nu, nv, lambda U, V: (xc + rad * (1 + bulge * np.sin(np.pi * V)) * np.cos(2 * np.pi * U), y0 + (y1 - y0) * V,
                                            zc + rad * (1 + bulge * np.sin(np.pi * V)) * np.sin(2 * np.pi * U)), flip=True), mat)

        X0, X1, ZW = -2600.0, 2600.0, 420.0            # street axis along x, facades at z = +-ZW
        # road: cobbles (displaced), two kerbs and pavements
        ph = rng.random(2) * 6.28
        add(_grid(n(260), n(36), lambda U, V: (X0 + (X1 - X0) * U, 2.5 * np.sin(U * 900 + ph[0]) * np.sin(V * 130 + ph[1]) - 2.0,
                                               -260 + 520 * V), flip=True), 0, 1)
        for sgn in (-1.0, 1.0):
            box(X0, X1, 0.0, 14.0, sgn * 260 - 12, sgn * 260 + 12, 1, (n(60), 1, 1))
            add(_grid(n(200), n(12), lambda U, V, sgn=sgn: (X0 + (X1 - X0) * U, 14.0 + 0.6 * np.sin(U * 400) * np.sin(V * 40),
                                                            sgn * 272 + sgn * (ZW - 272) * V), flip=sgn > 0), 2, 0)
        # facades: 20 modules per side, 260 wide, 3-5 storeys
        nmod = 20
        for side, sgn in enumerate((-1.0, 1.0)):
            for m in range(nmod):
                mrng = np.random.default_rng(seed + 1000 * side + m)
                xa = X0 + (X1 - X0) * m / nmod
                xb = xa + (X1 - X0) / nmod
                storeys = int(mrng.integers(3, 6))
                hs = 300.0 + 40.0 * mrng.random()
                height = storeys * hs + 80.0
                zf = sgn * ZW
                wall_mat = 3 + int(mrng.integers(0, 8))
                amp, f1, f2 = 3.0 + 5.0 * mrng.random(), 30 + 60 * mrng.random(), 20 + 40 * mrng.random()
                # relief wall facing the street
                add(_grid(n(40), n(14 * storeys), lambda U, V, xa=xa, xb=xb, height=height, zf=zf, amp=amp, f1=f1, f2=f2, sgn=sgn: (
                    xa + (xb - xa) * U, height * V, zf - sgn * amp * (np.sin(U * f1) * np.sin(V * f2 * storeys / 3.0) + 1.0)),
                    flip=sgn < 0), wall_mat, 1 + (m % 3))
                # roof slab and side fins between modules (occlusion in depth along the street)
                box(xa, xb, height, height + 25.0, zf - 40 * (sgn < 0) - 0.0, zf + 40 * (sgn > 0) + 0.0, 11, (n(8), 1, 1))
                box(xa - 6, xa + 6, 0.0, height, min(zf, zf - sgn * 30), max(zf, zf - sgn * 30), 12, (1, n(4 * storeys), 1))
                # cornice
                add(_grid(n(30), n(6), lambda U, V, xa=xa, xb=xb, height=height, zf=zf, sgn=sgn: (
                    xa + (xb - xa) * U, height - 30 + 30 * V, zf - sgn * (10 + 28 * np.sin(np.pi * V) ** 2)), flip=sgn < 0), 13)
                nwin = 3
                for st_ in range(storeys):
                    y0 = 70.0 + st_ * hs
                    for wv in range(nwin):
                        xc = xa + (xb - xa) * (wv + 0.5) / nwin
                        ww, wh = 46.0, 150.0
                        # window: sill, two jambs, lintel, glass pane set back from the wall
                        box(xc - ww - 8, xc + ww + 8, y0 - 10, y0, min(zf - sgn * 26, zf), max(zf - sgn * 26, zf), 14, (n(3), 1, 1))
                        box(xc - ww - 8, xc - ww, y0, y0 + wh, min(zf - sgn * 16, zf), max(zf - sgn * 16, zf), 14, (1, n(3), 1))
                        box(xc + ww, xc + ww + 8, y0, y0 + wh, min(zf - sgn * 16, zf), max(zf - sgn * 16, zf), 14, (1, n(3), 1))
                        box(xc - ww - 8, xc + ww + 8, y0 + wh, y0 + wh + 12, min(zf - sgn * 20, zf), max(zf - sgn * 20, zf), 14, (n(3), 1, 1))
                        add(_grid(n(4), n(8), lambda U, V, xc=xc, ww=ww, y0=y0, wh=wh, zf=zf, sgn=sgn: (
                            xc - ww + 2 * ww * U, y0 + wh * V, zf - sgn * 4.0 + 0 * U), flip=sgn < 0), 15)
                    # balcony on some storeys: slab + a row of balusters + rail
                    if st_ > 0 and mrng.random() < 0.55:
                        yb = y0 - 14.0
                        z_in, z_out = zf, zf - sgn * 70.0
                        box(xa + 20, xb - 20, yb, yb + 8, min(z_in, z_out), max(z_in, z_out), 16, (n(10), 1, n(2)))
                        nb = n(22)
                        for b in range(nb):
                            xbp = xa + 26 + (xb - xa - 52) * b / max(1, nb - 1)
                            cylinder(xbp, z_out + sgn * 4, yb + 8, yb + 60, 2.2, 17, max(5, n(6)), max(1, n(3)), bulge=0.5)
                        box(xa + 20, xb - 20, yb + 60, yb + 66, min(z_out, z_out + sgn * 8), max(z_out, z_out + sgn * 8), 17, (n(10), 1, 1))
                # ground floor awning (sagging cloth)
                if mrng.random() < 0.6:
                    add(_grid(n(24), n(10), lambda U, V, xa=xa, xb=xb, zf=zf, sgn=sgn: (
                        xa + 30 + (xb - xa - 60) * U, 250 - 60 * V - 10 * np.sin(np.pi * U) * V - 4 * np.sin(U * 50) * V, zf - sgn * 130 * V),
                        flip=sgn > 0), 18 + (m % 6), 4 + (m % 4))
        # lamp posts with emissive lanterns, planters
        for k in range(16):
            xc = X0 + 160 + (X1 - X0 - 320) * k / 15.0
            zc = 300.0 if k % 2 else -300.0
            cylinder(xc, zc, 14.0, 420.0, 5.0, 24, max(6, n(10)), max(2, n(12)))
            v, i, tc = _grid(max(6, n(12)), max(4, n(8)), lambda U, V, xc=xc, zc=zc: (
                xc + 22 * np.sin(np.pi * V) * np.cos(2 * np.pi * U), 440 + 26 * np.cos(np.pi * V), zc + 22 * np.sin(np.pi * V) * np.sin(2 * np.pi * U)))
            M.meshes.append(TriangleMesh(v, i, mk((1.0, 0.9, 0.7)) if material != "app" else app_default((1.0, 0.9, 0.7), (5.0, 4.2, 3.0)), tc, -1))
            box(xc + 60, xc + 150, 14.0, 60.0, zc - 30, zc + 30, 25, (n(4), n(2), n(2)), 2)
        # trees: trunk, a few branches, crowns of opaque foliage cards
        for k in range(14):
            trng = np.random.default_rng(seed + 5000 + k)
            xc = X0 + 300 + (X1 - X0 - 600) * k / 13.0 + 40 * (trng.random() - 0.5)
            zc = (200.0 if k % 2 else -200.0) + 30 * (trng.random() - 0.5)
            th = 330.0 + 120.0 * trng.random()
            cylinder(xc, zc, 14.0, th, 14.0, 26, max(6, n(12)), max(3, n(14)), bulge=-0.25)
            ncards = n(3600) * 4 // 4
            c = np.stack([xc + 170 * trng.normal(size=ncards) * 0.55, th + 110 + 130 * trng.normal(size=ncards) * 0.55,
                          zc + 170 * trng.normal(size=ncards) * 0.55], axis=1)
            a = trng.normal(size=(ncards, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
            b = np.cross(a, trng.normal(size=(ncards, 3))); b /= np.linalg.norm(b, axis=1, keepdims=True)
            sz = (9.0 + 9.0 * trng.random(ncards))[:, None]
            quad = np.stack([c - a * sz - b * sz, c + a * sz - b * sz, c + a * sz + b * sz, c - a * sz + b * sz], axis=1)   # (n,4,3)
            v = quad.reshape(-1, 3).astype(np.float32)
            base = (4 * np.arange(ncards, dtype=np.uint32))[:, None]
            idx = np.concatenate([base + np.uint32([0, 1, 2]), base + np.uint32([0, 2, 3])], axis=0).astype(np.uint32)
            tcq = np.tile(np.float32([[0, 0], [1, 0], [1, 1], [0, 1]]), (ncards, 1))
            M.meshes.append(TriangleMesh(v, idx, leaf_mats[k % 4], tcq, -1))
        # one mesh per (material, texture), as loadOBJ would deliver a modelled street (Model.cpp:166-206)
        groups = {}
        for mesh in M.meshes:
            groups.setdefault((id(mesh.material), mesh.texture_id), []).append(mesh)
        merged = []
        for (_, tex), ms in groups.items():
            v, i = _merge([(m_.vertex, m_.index) for m_ in ms])
            tc = np.concatenate([m_.texcoord for m_ in ms]).astype(np.float32)
            merged.append(TriangleMesh(v, i, ms[0].material, tc, tex))
        M.meshes = merged
        return M

    s = np.sqrt(target_triangles / 3800000.0)
    best, best_err = None, None
    for _ in range(20):
        model = build(s)
        cnt = model.num_triangles
        err = abs(cnt - target_triangles)
        if best is None or err < best_err:
            best, best_err = model, err
        if err <= tolerance * target_triangles:
            break
        s *= (target_triangles / cnt) ** 0.4
    return best


def ambient_probe(width, height, value=2.5):
    """loadColor (PT_sv5_/main.cpp:175-187): a constant-colour probe at frame resolution."""
    data = np.empty((height, width, 4), dtype=np.float32)
    data[..., :3] = np.float32(value)
    data[..., 3] = 1.0
    return data


def sky_probe(width=64, height=32, seed=7):
    """Small HDR-like probe with a bright sun blob (for importance-sampling tests)."""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:height, 0:width]
    u = (u + 0.5) / width
    v = (v + 0.5) / height
    sky = 0.4 + 0.6 * (1 - v)
    sun = 40.0 * np.exp(-((u - 0.3) ** 2 + (v - 0.25) ** 2) / 0.003)
    lum = sky + sun + 0.05 * rng.random((height, width))
    data = np.empty((height, width, 4), dtype=np.float32)
    data[..., 0] = lum * 1.0
    data[..., 1] = lum * 0.95
    data[..., 2] = lum * 0.85 + 0.1 * (1 - v)
    data[..., 3] = 1.0
    return data


def pack_model(model: Model):
    """Model -> (fovpt_mesh_desc[], n, fovpt_texture_desc[], nt, keepalive) for fovpt_set_scene."""
    import ctypes as C
    from .abi import MeshDesc, TextureDesc

    keep = []
    md = (MeshDesc * max(1, len(model.meshes)))()
    for k, m in enumerate(model.meshes):
        v = np.ascontiguousarray(m.vertex, dtype=np.float32)
        i = np.ascontiguousarray(m.index, dtype=np.uint32)
        keep += [v, i]
        md[k].vertex = v.ctypes.data
        md[k].index = i.ctypes.data
        md[k].normal = None
        if m.texcoord is not None:
            t = np.ascontiguousarray(m.texcoord, dtype=np.float32)
            keep.append(t)
            md[k].texcoord = t.ctypes.data
        else:
            md[k].texcoord = None
        md[k].num_vertices = v.shape[0]
        md[k].num_triangles = i.shape[0]
        md[k].texture_id = int(m.texture_id)
        md[k].material = m.material
    td = (TextureDesc * max(1, len(model.textures)))()
    for k, t in enumerate(model.textures):
        px = np.ascontiguousarray(t, dtype=np.uint32)
        keep.append(px)
        td[k].pixel = px.ctypes.data
        td[k].width = px.shape[1]
        td[k].height = px.shape[0]
    return md, len(model.meshes), td, len(model.textures), keep


def box_mesh(pos, extend, material) -> TriangleMesh:
    """addBox (PT_sv5_/Model.cpp:219-291): 12 triangles over 36 unshared vertices, in the reference's
    face and winding order (front, back, left, right, top, bottom)."""
    sx = (-1, 1, 1, -1, -1, 1, 1, -1)
    sy = (-1, -1, 1, 1, -1, -1, 1, 1)
    sz = (1, 1, 1, 1, -1, -1, -1, -1)
    P = [(np.float32(sx[k]) * np.float32(extend[0]) + np.float32(pos[0]),
          np.float32(sy[k]) * np.float32(extend[1]) + np.float32(pos[1]),
          np.float32(sz[k]) * np.float32(extend[2]) + np.float32(pos[2])) for k in range(8)]
    A, B, C_, D, E, F, G, H = range(8)
    tri = [(A, B, C_), (A, C_, D), (E, H, G), (E, G, F), (E, A, D), (E, D, H),
           (B, F, G), (B, G, C_), (D, C_, G), (D, G, H), (E, A, B), (E, B, F)]
    v = np.array([P[i] for t in tri for i in t], np.float32)
    idx = np.arange(36, dtype=np.uint32).reshape(12, 3)
    return TriangleMesh(v, idx, material, np.zeros((36, 2), np.float32), -1)
