"""OBJ ingestion (fovpathtracing_optixcodelatest_amd/loaders.py) against the behaviour of the reference's
loadOBJ (PT_sv5_/Model.cpp:138-217) on hand-checkable files."""
import os

import numpy as np
import pytest

from fovpathtracing_optixcodelatest_amd import loaders

OBJ = """# two shapes, two materials, quads, negative indices, all corner syntaxes
mtllib test.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
o wall
usemtl red
f 1/1/1 2/2/1 3/3/1 4/4/1
usemtl lamp
f 5//1 6//1 -5//1
g floor
usemtl red
f 1 2 6 5
f -1 -2 -6
"""
MTL = """newmtl red
Kd 0.8 0.1 0.2
map_Kd -s 1 1 1 tex.ppm
newmtl lamp
Kd 1 1 1
Ke 5 4 3
"""


@pytest.fixture()
def obj_dir(tmp_path):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    px = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8)   # 2x2: R G / B W
    with open(tmp_path / "tex.ppm", "wb") as f:
        f.write(b"P6\n2 2\n255\n" + px.tobytes())
    return str(tmp_path)


def test_obj_meshes_split_per_shape_and_material(obj_dir):
    m = loaders.load_obj(os.path.join(obj_dir, "test.obj"))
    # shape "wall": materials red (id 0) then lamp (id 1); shape "floor": red
    assert len(m.meshes) == 3
    wall_red, wall_lamp, floor = m.meshes
    assert wall_red.index.tolist() == [[0, 1, 2], [0, 2, 3]]                 # quad fanned from its first corner
    assert wall_red.vertex.tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]
    assert wall_red.texcoord.tolist() == [[0, 0], [1, 0], [1, 1], [0, 1]]
    assert wall_lamp.index.tolist() == [[0, 1, 2]]
    assert wall_lamp.vertex.tolist() == [[0, 0, 1], [1, 0, 1], [1, 0, 0]]    # -5 of 6 vertices = vertex 2
    assert wall_lamp.texcoord is None
    assert floor.index.shape == (3, 3) and floor.vertex.shape[0] == 4        # corners shared between its faces
    assert floor.index.tolist() == [[0, 1, 2], [0, 2, 3], [2, 3, 0]]         # -1,-2,-6 -> vertices 6,5,1 -> ids 2,3,0


def test_obj_materials_are_ctor_defaults_with_color_and_emission(obj_dir):
    m = loaders.load_obj(os.path.join(obj_dir, "test.obj"))
    red, lamp = m.meshes[0].material, m.meshes[1].material
    assert red.color.tolist() == pytest.approx([0.8, 0.1, 0.2]) and red.emission.tolist() == [0, 0, 0]
    assert lamp.emission.tolist() == [5, 4, 3]
    for mat in (red, lamp):                                                   # Material.h:13-38 stays in force
        assert (mat.transmission, mat.metallic, mat.eta, mat.roughness) == pytest.approx((0.4, 0.5, 1.4, 1.0))


def test_obj_diffuse_texture_is_loaded_once_and_mirrored(obj_dir):
    m = loaders.load_obj(os.path.join(obj_dir, "test.obj"))
    assert len(m.textures) == 1
    assert m.meshes[0].texture_id == 0 and m.meshes[2].texture_id == 0 and m.meshes[1].texture_id == -1
    t = m.textures[0]
    assert t.shape == (2, 2)
    # file rows were [R G] / [B W]; after the y mirror row 0 is [B W]
    assert [hex(x) for x in t[0]] == ["0xffff0000", "0xffffffff"] and [hex(x) for x in t[1]] == ["0xff0000ff", "0xff00ff00"]


def test_obj_missing_file_and_missing_texture(tmp_path):
    with pytest.raises((RuntimeError, OSError)):
        loaders.load_obj(str(tmp_path / "nope.obj"))
    (tmp_path / "a.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    (tmp_path / "a.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd missing.png\n")
    m = loaders.load_obj(str(tmp_path / "a.obj"))
    assert len(m.meshes) == 1 and m.meshes[0].texture_id == -1 and not m.textures   # Model.cpp:129-131


def test_loaded_model_packs_for_the_c_abi(obj_dir):
    from fovpathtracing_optixcodelatest_amd.scenes import pack_model
    m = loaders.load_obj(os.path.join(obj_dir, "test.obj"))
    md, n, td, nt, keep = pack_model(m)
    assert n == 3 and nt == 1 and md[0].num_triangles == 2 and md[0].texture_id == 0 and td[0].width == 2
