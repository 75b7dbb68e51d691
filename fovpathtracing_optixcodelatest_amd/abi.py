"""ctypes mirror of include/fovpt.h (the C ABI of libfovpt).

Layouts follow the reference's device-visible structs: Material (PT_sv5_/Material.h:48-69),
Probe (PT_sv5_/Probe.cuh:6-21), LaunchParams (PT_sv5_/LaunchParams.h:49-91).
"""
import ctypes as C

FOVPT_OK = 0
MATERIAL_FLAG_SHADOW_CATCHER = 1
OPT_SKY_MISS, OPT_RUSSIAN_ROULETTE = 1, 2      # fovpt_config.options (include/fovpt.h)

OP_SIN, OP_COS, OP_ACOS, OP_ATAN2, OP_LOG, OP_POW, OP_SQRT, OP_DIV, OP_RSQRTD, OP_UNORM8, OP_HALFPLUS = range(1, 12)


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def set(self, v):
        self.x, self.y, self.z = float(v[0]), float(v[1]), float(v[2])
        return self

    def tolist(self):
        return [self.x, self.y, self.z]


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Int2(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class UInt2(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]


class UInt3(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class Material(C.Structure):
    """fovpt_material == Material (PT_sv5_/Material.h).  Construct with Material.reference_default()
    to get the non-trivial constructor defaults of Material.h:13-38."""
    _fields_ = [
        ("emission", Float3), ("color", Float3), ("absorption", Float3),
        ("eta", C.c_float), ("metallic", C.c_float), ("subsurface", C.c_float),
        ("specular", C.c_float), ("roughness", C.c_float), ("specularTint", C.c_float),
        ("anisotropic", C.c_float), ("sheen", C.c_float), ("sheenTint", C.c_float),
        ("clearcoat", C.c_float), ("clearcoatGloss", C.c_float), ("transmission", C.c_float),
        ("bump", C.c_float), ("bumpTile", Float3), ("flags", C.c_int32),
    ]

    @classmethod
    def reference_default(cls):
        m = cls()
        m.color.set((1.0, 0.0, 0.0))
        m.emission.set((1.0, 1.0, 1.0))
        m.absorption.set((1.0, 1.0, 1.0))
        m.eta = 1.4
        m.metallic = 0.5
        m.subsurface = 0.0
        m.specular = 1.0
        m.roughness = 1.0
        m.specularTint = 1.0
        m.anisotropic = 0.0
        m.sheen = 0.0
        m.sheenTint = 0.0
        m.clearcoat = 0.0
        m.clearcoatGloss = 1.0
        m.transmission = 0.4
        m.bump = 0.0
        m.bumpTile.set((1.0, 1.0, 1.0))
        m.flags = 0
        return m

    def copy(self):
        m = Material()
        C.memmove(C.byref(m), C.byref(self), C.sizeof(Material))
        return m


class Probe(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("data", C.c_void_p),
        ("offset", Float3), ("_pad0", C.c_uint32),
        ("pdfValuesX", C.c_void_p), ("cdfValuesX", C.c_void_p),
        ("pdfValuesY", C.c_void_p), ("cdfValuesY", C.c_void_p),
    ]


class _Frame(C.Structure):
    _fields_ = [
        ("accum_buffer", C.c_void_p), ("frame_buffer", C.c_void_p),
        ("color_buffer", C.c_void_p), ("normal_buffer", C.c_void_p), ("albedo_buffer", C.c_void_p),
        ("size", Int2), ("subframe_index", C.c_uint32), ("factor", UInt3),
        ("fillSize", C.c_int32), ("_pad0", C.c_uint32), ("c", UInt2),
        ("r_inner", C.c_float), ("r_outer", C.c_float), ("offset", UInt2),
        ("redraw", C.c_uint32), ("_pad1", C.c_uint32),
    ]


class _Camera(C.Structure):
    _fields_ = [("eye", Float3), ("U", Float3), ("V", Float3), ("W", Float3)]


class LaunchParams(C.Structure):
    _fields_ = [
        ("frame", _Frame), ("camera", _Camera),
        ("samples_per_launch", C.c_uint32), ("_pad2", C.c_uint32),
        ("traversable", C.c_uint64), ("probe", Probe),
        ("viewportSize", Int2), ("white", C.c_float), ("_pad3", C.c_uint32),
    ]

    def copy(self):
        m = LaunchParams()
        C.memmove(C.byref(m), C.byref(self), C.sizeof(LaunchParams))
        return m


class MeshDesc(C.Structure):
    _fields_ = [
        ("vertex", C.c_void_p), ("normal", C.c_void_p), ("texcoord", C.c_void_p), ("index", C.c_void_p),
        ("num_vertices", C.c_uint32), ("num_triangles", C.c_uint32),
        ("texture_id", C.c_int32), ("material", Material),
    ]


class ModelMesh(C.Structure):
    """fovpt_model_mesh: one TriangleMesh of a model loaded by fovpt_model_load_obj (arrays owned by the model)."""
    _fields_ = [
        ("vertex", C.c_void_p), ("normal", C.c_void_p), ("texcoord", C.c_void_p), ("index", C.c_void_p),
        ("num_vertices", C.c_uint32), ("num_normals", C.c_uint32), ("num_texcoords", C.c_uint32), ("num_triangles", C.c_uint32),
        ("material", Material), ("diffuse_texture_id", C.c_int32),
    ]


class TextureDesc(C.Structure):
    _fields_ = [("pixel", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class Config(C.Structure):
    _fields_ = [
        ("uniform", C.c_int32), ("r_inner", C.c_int32), ("r_outer", C.c_int32),
        ("spp_periphery", C.c_int32), ("spp_middle", C.c_int32), ("spp_fovea", C.c_int32),
        ("spp_uniform", C.c_int32), ("max_depth", C.c_int32), ("accumulate", C.c_int32),
        ("rank", C.c_int32), ("world", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
        ("profile", C.c_int32), ("write_guides", C.c_int32), ("options", C.c_int32),
        ("frames_in_flight", C.c_int32),      # 0 = library default (2), 1, 2
        ("chains_per_frame", C.c_int32),      # 0 / 1 = one chain per frame, 2 = two (for callers that synchronise every frame)
    ]

    @classmethod
    def reference_default(cls):
        """PT_sv5_ as shipped: FOV_ON, radii 74/241, spp 8/16/32, uniform spp 4, depth cap 4."""
        c = cls()
        c.uniform = 0
        c.r_inner, c.r_outer = 74, 241
        c.spp_periphery, c.spp_middle, c.spp_fovea, c.spp_uniform = 8, 16, 32, 4
        c.max_depth = 4
        c.accumulate = 0
        c.rank, c.world = 0, 1
        c.tile_w, c.tile_h = 8, 4
        return c

    def copy(self):
        m = Config()
        C.memmove(C.byref(m), C.byref(self), C.sizeof(Config))
        return m


class Stats(C.Structure):
    _fields_ = [
        ("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64), ("frames", C.c_uint64),
        ("ms_generate", C.c_double), ("ms_trace", C.c_double), ("ms_shade", C.c_double),
        ("ms_shadow", C.c_double), ("ms_resolve", C.c_double),
        ("n_trace_launches", C.c_uint64), ("n_shadow_launches", C.c_uint64),
        ("num_triangles", C.c_uint64), ("num_bvh_nodes", C.c_uint64), ("bvh_max_depth", C.c_uint64),
        ("bvh_bytes", C.c_uint64), ("tri_bytes", C.c_uint64), ("ms_bvh_build", C.c_double),
    ]


class FramePtrs(C.Structure):
    _fields_ = [
        ("frame_buffer", C.c_void_p), ("accum_buffer", C.c_void_p), ("color_buffer", C.c_void_p),
        ("normal_buffer", C.c_void_p), ("albedo_buffer", C.c_void_p),
    ]


assert C.sizeof(Material) == 104
assert C.sizeof(Probe) == 64
assert C.sizeof(LaunchParams) == 248
assert LaunchParams.camera.offset == 104 and LaunchParams.traversable.offset == 160
assert LaunchParams.probe.offset == 168 and LaunchParams.viewportSize.offset == 232
assert _Frame.c.offset == 72 and _Frame.offset.offset == 88 and _Frame.size.offset == 40
