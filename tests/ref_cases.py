"""The unit-level cases on which the oracle is pinned to the reference's own code.

Every case has `gen(rng) -> inputs` (seeded numpy arrays) and `run(L, prefix, inputs) -> outputs`, where L is a
ctypes library offering the flat entry points `<prefix>name(...)`:

    prefix "ref_"  oracle/_ref/libfovpt_ref.so  = the reference's headers compiled as they lie (oracle/ref_shim.cpp)
    prefix "orc_"  oracle/libfovpt_oracle.so    = the restatement under test

tests/golden/make_ref_golden.py runs the cases on the reference and commits inputs + outputs as
tests/golden/ref_vectors.npz; tests/test_ref_pin_cpu.py replays them on the oracle (bit-exact in libm mode) and
on the GPU kernels (within the detmath-vs-libm budget).  TEST INFRASTRUCTURE ONLY.
"""
import ctypes as C

import numpy as np


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


_ALIAS = {"orc_probe_sample": "orc_probe_sample2", "orc_make_color": "orc_make_color_raw"}


def _fn(L, prefix, name, restype=None):
    full = prefix + name
    f = getattr(L, _ALIAS.get(full, full))
    f.restype = restype
    return f


def unit_dirs(rng, n):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    # axis-aligned and degenerate directions too
    special = np.float32([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [0.6, 0.8, 0], [0, 0.6, 0.8]])
    d[:len(special)] = special
    return d


def hdr_probe(rng, w, h):
    """A seeded HDR-like probe: smooth sky gradient, a sun of a few very bright texels, a dark ground."""
    v = (np.arange(h, dtype=np.float64)[:, None] + 0.5) / h
    u = (np.arange(w, dtype=np.float64)[None, :] + 0.5) / w
    sky = 0.3 + 1.5 * np.clip(0.5 - v, 0, 1) + 0.2 * np.sin(6.283 * u) ** 2
    img = np.stack([sky * 0.8, sky * 0.9, sky * 1.2, np.ones_like(sky)], axis=-1)
    img[v[:, 0] > 0.55] *= 0.15
    for _ in range(3):
        j, i = int(rng.integers(1, h // 2)), int(rng.integers(0, w))
        img[j, i, :3] += rng.uniform(200, 4000)
    img[..., :3] *= rng.uniform(0.5, 1.5, size=(h, w, 1))
    return img.astype(np.float32)


# ---- reference Model* -> arrays (oracle/ref_shim.cpp ref_model_*) ----------------------------------
def dump_model(L, h):
    L.ref_model_num_meshes.argtypes = [C.c_void_p]
    L.ref_model_num_textures.argtypes = [C.c_void_p]
    out = {"num_meshes": np.int32(L.ref_model_num_meshes(h)), "num_textures": np.int32(L.ref_model_num_textures(h))}
    for k in range(int(out["num_meshes"])):
        cnt = np.zeros(5, np.int32)
        mat = np.zeros(104, np.uint8)
        L.ref_model_mesh_info(C.c_void_p(h), k, _p(cnt), _p(mat))
        v, n, t, i = (np.zeros((cnt[0], 3), np.float32), np.zeros((cnt[1], 3), np.float32), np.zeros((cnt[2], 2), np.float32),
                      np.zeros((cnt[3], 3), np.uint32))
        L.ref_model_mesh_data(C.c_void_p(h), k, _p(v), _p(n), _p(t), _p(i))
        out.update({"mesh%d_vertex" % k: v, "mesh%d_normal" % k: n, "mesh%d_texcoord" % k: t, "mesh%d_index" % k: i,
                    "mesh%d_material" % k: mat, "mesh%d_texture_id" % k: np.int32(cnt[4])})
    for k in range(int(out["num_textures"])):
        wh = np.zeros(2, np.int32)
        L.ref_model_texture_info(C.c_void_p(h), k, _p(wh))
        px = np.zeros((wh[1], wh[0]), np.uint32)
        L.ref_model_texture_data(C.c_void_p(h), k, _p(px))
        out["texture%d" % k] = px
    return out


# ---- cases --------------------------------------------------------------------------------------
def gen_rng(rng):
    return dict(tea_a=rng.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32),
                tea_b=np.concatenate([np.arange(8, dtype=np.uint32), rng.integers(0, 2**32, 56, dtype=np.uint64).astype(np.uint32)]),
                lcg_seeds=rng.integers(0, 2**32, 8, dtype=np.uint64).astype(np.uint32),
                random_seeds=np.concatenate([np.int32([0, 1, -1, 2**31 - 1, -2**31]), rng.integers(-2**31, 2**31, 11).astype(np.int32)]))


def run_rng(L, prefix, inp):
    tea = _fn(L, prefix, "tea4", C.c_uint32)
    out = dict(tea=np.array([tea(C.c_uint32(int(a)), C.c_uint32(int(b))) for a, b in zip(inp["tea_a"], inp["tea_b"])], np.uint32))
    n = 64
    lcg_u, lcg_f = [], []
    for s in inp["lcg_seeds"]:
        u, f = np.empty(n, np.uint32), np.empty(n, np.float32)
        _fn(L, prefix, "lcg_stream")(C.c_uint32(int(s)), n, _p(u), _p(f))
        lcg_u.append(u); lcg_f.append(f)
    out["lcg"], out["rnd"] = np.stack(lcg_u), np.stack(lcg_f)
    ru, rf, s2, st = [], [], [], []
    for s in inp["random_seeds"]:
        u, f = np.empty(n, np.uint32), np.empty(n, np.float32)
        _fn(L, prefix, "random_stream")(C.c_int(int(s)), n, _p(u), _p(f))
        ru.append(u); rf.append(f)
        o2, sa = np.empty((n, 2), np.float32), np.empty(2, np.uint32)
        _fn(L, prefix, "sample2d_stream")(C.c_int(int(s)), n, _p(o2), _p(sa))
        s2.append(o2); st.append(sa)
    out["rand"], out["randf"], out["sample2d"], out["sample2d_state"] = np.stack(ru), np.stack(rf), np.stack(s2), np.stack(st)
    return out


def gen_samplers(rng):
    n = 2048
    w = unit_dirs(rng, n)
    w[8:16] *= np.float32(1e-3)                    # BasisFromVector / SafeNormalize do not assume unit length
    a = rng.normal(size=(n, 3)).astype(np.float32)
    a[:4] = 0.0
    a[4:8] = np.float32(1e-25)                      # dot underflows to 0 -> fallback
    return dict(w=w, a=a, u2=rng.uniform(0, 1, (n, 2)).astype(np.float32),
                rgba=rng.uniform(0, 50, (n, 4)).astype(np.float32), seeds=rng.integers(-2**31, 2**31, 4).astype(np.int32))


def run_samplers(L, prefix, inp):
    n = inp["w"].shape[0]
    u, v = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    _fn(L, prefix, "basis_from_vector")(n, _p(inp["w"]), _p(u), _p(v))
    sn = np.empty((n, 3), np.float32)
    _fn(L, prefix, "safe_normalize")(n, _p(inp["a"]), _p(sn))
    cos = np.empty((n, 3), np.float32)
    _fn(L, prefix, "cosine_hemisphere")(n, _p(inp["u2"]), _p(cos))
    lum = np.empty(n, np.float32)
    _fn(L, prefix, "luminance")(n, _p(inp["rgba"]), _p(lum))
    uh, us = [], []
    for s in inp["seeds"]:
        o, st = np.empty((256, 3), np.float32), np.empty(2, np.uint32)
        _fn(L, prefix, "uniform_hemisphere")(C.c_int(int(s)), 256, _p(o), _p(st))
        uh.append(o); us.append(st)
    return dict(basis_u=u, basis_v=v, safe_normalize=sn, cosine_hemisphere=cos, luminance=lum,
                uniform_hemisphere=np.stack(uh), uniform_hemisphere_state=np.stack(us))


def gen_probe(rng):
    """Two probes (SURVEY 8c golden vector 3): a seeded HDR-like 64x32 one and a constant one; the CDF tables are
    INPUTS here (Probe.h's BuildCDF is not compilable), built by sequential fp32 sums in numpy."""
    out = {}
    for name, img in (("hdr", hdr_probe(rng, 64, 32)), ("const", np.tile(np.float32([2.5, 2.5, 2.5, 1.0]), (16, 48, 1)))):
        lum = (img[..., 0] * np.float32(0.3) + img[..., 1] * np.float32(0.6) + img[..., 2] * np.float32(0.1)).astype(np.float32)
        cdfx = np.cumsum(lum, axis=1, dtype=np.float32)
        tot = cdfx[:, -1].copy()
        inv = (np.float32(1.0) / tot).astype(np.float32)
        pdfx = (lum * inv[:, None]).astype(np.float32)
        cdfx = (cdfx * inv[:, None]).astype(np.float32)
        cdfy = np.cumsum(tot, dtype=np.float32)
        pdfy = (tot / cdfy[-1]).astype(np.float32)
        cdfy = (cdfy / cdfy[-1]).astype(np.float32)
        out.update({name + "_data": img, name + "_pdfx": pdfx, name + "_cdfx": cdfx, name + "_pdfy": pdfy, name + "_cdfy": cdfy})
    n = 4096
    out["dirs"] = unit_dirs(rng, n)
    uv = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    uv[:6] = np.float32([[0, 0], [1, 1], [0.5, 0.5], [0.999999, 0.0], [0.25, 1.0], [1.0, 0.25]])
    out["uv"] = uv
    out["lb_values"] = np.concatenate([np.float32([0.0, 1.0, 0.999999, 1e-9]), rng.uniform(0, 1, 508).astype(np.float32)])
    out["seeds"] = np.int32([7, 123456789])
    return out


def run_probe(L, prefix, inp):
    n = inp["dirs"].shape[0]
    out = {}
    uv = np.empty((n, 2), np.float32)
    _fn(L, prefix, "probe_dir_to_uv")(n, _p(inp["dirs"]), _p(uv))
    out["dir_to_uv"] = uv
    d = np.empty((n, 3), np.float32)
    _fn(L, prefix, "probe_uv_to_dir")(n, _p(inp["uv"]), _p(d))
    out["uv_to_dir"] = d
    for name in ("hdr", "const"):
        data = inp[name + "_data"]
        h, w = data.shape[:2]
        ev = np.empty((n, 4), np.float32)
        _fn(L, prefix, "probe_eval")(w, h, _p(data), n, _p(inp["uv"]), _p(ev))
        out[name + "_eval"] = ev
        m = len(inp["lb_values"])
        lb = np.empty(m, np.int32)
        _fn(L, prefix, "lower_bound")(_p(inp[name + "_cdfy"]), 0, h, m, _p(inp["lb_values"]), _p(lb))
        out[name + "_lower_bound_rows"] = lb
        lb2 = np.empty(m, np.int32)
        row = h // 3
        _fn(L, prefix, "lower_bound")(_p(inp[name + "_cdfx"]), row * w, (row + 1) * w, m, _p(inp["lb_values"]), _p(lb2))
        out[name + "_lower_bound_cols"] = lb2
        for k, seed in enumerate(inp["seeds"]):
            dd, cc, pp, st = np.empty((4096, 3), np.float32), np.empty((4096, 3), np.float32), np.empty(4096, np.float32), np.empty(2, np.uint32)
            _fn(L, prefix, "probe_sample")(w, h, _p(data), _p(inp[name + "_pdfx"]), _p(inp[name + "_cdfx"]), _p(inp[name + "_pdfy"]),
                                           _p(inp[name + "_cdfy"]), C.c_int(int(seed)), 4096, _p(dd), _p(cc), _p(pp), _p(st))
            out["%s_sample%d_dir" % (name, k)], out["%s_sample%d_color" % (name, k)] = dd, cc
            out["%s_sample%d_pdf" % (name, k)], out["%s_sample%d_state" % (name, k)] = pp, st
    return out


def gen_color(rng):
    n = 4096
    rgb = rng.uniform(0, 1.2, (n, 3)).astype(np.float32)
    rgb[:8] = np.float32([[0, 0, 0], [1, 1, 1], [0.0031308, 0.0031307, 0.0031309], [0.5, 0.25, 0.75], [2, -1, 0.5],
                          [1e-6, 0.999999, 0.003], [0.2, 0.4, 0.6], [0.73, 0.73, 0.73]])
    x = np.concatenate([np.float32([-1, 0, 1, 2, 0.99609375, 0.996, 0.5, 1.0 / 256]), rng.uniform(-0.1, 1.1, n - 8).astype(np.float32)])
    return dict(rgb=rgb, x=x)


def run_color(L, prefix, inp):
    n = inp["rgb"].shape[0]
    mc = np.empty(n, np.uint32)
    _fn(L, prefix, "make_color")(n, _p(inp["rgb"]), _p(mc))
    clamped = np.clip(inp["rgb"], 0, 1).astype(np.float32)
    srgb = np.empty((n, 3), np.float32)
    _fn(L, prefix, "to_srgb")(n, _p(clamped), _p(srgb))
    q = np.empty(n, np.uint8)
    _fn(L, prefix, "quantize8")(n, _p(inp["x"]), _p(q))
    return dict(make_color=mc, to_srgb=srgb, quantize8=q)


def gen_vec(rng):
    n = 1024
    a = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(3)
    b = rng.normal(size=(n, 3)).astype(np.float32)
    b[np.abs(b) < 1e-3] = 0.5
    s = rng.uniform(0.05, 4, n).astype(np.float32)
    return dict(a=a, b=b, s=s)


def run_vec(L, prefix, inp):
    n = inp["a"].shape[0]
    out = {}
    for op, name in enumerate(["normalize", "cross", "div_scalar", "lerp", "faceforward", "clamp_0_10", "mul", "scalar_minus", "div"]):
        o = np.empty((n, 3), np.float32)
        _fn(L, prefix, "vec3_op")(op, n, _p(inp["a"]), _p(inp["b"]), _p(inp["s"]), _p(o))
        out[name] = o
    d, l = np.empty(n, np.float32), np.empty(n, np.float32)
    _fn(L, prefix, "vec3_dot_length")(n, _p(inp["a"]), _p(inp["b"]), _p(d), _p(l))
    out["dot"], out["length"] = d, l
    return out


def gen_camera(rng):
    cams = [((-1293.07, 154.681, -0.7304), (-1000.0, 120.0, 0.0), (0, 1, 0), 45.0, 16 / 9.0),     # main.cpp:240-251 style
            ((278, 273, -800), (278, 273, 0), (0, 1, 0), 40.0, 1.0),
            ((0, 0, 5), (0.3, -0.2, 0), (0.1, 1, 0.05), 60.0, 1.5),
            ((10, 20, 30), (-5, 2, 7), (0, 0, 1), 35.0, 0.8)]
    return dict(cams=np.float32([list(e) + list(l) + list(u) + [f, a] for e, l, u, f, a in cams]))


def run_camera(L, prefix, inp):
    out = []
    f = _fn(L, prefix, "camera_uvw")
    for row in inp["cams"]:
        e, l, u = _f32(row[0:3]), _f32(row[3:6]), _f32(row[6:9])
        U, V, W = np.empty(3, np.float32), np.empty(3, np.float32), np.empty(3, np.float32)
        f(_p(e), _p(l), _p(u), C.c_float(float(row[9])), C.c_float(float(row[10])), _p(U), _p(V), _p(W))
        out.append(np.concatenate([U, V, W]))
    return dict(uvw=np.stack(out))


def gen_material(rng):
    return dict(eta=np.float32([0.0, 0.0, 0.0, 1.4, 1.0, 2.5]), specular=np.float32([0.5, 1.0, 0.0, 1.0, 0.3, 0.9]))


def run_material(L, prefix, inp):
    f = _fn(L, prefix, "material_ior", C.c_float)
    return dict(ior=np.float32([f(C.c_float(float(e)), C.c_float(float(s))) for e, s in zip(inp["eta"], inp["specular"])]))


CASES = {
    "rng": (gen_rng, run_rng),               # a19: tea<4>, lcg, rnd, Random, Sample2D
    "samplers": (gen_samplers, run_samplers),  # a20: BasisFromVector, SafeNormalize, hemisphere samplers, Luminance
    "probe": (gen_probe, run_probe),         # a14: ProbeDirToUV, ProbeUVToDir, ProbeEval, LowerBound, ProbeSample
    "color": (gen_color, run_color),         # a8: toSRGB, quantizeUnsigned8Bits, make_color
    "vec": (gen_vec, run_vec),               # sutil/vec_math.h subset
    "camera": (gen_camera, run_camera),      # a21: Camera::UVWFrame
    "material": (gen_material, run_material),  # a3: GetIndexOfRefraction
}

# outputs that go through libm's sinf/cosf/acosf/atan2f/powf/tanf: bit-exact against the oracle in libm mode; against
# the oracle in detmath mode (what the GPU computes) they differ by the libm-vs-correctly-rounded budget
TRANSCENDENTAL = {"samplers/cosine_hemisphere", "samplers/uniform_hemisphere", "probe/dir_to_uv", "probe/uv_to_dir",
                  "color/make_color", "color/to_srgb", "camera/uvw"}


def is_transcendental(key):
    return key in TRANSCENDENTAL or (key.startswith("probe/") and ("_sample" in key) and (key.endswith("_dir") or key.endswith("_pdf")))


def run_all(L, prefix, inputs=None, seed=20260204):
    """-> (inputs, outputs), flat dicts keyed 'case/name'."""
    rng = np.random.default_rng(seed)
    all_in, all_out = {}, {}
    for case, (gen, run) in CASES.items():
        inp = gen(rng) if inputs is None else {k.split("/", 1)[1]: v for k, v in inputs.items() if k.startswith(case + "/")}
        out = run(L, prefix, inp)
        all_in.update({case + "/" + k: v for k, v in inp.items()})
        all_out.update({case + "/" + k: v for k, v in out.items()})
    return all_in, all_out
