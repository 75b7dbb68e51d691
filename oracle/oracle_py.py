"""ctypes access to the CPU oracle (oracle/libfovpt_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from fovpathtracing_optixcodelatest_amd import abi
from fovpathtracing_optixcodelatest_amd.scenes import pack_model

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfovpt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "fovpt_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_num_triangles.restype = C.c_uint64
        L.orc_scene_num_triangles.argtypes = [C.c_void_p]
        L.orc_launch.argtypes = [C.c_void_p, C.POINTER(abi.LaunchParams), C.c_uint32, C.c_uint32,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.LaunchParams), C.POINTER(abi.Config),
                                 C.c_int, C.c_int, C.c_void_p]
        L.orc_tea4.restype = C.c_uint32
        L.orc_tea4.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_math.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_lib_counts.argtypes = [C.c_void_p, C.c_int]
        L.orc_camera_uvw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def set_lib_counting(on: bool):
    """Counts.lib_* need one BSDF evaluation more than the reference does on occluded shadow rays; the timed CPU
    baseline turns the counting off (the counts of such a run are meaningless)."""
    lib().orc_set_lib_counting(1 if on else 0)


def set_math_mode(detmath: bool):
    lib().orc_set_math_mode(1 if detmath else 0)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    def __init__(self, model):
        md, n, td, nt, keep = pack_model(model)
        self._h = lib().orc_scene_create(C.cast(md, C.c_void_p), n, C.cast(td, C.c_void_p), nt)
        self.num_triangles = lib().orc_scene_num_triangles(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    def trace(self, origins, dirs, brute=False):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        prim = np.empty(n, np.uint32)
        tuv = np.empty((n, 3), np.float32)
        occ = np.empty(n, np.uint8)
        lib().orc_trace(self._h, n, _p(o), _p(d), int(brute), _p(prim), _p(tuv), _p(occ))
        return prim, tuv, occ


def build_cdf(data):
    """ProbeData::BuildCDF restatement.  data: (H,W,4) float32 -> pdfX, cdfX, pdfY, cdfY"""
    data = np.ascontiguousarray(data, np.float32)
    h, w = data.shape[:2]
    pdfx = np.empty((h, w), np.float32)
    cdfx = np.empty((h, w), np.float32)
    pdfy = np.empty(h, np.float32)
    cdfy = np.empty(h, np.float32)
    lib().orc_build_cdf(w, h, _p(data), _p(pdfx), _p(cdfx), _p(pdfy), _p(cdfy))
    return pdfx, cdfx, pdfy, cdfy


class HostProbe:
    """fovpt_probe with HOST pointers, for the oracle."""

    def __init__(self, data, cdf=None):
        self.data = np.ascontiguousarray(data, np.float32)
        self.pdfx, self.cdfx, self.pdfy, self.cdfy = cdf if cdf is not None else build_cdf(self.data)
        p = abi.Probe()
        p.height, p.width = self.data.shape[:2]
        p.data = self.data.ctypes.data
        p.pdfValuesX, p.cdfValuesX = self.pdfx.ctypes.data, self.cdfx.ctypes.data
        p.pdfValuesY, p.cdfValuesY = self.pdfy.ctypes.data, self.cdfy.ctypes.data
        self.struct = p


def camera_uvw(eye, lookat, up, fovy, aspect):
    e, l, u = (np.asarray(x, np.float32) for x in (eye, lookat, up))
    U, V, W = np.empty(3, np.float32), np.empty(3, np.float32), np.empty(3, np.float32)
    lib().orc_camera_uvw(_p(e), _p(l), _p(u), fovy, aspect, _p(U), _p(V), _p(W))
    return U, V, W


class OracleFrame:
    """Host frame buffers + LaunchParams for the oracle; mirrors what SampleRenderer holds."""

    def __init__(self, width, height, probe: HostProbe, camera: dict, gaze=None, subframe_index=0):
        self.w, self.h = width, height
        self.accum = np.zeros((height, width, 4), np.float32)
        self.frame = np.zeros((height, width), np.uint32)
        self.normal = np.zeros((height, width, 4), np.float32)
        self.color = np.zeros((height, width, 4), np.float32)
        self.albedo = np.zeros((height, width, 4), np.float32)
        self.probe = probe
        lp = abi.LaunchParams()
        lp.frame.accum_buffer = self.accum.ctypes.data
        lp.frame.frame_buffer = self.frame.ctypes.data
        lp.frame.normal_buffer = self.normal.ctypes.data
        lp.frame.color_buffer = self.color.ctypes.data
        lp.frame.albedo_buffer = self.albedo.ctypes.data
        lp.frame.size.x, lp.frame.size.y = width, height
        lp.frame.subframe_index = subframe_index
        gx, gy = gaze if gaze is not None else (width // 2, height // 2)
        lp.frame.c.x, lp.frame.c.y = gx, gy
        U, V, W = camera_uvw(camera["eye"], camera["lookat"], camera["up"], camera["fovy"], width / float(height))
        lp.camera.eye.set(camera["eye"])
        lp.camera.U.set(U)
        lp.camera.V.set(V)
        lp.camera.W.set(W)
        lp.probe = probe.struct
        self.lp = lp


class Counts(tuple):
    """(radiance_rays, shadow_rays, paths) as the REFERENCE traces them, plus .lib_radiance / .lib_shadow: the rays
    libfovpt traces for the same frame (it skips the reference's discarded last segment and shadow rays that cannot
    change a bit of the image; fovpt_oracle.cpp, g_lib_radiance).  The library's device counters must equal them."""
    lib_radiance = 0
    lib_shadow = 0


def _counts(cnt):
    lc = np.zeros(2, np.uint64)
    lib().orc_lib_counts(_p(lc), 1)
    c = Counts(int(x) for x in cnt)
    c.lib_radiance, c.lib_shadow = int(lc[0]), int(lc[1])
    return c


def render(scene: OracleScene, frame: OracleFrame, cfg: abi.Config, brute=False, nthreads=None):
    """SampleRenderer::render() on the CPU.  Returns Counts(radiance_rays, shadow_rays, paths)."""
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    cnt = np.zeros(3, np.uint64)
    lib().orc_lib_counts(None, 1)
    rc = lib().orc_render(scene._h, C.byref(frame.lp), C.byref(cfg), int(brute), int(nthreads), _p(cnt))
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return _counts(cnt)


def launch(scene: OracleScene, frame: OracleFrame, width, height, max_depth=4, accumulate=0, brute=False, nthreads=None):
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    cnt = np.zeros(3, np.uint64)
    lib().orc_lib_counts(None, 1)
    rc = lib().orc_launch(scene._h, C.byref(frame.lp), width, height, max_depth, accumulate, int(brute), int(nthreads), _p(cnt))
    if rc != 0:
        raise RuntimeError("orc_launch failed: %d" % rc)
    return _counts(cnt)


def math_op(op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    out = np.empty_like(a)
    lib().orc_math(op, a.size, _p(a), _p(bb), _p(out))
    return out


# ---- unit-level entry points -----------------------------------------------------------------
def tea4(a, b):
    return int(lib().orc_tea4(a & 0xFFFFFFFF, b & 0xFFFFFFFF))


def lcg_stream(seed, n):
    u = np.empty(n, np.uint32)
    f = np.empty(n, np.float32)
    lib().orc_lcg_stream(C.c_uint32(seed & 0xFFFFFFFF), n, _p(u), _p(f))
    return u, f


def random_stream(seed, n):
    u = np.empty(n, np.uint32)
    f = np.empty(n, np.float32)
    s = seed if seed < 0x80000000 else seed - (1 << 32)
    lib().orc_random_stream(C.c_int(s), n, _p(u), _p(f))
    return u, f


def probe_sample(probe: "HostProbe", seed, n):
    d = np.empty((n, 3), np.float32)
    c = np.empty((n, 3), np.float32)
    p = np.empty(n, np.float32)
    lib().orc_probe_sample(C.byref(probe.struct), C.c_int(seed), n, _p(d), _p(c), _p(p))
    return d, c, p


def probe_dir_to_uv(dirs):
    d = np.ascontiguousarray(dirs, np.float32)
    uv = np.empty((d.shape[0], 2), np.float32)
    lib().orc_probe_dir_to_uv(d.shape[0], _p(d), _p(uv))
    return uv


def bsdf_table(material, N, view, albedo, etaI, etaO, seeds):
    n = len(seeds)
    N, view, albedo = (np.ascontiguousarray(x, np.float32) for x in (N, view, albedo))
    etaI, etaO = np.ascontiguousarray(etaI, np.float32), np.ascontiguousarray(etaO, np.float32)
    seeds = np.ascontiguousarray(seeds, np.int32)
    light = np.empty((n, 3), np.float32)
    pdf = np.empty(n, np.float32)
    typ = np.empty(n, np.int32)
    ev = np.empty((n, 3), np.float32)
    pdf2 = np.empty(n, np.float32)
    rng = np.empty((n, 2), np.uint32)
    lib().orc_bsdf_table(C.byref(material), n, _p(N), _p(view), _p(albedo), _p(etaI), _p(etaO), _p(seeds),
                         _p(light), _p(pdf), _p(typ), _p(ev), _p(pdf2), _p(rng))
    return dict(light=light, pdf=pdf, type=typ, eval=ev, pdf_again=pdf2, rng_after=rng)


def make_color(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.empty(rgb.shape[0], np.uint32)
    lib().orc_make_color(rgb.shape[0], _p(rgb), _p(out))
    return out


def tex2d(texture_u32, uv):
    px = np.ascontiguousarray(texture_u32, np.uint32)
    td = abi.TextureDesc()
    td.pixel, td.width, td.height = px.ctypes.data, px.shape[1], px.shape[0]
    uv = np.ascontiguousarray(uv, np.float32)
    out = np.empty((uv.shape[0], 4), np.float32)
    lib().orc_tex2d(C.byref(td), uv.shape[0], _p(uv), _p(out))
    return out
