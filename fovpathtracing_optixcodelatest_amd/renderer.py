"""Host-side mirror of the reference's renderer interface for the foveated launch path.

Same names, argument meaning and error behaviour as PT_sv5_/SimplePathtracer.h:45-72 and the
types it consumes (sutil::Camera, ProbeData), on top of the C ABI in include/fovpt.h.  All
compute happens in libfovpt.so (HIP, gfx950); nothing here touches pixels.
"""
import ctypes as C

import numpy as np

from . import abi, lib
from .scenes import Model, pack_model


class Camera:
    """sutil::Camera (sutil/Camera.h:40-100): eye, lookat, up, fovY (degrees), aspect ratio."""

    def __init__(self, eye=(1.0, 1.0, 1.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=35.0, aspectRatio=1.0):
        self.m_eye, self.m_lookat, self.m_up = tuple(eye), tuple(lookat), tuple(up)
        self.m_fovY, self.m_aspectRatio = float(fovY), float(aspectRatio)

    def eye(self):
        return self.m_eye

    def setEye(self, v):
        self.m_eye = tuple(v)

    def setAspectRatio(self, a):
        self.m_aspectRatio = float(a)

    def UVWFrame(self):
        """sutil/Camera.cpp:32-44, evaluated by the library's host helper."""
        e, l, u = abi.Float3().set(self.m_eye), abi.Float3().set(self.m_lookat), abi.Float3().set(self.m_up)
        U, V, W = abi.Float3(), abi.Float3(), abi.Float3()
        rc = lib.load().fovpt_camera_uvw(C.byref(e), C.byref(l), C.byref(u), self.m_fovY, self.m_aspectRatio,
                                         C.byref(U), C.byref(V), C.byref(W))
        lib.check(None, rc)
        return U, V, W


class ProbeData:
    """PT_sv5_/Probe.h:7-86.  data: (H,W,4) float32; BuildCDF() fills the four tables."""

    def __init__(self, data=None):
        self.valid = False
        self.offset = (0.0, 0.0, 0.0)
        self.data = None
        self.width = self.height = 0
        if data is not None:
            self.data = np.ascontiguousarray(data, np.float32)
            self.height, self.width = self.data.shape[:2]

    def BuildCDF(self):
        h, w = self.height, self.width
        self.pdfValuesX = np.empty((h, w), np.float32)
        self.cdfValuesX = np.empty((h, w), np.float32)
        self.pdfValuesY = np.empty(h, np.float32)
        self.cdfValuesY = np.empty(h, np.float32)
        rc = lib.load().fovpt_probe_build_cdf(w, h, self.data.ctypes.data, self.pdfValuesX.ctypes.data,
                                              self.cdfValuesX.ctypes.data, self.pdfValuesY.ctypes.data,
                                              self.cdfValuesY.ctypes.data)
        lib.check(None, rc)
        self.valid = True
        return self


class OutputBuffer:
    """Shape of sutil::CUDAOutputBuffer<uint32_t> (sutil/CUDAOutputBuffer.h:54-94): anything with
    map() -> device pointer of W*H uint32 and unmap().  Wraps a caller-owned device pointer."""

    def __init__(self, device_ptr):
        self._p = int(device_ptr)

    def map(self):
        return self._p

    def unmap(self):
        pass


class SampleRenderer:
    """class SampleRenderer, PT_sv5_/SimplePathtracer.h:45-188 (public part)."""

    def __init__(self, model: Model, device: int = 0):
        self._L = lib.load()
        self._ctx = C.c_void_p()
        lib.check(None, self._L.fovpt_create(C.byref(self._ctx), device))
        self.launchParams = abi.LaunchParams()
        self.model = model
        md, n, td, nt, keep = pack_model(model)
        trav = C.c_uint64()
        self._check(self._L.fovpt_set_scene(self._ctx, C.cast(md, C.c_void_p), n, C.cast(td, C.c_void_p), nt, C.byref(trav)))
        self.launchParams.traversable = trav.value                      # SimplePathtracer.cpp:61
        self.lastSetCamera = Camera()
        self._frame_ptrs = abi.FramePtrs()

    # -- plumbing -----------------------------------------------------------------------
    def _check(self, rc):
        lib.check(self._ctx, rc)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.fovpt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._L.fovpt_stream(self._ctx)

    @property
    def config(self) -> abi.Config:
        c = abi.Config()
        self._check(self._L.fovpt_get_config(self._ctx, C.byref(c)))
        return c

    @config.setter
    def config(self, c: abi.Config):
        self._check(self._L.fovpt_set_config(self._ctx, C.byref(c)))

    def stats(self) -> abi.Stats:
        s = abi.Stats()
        self._check(self._L.fovpt_get_stats(self._ctx, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self._L.fovpt_reset_stats(self._ctx))

    def synchronize(self):
        self._check(self._L.fovpt_synchronize(self._ctx))

    # -- the reference's public interface ---------------------------------------------------
    def render(self, target=None):
        """render() / render(CUDAOutputBuffer&), SimplePathtracer.cpp:77-226.  Like the reference,
        render(target) repoints launchParams.frame.frame_buffer at the mapped target and leaves it
        there (:218-219).  Synchronises before returning (CUDA_SYNC_CHECK, :212)."""
        if target is not None:
            self.launchParams.frame.frame_buffer = target.map()
        self._check(self._L.fovpt_render(self._ctx, C.byref(self.launchParams)))
        self.synchronize()
        if target is not None:
            target.unmap()

    def render_async(self):
        """render() without the trailing synchronisation (for back-to-back timed frames)."""
        self._check(self._L.fovpt_render(self._ctx, C.byref(self.launchParams)))

    # -- multi-GPU gather of the owned pixels (include/fovpt.h, fovpt_gather_*) -------------
    def gather_plan(self):
        """Partition of the frame's pixels by owning rank for the current config / frame size / gaze -> counts per rank."""
        world = max(1, self.config.world)
        counts = (C.c_uint32 * world)()
        self._check(self._L.fovpt_gather_plan(self._ctx, C.byref(self.launchParams), counts, world))
        return [int(x) for x in counts]

    def gather_pack(self, frame_ptr, packed_ptr):
        self._check(self._L.fovpt_gather_pack(self._ctx, frame_ptr, packed_ptr))

    def gather_unpack(self, gathered_ptr, stride, frame_ptr):
        self._check(self._L.fovpt_gather_unpack(self._ctx, gathered_ptr, stride, frame_ptr))

    # -- the RCCL transport of that gather inside the library (fovpt_comm_*, fovpt_gather_frame): what a C++ host uses
    @staticmethod
    def comm_unique_id():
        lib.share_torch_rccl()
        buf = C.create_string_buffer(128)
        lib.check(None, lib.load().fovpt_comm_get_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        lib.share_torch_rccl()
        self._check(self._L.fovpt_comm_init(self._ctx, C.c_char_p(unique_id), rank, world))

    def comm_destroy(self):
        self._check(self._L.fovpt_comm_destroy(self._ctx))

    def gather_frame(self, root, frame_ptr, full_frame_ptr):
        self._check(self._L.fovpt_gather_frame(self._ctx, C.byref(self.launchParams), root, frame_ptr, full_frame_ptr))

    def launch(self, width, height):
        """One optixLaunch with the current launchParams (SimplePathtracer.cpp:148-157)."""
        self._check(self._L.fovpt_launch(self._ctx, C.byref(self.launchParams), width, height))

    def resize(self, newSize):
        """SimplePathtracer.cpp:228-274."""
        w, h = int(newSize[0]), int(newSize[1])
        if w == 0 or h == 0:
            return
        self._check(self._L.fovpt_resize(self._ctx, w, h, C.byref(self._frame_ptrs)))
        f = self.launchParams.frame
        f.size.x, f.size.y = w, h
        f.frame_buffer = self._frame_ptrs.frame_buffer
        f.accum_buffer = self._frame_ptrs.accum_buffer
        f.normal_buffer = self._frame_ptrs.normal_buffer
        f.color_buffer = self._frame_ptrs.color_buffer
        f.albedo_buffer = self._frame_ptrs.albedo_buffer

    def downloadPixels(self):
        """SimplePathtracer.cpp:276-280: always the renderer's own frame_buffer."""
        f = self.launchParams.frame
        out = np.empty((f.size.y, f.size.x), np.uint32)
        self._check(self._L.fovpt_download(self._ctx, self._frame_ptrs.frame_buffer, out.ctypes.data, out.nbytes))
        return out

    def downloadAccum(self):
        """The float4 accum_buffer (per-pixel radiance), the quantity parity is judged on."""
        f = self.launchParams.frame
        out = np.empty((f.size.y, f.size.x, 4), np.float32)
        self._check(self._L.fovpt_download(self._ctx, f.accum_buffer, out.ctypes.data, out.nbytes))
        return out

    def download(self, device_ptr, array):
        self._check(self._L.fovpt_download(self._ctx, device_ptr, array.ctypes.data, array.nbytes))
        return array

    def setCamera(self, camera: Camera):
        """SimplePathtracer.cpp:282-289: aspect ratio is recomputed from the frame size."""
        self.lastSetCamera = camera
        f = self.launchParams.frame
        self.lastSetCamera.setAspectRatio(f.size.x / float(f.size.y) if f.size.y else 1.0)
        U, V, W = self.lastSetCamera.UVWFrame()
        cam = self.launchParams.camera
        cam.U, cam.V, cam.W = U, V, W
        cam.eye.set(self.lastSetCamera.eye())

    def setCameraFov(self, eye, forward, up, angle_left, angle_right, angle_up, angle_down):
        """Per-eye asymmetric frustum from OpenXR-style half angles (XrFovf, radians; left/down negative), as the
        VR callers of the path compute them (OtherProjects_01/11HelloRaytracingOpenXR/main.cpp:891-897).  The raygen
        maps d in [-1,1]^2 to dir = d.x*U + d.y*V + W (deviceProgram.cu:483-491), so an off-centre frustum is W
        shifted by the tangent-space centre and U, V scaled by the tangent-space half extents."""
        f = np.array(forward, np.float64)
        f = f / np.linalg.norm(f)
        r_ = np.cross(f, np.array(up, np.float64))
        r_ = r_ / np.linalg.norm(r_)
        u = np.cross(r_, f)
        tl, tr, tu, td = (float(np.tan(a)) for a in (angle_left, angle_right, angle_up, angle_down))
        cam = self.launchParams.camera
        cam.eye.set(eye)
        cam.U.set((0.5 * (tr - tl)) * r_)
        cam.V.set((0.5 * (tu - td)) * u)
        cam.W.set(f + (0.5 * (tr + tl)) * r_ + (0.5 * (tu + td)) * u)

    def setProbe(self, probe: ProbeData):
        """SimplePathtracer.cpp:292-308; raises like CUDAProbeData::createBuffer (Probe.h:104-105)."""
        if not probe.valid:
            raise RuntimeError("Probe Data is not valid")
        off = abi.Float3().set(probe.offset)
        out = abi.Probe()
        self._check(self._L.fovpt_set_probe(self._ctx, probe.width, probe.height, probe.data.ctypes.data,
                                            probe.pdfValuesX.ctypes.data, probe.cdfValuesX.ctypes.data,
                                            probe.pdfValuesY.ctypes.data, probe.cdfValuesY.ctypes.data,
                                            C.byref(off), C.byref(out)))
        self.launchParams.probe = out

    def setProbeData(self, data, offset=(0.0, 0.0, 0.0)):
        """loadColor / loadProbe + BuildCDF + setProbe in one step (main.cpp:161-187, 306), with BuildCDF
        run on the device (fovpt_set_probe_data).  Returns the device-side tables for inspection."""
        data = np.ascontiguousarray(data, np.float32)
        h, w = data.shape[:2]
        off = abi.Float3().set(offset)
        out = abi.Probe()
        self._check(self._L.fovpt_set_probe_data(self._ctx, w, h, data.ctypes.data, C.byref(off), C.byref(out)))
        self.launchParams.probe = out
        return out

    def debug_trace(self, origins, dirs):
        """The production traversal kernel on a batch of rays -> (global prim id or 0xffffffff, (t, u, v), occluded 0 / 1)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        prim, tuv, occ = np.empty(n, np.uint32), np.empty((n, 3), np.float32), np.empty(n, np.uint8)
        self._check(self._L.fovpt_debug_trace(self._ctx, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data, tuv.ctypes.data, occ.ctypes.data))
        return prim, tuv, occ

    def debug_math(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else None
        out = np.empty_like(a)
        self._check(self._L.fovpt_debug_math(self._ctx, op, a.ctypes.data, bb.ctypes.data if bb is not None else None,
                                             out.ctypes.data, a.size))
        return out
