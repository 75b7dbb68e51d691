"""Loader for libfovpt.so (the HIP C-ABI library, include/fovpt.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("FOVPT_SO") or os.path.join(CSRC, "libfovpt.so")   # FOVPT_SO: A/B builds of the same library

EXPORTS = [
    "fovpt_create", "fovpt_destroy", "fovpt_last_error", "fovpt_set_scene", "fovpt_set_probe", "fovpt_set_probe_data",
    "fovpt_resize", "fovpt_get_config", "fovpt_set_config", "fovpt_launch", "fovpt_render",
    "fovpt_synchronize", "fovpt_download", "fovpt_get_stats", "fovpt_reset_stats", "fovpt_stream",
    "fovpt_probe_build_cdf", "fovpt_camera_uvw", "fovpt_debug_math", "fovpt_debug_buffer", "fovpt_debug_trace",
    "fovpt_gather_plan", "fovpt_gather_pack", "fovpt_gather_unpack",
    "fovpt_comm_get_unique_id", "fovpt_comm_init", "fovpt_comm_destroy", "fovpt_gather_frame",
    "fovpt_model_load_obj", "fovpt_model_load_gltf", "fovpt_model_destroy", "fovpt_model_counts", "fovpt_model_get_mesh", "fovpt_model_get_texture",
    "fovpt_image_load_float4", "fovpt_image_free", "fovpt_image_load_rgba8", "fovpt_image_free_rgba8",
]


class FovptError(RuntimeError):
    """The std::runtime_error of the reference (sutil::Exception, sutil/Exception.h:245)."""

    def __init__(self, code, msg):
        super().__init__("libfovpt error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libfovpt.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's, found by file name
    through an RPATH).  Imported FIRST, torch's copy also serves libfovpt.so; imported AFTER libfovpt.so has
    pulled in /opt/rocm's copy, torch loads a second HIP/HSA runtime into the process, which finds "No HIP
    GPUs".  So when torch is installed but not imported yet, its copy is loaded here, and either import order
    ends with one runtime.  Without torch (C++ callers, plain Python) the library binds to /opt/rocm as linked.
    FOVPT_SYSTEM_HIP=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FOVPT_SYSTEM_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def share_torch_rccl():
    """The same for RCCL, before the library's own transport (fovpt_comm_*) loads one: PyTorch bundles a librccl.so of its own
    (another file than /opt/rocm's librccl.so.1) and the library takes whichever copy the process already holds: with torch
    installed but not imported yet, torch's copy is loaded here, so that either order ends with one RCCL.  Loaded RTLD_LOCAL, as
    the library loads it: RCCL brings librocm_smi64 along, whose `amd::smi` globals also exist in /opt/rocm's libamd_smi.so
    (which torch's device queries load) -- made global, the two run their static destructors on one object and the process
    aborts with a double free when it exits (round 4: this transport first, `import torch` afterwards).
    FOVPT_SYSTEM_HIP=1 or FOVPT_RCCL_LIB skip this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("FOVPT_SYSTEM_HIP") or os.environ.get("FOVPT_RCCL_LIB"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    rccl = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(rccl):
        try:
            C.CDLL(rccl)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FovptError(-100, "%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % SO_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(SO_PATH)
    vp, i32, u32, u64, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_size_t
    L.fovpt_create.argtypes = [C.POINTER(vp), i32]
    L.fovpt_destroy.argtypes = [vp]
    L.fovpt_destroy.restype = None
    L.fovpt_last_error.argtypes = [vp]
    L.fovpt_last_error.restype = C.c_char_p
    L.fovpt_set_scene.argtypes = [vp, vp, i32, vp, i32, C.POINTER(u64)]
    L.fovpt_set_probe.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(abi.Probe)]
    L.fovpt_set_probe_data.argtypes = [vp, i32, i32, vp, vp, C.POINTER(abi.Probe)]
    L.fovpt_resize.argtypes = [vp, i32, i32, C.POINTER(abi.FramePtrs)]
    L.fovpt_get_config.argtypes = [vp, C.POINTER(abi.Config)]
    L.fovpt_set_config.argtypes = [vp, C.POINTER(abi.Config)]
    L.fovpt_launch.argtypes = [vp, C.POINTER(abi.LaunchParams), u32, u32]
    L.fovpt_render.argtypes = [vp, C.POINTER(abi.LaunchParams)]
    L.fovpt_synchronize.argtypes = [vp]
    L.fovpt_download.argtypes = [vp, vp, vp, sz]
    L.fovpt_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.fovpt_reset_stats.argtypes = [vp]
    L.fovpt_stream.argtypes = [vp]
    L.fovpt_stream.restype = vp
    L.fovpt_probe_build_cdf.argtypes = [i32, i32, vp, vp, vp, vp, vp]
    L.fovpt_camera_uvw.argtypes = [C.POINTER(abi.Float3), C.POINTER(abi.Float3), C.POINTER(abi.Float3),
                                   C.c_float, C.c_float, C.POINTER(abi.Float3), C.POINTER(abi.Float3), C.POINTER(abi.Float3)]
    L.fovpt_gather_plan.argtypes = [vp, C.POINTER(abi.LaunchParams), vp, i32]
    L.fovpt_gather_pack.argtypes = [vp, vp, vp]
    L.fovpt_gather_unpack.argtypes = [vp, vp, u32, vp]
    L.fovpt_comm_get_unique_id.argtypes = [vp]
    L.fovpt_comm_init.argtypes = [vp, vp, i32, i32]
    L.fovpt_comm_destroy.argtypes = [vp]
    L.fovpt_gather_frame.argtypes = [vp, C.POINTER(abi.LaunchParams), i32, vp, vp]
    L.fovpt_debug_math.argtypes = [vp, i32, vp, vp, vp, sz]
    L.fovpt_debug_buffer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(sz)]
    L.fovpt_debug_trace.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.fovpt_model_load_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fovpt_model_load_gltf.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fovpt_model_destroy.argtypes = [vp]
    L.fovpt_model_destroy.restype = None
    L.fovpt_model_counts.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.fovpt_model_get_mesh.argtypes = [vp, i32, C.POINTER(abi.ModelMesh)]
    L.fovpt_model_get_texture.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.fovpt_image_load_float4.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]
    L.fovpt_image_free.argtypes = [vp]
    L.fovpt_image_free.restype = None
    L.fovpt_image_load_rgba8.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]
    L.fovpt_image_free_rgba8.argtypes = [vp]
    L.fovpt_image_free_rgba8.restype = None
    for name in EXPORTS:
        if name not in ("fovpt_destroy", "fovpt_last_error", "fovpt_stream", "fovpt_model_destroy", "fovpt_image_free", "fovpt_image_free_rgba8"):
            getattr(L, name).restype = i32
    _lib = L
    return L


LOADER_SO_PATH = os.path.join(CSRC, "libfovpt_loader.so")
LOADER_EXPORTS = [n for n in EXPORTS if n.startswith("fovpt_model_") or n.startswith("fovpt_image_")] + ["fovpt_last_error"]
_loader = None


def load_loader():
    """The scene / image ingestion of the C ABI (fovpt_model_*, fovpt_image_*).  If libfovpt.so is already loaded, it; otherwise the
    HOST-ONLY libfovpt_loader.so (same code, g++, no HIP runtime), so that reading OBJ / glTF / JPEG / HDR files needs no GPU
    stack; FOVPT_SO (an A/B build of the whole library) takes precedence.  The render path never goes through here."""
    global _loader
    if _lib is not None or os.environ.get("FOVPT_SO") or not os.path.exists(LOADER_SO_PATH):
        return load()
    if _loader is not None:
        return _loader
    L = C.CDLL(LOADER_SO_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.fovpt_last_error.argtypes = [vp]
    L.fovpt_last_error.restype = C.c_char_p
    L.fovpt_model_load_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fovpt_model_load_gltf.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fovpt_model_destroy.argtypes = [vp]
    L.fovpt_model_destroy.restype = None
    L.fovpt_model_counts.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.fovpt_model_get_mesh.argtypes = [vp, i32, C.POINTER(abi.ModelMesh)]
    L.fovpt_model_get_texture.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.fovpt_image_load_float4.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]
    L.fovpt_image_free.argtypes = [vp]
    L.fovpt_image_free.restype = None
    L.fovpt_image_load_rgba8.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]
    L.fovpt_image_free_rgba8.argtypes = [vp]
    L.fovpt_image_free_rgba8.restype = None
    for name in LOADER_EXPORTS:
        if name not in ("fovpt_last_error", "fovpt_model_destroy", "fovpt_image_free", "fovpt_image_free_rgba8"):
            getattr(L, name).restype = i32
    _loader = L
    return L


def check(ctx, rc, L=None):
    """Raises FovptError with the library's text; `L`: the library the failing call was made in (the loader library keeps its own
    error text)."""
    if rc != 0:
        msg = (L or load()).fovpt_last_error(ctx)
        raise FovptError(rc, msg.decode("utf-8", "replace") if msg else "")
