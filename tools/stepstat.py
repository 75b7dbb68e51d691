"""Diagnostic: SIMD utilisation of the traversal kernel (needs a -DFOVPT_V_STEPSTAT=1 build, FOVPT_SO=...).

Per ray kind: wave-level node / leaf steps, and how many of the 16 rays of a wave take part in each."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes, lib
W, H = 1920, 1080
street = os.environ.get("FOVPT_SCENE") == "street"
model = scenes.street(int(os.environ.get("FOVPT_TRIS", "3800000"))) if street else scenes.atrium(int(os.environ.get("FOVPT_TRIS", "262144")))
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.STREET_CAMERA if street else scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.sky_probe(512, 256, seed=5) if street else scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.max_depth = int(os.environ.get("FOVPT_DEPTH", "4"))
r.config = cfg
r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
r.reset_stats()
r.render()
st = r.stats()
L = lib.load()
p, n = C.c_void_p(), C.c_size_t()
lib.check(r._ctx, L.fovpt_debug_buffer(r._ctx, b"counters", C.byref(p), C.byref(n)))
raw = np.empty(n.value, np.uint8)
r.download(p.value, raw)
off = raw.size - 64          # the diag block is the tail of struct Counters
diag = raw[off:off + 64].view(np.uint64).reshape(2, 4)
rays = {"closest": st.radiance_rays, "any-hit": st.shadow_rays}
for k, name in enumerate(("closest", "any-hit")):
    ns, nq, ls, lq = (int(x) for x in diag[k])
    print("%-8s rays %9d | node: wave steps %10d, rays/step %5.2f of 16, steps/ray %5.1f | leaf: wave steps %9d, rays/step %5.2f, steps/ray %4.1f"
          % (name, rays[name], ns, nq / max(ns, 1), nq / max(rays[name], 1), ls, lq / max(ls, 1), lq / max(rays[name], 1)))
