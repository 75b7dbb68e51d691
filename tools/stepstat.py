"""Diagnostic: per-shadow-ray traversal step histogram of the last bounce (needs a -DFOVPT_V_STEPSTAT=1 build)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes, lib
W, H = 1920, 1080
model = scenes.atrium(262144)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
r.config = cfg
r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
r.render()
L = lib.load()
def buf(name, dtype):
    p, n = C.c_void_p(), C.c_size_t()
    lib.check(r._ctx, L.fovpt_debug_buffer(r._ctx, name.encode(), C.byref(p), C.byref(n)))
    a = np.empty(n.value // np.dtype(dtype).itemsize, dtype)
    r.download(p.value, a)
    return a
cnt = buf("counters", np.uint32)
sq_counts = cnt[64 * 8:64 * 8 + 64 * 8].reshape(64, 8)
print("shadow queue sizes per iteration:", sq_counts[:4].sum(1))
occ = buf("sq_occ", np.float32).reshape(-1, 4)
vis = buf("sq_vis", np.float32).reshape(-1, 4)
cap = occ.shape[0] // 8
steps, isocc = [], []
for s in range(8):
    n = sq_counts[3][s]
    st = occ[s * cap:s * cap + n, 3].view(np.uint32)
    steps.append(st); isocc.append(vis[s * cap:s * cap + n, 3])
steps = np.concatenate(steps); isocc = np.concatenate(isocc)
nodes, leaves = steps & 0xFFFF, steps >> 16
print("last bounce shadow rays:", steps.size, "occluded fraction %.3f" % isocc.mean())
for name, v in (("node visits", nodes), ("leaf visits", leaves)):
    print(name, "mean %.1f median %d p90 %d p99 %d max %d" % (v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
for lab, m in (("occluded", isocc > 0.5), ("visible", isocc < 0.5)):
    print(lab, "n", m.sum(), "nodes mean %.1f max %d  leaves mean %.1f max %d" % (nodes[m].mean(), nodes[m].max(), leaves[m].mean(), leaves[m].max()))
