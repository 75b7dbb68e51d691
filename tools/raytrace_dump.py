"""Diagnostic (needs a -DFOVPT_V_STEPSTAT=1 build, FOVPT_SO=...): dumps, for the closest-hit rays of bounce B of the
C3 frame in queue order, the node steps of every node phase (between two leaf visits) -> gpurun_out/raytrace_b<B>.npz,
the input of tools/raysim.py (offline simulation of wave scheduling policies)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes, lib
W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
model = scenes.atrium(ntri)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.max_depth = B + 1                   # bounce B is the last traced segment: its rays stay in the queue
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
raw = buf("counters", np.uint8)
stride = (raw.size - 24 - 64) // 8 // 4
sh = raw[:8 * stride * 4].view(np.uint32).reshape(8, stride)
nq = sh[:, B]                             # radiance queue sizes of iteration B, per shard
q = "queue_b" if B % 2 else "queue_a"
qd = buf(q + "_d", np.float32).reshape(-1, 4); cap = qd.shape[0] // 8
qo = buf(q + "_o", np.float32).reshape(-1, 4)
tr = buf("trace", np.uint8).reshape(-1, 16)
hit = buf("hit", np.float32).reshape(-1, 4)
sel = np.concatenate([np.arange(s * cap, s * cap + nq[s]) for s in range(8)])
st = qd[sel, 3].view(np.uint32)
print("bounce", B, "rays", sel.size, "node %.2f leaf %.2f no-hit %.2f skipped %.2f" % ((st & 0xfff).mean(), ((st >> 12) & 0xff).mean(), ((st >> 20) & 63).mean(), (st >> 26).mean()), "miss frac %.3f" % (hit[sel, 3].view(np.uint32) == 0xffffffff).mean())
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/raytrace_b%d_%d.npz" % (B, ntri), steps=st, trace=tr[sel], shard_sizes=nq, miss=(hit[sel, 3].view(np.uint32) == 0xffffffff),
                    o=qo[sel, :3].astype(np.float32), d=qd[sel, :3].astype(np.float16), t=hit[sel, 0].astype(np.float32))
