"""Diagnostic (needs a -DFOVPT_V_STEPSTAT=1 build, FOVPT_SO=...): per-ray traversal steps of the bounce-1
closest-hit rays in queue order, and what a wave of 16 consecutive rays would cost under other orders."""
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
cfg.max_depth = 2                       # bounce 1 is the last traced segment: its rays stay in ray_o / ray_d
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
n1 = sh[:, 1]                             # radiance queue sizes of iteration 1, per shard
qo = buf("queue_b_o", np.float32).reshape(-1, 4); qd = buf("queue_b_d", np.float32).reshape(-1, 4); cap = qo.shape[0] // 8
sel = np.concatenate([np.arange(s * cap, s * cap + n1[s]) for s in range(8)])
slots = qo[sel, 3].view(np.uint32)
st = qd[sel, 3].view(np.uint32)                  # node | leaf << 16 steps, written by the STEPSTAT traversal
steps = (st & 0xfff) + 2.7 * ((st >> 12) & 0xff)         # a leaf step costs about 2.7 node steps
d = qd[sel, :3]; o = qo[sel, :3]
print("bounce-1 rays:", slots.size, "mean cost %.1f" % steps.mean(), "node %.2f leaf %.2f no-hit steps %.2f skipped pops %.2f" % ((st & 0xfff).mean(), ((st >> 12) & 0xff).mean(), ((st >> 20) & 63).mean(), (st >> 26).mean()))
def wave_cost(order, label):
    s = steps[order]
    n = s.size // 16 * 16
    m = s[:n].reshape(-1, 16)
    print("%-40s lane use %.3f" % (label, m.mean() / m.max(1).mean()))
idx = np.arange(slots.size)
wave_cost(idx, "queue order")
octant = (d[:, 0] > 0).astype(int) | ((d[:, 1] > 0).astype(int) << 1) | ((d[:, 2] > 0).astype(int) << 2)
blk = idx // 256
wave_cost(np.lexsort((octant, blk)), "octant within 256-blocks")
wave_cost(np.lexsort((steps, blk)), "by true cost within 256-blocks (bound)")
wave_cost(np.argsort(steps, kind="stable"), "by true cost globally (bound)")
up = np.digitize(d[:, 1], [-0.5, 0.0, 0.5])
wave_cost(np.lexsort((up, blk)), "4 elevation bins within 256-blocks")
wave_cost(np.lexsort((octant, idx // 1024)), "octant within 1024-blocks")
rng = np.random.default_rng(0)
wave_cost(rng.permutation(slots.size), "random order")
for name, key in (("dir.y", d[:, 1]), ("origin.y", o[:, 1]), ("|dir.x|", np.abs(d[:, 0]))):
    print("corr(cost, %s) = %.3f" % (name, np.corrcoef(steps, key)[0, 1]))
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], steps=st, d=d.astype(np.float16), o=o.astype(np.float32))
