"""Diagnostic (needs a -DFOVPT_V_STEPSTAT=1 build, FOVPT_SO=...): traversal steps per closest-hit ray of bounce 0 and bounce 1
-- node steps, leaf steps, node steps in which no child was hit, popped entries dropped without a step (pruning build) --
on the C3 frame of the atrium or, with FOVPT_SCENE=street, of the street.  usage: stepcount.py [triangles]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes, lib
W, H = 1920, 1080
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
street = os.environ.get("FOVPT_SCENE") == "street"
model = scenes.street(ntri, material="app") if street else scenes.atrium(ntri)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.STREET_CAMERA if street else scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.sky_probe(W, H, seed=11) if street else scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.max_depth = 2                       # bounce 1 is the last traced segment: both queues still hold their rays
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
s = r.stats()
print("scene", "street" if street else "atrium", model.num_triangles, "tris, bvh nodes", s.num_bvh_nodes, "depth", s.bvh_max_depth)
per_slot = {}
for b, q in ((0, "queue_a_d"), (1, "queue_b_d")):
    qd = buf(q, np.float32).reshape(-1, 4); cap = qd.shape[0] // 8
    qo = buf(q.replace("_d", "_o"), np.float32).reshape(-1, 4)
    n = sh[:, b]
    sel = np.concatenate([np.arange(k * cap, k * cap + n[k]) for k in range(8)])
    st = qd[sel, 3].view(np.uint32)
    nn, nl, n0, nk = st & 0xfff, (st >> 12) & 0xff, (st >> 20) & 63, st >> 26
    per_slot[b] = (qo[sel, 3].view(np.uint32), nn.astype(np.float64) + 1.4 * nl, qd[sel, :3], qo[sel, :3])
    w = nn[: nn.size // 16 * 16].reshape(-1, 16)
    print("bounce %d: %d rays, node steps %.2f leaf steps %.2f no-hit node steps %.2f (%.1f %%) dropped pops %.2f | wave: max node steps %.1f, lane use %.3f"
          % (b, st.size, nn.mean(), nl.mean(), n0.mean(), 100.0 * n0.sum() / max(1, nn.sum()), nk.mean(), w.max(1).mean(), w.mean() / w.max(1).mean()))

# how well does what is known when a bounce-1 ray is queued predict its cost?  (an order of a block's rays by estimated length is
# the idle-lane reduction that has not been tried: DESIGN.md section 8)
s0, c0, _, _ = per_slot[0]; s1, c1, d1, o1 = per_slot[1]
cost0 = np.zeros(int(max(s0.max(), s1.max())) + 1); cost0[s0] = c0
prev = cost0[s1]
print("bounce-1 cost (node + 1.4 leaf steps) against: the same path's bounce-0 cost %.3f, dir.y %.3f, |dir.y| %.3f, origin.y %.3f, |dir.x| %.3f, |dir.z| %.3f"
      % tuple(np.corrcoef(c1, k)[0, 1] for k in (prev, d1[:, 1], np.abs(d1[:, 1]), o1[:, 1], np.abs(d1[:, 0]), np.abs(d1[:, 2]))))
blk = np.arange(c1.size) // 256
def lane_use(order, label):
    m = c1[order][: c1.size // 16 * 16].reshape(-1, 16)
    print("  %-46s lane use of static rounds %.3f" % (label, m.mean() / m.max(1).mean()))
lane_use(np.arange(c1.size), "queue order")
lane_use(np.lexsort((prev, blk)), "by the path's bounce-0 cost within 256-blocks")
lane_use(np.lexsort((d1[:, 1], blk)), "by dir.y within 256-blocks")
lane_use(np.lexsort((c1, blk)), "by true cost within 256-blocks (bound)")
