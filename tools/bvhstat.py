"""Shape of the 4-wide hierarchy the GPU build produces (FOVPT_SO=... for a variant): children per node, triangles
per leaf, and the SAH cost with a leaf step priced at 2.7 node steps (what it costs the traversal kernel)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import renderer, scenes, lib
model = scenes.atrium(int(sys.argv[1]) if len(sys.argv) > 1 else 262144)
r = renderer.SampleRenderer(model)
L = lib.load()
p, n = C.c_void_p(), C.c_size_t()
lib.check(r._ctx, L.fovpt_debug_buffer(r._ctx, b"bvh_nodes", C.byref(p), C.byref(n)))
raw = np.empty(n.value // 4, np.float32)
r.download(p.value, raw)
ch = raw.reshape(-1, 4, 8)                                  # node, child, (lo.xyz, hi.xyz, code, pad)
lo, hi, code = ch[..., 0:3], ch[..., 3:6], ch[..., 6].view(np.int32)
used = np.isfinite(lo[..., 0]) & (hi[..., 0] >= lo[..., 0])
ext = np.where(used[..., None], hi - lo, 0.0).astype(np.float64)
area = ext[..., 0] * ext[..., 1] + ext[..., 1] * ext[..., 2] + ext[..., 2] * ext[..., 0]
leaf = used & (code < 0)
inner = used & (code >= 0)
cnt = ((~code) & 7) + 1
root_lo = np.where(used[0][:, None], lo[0], np.inf).min(0); root_hi = np.where(used[0][:, None], hi[0], -np.inf).max(0)
e = (root_hi - root_lo).astype(np.float64); root_area = e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
print("wide nodes %d, children per node %.2f (hist %s)" % (ch.shape[0], used.sum(1).mean(), np.bincount(used.sum(1), minlength=5)[1:].tolist()))
print("leaves %d, triangles per leaf %.2f (hist %s)" % (leaf.sum(), cnt[leaf].mean(), np.bincount(cnt[leaf], minlength=5)[1:].tolist()))
# expected steps of a random long ray: it enters a node's box with probability area / root_area
node_area = np.zeros(ch.shape[0]); node_area[0] = root_area
node_area[code[inner]] = area[inner]
print("SAH: expected node steps %.2f, leaf steps %.2f, cost %.2f node-step equivalents"
      % (node_area.sum() / root_area, area[leaf].sum() / root_area, (node_area.sum() + 2.7 * area[leaf].sum()) / root_area))
