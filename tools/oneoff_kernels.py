"""Drive the per-plan / per-probe kernels at production sizes (for `rocprofv3 --kernel-trace --stats`):
BuildCDF on a 4096 x 2048 probe and gather plans of a 3840 x 2160 foveated frame on 8 ranks with a moving gaze.

usage: [FOVPT_SO=...] python tools/oneoff_kernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes

W, H = 3840, 2160
model = scenes.atrium(20000)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
rng = np.random.default_rng(3)
data = rng.random((2048, 4096, 4), dtype=np.float32)
for _ in range(3):
    r.setProbeData(data)
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 296, 964
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.rank, cfg.world = 3, 8
r.config = cfg
for k in range(10):                                  # the plan is cached per gaze: move it
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = 1000 + 97 * k, 700 + 41 * k
    counts = r.gather_plan()
r.synchronize()
print("plan counts", counts, "sum", sum(counts))
