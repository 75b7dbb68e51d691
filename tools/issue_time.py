"""Host side of a frame: how long does the caller's thread take to ISSUE a frame (render_async returns when every launch is queued)
against how long the GPU takes to run it?  1/N shards of C3, two frames in flight.  usage: issue_time.py [world ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
r = renderer.SampleRenderer(scenes.atrium(262144)); r.resize((W, H))
cam = scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.ambient_probe(W, H, 2.5)).BuildCDF())
for world in [int(a) for a in sys.argv[1:]] or [1, 8]:
    cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    cfg.rank, cfg.world = 0, world
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
    for _ in range(6):
        r.launchParams.frame.subframe_index = 0; r.render()
    n = 200
    t = time.time()
    for _ in range(n):
        r.launchParams.frame.subframe_index = 0; r.render_async()
    t_issue = time.time() - t
    r.synchronize(); dt = time.time() - t
    # the same calls with nothing to wait for: the GPU idle between frames
    t1 = 0.0
    for _ in range(50):
        r.synchronize()
        a = time.time(); r.launchParams.frame.subframe_index = 0; r.render_async(); t1 += time.time() - a
    r.synchronize()
    print("world %d: issue %.3f ms/frame while the queues are full, %.3f ms/frame into idle queues; frames complete every %.3f ms" % (world, t_issue / n * 1e3, t1 / 50 * 1e3, dt / n * 1e3))
