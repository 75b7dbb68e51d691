"""Frame time of the C3 frame for a caller that synchronises after every frame (the reference's main loop) and for one that does
not, for chains_per_frame = 1 / 2 and frames_in_flight = 1 / 2.  usage: python tools/latency.py [atrium|street] [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
W, H = 1920, 1080
street = len(sys.argv) > 1 and sys.argv[1] == "street"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
model = scenes.street(3800000, material="app") if street else scenes.atrium(262144)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.STREET_CAMERA if street else scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.sky_probe(512, 256, seed=5) if street else scenes.ambient_probe(W, H, 2.5)).BuildCDF())
for chains, fif in ((1, 1), (1, 2), (2, 1)):
    cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148, 482
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    cfg.chains_per_frame, cfg.frames_in_flight = chains, fif
    r.config = cfg
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
    for _ in range(5):
        r.launchParams.frame.subframe_index = 0; r.render()
    t = time.time()
    for _ in range(frames):
        r.launchParams.frame.subframe_index = 0; r.render()                      # render_async + synchronise
    sync = (time.time() - t) / frames
    t = time.time()
    for _ in range(frames):
        r.launchParams.frame.subframe_index = 0; r.render_async()
    r.synchronize()
    back = (time.time() - t) / frames
    print("chains %d, frames in flight %d: synchronised every frame %.3f ms, back to back %.3f ms" % (chains, fif, sync * 1e3, back * 1e3), flush=True)
