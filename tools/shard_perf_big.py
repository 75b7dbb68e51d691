"""Per-rank frame time of the tile-sharded C4 / C5 frames on ONE GPU (rank 0 and the last rank of world N), with pack, without
the transport.  usage: shard_perf_big.py [atrium|street]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes
kind = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if kind == "street":
    model, cam, probe_of = scenes.street(3800000, material="app"), scenes.STREET_CAMERA, lambda W, H: scenes.sky_probe(512, 256, seed=5)
else:
    model, cam, probe_of = scenes.atrium(3800000, material="app"), scenes.ATRIUM_CAMERA, lambda W, H: scenes.ambient_probe(W, H, 2.5)
r = renderer.SampleRenderer(model)
for name, (W, H), radii, depth, worlds in (("C4", (2560, 1440), (197, 643), 4, (1, 2, 4)), ("C5 eye", (2160, 2160), (296, 964), 8, (1, 2, 4, 8))):
    r.resize((W, H))
    r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
    r.setProbe(renderer.ProbeData(probe_of(W, H)).BuildCDF())
    for world in worlds:
        for rank in sorted(set((0, world - 1))):
            cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = radii
            cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
            cfg.max_depth = depth
            cfg.rank, cfg.world = rank, world
            r.config = cfg
            r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
            counts = r.gather_plan()
            buf = torch.zeros((max(counts) + 63) // 64 * 64, dtype=torch.int32, device="cuda")
            for _ in range(3):
                r.launchParams.frame.subframe_index = 0; r.render()
            r.reset_stats(); n = 20
            t = time.time()
            for _ in range(n):
                r.launchParams.frame.subframe_index = 0; r.render_async()
                if world > 1: r.gather_pack(r.launchParams.frame.frame_buffer, buf.data_ptr())
            r.synchronize(); dt = (time.time() - t) / n
            s = r.stats()
            print("%s %s world %d rank %d: %.3f ms/frame, rays %d, packed %d B" % (name, kind, world, rank, dt * 1e3, (s.radiance_rays + s.shadow_rays) // n, 4 * max(counts)), flush=True)
