"""Diagnostic: where the cycles of a traversal step go (needs a -DFOVPT_V_CYCLES=1|2 build, FOVPT_SO=...).

The build stamps s_memtime (shader cycles) inside node_step / leaf_step of k_traverse:
  node step:  gap (loop overhead since the previous step) | load (address -> both 16-byte loads arrived)
              | alu (box test, hit mask, rank) | lds (stack write, wave barrier, pop)
  leaf step:  gap | load (triangle record arrived) | rest (Moeller-Trumbore, merge, pop)
in a SAMPLE of the waves (wave 0 of ~256 workgroups spread over the grid: stamping every wave slows the launch 4-8 x) and sums them per wave
(wave-level: a wave steps its 16 rays in lockstep).  With FOVPT_V_CYCLES=2 it also fills histograms of the load wait and of
whole steps.  Every wave records when it started and ended (s_memrealtime), which gives the ramp and the tail of a launch.

usage: FOVPT_SO=build/libfovpt_cyc1.so python tools/stepcycles.py [atrium|street] [frames]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes, lib

W, H = (int(x) for x in os.environ.get("FOVPT_SIZE", "1920,1080").split(","))
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
street = which == "street"
ntri = int(os.environ.get("FOVPT_TRIS", "3800000" if street else "262144"))
model = scenes.street(ntri) if street else scenes.atrium(ntri)
r = renderer.SampleRenderer(model); r.resize((W, H))
cam = scenes.STREET_CAMERA if street else scenes.ATRIUM_CAMERA
r.setCamera(renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], W / H))
r.setProbe(renderer.ProbeData(scenes.sky_probe(512, 256, seed=5) if street else scenes.ambient_probe(W, H, 2.5)).BuildCDF())
cfg = abi.Config.reference_default(); cfg.r_inner, cfg.r_outer = 148 * H // 1080, 482 * H // 1080
cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
cfg.max_depth = int(os.environ.get("FOVPT_DEPTH", "4"))
cfg.profile = int(os.environ.get("FOVPT_PROFILE", "2"))        # 2: every kernel alone (no overlap between the two streams)
r.config = cfg
r.launchParams.frame.c.x, r.launchParams.frame.c.y = W // 2, H // 2
for _ in range(2):
    r.launchParams.frame.subframe_index = 0
    r.render()
r.reset_stats()
L = lib.load()
BASE = 8 * 128 * 4 + 3 * 8 + 2 * 4 * 8
NW = 32768
SLOTS = 256 * 32                       # wave slots of the chip: 256 CUs x 4 SIMDs x 8
cyc = np.zeros((2, 8, 16), np.uint64)
hist = np.zeros((2, 8, 3, 64), np.uint64)
wt = []
seen = []
for f in range(frames):
    r.launchParams.frame.subframe_index = 0
    r.render()
    p, n = C.c_void_p(), C.c_size_t()
    lib.check(r._ctx, L.fovpt_debug_buffer(r._ctx, b"counters", C.byref(p), C.byref(n)))
    if (p.value, n.value) not in seen:
        seen.append((p.value, n.value))
st = r.stats()
HB = 2 * 8 * 3 * 64 * 4
for (ptr, n) in seen:          # the two state sets accumulate separately since reset_stats()
    raw = np.empty(n, np.uint8)
    r.download(ptr, raw)
    assert n >= BASE + cyc.nbytes + HB + 8 * NW * 16, "not a FOVPT_V_CYCLES build (counters block is %d bytes)" % n
    cyc += raw[BASE:BASE + cyc.nbytes].view(np.uint64).reshape(cyc.shape)
    hist += raw[BASE + cyc.nbytes:BASE + cyc.nbytes + HB].view(np.uint32).reshape(hist.shape)
    wt.append(raw[BASE + cyc.nbytes + HB:BASE + cyc.nbytes + HB + 8 * NW * 16].view(np.uint64).reshape(8, NW, 2).copy())
print("scene %s, %d triangles, %d frame(s), profile %d; rays/frame: closest %d, any-hit %d"
      % (which, model.num_triangles, frames, cfg.profile, st.radiance_rays // frames, st.shadow_rays // frames))
print("ms/frame (stage timers): closest %.3f any-hit %.3f shade %.3f" % (st.ms_trace / frames, st.ms_shadow / frames, st.ms_shade / frames))
names = ("closest", "any-hit")
for k in range(2):
    for it in range(4):
        v = [float(x) for x in cyc[k, it]]
        n_node, gap, load, alu, lds, n_leaf, lgap, lload, lrest, cal, ncal, life, real, waves = v[:14]
        if waves == 0 or n_node == 0:
            if wt and (wt[-1][k * 4 + it][:, 1] > 0).any():
                print("%-7s it %d |" % (names[k], it))
                node = leaf = 0
            else:
                continue
        clock = life / (real / 100e6) / 1e9 if real else float("nan")        # s_memrealtime ticks at 100 MHz
        if n_node:
          node = gap + load + alu + lds
          leaf = lgap + lload + lrest
          print("%-7s it %d | sampled waves %4d  clock %.2f GHz  wave life %7.0f cyc (%.1f us)  in steps %4.1f %%  stamp %3.0f cyc"
              % (names[k], it, waves, clock, life / waves, life / waves / clock / 1e3, 100 * (node + leaf) / life, cal / max(ncal, 1)))
          print("          node steps/wave %6.1f  cyc/step %6.0f = gap %4.0f + load %4.0f + alu %4.0f + lds %4.0f   (%.0f / %.0f / %.0f / %.0f %%)"
              % (n_node / waves, node / n_node, gap / n_node, load / n_node, alu / n_node, lds / n_node,
                 100 * gap / node, 100 * load / node, 100 * alu / node, 100 * lds / node))
        if n_leaf:
            print("          leaf steps/wave %6.1f  cyc/step %6.0f = gap %4.0f + load %4.0f + rest %4.0f" % (n_leaf / waves, leaf / n_leaf, lgap / n_leaf, lload / n_leaf, lrest / n_leaf))
        for hi, (hname, width) in enumerate((("node load wait", 16), ("node step", 32), ("leaf step", 32))):
            h = hist[k, it, hi].astype(np.float64)
            if h.sum() == 0:
                continue
            c = np.cumsum(h) / h.sum()
            q = [int(np.searchsorted(c, x)) * width for x in (0.1, 0.5, 0.9, 0.99)]
            print("          hist %-14s p10 %5d  p50 %5d  p90 %5d  p99 %5d cyc (bins of %d; last bin = overflow: %.1f %%)" % (hname, *q, width, 100 * h[-1] / h.sum()))
            if os.environ.get("FOVPT_HIST_FULL"):
                print("            " + " ".join("%d" % x for x in h))
        # when the waves of the LAST launch of this kind / iteration started and ended (10 ns ticks)
        w = wt[-1][k * 4 + it]
        w = w[w[:, 1] > 0].astype(np.float64)
        if len(w):
            t0, t1 = w[:, 0].min(), w[:, 1].max()
            span = (t1 - t0) / 100.0
            lifeus = (w[:, 1] - w[:, 0]) / 100.0
            endus = (w[:, 1] - t0) / 100.0
            startus = (w[:, 0] - t0) / 100.0
            slots = min(len(w), SLOTS)             # one-wave workgroups: more waves than slots, the dispatcher starts one as one ends
            print("          launch: %d waves, first start -> last end %.1f us; wave start mean %.1f (p99 %.1f) us, wave end mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us;"
                  " life mean %.1f us; wave slots busy %.1f %% of the span" % (len(w), span, startus.mean(), np.percentile(startus, 99), endus.mean(),
                  np.percentile(endus, 10), np.percentile(endus, 50), np.percentile(endus, 90), endus.max(), lifeus.mean(), 100 * lifeus.sum() / (slots * span)))
