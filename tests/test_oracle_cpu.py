"""CPU tests of the oracle (the checker) itself: known answers, self-consistency, golden frames.

The reference ships no tests for this path (parity unpinned upstream); what can be pinned is
pinned here: integer RNG streams against an independent pure-Python restatement, traversal
against brute force, the deterministic math against libm, and regression frames.
"""
import json
import os

import numpy as np
import pytest

from fovpathtracing_optixcodelatest_amd import abi, scenes
from common import cfg_foveated

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rng_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    for a, b, want in kat["tea4"]:
        assert oracle.tea4(a, b) == want
    for seed, want in kat["lcg"].items():
        u, f = oracle.lcg_stream(int(seed), len(want))
        assert u.tolist() == want
        assert np.array_equal(f, (np.array(want, np.float64) / 16777216.0).astype(np.float32))   # rnd(), random.h:101-104
    for seed, want in kat["random"].items():
        u, f = oracle.random_stream(int(seed), len(want))
        assert u.tolist() == want
        ref = np.clip((np.array(want, np.uint32).astype(np.float32) * np.float32(2.0 ** -32)), np.float32(0), np.float32(0.999999))
        assert np.array_equal(f, ref.astype(np.float32))


def _random_rays(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("which", ["cornell", "atrium"])
def test_bvh_equals_brute_force(oracle, which):
    if which == "cornell":
        model, lo, hi, n = scenes.cornell_box(), 0.0, 555.0, 20000
    else:
        model, lo, hi, n = scenes.atrium(6000), -900.0, 900.0, 1500
    S = oracle.OracleScene(model)
    o, d = _random_rays(n, lo, hi, 3)
    p1, t1, c1 = S.trace(o, d, brute=False)
    p2, t2, c2 = S.trace(o, d, brute=True)
    assert np.array_equal(p1, p2) and np.array_equal(c1, c2)
    assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
    assert (p1 != 0xFFFFFFFF).sum() > n // 4


def test_closest_hit_tie_break_lowest_primitive(oracle):
    """Two coincident triangles: the lower global primitive id wins (intersection contract)."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    m = scenes.Model([scenes.TriangleMesh(v, np.array([[0, 1, 2]], np.uint32), scenes.matte((1, 1, 1))),
                      scenes.TriangleMesh(v.copy(), np.array([[0, 1, 2]], np.uint32), scenes.matte((1, 0, 0)))])
    S = oracle.OracleScene(m)
    o = np.array([[0.2, 0.2, 1.0], [0.2, 0.2, -1.0]], np.float32)
    d = np.array([[0, 0, -1.0], [0, 0, 1.0]], np.float32)
    for brute in (False, True):
        p, t, occ = S.trace(o, d, brute=brute)
        assert p.tolist() == [0, 0]
        # seen from +z the triangle is counter-clockwise = front face -> occludes; from -z it is culled
        assert occ.tolist() == [1, 0]


def test_golden_frames(oracle):
    g = np.load(os.path.join(GOLD, "frames.npz"))
    S = oracle.OracleScene(scenes.cornell_box())
    F = oracle.OracleFrame(64, 64, oracle.HostProbe(scenes.ambient_probe(64, 32, 0.2)), scenes.CORNELL_CAMERA)
    cfg = abi.Config.reference_default()
    cfg.uniform, cfg.spp_uniform, cfg.max_depth = 1, 4, 3
    cnt = oracle.render(S, F, cfg, nthreads=4)        # threads must not change anything
    assert np.array_equal(F.accum.view(np.uint32), g["cornell_accum"].view(np.uint32))
    assert np.array_equal(F.frame, g["cornell_frame"])
    assert list(cnt) == g["cornell_counts"].tolist()
    F = oracle.OracleFrame(128, 72, oracle.HostProbe(scenes.sky_probe()), scenes.CORNELL_CAMERA)
    cfg = abi.Config.reference_default()
    cfg.r_inner, cfg.r_outer = 10, 30
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    cnt = oracle.render(S, F, cfg, brute=True)
    assert np.array_equal(F.accum.view(np.uint32), g["fov_accum"].view(np.uint32))
    assert np.array_equal(F.frame, g["fov_frame"])
    assert list(cnt) == g["fov_counts"].tolist()


def test_render_equals_three_launches(oracle):
    """render() == the three optixLaunch calls of SimplePathtracer.cpp:137-209, in order."""
    S = oracle.OracleScene(scenes.cornell_box())
    probe = oracle.HostProbe(scenes.sky_probe())
    A = oracle.OracleFrame(96, 64, probe, scenes.CORNELL_CAMERA)
    cfg = abi.Config.reference_default()
    cfg.r_inner, cfg.r_outer = 8, 20
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 4
    oracle.render(S, A, cfg)
    B = oracle.OracleFrame(96, 64, probe, scenes.CORNELL_CAMERA)
    f = B.lp.frame
    f.factor.x, f.factor.y, f.factor.z, f.fillSize = 4, 4, 1, 4
    f.r_inner, f.r_outer, f.offset.x, f.offset.y, f.redraw = 20.0, 1e9, 0, 0, 0
    B.lp.samples_per_launch = 1
    oracle.launch(S, B, 96 // 4, 64 // 4)
    f.factor.x, f.factor.y, f.fillSize = 2, 2, 2
    f.r_inner, f.r_outer, f.offset.x, f.offset.y, f.redraw = 8.0, 22.0, 48 - 22, 32 - 22, 1
    B.lp.samples_per_launch = 2
    oracle.launch(S, B, 22, 22)
    f.factor.x, f.factor.y, f.fillSize = 1, 1, 1
    f.r_inner, f.r_outer, f.offset.x, f.offset.y = 0.0, 9.0, 48 - 9, 32 - 9
    B.lp.samples_per_launch = 4
    oracle.launch(S, B, 18, 18)
    assert np.array_equal(A.accum.view(np.uint32), B.accum.view(np.uint32))
    assert np.array_equal(A.frame, B.frame)
    assert A.lp.frame.subframe_index == 1            # :210-211
    # every pixel got a writer and the three rings have distinct footprints
    assert (A.accum[..., 3] == 1.0).all()


def _ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("op,lo,hi,b", [
    (abi.OP_SIN, -7.0, 7.0, None), (abi.OP_COS, -7.0, 7.0, None), (abi.OP_ACOS, -1.0, 1.0, None),
    (abi.OP_LOG, 1e-6, 4.0, None), (abi.OP_ATAN2, -3.0, 3.0, "rand"), (abi.OP_POW, 0.0, 1.0, 1.0 / 2.4),
])
def test_detmath_matches_libm_within_one_ulp(oracle, op, lo, hi, b):
    rng = np.random.default_rng(op)
    a = rng.uniform(lo, hi, 100000).astype(np.float32)
    bb = None
    if b == "rand":
        bb = rng.uniform(lo, hi, a.size).astype(np.float32)
    elif b is not None:
        bb = np.full_like(a, np.float32(b))
    oracle.set_math_mode(True)
    det = oracle.math_op(op, a, bb)
    oracle.set_math_mode(False)
    ref = oracle.math_op(op, a, bb)
    oracle.set_math_mode(True)
    if op in (abi.OP_SIN, abi.OP_COS):
        # near zeros of sin/cos an ulp is tiny; compare absolutely there
        ok = (_ulp_diff(det, ref) <= 1) | (np.abs(det - ref) <= 1e-7)
        assert ok.all()
    else:
        assert _ulp_diff(det, ref).max() <= 1
    # agreement with libm in the vast majority of arguments (glibc's acosf/atan2f are 1-ulp, not
    # correctly rounded, hence the lower bar for them)
    assert (det.view(np.uint32) == ref.view(np.uint32)).mean() > (0.75 if op in (abi.OP_ACOS, abi.OP_ATAN2) else 0.97)


def test_detmath_is_correctly_rounded(oracle):
    """include/fovpt_detmath.h against float64 numpy rounded once to float32."""
    rng = np.random.default_rng(9)
    x = rng.uniform(-7, 7, 100000).astype(np.float32)
    u = rng.uniform(-1, 1, 100000).astype(np.float32)
    p = rng.uniform(1e-6, 4, 100000).astype(np.float32)
    q = rng.uniform(0, 1, 100000).astype(np.float32)
    y = rng.uniform(-3, 3, 100000).astype(np.float32)
    e = np.full_like(q, np.float32(1.0) / np.float32(2.4))
    f64 = np.float64
    cases = [
        (abi.OP_SIN, x, None, np.sin(x.astype(f64))), (abi.OP_COS, x, None, np.cos(x.astype(f64))),
        (abi.OP_ACOS, u, None, np.arccos(u.astype(f64))), (abi.OP_LOG, p, None, np.log(p.astype(f64))),
        (abi.OP_ATAN2, y, x, np.arctan2(y.astype(f64), x.astype(f64))), (abi.OP_POW, q, e, np.power(q.astype(f64), e.astype(f64))),
    ]
    oracle.set_math_mode(True)
    for op, a, b, ref in cases:
        got = oracle.math_op(op, a, b)
        assert (got.view(np.uint32) == ref.astype(np.float32).view(np.uint32)).mean() > 0.9999, op


def test_detmath_vs_libm_image(oracle):
    """The pure restatement (libm) and the deterministic-math contract give the same picture."""
    S = oracle.OracleScene(scenes.cornell_box())
    cfg = abi.Config.reference_default()
    cfg.uniform, cfg.spp_uniform, cfg.max_depth = 1, 4, 3
    imgs = []
    for det in (True, False):
        oracle.set_math_mode(det)
        F = oracle.OracleFrame(96, 96, oracle.HostProbe(scenes.sky_probe()), scenes.CORNELL_CAMERA)
        oracle.render(S, F, cfg)
        imgs.append(F.accum[..., :3].astype(np.float64))
    oracle.set_math_mode(True)
    rel = np.sqrt(((imgs[0] - imgs[1]) ** 2).sum() / (imgs[1] ** 2).sum())
    assert rel <= 1e-4, rel


def test_build_cdf_properties_and_host_helper(oracle):
    from fovpathtracing_optixcodelatest_amd import renderer
    data = scenes.sky_probe(64, 32)
    pdfx, cdfx, pdfy, cdfy = oracle.build_cdf(data)
    assert (np.diff(cdfy) >= 0).all() and abs(cdfy[-1] - 1.0) < 1e-6
    assert (np.diff(cdfx, axis=1) >= 0).all() and np.allclose(cdfx[:, -1], 1.0, atol=1e-5)
    assert abs(pdfy.sum() - 1.0) < 1e-4
    p = renderer.ProbeData(data).BuildCDF()     # the library's host-side BuildCDF (fovpt_probe_build_cdf)
    for a, b in zip((p.pdfValuesX, p.cdfValuesX, p.pdfValuesY, p.cdfValuesY), (pdfx, cdfx, pdfy, cdfy)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(RuntimeError):
        # setProbe on an un-built probe throws like CUDAProbeData::createBuffer (Probe.h:104-105);
        # checked before any device call
        renderer.SampleRenderer.setProbe(object.__new__(renderer.SampleRenderer), renderer.ProbeData(data))


def test_camera_uvw_matches_host_helper(oracle):
    from fovpathtracing_optixcodelatest_amd import renderer
    for cam, aspect in ((scenes.CORNELL_CAMERA, 1.0), (scenes.ATRIUM_CAMERA, 16 / 9), (scenes.ATRIUM_CAMERA, 0.5)):
        U, V, W = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], aspect)
        u, v, w = renderer.Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], aspect).UVWFrame()
        assert np.array_equal(np.float32(u.tolist()), U) and np.array_equal(np.float32(v.tolist()), V) and np.array_equal(np.float32(w.tolist()), W)
        assert abs(np.dot(U, V)) < 1e-2 and abs(np.dot(U, W)) < 1e-2


def test_probe_sample_properties(oracle):
    probe = oracle.HostProbe(scenes.sky_probe(64, 32))
    d, c, p = oracle.probe_sample(probe, 77, 4096)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    assert (p >= 0).all() and np.isfinite(p).all()
    # importance sampling: the sun texels (bright) are drawn far more often than their area share
    bright = c[:, 0] > 5.0
    assert bright.mean() > 0.3
    uv = oracle.probe_dir_to_uv(d)
    assert (uv >= 0).all() and (uv <= 1.0001).all()


def _coverage_numpy(w, h, cx, cy, r_in, r_out):
    """Which pixels the three launches of render() write (SimplePathtracer.cpp:133-209, deviceProgram.cu:433-440,
    :545-554), restated with numpy: launch grid -> pixel index in uint32 -> ring test in binary32 -> clamped block fill."""
    written = np.zeros((h, w), bool)
    u32 = lambda a: np.asarray(a, np.int64) & 0xffffffff
    passes = [(4, 4, float(r_out), 1e9, 0, 0, w // 4, h // 4),
              (2, 2, float(r_in), float(r_out + 2), cx - (r_out + 2), cy - (r_out + 2), r_out + 2, r_out + 2),
              (1, 1, 0.0, float(r_in + 1), cx - (r_in + 1), cy - (r_in + 1), 2 * (r_in + 1), 2 * (r_in + 1))]
    for f, fill, lo, hi, offx, offy, gw, gh in passes:
        ly, lx = np.meshgrid(np.arange(gh, dtype=np.int64), np.arange(gw, dtype=np.int64), indexing="ij")
        ix, iy = u32(lx * f + u32(offx)), u32(ly * f + u32(offy))
        dx = ix.astype(np.float32) - np.float32(u32(cx)); dy = iy.astype(np.float32) - np.float32(u32(cy))
        rng_ = np.sqrt(dx * dx + dy * dy + np.float32(0.0))
        alive = ~((rng_ < np.float32(lo)) | (rng_ > np.float32(hi)))
        for i in range(fill):
            for j in range(fill):
                px = np.minimum(u32(lx * f + i + u32(offx)), w - 1); py = np.minimum(u32(ly * f + j + u32(offy)), h - 1)
                written[py[alive], px[alive]] = True
    return written


def test_foveation_coverage_against_numpy(oracle):
    """With nothing to hit and a constant probe every written pixel holds the probe colour, so the frame shows exactly
    WHICH pixels the three passes write: compared with a numpy restatement of the launch geometry, gaze on and off the
    frame centre, at the border and with offsets that wrap below zero."""
    tri = np.float32([[0, 0, 900], [1, 0, 900], [0, 1, 900]])           # behind the camera
    model = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri, index=np.uint32([[0, 1, 2]]), material=scenes.matte((0.5, 0.5, 0.5)))])
    cam = dict(eye=(0.0, 0.0, 10.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovy=40.0)
    w, h = 256, 144
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.ambient_probe(32, 16, 0.25))
    for (cx, cy), (ri, ro) in (((128, 72), (20, 64)), ((60, 100), (17, 41)), ((5, 3), (12, 30)), ((250, 140), (25, 70)), ((128, 72), (74, 241))):
        F = oracle.OracleFrame(w, h, hp, cam, gaze=(cx, cy))
        cfg = cfg_foveated(ri, ro, (1, 2, 4))
        oracle.render(S, F, cfg)
        want = _coverage_numpy(w, h, cx, cy, ri, ro)
        got = F.frame != 0
        assert np.array_equal(got, want), ((cx, cy), (ri, ro), int((got != want).sum()))
        assert 0.3 < want.mean() <= 1.0
        assert np.array_equal(F.accum[want][:, :3], np.full((int(want.sum()), 3), 0.25, np.float32))
        assert (F.accum[~want] == 0).all()


def test_camera_rays_and_jitter_against_python(oracle):
    """Seed = tea<4>(launch index in the FULL frame width, subframe), jitter = two LCG draws with x first, the pixel ->
    direction map through U, V, W (deviceProgram.cu:411, :479-487): restated with the pure-Python integer RNG of
    tests/golden/make_golden.py and binary64, and checked through visibility -- which pixels of a 1-spp frame see an
    emissive rectangle whose silhouette falls between pixel centres."""
    sys_path_golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(sys_path_golden, "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    a, b = 2.37, 1.41
    quad = np.float32([[-a, -b, 0], [a, -b, 0], [a, b, 0], [-a, b, 0]])
    model = scenes.Model(meshes=[scenes.TriangleMesh(vertex=quad, index=np.uint32([[0, 1, 2], [0, 2, 3]]),
                                                     material=scenes.matte((0.5, 0.5, 0.5), emission=(5.0, 5.0, 5.0)))])
    cam = dict(eye=(0.3, -0.2, 10.0), lookat=(0.0, 0.1, 0.0), up=(0.0, 1.0, 0.0), fovy=40.0)
    w, h = 96, 64
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.ambient_probe(32, 16, 0.25))
    from common import cfg_uniform
    for subframe in (0, 7):
        F = oracle.OracleFrame(w, h, hp, cam, subframe_index=subframe)
        oracle.render(S, F, cfg_uniform(1, max_depth=1))
        saw = ~np.all(F.accum[..., :3] == np.float32(0.25), axis=2)          # anything but the backplate
        U, V, W = (np.float64(x) for x in oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h))
        eye = np.float64(cam["eye"])
        want = np.zeros((h, w), bool); sure = np.zeros((h, w), bool)
        for ly in range(h):
            for lx in range(w):
                # the shipped FOV_OFF branch renders with subframe_index 0 whatever the caller's counter says (SimplePathtracer.cpp:93)
                seed = G.tea4(ly * w + lx, 0)
                jx, jy = (v / float(1 << 24) for v in G.lcg_stream(seed, 2))
                dx, dy = 2.0 * (lx + jx) / w - 1.0, 2.0 * (ly + jy) / h - 1.0
                d = dx * U + dy * V + W
                t = -eye[2] / d[2]
                x, y = eye[0] + t * d[0], eye[1] + t * d[1]
                want[ly, lx] = abs(x) < a and abs(y) < b
                sure[ly, lx] = min(abs(abs(x) - a), abs(abs(y) - b)) > 1e-4
        assert sure.mean() > 0.99 and 0.1 < want.mean() < 0.8
        assert np.array_equal(saw[sure], want[sure]), int((saw != want)[sure].sum())
        # the jitter matters: pixel centres alone (no jitter) give a different silhouette
        edge = np.zeros((h, w), bool)
        for ly in range(h):
            for lx in range(w):
                d = (2.0 * (lx + 0.5) / w - 1.0) * U + (2.0 * (ly + 0.5) / h - 1.0) * V + W
                t = -eye[2] / d[2]
                edge[ly, lx] = abs(eye[0] + t * d[0]) < a and abs(eye[1] + t * d[1]) < b
        assert (edge != want).sum() > 10


def test_probe_sampling_against_numpy(oracle):
    """BuildCDF (Probe.h:29-77) and ProbeSample (Probe.cuh:138-169) restated with numpy in binary64 -- cumulative sums,
    searchsorted(side="left") for LowerBound, the pdf and direction formulae -- on the oracle's own Random stream
    (pinned separately by tests/golden/rng_kat.json)."""
    rng = np.random.default_rng(9)
    for data in (scenes.sky_probe(64, 32), np.concatenate([rng.uniform(0.0, 4.0, (24, 40, 3)), np.ones((24, 40, 1))], axis=2).astype(np.float32)):
        h, w = data.shape[:2]
        probe = oracle.HostProbe(data)
        lum = 0.3 * data[..., 0].astype(np.float64) + 0.6 * data[..., 1] + 0.1 * data[..., 2]     # maths.h:165
        row_sum = lum.sum(1)
        assert np.allclose(probe.pdfx, lum / row_sum[:, None], rtol=2e-5, atol=1e-9)
        assert np.allclose(probe.cdfx, np.cumsum(lum, 1) / row_sum[:, None], rtol=2e-5, atol=1e-7)
        assert np.allclose(probe.pdfy, row_sum / row_sum.sum(), rtol=2e-5) and np.allclose(probe.cdfy, np.cumsum(row_sum) / row_sum.sum(), rtol=2e-5)
        n = 5000
        d, c, p = oracle.probe_sample(probe, 1234, n)
        _, f = oracle.random_stream(1234, 2 * n)                     # Sample2D = two Randf() draws per sample (sample.h:254-259)
        r1, r2 = f[0::2], f[1::2]
        row = np.minimum(np.searchsorted(probe.cdfy, r1, side="left"), h)
        inside = row < h
        rowc = np.minimum(row, h - 1)
        col = np.array([np.searchsorted(probe.cdfx[rr], x, side="left") for rr, x in zip(rowc, r2)])
        inside &= col < w
        colc = np.minimum(col, w - 1)
        assert inside.mean() > 0.999
        assert np.array_equal(c[inside], data[rowc, colc, :3][inside])
        theta, phi = rowc / float(h) * np.pi, colc / float(w) * 2.0 * np.pi          # texel corner, no +0.5
        want_d = np.stack([-np.sin(theta) * np.cos(phi), np.cos(theta), -np.sin(theta) * np.sin(phi)], 1)
        assert np.abs(d[inside] - want_d[inside]).max() < 2e-6
        st = np.sin(theta)
        with np.errstate(all="ignore"):
            want_p = np.where(np.float32(st) == 0.0, 0.0, probe.pdfx[rowc, colc].astype(np.float64) * probe.pdfy[rowc] * w * h / (2.0 * np.pi * np.pi * st))
        ok = inside & (row > 0)                                       # row 0: sin(0) == 0 -> pdf 0 in both
        assert np.allclose(p[ok], want_p[ok], rtol=1e-4) and (p[inside & (row == 0)] == 0).all()


def test_bsdf_table_sanity(oracle):
    rng = np.random.default_rng(5)
    n = 4096
    N = rng.normal(size=(n, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    view = rng.normal(size=(n, 3)); view /= np.linalg.norm(view, axis=1, keepdims=True)
    flip = (N * view).sum(1) < 0
    view[flip] *= -1                                  # wo on the side of the (face-forwarded) normal
    alb = rng.uniform(0.05, 1.0, (n, 3))
    for mat in (abi.Material.reference_default(), scenes.matte((0.7, 0.7, 0.7)), scenes.diffuse_only((0.5, 0.5, 0.5))):
        t = oracle.bsdf_table(mat, N, view, alb, np.ones(n), np.full(n, 1.4), np.arange(n))
        ok = t["pdf"] > 0
        assert ok.mean() > 0.5
        assert np.isfinite(t["eval"][ok]).all() and (t["eval"][ok] >= 0).all()
        assert np.allclose(np.linalg.norm(t["light"][ok], axis=1), 1.0, atol=1e-3)
        nonspec = ok & (t["type"] != 2)               # eSpecular returns its own pdf (Disney.cuh:236-241)
        assert np.array_equal(t["pdf"][nonspec], t["pdf_again"][nonspec])
        if mat.transmission == 0.0:
            assert (t["type"][ok] != 2).all()


def test_bsdf_against_an_independent_binary64_restatement(oracle):
    """BSDFEval / BSDFPdf of the oracle (binary32, the reference's operation order) against tests/disney_f64.py
    (binary64 numpy, written separately from Disney.cuh's formulae) on the directions BSDFSample draws: every lobe --
    diffuse, GGX, clearcoat, subsurface, transmission with and without total internal reflection."""
    import disney_f64 as D
    rng = np.random.default_rng(17)
    n = 20000
    N = rng.normal(size=(n, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    view = rng.normal(size=(n, 3)); view /= np.linalg.norm(view, axis=1, keepdims=True)
    view[(N * view).sum(1) < 0] *= -1
    alb = rng.uniform(0.0, 1.0, (n, 3)); alb[:50] = 0.0                     # black albedo: the Ctint fallback
    mats = [abi.Material.reference_default(), scenes.matte((0.7, 0.6, 0.5)), scenes.diffuse_only((0.5, 0.5, 0.5))]
    for kw in (dict(subsurface=0.6, transmission=0.0, metallic=0.2, roughness=0.3, clearcoat=0.8, clearcoatGloss=0.3, specularTint=0.4, specular=0.7),
               dict(subsurface=0.3, transmission=0.7, metallic=0.0, roughness=0.05, clearcoat=0.0, clearcoatGloss=1.0, specularTint=0.0, specular=1.0),
               dict(subsurface=0.0, transmission=1.0, metallic=0.6, roughness=0.6, clearcoat=1.0, clearcoatGloss=0.0, specularTint=1.0, specular=0.2)):
        m = abi.Material.reference_default()
        m.color.set((0.8, 0.3, 0.1))
        for k, v in kw.items():
            setattr(m, k, v)
        mats.append(m)
    checked = 0
    for mat in mats:
        for eta_i, eta_o in ((1.0, 1.4), (1.4, 1.0)):                        # entering, leaving (TIR happens)
            t = oracle.bsdf_table(mat, N, view, alb, np.full(n, eta_i), np.full(n, eta_o), np.arange(n) * 7 + 1)
            L = t["light"].astype(np.float64)
            ok = (t["pdf"] > 0) & (np.abs(np.linalg.norm(L, axis=1) - 1.0) < 1e-3)
            Nn, Vv = N.astype(np.float32).astype(np.float64), view.astype(np.float32).astype(np.float64)
            ev, near_e = D.bsdf_eval(mat, alb.astype(np.float32).astype(np.float64), eta_i, eta_o, Nn, Vv, L)
            pdf, near_p = D.bsdf_pdf(mat, eta_i, eta_o, Nn, Vv, L)
            use = ok & ~near_e & ~near_p & np.isfinite(ev).all(1)
            assert use.sum() > 0.5 * ok.sum() > 0
            scale = np.maximum(np.abs(ev[use]).max(1), 1e-6)
            # binary32 rounding of N.H is amplified by ~1/a^2 at the GGX peak of a smooth surface
            tol = 2e-4 * max(1.0, 0.01 / max(0.001, mat.roughness) ** 2)
            assert (np.abs(t["eval"][use] - ev[use]).max(1) / scale).max() < tol, (mat.transmission, mat.roughness)
            assert (np.abs(t["pdf_again"][use] - pdf[use]) / np.maximum(pdf[use], 1e-6)).max() < tol
            checked += int(use.sum())
    assert checked > 100000


def test_bsdf_sample_draw_order_against_python(oracle):
    """BSDFSample (Disney.cuh:197-315): which lobe is sampled, in which order the random numbers are consumed and where
    the stream stands afterwards -- against the scalar Python statement in tests/disney_f64.py, which runs its own
    integer `Random`.  Samples decided within 1e-5 of a branch threshold are left out."""
    import disney_f64 as D
    rng = np.random.default_rng(31)
    n = 3000
    N = rng.normal(size=(n, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    view = rng.normal(size=(n, 3)); view /= np.linalg.norm(view, axis=1, keepdims=True)
    view[(N * view).sum(1) < 0] *= -1
    N32, V32 = N.astype(np.float32), view.astype(np.float32)
    alb = np.full((n, 3), 0.5)
    mats = [abi.Material.reference_default(), scenes.matte((0.7, 0.6, 0.5))]
    for kw in (dict(subsurface=0.6, transmission=0.0, roughness=0.3), dict(subsurface=0.3, transmission=0.7, roughness=0.1),
               dict(subsurface=0.0, transmission=1.0, roughness=0.6)):
        m = abi.Material.reference_default()
        for k, v in kw.items():
            setattr(m, k, v)
        mats.append(m)
    seen_types, compared = set(), 0
    for mat in mats:
        for eta_i, eta_o in ((1.0, 1.4), (1.4, 1.0)):
            seeds = np.arange(n) * 13 + 5
            t = oracle.bsdf_table(mat, N32, V32, alb, np.full(n, eta_i), np.full(n, eta_o), seeds)
            for i in range(n):
                r = D.PyRandom(int(seeds[i]))
                light, typ, early, margin = D.bsdf_sample(mat, eta_i, eta_o, N32[i].astype(np.float64), V32[i].astype(np.float64), r)
                if margin < 1e-5:
                    continue
                assert (int(t["rng_after"][i, 0]), int(t["rng_after"][i, 1])) == (r.s1, r.s2), (i, typ)
                if light is None:
                    assert t["pdf"][i] == 0.0
                    continue
                assert np.abs(t["light"][i] - light).max() < 5e-5, (i, typ, margin)
                if early is not None:                                    # the specular branch returns its own pdf
                    assert t["type"][i] == 2 and abs(t["pdf"][i] - early) <= 2e-5 * max(early, 1e-3)
                elif t["pdf"][i] > 0:
                    assert t["type"][i] == typ
                seen_types.add(typ); compared += 1
    assert seen_types == {0, 1, 2} and compared > 25000


def test_whole_frames_against_the_python_path_tracer(oracle):
    """End to end: tests/mini_pt.py (scalar Python, binary64, brute-force hits, its own RNG, BSDF, probe sampling and path
    loop, written from the reference's text) renders small uniform frames of the Cornell box and of a scene with
    transmissive / subsurface / clearcoat materials; the C++ oracle must give the same radiance per pixel."""
    import mini_pt
    glass = abi.Material.reference_default()
    glass.color.set((0.9, 0.8, 0.7)); glass.emission.set((0.0, 0.0, 0.0)); glass.transmission = 0.8; glass.roughness = 0.2; glass.metallic = 0.0
    wax = abi.Material.reference_default()
    wax.color.set((0.3, 0.7, 0.4)); wax.emission.set((0.1, 0.0, 0.2)); wax.transmission = 0.0; wax.subsurface = 0.5
    wax.roughness = 0.4; wax.clearcoat = 0.7; wax.clearcoatGloss = 0.4; wax.metallic = 0.1
    mixed = scenes.Model(meshes=[scenes.box_mesh((0.0, -1.2, 0.0), (3.0, 0.2, 3.0), wax), scenes.box_mesh((-0.7, 0.0, 0.2), (0.6, 0.9, 0.6), glass),
                                 scenes.box_mesh((0.9, -0.3, -0.4), (0.5, 0.7, 0.5), scenes.matte((0.8, 0.3, 0.2)))])
    for m in mixed.meshes:
        m.texture_id = -1
    cases = [(scenes.cornell_box(), scenes.CORNELL_CAMERA, scenes.ambient_probe(16, 8, 0.2), (20, 20), 2, 3),
             (mixed, dict(eye=(0.5, 1.5, 6.0), lookat=(0.0, -0.2, 0.0), up=(0.0, 1.0, 0.0), fovy=35.0), scenes.sky_probe(32, 16, seed=5), (24, 16), 2, 4)]
    for model, cam, probe_data, (w, h), spp, depth in cases:
        S = oracle.OracleScene(model)
        hp = oracle.HostProbe(probe_data)
        F = oracle.OracleFrame(w, h, hp, cam)
        from common import cfg_uniform
        oracle.render(S, F, cfg_uniform(spp, max_depth=depth))
        uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
        want, doubtful = mini_pt.render_uniform(model, hp, uvw, cam["eye"], w, h, spp, depth)
        got = F.accum[..., :3].astype(np.float64)
        err = np.abs(got - want).max(2) / np.maximum(np.abs(want).max(2), 0.05)
        sure = ~doubtful
        assert sure.mean() > 0.8, sure.mean()
        # binary32 against binary64 over a whole path: almost every pixel agrees to 1e-3; a ray that grazes an edge may
        # take another branch in one of the two precisions
        # (measured: every pixel of both frames within 1e-5, median relative difference 5e-8)
        assert (err < 1e-3).mean() > 0.97 and err[sure].max() < 1e-3, (float((err < 1e-3).mean()), float(err[sure].max()))
        assert np.median(err) < 1e-5 and want.max() > 0.1


def test_shadow_catcher_against_the_python_path_tracer(oracle):
    """MATERIAL_FLAG_SHADOW_CATCHER (deviceProgram.cu:646-651, :691-694, SampleShadow :346-385): a primary hit turns occluded
    probe samples into alpha and the backplate shows through 1 - alpha; secondary rays pass through without using up depth."""
    import mini_pt
    from common import cfg_uniform
    model, cam = scenes.cornell_box(), scenes.CORNELL_CAMERA
    model.meshes[0].material.flags = abi.MATERIAL_FLAG_SHADOW_CATCHER      # floor + ceiling + back wall
    w, h, spp, depth = 20, 20, 2, 3
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.sky_probe(16, 8, seed=6))
    F = oracle.OracleFrame(w, h, hp, cam)
    oracle.render(S, F, cfg_uniform(spp, max_depth=depth))
    uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
    want, doubtful = mini_pt.render_uniform(model, hp, uvw, cam["eye"], w, h, spp, depth)
    plain = scenes.cornell_box()
    want_plain, _ = mini_pt.render_uniform(plain, hp, uvw, cam["eye"], w, h, spp, depth)
    got = F.accum[..., :3].astype(np.float64)
    err = np.abs(got - want).max(2) / np.maximum(np.abs(want).max(2), 0.05)
    assert (err < 1e-3).mean() > 0.97 and err[~doubtful].max() < 1e-3 and np.median(err) < 1e-5, (float((err < 1e-3).mean()), float(np.median(err)))
    assert np.abs(want - want_plain).max() > 0.05                           # the flag changes the picture


def test_textured_scene_against_the_python_path_tracer(oracle):
    """The procedural atrium (textured walls and floor, texcoords interpolated at the hit, bilinear wrap lookup; all-lobes
    'application default' materials) through tests/mini_pt.py."""
    import mini_pt
    from common import cfg_uniform
    model, cam = scenes.atrium(1500), scenes.ATRIUM_CAMERA
    assert any(m.texture_id >= 0 for m in model.meshes)
    w, h, spp, depth = 24, 14, 1, 3
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.ambient_probe(16, 8, 2.5))
    F = oracle.OracleFrame(w, h, hp, cam)
    oracle.render(S, F, cfg_uniform(spp, max_depth=depth))
    uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
    want, doubtful = mini_pt.render_uniform(model, hp, uvw, cam["eye"], w, h, spp, depth)
    got = F.accum[..., :3].astype(np.float64)
    err = np.abs(got - want).max(2) / np.maximum(np.abs(want).max(2), 0.05)
    # (one pixel looks at two coincident triangles that binary32 and binary64 order differently: flagged doubtful)
    assert (err < 1e-3).mean() > 0.95 and np.median(err) < 1e-5 and err[~doubtful].max() < 1e-3, (float((err < 1e-3).mean()), float(err[~doubtful].max()))


_FUZZ_PY = range(int(os.environ.get("FOVPT_FUZZPY_FROM", "0")), int(os.environ.get("FOVPT_FUZZPY_TO", "10")))


@pytest.mark.parametrize("seed", _FUZZ_PY)
def test_random_scenes_against_the_python_path_tracer(oracle, seed):
    """Seeded random triangle soups, materials over the whole range of every Disney parameter (eta 0 included: the
    specular-derived index of refraction), random probes, cameras, sample counts and depths -- oracle against tests/mini_pt.py."""
    import mini_pt
    from common import cfg_uniform
    rng = np.random.default_rng(1000 + seed)
    meshes = []
    for _ in range(int(rng.integers(2, 5))):
        n = int(rng.integers(3, 14))
        centre = rng.uniform(-1.5, 1.5, (n, 1, 3))
        tri = (centre + rng.uniform(-1.2, 1.2, (n, 3, 3))).astype(np.float32)
        m = abi.Material.reference_default()
        m.color.set(tuple(rng.uniform(0.0, 1.0, 3))); m.emission.set(tuple(rng.uniform(0.0, 1.5, 3) * (rng.random() < 0.5)))
        m.eta = float(rng.choice([0.0, 1.0, 1.2, 1.5, 2.4])); m.metallic = float(rng.uniform(0, 1)); m.subsurface = float(rng.choice([0.0, rng.uniform(0, 1)]))
        m.specular = float(rng.uniform(0, 1)); m.roughness = float(rng.choice([0.0, 0.05, rng.uniform(0.1, 1.0)])); m.specularTint = float(rng.uniform(0, 1))
        m.clearcoat = float(rng.choice([0.0, rng.uniform(0, 1)])); m.clearcoatGloss = float(rng.uniform(0, 1))
        m.transmission = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        m.flags = abi.MATERIAL_FLAG_SHADOW_CATCHER if rng.random() < 0.2 else 0
        meshes.append(scenes.TriangleMesh(vertex=tri.reshape(-1, 3), index=np.arange(3 * n, dtype=np.uint32).reshape(n, 3), material=m))
    model = scenes.Model(meshes=meshes)
    cam = dict(eye=tuple(rng.uniform(-1, 1, 3) + np.float64([0, 0, 7])), lookat=tuple(rng.uniform(-0.5, 0.5, 3)), up=(0.0, 1.0, 0.0), fovy=float(rng.uniform(25, 60)))
    w, h, spp, depth = int(rng.integers(10, 18)), int(rng.integers(8, 14)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.sky_probe(16, 8, seed=seed) if rng.random() < 0.7 else scenes.ambient_probe(8, 4, float(rng.uniform(0.1, 3.0))))
    F = oracle.OracleFrame(w, h, hp, cam)
    oracle.render(S, F, cfg_uniform(spp, max_depth=depth))
    uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
    want, doubtful = mini_pt.render_uniform(model, hp, uvw, cam["eye"], w, h, spp, depth)
    got = F.accum[..., :3].astype(np.float64)
    fin = np.isfinite(want).all(2) & np.isfinite(got).all(2)
    assert np.array_equal(np.isfinite(want).all(2) | doubtful, np.isfinite(got).all(2) | doubtful)
    err = np.abs(got - want).max(2) / np.maximum(np.abs(want).max(2), 0.05)
    sure = fin & ~doubtful
    assert sure.mean() > 0.5
    assert (err[sure] < 2e-3).mean() > 0.97 and np.median(err[sure]) < 1e-4, (seed, float((err[sure] < 2e-3).mean()), float(np.median(err[sure])), float(err[sure].max()))


_FUZZ_PYF = range(int(os.environ.get("FOVPT_FUZZPYF_FROM", "0")), int(os.environ.get("FOVPT_FUZZPYF_TO", "4")))


@pytest.mark.parametrize("seed", _FUZZ_PYF)
def test_random_foveated_frames_against_the_python_path_tracer(oracle, seed):
    """Random frame sizes, gaze points (also off the frame: wrapping offsets), radii, per-pass sample counts, subframe
    indices and depths: the three launches of render() in the oracle and in tests/mini_pt.py."""
    import mini_pt
    rng = np.random.default_rng(500 + seed)
    model, cam = scenes.cornell_box(), scenes.CORNELL_CAMERA
    w, h = 4 * int(rng.integers(6, 13)), 4 * int(rng.integers(5, 9))
    gaze = (int(rng.integers(-3, w + 3)), int(rng.integers(-3, h + 3)))
    ri = int(rng.integers(2, 7)); ro = ri + int(rng.integers(2, 9))
    spp = tuple(int(x) for x in rng.integers(1, 4, 3))
    subframe, depth = int(rng.integers(0, 6)), int(rng.integers(1, 4))
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.sky_probe(16, 8, seed=seed))
    F = oracle.OracleFrame(w, h, hp, cam, gaze=gaze, subframe_index=subframe)
    oracle.render(S, F, cfg_foveated(ri, ro, spp, max_depth=depth))
    uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
    want, doubtful = mini_pt.render_foveated(model, hp, uvw, cam["eye"], w, h, gaze, ri, ro, spp, subframe, depth)
    written = ~np.isnan(want[..., 0])
    assert np.array_equal(written, F.accum[..., 3] == 1.0), (seed, gaze, ri, ro)
    sure = written & ~doubtful
    got = F.accum[..., :3].astype(np.float64)[sure]
    err = np.abs(got - want[sure]).max(1) / np.maximum(np.abs(want[sure]).max(1), 0.05)
    if written.sum() >= 50:                                              # (a gaze point off the frame can leave it almost empty)
        assert sure.sum() > 0.5 * written.sum()
    if sure.any():
        assert (err < 1e-3).mean() > 0.97 and np.median(err) < 1e-5, (seed, float((err < 1e-3).mean()), float(np.median(err)))


def test_a_foveated_frame_against_the_python_path_tracer(oracle):
    """The three launches of render() -- periphery blocks, middle ring, fovea, each with its sample count, seeds from the
    launch index, subframe 0 for the inner two -- through tests/mini_pt.py: radiance AND layout of a whole foveated frame."""
    import mini_pt
    model, cam = scenes.cornell_box(), scenes.CORNELL_CAMERA
    w, h, gaze, ri, ro = 48, 32, (27, 14), 5, 11
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.sky_probe(16, 8, seed=2))
    for subframe in (0, 3):
        F = oracle.OracleFrame(w, h, hp, cam, gaze=gaze, subframe_index=subframe)
        oracle.render(S, F, cfg_foveated(ri, ro, (1, 2, 3), max_depth=3))
        uvw = oracle.camera_uvw(cam["eye"], cam["lookat"], cam["up"], cam["fovy"], w / h)
        want, doubtful = mini_pt.render_foveated(model, hp, uvw, cam["eye"], w, h, gaze, ri, ro, (1, 2, 3), subframe, 3)
        written = ~np.isnan(want[..., 0])
        assert 0.5 < written.mean() < 1.0 and np.array_equal(written, F.accum[..., 3] == 1.0)     # same pixels written, same holes left
        got = F.accum[..., :3].astype(np.float64)[written]
        err = np.abs(got - want[written]).max(1) / np.maximum(np.abs(want[written]).max(1), 0.05)
        assert (err < 1e-3).mean() > 0.97 and err[~doubtful[written]].max() < 1e-3 and np.median(err) < 1e-5, (float((err < 1e-3).mean()), float(np.median(err)))


def test_accumulate_mode_formula_against_numpy(oracle):
    """PT_sv4_vmv2/deviceProgram.cu:545-553: with subframe_index > 0 and redraw == 0 a launch writes
    lerp(previous, clamp(new, 0, 10), 1 / (subframe_index + 1)); otherwise the new value.  `new` is what the same
    launch writes with accumulation off."""
    model, cam = scenes.cornell_box(), scenes.CORNELL_CAMERA
    w, h = 40, 24
    S = oracle.OracleScene(model)
    hp = oracle.HostProbe(scenes.sky_probe(16, 8, seed=4))
    prev = np.random.default_rng(8).uniform(0.0, 12.0, (h, w, 4)).astype(np.float32)

    def run(accumulate, subframe, redraw):
        F = oracle.OracleFrame(w, h, hp, cam, subframe_index=subframe)
        f = F.lp.frame
        f.factor.x, f.factor.y, f.factor.z, f.fillSize = 2, 2, 1, 2
        f.r_inner, f.r_outer, f.redraw = 0.0, 1e9, redraw
        F.lp.samples_per_launch = 2
        F.accum[...] = prev
        oracle.launch(S, F, w // 2, h // 2, max_depth=3, accumulate=accumulate)
        return F.accum.copy()

    for subframe, redraw in ((5, 0), (1, 0), (0, 0), (5, 1)):
        new = run(0, subframe, redraw)[..., :3]
        got = run(1, subframe, redraw)[..., :3]
        if subframe > 0 and not redraw:
            a = np.float32(1.0) / np.float32(subframe + 1)
            c = np.clip(new, np.float32(0.0), np.float32(10.0))
            want = prev[..., :3] + (c - prev[..., :3]) * a                   # lerp(a, b, t) = a + t * (b - a), binary32
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max()))
            assert not np.array_equal(got, new)
        else:
            assert np.array_equal(got, new)


def test_tone_mapping_against_numpy(oracle):
    """accum * 16 -> Reinhard (white 1) -> clamp -> sRGB OETF -> 8 bits (deviceProgram.cu:126-131, :586-597,
    cuda/helpers.h:35-62) in binary64 numpy; binary32 may land one code away right at a quantisation step."""
    rng = np.random.default_rng(23)
    rgb = np.concatenate([rng.uniform(0.0, 0.2, (20000, 3)), rng.uniform(0.0, 5.0, (5000, 3)), 10.0 ** rng.uniform(-6, 1, (5000, 3)),
                          np.float64([[0, 0, 0], [1e-5, 2e-5, 3e-5], [1e3, 1e3, 1e3], [-1, -2, -3]])]).astype(np.float32)
    got = oracle.make_color(rgb)
    c = rgb.astype(np.float64) * 16.0
    lum = 0.2126 * c[:, 0] + 0.7152 * c[:, 1] + 0.0722 * c[:, 2]
    t = np.clip(c / (1.0 + lum)[:, None], 0.0, 1.0)
    srgb = np.where(t < 0.0031308, 12.92 * t, 1.055 * t ** (1.0 / 2.4) - 0.055)
    q = np.minimum((np.clip(srgb, 0.0, 1.0) * 256.0).astype(np.int64), 255)
    want = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | (255 << 24)
    ch = lambda v, k: ((v >> (8 * k)) & 255).astype(np.int64)
    diff = np.stack([np.abs(ch(got.astype(np.int64), k) - ch(want, k)) for k in range(3)], 1)
    assert (got >> 24 == 255).all() and diff.max() <= 1 and (diff == 0).all(1).mean() > 0.995


def test_make_color_known_values(oracle):
    c = oracle.make_color(np.float32([[0, 0, 0], [1e6, 1e6, 1e6], [-0.01, -0.01, -0.01]]))
    assert c[0] == 0xFF000000 and c[2] == 0xFF000000
    assert (c[1] & 0xFF) >= 254 and (c[1] >> 24) == 255


def test_tex2d_contract(oracle):
    tex = np.array([[0xFF0000FF, 0xFF00FF00], [0xFFFF0000, 0xFFFFFFFF]], np.uint32)     # 2x2: R G / B W
    out = oracle.tex2d(tex, np.float32([[0.25, 0.25], [0.75, 0.25], [0.5, 0.5], [1.25, 0.25]]))
    assert np.allclose(out[0], [1, 0, 0, 1]) and np.allclose(out[1], [0, 1, 0, 1])
    assert np.allclose(out[2], [0.5, 0.5, 0.5, 1.0])            # centre: equal blend of the four texels
    assert np.allclose(out[3], out[0])                          # wrap addressing


def test_opt_in_extensions_behave(oracle):
    """The two opt-in modes of fovpt_config.options in the oracle (the GPU is held to it in tests/test_gpu_lifecycle.py):
    Russian roulette is unbiased (the frame mean stays, fewer rays are traced), sky radiance for escaped secondary rays
    only ever adds light, and options = 0 is the reference's frame."""
    size = (48, 32)
    model, probe = scenes.cornell_box(), scenes.sky_probe(64, 32, seed=4)
    S = oracle.OracleScene(model)
    means, rays = {}, {}
    for name, opt in (("plain", 0), ("rr", abi.OPT_RUSSIAN_ROULETTE), ("sky", abi.OPT_SKY_MISS)):
        F = oracle.OracleFrame(size[0], size[1], oracle.HostProbe(probe), scenes.CORNELL_CAMERA)
        cfg = abi.Config.reference_default()
        cfg.uniform, cfg.spp_uniform, cfg.max_depth, cfg.options = 1, 96, 8, opt
        cnt = oracle.render(S, F, cfg)
        means[name], rays[name] = F.accum[..., :3].astype(np.float64).mean(), cnt[0]
        if name == "plain":
            base = F.accum.copy()
        elif name == "sky":
            assert (F.accum[..., :3] >= base[..., :3] - 1e-6).all()
    assert rays["rr"] < 0.9 * rays["plain"]
    assert abs(means["rr"] - means["plain"]) < 0.03 * means["plain"]
    assert means["sky"] > means["plain"] * 1.001


def test_binary64_detours_of_the_reference_equal_their_binary32_forms(oracle):
    """maths.h writes 1.0 / sqrtf(x) and Disney.cuh 0.5 + y with binary64 literals: a binary64 operation on binary32 values,
    rounded back.  The compiler emits the binary32 operation for them in the HIP kernels (wavefront.hip: rounding twice is
    innocuous for + - * / when the wide format has >= 2 * 24 + 2 bits).  Here on the CPU: every binade, random significands, the range ends."""
    rng = np.random.default_rng(7)
    e = rng.integers(-126, 128, 400000)
    m = rng.integers(0, 1 << 23, 400000).astype(np.float64) / (1 << 23) + 1.0
    x = np.concatenate([np.ldexp(m, e).astype(np.float32), np.float32([np.finfo(np.float32).tiny, np.finfo(np.float32).max, 1.0, 2.0, 3.0])])
    s = np.sqrt(x)                                                           # binary32, correctly rounded (sqrtf)
    with np.errstate(over="ignore", divide="ignore"):
        wide = (1.0 / s.astype(np.float64)).astype(np.float32)
        narrow = np.float32(1.0) / s
    assert np.array_equal(wide.view(np.uint32), narrow.view(np.uint32))
    e = rng.integers(-149, 128, 400000)
    y = np.ldexp(m, e).astype(np.float32)
    y = np.concatenate([y, -y, np.float32([0.0, -0.5, 0.5, 2.0 ** -25, -(2.0 ** -25), 0.5 - 2.0 ** -25])])
    with np.errstate(over="ignore"):
        wide = (0.5 + y.astype(np.float64)).astype(np.float32)
        narrow = np.float32(0.5) + y
    assert np.array_equal(wide.view(np.uint32), narrow.view(np.uint32))
    assert np.array_equal(oracle.math_op(11, y).view(np.uint32), narrow.view(np.uint32))          # FOVPT_OP_HALFPLUS in the oracle
    assert np.array_equal(oracle.math_op(9, x).view(np.uint32), (np.float32(1.0) / np.sqrt(x)).view(np.uint32))   # FOVPT_OP_RSQRTD
