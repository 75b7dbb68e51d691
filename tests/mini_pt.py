"""An end-to-end SECOND statement of the path -- scalar Python, binary64, brute-force intersection -- written from
PT_sv5_/deviceProgram.cu (raygen :392-617, closest hit :619-732, miss :253-282, SampleLights :303-344), Probe.cuh and,
through tests/disney_f64.py, Disney.cuh.  It shares no code with oracle/ and is far too slow for anything but the
tiny frames of tests/test_oracle_cpu.py::test_whole_frames_against_the_python_path_tracer: a checker of the checker."""
import numpy as np

import disney_f64 as D

TMIN, TMAX = 0.01, 1e16
M32 = 0xFFFFFFFF


def tea4(v0, v1):                                                        # cuda/random.h:34-47
    s0 = 0
    for _ in range(4):
        s0 = (s0 + 0x9E3779B9) & M32
        v0 = (v0 + ((((v1 << 4) & M32) + 0xA341316C) & M32 ^ ((v1 + s0) & M32) ^ (((v1 >> 5) + 0xC8013EA4) & M32))) & M32
        v1 = (v1 + ((((v0 << 4) & M32) + 0xAD90777D) & M32 ^ ((v0 + s0) & M32) ^ (((v0 >> 5) + 0x7E95761E) & M32))) & M32
    return v0


def tex2d(tex, u, v):
    """The fp32 contract standing in for tex2D<float4> (include/fovpt.h): bilinear, wrap, texel centres at +0.5, RGBA8 / 255."""
    hh, ww = tex.shape
    x, y = u * ww - 0.5, v * hh - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    fx, fy = x - x0, y - y0
    out = np.zeros(3)
    for yy, wy in ((y0, 1.0 - fy), (y0 + 1, fy)):
        for xx, wx in ((x0, 1.0 - fx), (x0 + 1, fx)):
            p = int(tex[yy % hh, xx % ww])
            out += wx * wy * np.float64([p & 255, (p >> 8) & 255, (p >> 16) & 255]) / 255.0
    return out


class Scene:
    def __init__(self, model):
        self.v0, self.e1, self.e2, self.mat, self.tc, self.tex = [], [], [], [], [], []
        for mesh in model.meshes:
            v = mesh.vertex.astype(np.float64)
            textured = mesh.texture_id >= 0 and mesh.texcoord is not None
            for a, b, c in mesh.index:
                self.v0.append(v[a]); self.e1.append(v[b] - v[a]); self.e2.append(v[c] - v[a]); self.mat.append(mesh.material)
                self.tc.append(mesh.texcoord[[a, b, c]].astype(np.float64) if textured else None)
                self.tex.append(model.textures[mesh.texture_id] if textured else None)
        self.v0, self.e1, self.e2 = np.array(self.v0), np.array(self.e1), np.array(self.e2)

    def albedo(self, k, o, d):
        """material colour, or the texture at the barycentric texcoord of the hit (deviceProgram.cu:655-665)"""
        mat = self.mat[k]
        if self.tex[k] is None:
            return np.float64([mat.color.x, mat.color.y, mat.color.z])
        p = np.cross(d, self.e2[k]); inv = 1.0 / float(np.dot(self.e1[k], p)); s = o - self.v0[k]
        u = float(np.dot(s, p)) * inv
        v = float(np.dot(np.cross(s, self.e1[k]), d)) * inv
        t0, t1, t2 = self.tc[k]
        tc = (1.0 - u - v) * t0 + u * t1 + v * t2
        return tex2d(self.tex[k], tc[0], tc[1])

    def candidates(self, o, d):
        """Moeller-Trumbore against every triangle -> (t, det, valid) arrays; a candidate has tmin < t < tmax."""
        p = np.cross(d, self.e2)
        det = (self.e1 * p).sum(1)
        with np.errstate(all="ignore"):
            inv = 1.0 / det
            s = o - self.v0
            u = (s * p).sum(1) * inv
            q = np.cross(s, self.e1)
            v = (q * d).sum(1) * inv
            t = (self.e2 * q).sum(1) * inv
            ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > TMIN) & (t < TMAX)
        return t, det, ok

    def closest(self, o, d):
        t, _, ok = self.candidates(o, d)
        if not ok.any():
            return None
        tt = np.where(ok, t, np.inf)
        k = int(np.argmin(tt))                                            # argmin takes the lowest index among ties
        tt2 = tt.copy(); tt2[k] = np.inf
        self.near_tie = bool(tt2.min() - tt[k] <= 1e-6 * tt[k])          # coincident surfaces: binary32 may order them differently
        return k, float(tt[k])

    def occluded(self, o, d):                                            # any front-facing candidate (back faces culled, :241)
        _, det, ok = self.candidates(o, d)
        return bool((ok & (det > 0)).any())


class ProbeF64:
    def __init__(self, host_probe):
        self.data = host_probe.data.astype(np.float64)
        self.h, self.w = self.data.shape[:2]
        self.pdfx, self.cdfx, self.pdfy, self.cdfy = host_probe.pdfx, host_probe.cdfx, host_probe.pdfy, host_probe.cdfy

    def eval_dir(self, d):                                               # ProbeEval(ProbeDirToUV(d)), Probe.cuh:38-46,61-67
        theta = np.arccos(min(1.0, max(-1.0, d[1])))
        phi = 0.0 if (d[0] == 0.0 and d[2] == 0.0) else np.arctan2(d[2], d[0])
        u, v = (np.pi + phi) / (2.0 * np.pi), theta / np.pi
        px = min(max(int(u * self.w), 0), self.w - 1); py = min(max(int(v * self.h), 0), self.h - 1)
        fu, fv = u * self.w, v * self.h
        self.near_texel_edge = min(abs(fu - round(fu)), abs(fv - round(fv))) < 1e-5      # binary32 may land in the neighbouring texel
        return self.data[py, px, :3]

    def sample(self, rnd):                                               # ProbeSample, Probe.cuh:138-169
        r1, r2 = rnd.randf(), rnd.randf()
        row = int(np.searchsorted(self.cdfy, np.float32(r1), side="left"))
        row_c = min(row, self.h - 1)
        col = int(np.searchsorted(self.cdfx[row_c], np.float32(r2), side="left"))
        col_c = min(col, self.w - 1)
        color = self.data[row_c, col_c, :3]
        pdf = float(self.pdfx[row_c, col_c]) * float(self.pdfy[row_c])
        theta, phi = row / float(self.h) * np.pi, col / float(self.w) * 2.0 * np.pi
        st = np.sin(theta)
        pdf = 0.0 if np.float32(st) == 0.0 else pdf * self.w * self.h / (2.0 * np.pi * np.pi * st)
        edge = row >= self.h or col >= self.w
        return np.float64([-st * np.cos(phi), np.cos(theta), -st * np.sin(phi)]), color, pdf, edge


def render_uniform(model, probe, cam_uvw, eye, w, h, spp, max_depth):
    """FOV_OFF frame (SimplePathtracer.cpp:85-131): factor 1, offset 0, every launch index alive, subframe 0.
    -> accum (h, w, 3) binary64, doubtful (h, w) bool: a branch was decided within rounding of its threshold."""
    accum = np.zeros((h, w, 3)); doubtful = np.zeros((h, w), bool)
    launch(model, probe, cam_uvw, eye, w, h, accum, doubtful, (w, h), 1, 1, (0, 0), (0, 0), 0.0, 1e9, spp, 0, max_depth)
    return accum, doubtful


def render_foveated(model, probe, cam_uvw, eye, w, h, gaze, r_inner, r_outer, spp_pmf, subframe, max_depth):
    """FOV_ON frame (SimplePathtracer.cpp:133-209): periphery, middle ring, fovea, later launches overwrite earlier ones;
    only the periphery launch uses the running subframe index.  Pixels no launch writes stay NaN."""
    accum = np.full((h, w, 3), np.nan); doubtful = np.zeros((h, w), bool)
    cx, cy = gaze
    launch(model, probe, cam_uvw, eye, w, h, accum, doubtful, (w // 4, h // 4), 4, 4, (0, 0), gaze, float(r_outer), 1e9, spp_pmf[0], subframe, max_depth)
    launch(model, probe, cam_uvw, eye, w, h, accum, doubtful, (r_outer + 2, r_outer + 2), 2, 2, (cx - (r_outer + 2), cy - (r_outer + 2)), gaze,
           float(r_inner), float(r_outer + 2), spp_pmf[1], 0, max_depth)
    launch(model, probe, cam_uvw, eye, w, h, accum, doubtful, (2 * (r_inner + 1), 2 * (r_inner + 1)), 1, 1, (cx - (r_inner + 1), cy - (r_inner + 1)), gaze,
           0.0, float(r_inner + 1), spp_pmf[2], 0, max_depth)
    return accum, doubtful


def launch(model, probe, cam_uvw, eye, w, h, accum, doubtful, grid, factor, fill, offset, gaze, r_in, r_out, spp, subframe, max_depth):
    """One optixLaunch of __raygen__renderFrame over `grid` launch indices, writing its block fills into accum."""
    sc, pr = Scene(model), ProbeF64(probe)
    U, V, W = (np.float64(x) for x in cam_uvw)
    eye = np.float64(eye)
    for ly in range(grid[1]):
        for lx in range(grid[0]):
            ix, iy = (lx * factor + offset[0]) & M32, (ly * factor + offset[1]) & M32        # uint arithmetic (:433)
            rng_ = np.sqrt(np.float32(np.float32(ix) - np.float32(gaze[0] & M32)) ** 2 + np.float32(np.float32(iy) - np.float32(gaze[1] & M32)) ** 2)
            if rng_ < np.float32(r_in) or rng_ > np.float32(r_out):
                continue
            seed = tea4(ly * w + lx, subframe)                                                # launch index, FULL frame width (:411)
            result, alpha_sum, backplate = np.zeros(3), np.zeros(3), np.zeros(3)
            doubt = False
            for _ in range(spp):
                rnd = D.PyRandom(seed)                                    # seeded before the jitter draws (:466)
                seed = (1664525 * seed + 1013904223) & M32; jx = (seed & 0xFFFFFF) / float(1 << 24)
                seed = (1664525 * seed + 1013904223) & M32; jy = (seed & 0xFFFFFF) / float(1 << 24)
                dx, dy = 2.0 * (float(np.float32(ix)) + jx) / w - 1.0, 2.0 * (float(np.float32(iy)) + jy) / h - 1.0
                d = dx * U + dy * V + W
                d /= np.linalg.norm(d)
                o = eye.copy()
                backplate = pr.eval_dir(d)
                doubt |= pr.near_texel_edge
                thr, ray_eta, done, secondary, depth = np.ones(3), 1.0, False, False, 0
                direct, indirect, alpha = np.zeros(3), np.zeros(3), np.zeros(3)
                while True:
                    radiance = np.zeros(3)
                    hit = sc.closest(o, d)
                    if hit is None:
                        done = True
                    else:
                        k, t = hit
                        mat = sc.mat[k]
                        doubt |= sc.near_tie
                        catcher = (mat.flags & 1) != 0                    # MATERIAL_FLAG_SHADOW_CATCHER (Material.h:9)
                    if hit is not None and catcher and secondary:
                        o = o + hit[1] * d                                # pass through, the bounce does not count (:646-651)
                        depth -= 1
                    elif hit is not None:
                        n0 = np.cross(sc.e1[k], sc.e2[k]); n0 /= np.linalg.norm(n0)
                        facing = float(np.dot(-d, n0))
                        doubt |= abs(facing) < 1e-6
                        N = n0 if facing >= 0.0 else -n0
                        P = o + t * d
                        albedo = sc.albedo(k, o, d)
                        out_eta = (mat.eta if mat.eta != 0.0 else 2.0 / (1.0 - np.sqrt(0.08 * mat.specular)) - 1.0) if ray_eta == 1.0 else 1.0
                        wo = -d
                        # SampleLights
                        wi, sky, sky_pdf, edge = pr.sample(rnd)
                        doubt |= edge
                        val = np.zeros(3)
                        if sc.occluded(P, wi) == catcher:                 # SampleLights counts visible, SampleShadow occluded samples
                            pdf, near_p = D.bsdf_pdf(mat, ray_eta, out_eta, N[None], wo[None], wi[None])
                            f, near_e = D.bsdf_eval(mat, albedo[None], ray_eta, out_eta, N[None], wo[None], wi[None])
                            doubt |= bool(near_p[0] or near_e[0])
                            if pdf[0] > 0.0:
                                wgt = 0.5 * sky_pdf / (0.5 * pdf[0] + 0.5 * sky_pdf)
                                if wgt > 0.0:
                                    val = wgt * sky * f[0] * abs(float(np.dot(wi, N))) / sky_pdf
                        if not catcher:
                            radiance = radiance + thr * val
                            alpha = np.ones(3)
                        else:
                            alpha = alpha + thr * val                     # (:691-694)
                        if not secondary:
                            radiance = radiance + np.float64([mat.emission.x, mat.emission.y, mat.emission.z])
                        light, typ, early, margin = D.bsdf_sample(mat, ray_eta, out_eta, N, wo, rnd)
                        doubt |= margin < 1e-5
                        if light is None:
                            pdf_s = 0.0
                        elif early is not None:
                            pdf_s = early
                        else:
                            pp, near_p = D.bsdf_pdf(mat, ray_eta, out_eta, N[None], wo[None], light[None])
                            pdf_s = float(pp[0]); doubt |= bool(near_p[0])
                        if pdf_s <= 0.0:
                            done = True
                        else:
                            f, near_e = D.bsdf_eval(mat, albedo[None], ray_eta, out_eta, N[None], wo[None], light[None])
                            doubt |= bool(near_e[0])
                            if float(np.dot(light, N)) <= 0.0:
                                ray_eta = out_eta
                            thr = thr * f[0] * abs(float(np.dot(N, light))) / pdf_s
                            o, d, secondary = P, light, True
                    if done or depth >= max_depth:                       # before the radiance is added (:515)
                        break
                    if depth == 0:
                        direct = direct + radiance
                    else:
                        indirect = indirect + radiance
                    depth += 1
                result = result + direct + indirect
                alpha_sum = alpha_sum + alpha
            a = alpha_sum / spp
            colour = (backplate * spp * (1.0 - a) + result) / spp             # backplate of the last sample (:558-560)
            for i in range(fill):
                for j in range(fill):
                    px = min((lx * factor + i + offset[0]) & M32, w - 1); py = min((ly * factor + j + offset[1]) & M32, h - 1)   # clamp (:554)
                    accum[py, px] = colour
                    doubtful[py, px] = doubt
