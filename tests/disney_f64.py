"""A second, independently written statement of the reference's Disney BSDF evaluation -- vectorised numpy in
binary64, from the formulae of PT_sv5_/Disney.cuh (BSDFPdf :152-193, BSDFEval :318-427, helpers :49-103) -- used
only to cross-check the C++ oracle (tests/test_oracle_cpu.py).  It is a checker of the checker: neither the
product nor the oracle uses it.  `near_branch` marks samples too close to one of the formulae's branch conditions
for a binary32 / binary64 comparison to be meaningful."""
import numpy as np

PI = np.float64(np.float32(3.141592653589793))          # kPi is a float literal (maths.h:29)
INV_PI = np.float64(np.float32(1.0) / np.float32(3.141592653589793))
INV_2PI = np.float64(np.float32(1.0) / (np.float32(2.0) * np.float32(3.141592653589793)))


def _dot(a, b):
    return (a * b).sum(-1)


def _lerp(a, b, t):
    return a + (b - a) * t


def schlick(u):                                          # :49-54
    m = np.clip(1.0 - u, 0.0, 1.0)
    return m ** 5


def gtr1(ndh, a):                                        # :56-62
    a2 = a * a
    with np.errstate(all="ignore"):
        t = 1.0 + (a2 - 1.0) * ndh * ndh
        v = (a2 - 1.0) / (PI * np.log(a2) * t)
    return np.where(a >= 1.0, INV_PI, v)


def gtr2(ndh, a):                                        # :64-69
    a2 = a * a
    t = 1.0 + (a2 - 1.0) * ndh * ndh
    return a2 / (PI * t * t)


def smith_ggx(ndv, alpha):                               # :71-76
    a, b = alpha * alpha, ndv * ndv
    return 1.0 / (ndv + np.sqrt(a + b - a * b))


def fresnel(vdn, eta_i, eta_t):                          # :79-97 ; returns (F, sin2T)
    s2 = (eta_i / eta_t) ** 2 * (1.0 - vdn * vdn)
    ldn = np.sqrt(np.maximum(0.0, 1.0 - s2))
    eta = eta_t / eta_i
    with np.errstate(all="ignore"):
        r1 = (vdn - eta * ldn) / (vdn + eta * ldn)
        r2 = (ldn - eta * vdn) / (ldn + eta * vdn)
    return np.where(s2 > 1.0, 1.0, 0.5 * (r1 * r1 + r2 * r2)), s2


def bsdf_pdf(mat, eta_i, eta_o, N, V, L):
    ndl = _dot(L, N)
    below = _lerp(INV_2PI * mat.subsurface * 0.5, 0.0, mat.transmission)
    F, s2 = fresnel(_dot(N, V), eta_i, eta_o)
    a = max(0.001, mat.roughness)
    h = L + V
    m = _dot(h, h)
    with np.errstate(all="ignore"):
        h = np.where((m > 0)[:, None], h / np.sqrt(m)[:, None], 0.0)
    cth = np.abs(_dot(h, N))
    pdf_half = gtr2(cth, a) * cth
    pdf_spec = 0.25 * pdf_half / np.maximum(1e-6, _dot(L, h))
    pdf_diff = np.abs(ndl) * INV_PI * (1.0 - mat.subsurface)
    above = _lerp(_lerp(pdf_diff, pdf_spec, 0.5), pdf_spec * F, mat.transmission)
    near = (np.abs(ndl) < 1e-3) | (np.abs(s2 - 1.0) < 1e-3) | (_dot(L, h) < 1e-4)
    return np.where(ndl <= 0.0, below, above), near


def bsdf_eval(mat, albedo, eta_i, eta_o, N, V, L):
    ndl, ndv = _dot(N, L), _dot(N, V)
    h = L + V
    with np.errstate(all="ignore"):
        H = h / np.linalg.norm(h, axis=1, keepdims=True)
    ndh, ldh = _dot(N, H), _dot(L, H)
    lum = 0.3 * albedo[:, 0] + 0.6 * albedo[:, 1] + 0.1 * albedo[:, 2]
    with np.errstate(all="ignore"):
        tint = np.where((lum > 0)[:, None], albedo / lum[:, None], 1.0)
    cspec0 = _lerp(mat.specular * 0.08 * _lerp(1.0, tint, mat.specularTint), albedo, mat.metallic)
    a = max(0.001, mat.roughness)
    ds = gtr2(ndh, a)
    with np.errstate(all="ignore"):
        gs = smith_ggx(ndv, a) * smith_ggx(ndl, a)
    bsdf = np.zeros_like(albedo)
    near = np.abs(ndl) < 1e-3
    if mat.transmission > 0.0:
        F, s2 = fresnel(ndv, eta_i, eta_o)
        with np.errstate(all="ignore"):
            below = (mat.transmission * (1.0 - F) / np.abs(ndl) * (1.0 - mat.metallic))[:, None] * np.ones(3)
        FH, s2h = fresnel(ldh, eta_i, eta_o)
        above = (gs * ds)[:, None] * _lerp(cspec0, 1.0, FH[:, None])
        bsdf = np.where((ndl <= 0)[:, None], below, above)
        near = near | np.where(ndl <= 0, np.abs(s2 - 1.0) < 1e-3, np.abs(s2h - 1.0) < 1e-3)
    brdf = np.zeros_like(albedo)
    if mat.transmission < 1.0:
        below = np.zeros_like(albedo)
        if mat.subsurface > 0.0:
            s = np.sqrt(np.float64([mat.color.x, mat.color.y, mat.color.z]))
            fd = (1.0 - 0.5 * schlick(np.abs(ndl))) * (1.0 - 0.5 * schlick(ndv))
            below = INV_PI * s[None, :] * mat.subsurface * fd[:, None] * (1.0 - mat.metallic)
        FH = schlick(ldh)
        fs = _lerp(cspec0, 1.0, FH[:, None])
        FL, FV = schlick(ndl), schlick(ndv)
        fd90 = 0.5 + 2.0 * ldh * ldh * mat.roughness
        fd = _lerp(1.0, fd90, FL) * _lerp(1.0, fd90, FV)
        dr = gtr1(ndh, _lerp(0.1, 0.001, mat.clearcoatGloss))
        fc = _lerp(0.04, 1.0, FH)
        with np.errstate(all="ignore"):
            gr = smith_ggx(ndl, 0.25) * smith_ggx(ndv, 0.25)
        with np.errstate(all="ignore"):
            above = (INV_PI * fd)[:, None] * albedo * (1.0 - mat.metallic) * (1.0 - mat.subsurface) + (gs * ds)[:, None] * fs \
                + (mat.clearcoat * gr * fc * dr)[:, None]
        brdf = np.where((ndl <= 0)[:, None], below, above)
    return _lerp(brdf, bsdf, mat.transmission), near


# ---- BSDFSample (Disney.cuh:197-315) as scalar Python: the order of the random draws and the branch structure -------
M32 = 0xFFFFFFFF
TWO_PI = 2.0 * float(PI)


class PyRandom:
    """class Random (maths.h:170-227) on Python integers."""

    def __init__(self, seed):
        self.s1 = (315645664 + seed) & M32
        self.s2 = self.s1 ^ 0x13AB45FE

    def rand(self):
        rot5 = ((self.s1 << 5) | (self.s1 >> 27)) & M32
        self.s1 = ((self.s2 ^ rot5) ^ ((self.s1 * self.s2) & M32)) & M32
        rot12 = ((self.s2 << 12) | (self.s2 >> 20)) & M32
        self.s2 = (self.s1 ^ rot12) & M32
        return self.s1

    def randf(self):
        f = float(np.float32(self.rand()) * (np.float32(1.0) / np.float32(0xFFFFFFFF)))
        return min(max(f, 0.0), float(np.float32(0.999999)))


def basis_from_vector(w):                                # maths.h:94-108
    if abs(w[0]) > abs(w[1]):
        il = 1.0 / np.sqrt(w[0] * w[0] + w[2] * w[2])
        u = np.float64([-w[2] * il, 0.0, w[0] * il])
    else:
        il = 1.0 / np.sqrt(w[1] * w[1] + w[2] * w[2])
        u = np.float64([0.0, w[2] * il, -w[1] * il])
    return u, np.cross(w, u)


def _fr1(vdn, eta_i, eta_t):
    F, _ = fresnel(np.float64([vdn]), eta_i, eta_t)
    return float(F[0])


def bsdf_sample(mat, eta_i, eta_o, N, view, rnd):
    """-> (light or None, type 0 reflected / 1 transmitted / 2 specular, early_pdf or None, margin): `margin` is the
    smallest distance of a random draw or a sign test from the threshold that decided a branch."""
    U, V = basis_from_vector(N)
    margin = 1.0

    def draw_below(limit):
        nonlocal margin
        x = rnd.randf()
        margin = min(margin, abs(x - limit))
        return x < limit

    def ggx_reflect(r1, r2):
        nonlocal margin
        a = max(0.001, mat.roughness)
        phi = r1 * TWO_PI
        ct = np.sqrt((1.0 - r2) / (1.0 + (a * a - 1.0) * r2))
        st = np.sqrt(max(0.0, 1.0 - ct * ct))
        half = U * (st * np.cos(phi)) + V * (st * np.sin(phi)) + N * ct
        hv = float(np.dot(half, view))
        margin = min(margin, abs(hv))
        if hv <= 0.0:
            half = -half
        return 2.0 * float(np.dot(view, half)) * half - view

    if draw_below(mat.transmission):
        F = _fr1(float(np.dot(N, view)), eta_i, eta_o)
        if draw_below(F):
            r1, r2 = rnd.randf(), rnd.randf()
            return ggx_reflect(r1, r2), 0, None, margin
        eta = eta_i / eta_o
        ci = float(np.dot(N, view))
        s2t = eta * eta * max(0.0, 1.0 - ci * ci)
        margin = min(margin, abs(s2t - 1.0))
        if s2t >= 1.0:
            return None, 2, 0.0, margin
        ct = np.sqrt(1.0 - s2t)
        return eta * -view + (eta * ci - ct) * N, 2, (1.0 - F) * mat.transmission, margin
    r1, r2 = rnd.randf(), rnd.randf()
    if draw_below(0.5):
        if draw_below(mat.subsurface):
            z = rnd.randf()
            w = np.sqrt(1.0 - z * z)
            phi = TWO_PI * rnd.randf()
            return U * (np.cos(phi) * w) + V * (np.sin(phi) * w) - N * z, 1, None, margin
        r, th = np.sqrt(r1), TWO_PI * r2
        x, y = r * np.cos(th), r * np.sin(th)
        return U * x + V * y + N * np.sqrt(max(0.0, 1.0 - x * x - y * y)), 0, None, margin
    return ggx_reflect(r1, r2), 0, None, margin
