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
        above = (INV_PI * fd)[:, None] * albedo * (1.0 - mat.metallic) * (1.0 - mat.subsurface) + (gs * ds)[:, None] * fs \
            + (mat.clearcoat * gr * fc * dr)[:, None]
        brdf = np.where((ndl <= 0)[:, None], below, above)
    return _lerp(brdf, bsdf, mat.transmission), near
