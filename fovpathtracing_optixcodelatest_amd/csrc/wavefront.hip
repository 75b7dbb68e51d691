// wavefront.hip -- the foveated path-tracing launch as a wavefront pipeline on gfx950.
//
// What the reference runs as ONE OptiX raygen thread per launch index with recursive
// optixTrace calls (PT_sv5_/deviceProgram.cu:392-732) is split here into queue-driven stages,
// one sample *slot* = (pass, launch index, sample number) per work item:
//
//   generate      __raygen__renderFrame :394-495   seed, ring test, jitter, camera ray, backplate
//   trace         optixTrace RADIANCE   :196-222   LBVH closest hit, LDS-staged stack
//   shade         __closesthit__/__miss__radiance :253-282,619-732 + SampleLights :303-344
//                 + Disney BSDF (Disney.cuh) + probe NEE (Probe.cuh); emits the shadow ray and
//                 the continuation ray, ballot-compacted into the next queues
//   shadow        optixTrace OCCLUSION  :224-248,284-300   any front-facing candidate
//   resolve       :541-616              ordered per-launch sample reduction, pass-ordered
//                                       block fill, exposure, Reinhard, sRGB, rgba8
//
// The three foveation passes of SampleRenderer::render() (SimplePathtracer.cpp:133-213) are
// one job: their slots share the queues, and resolve gives every pixel to its last writer in
// the reference's launch order (P, then M, then F; within a launch ascending y, x).
//
// Arithmetic: fp32 with the reference's operation order, -ffp-contract=off, IEEE divide and
// sqrt, transcendental functions from include/fovpt_detmath.h.  The only fused multiply-adds
// are the explicit ones in the (conservative) box test.
#include "fovpt_device.h"
#include <hip/hip_ext.h>
#include "../../include/fovpt_detmath.h"

// waves per SIMD the traversal kernel is compiled for (caps VGPRs at 64); measured best
#ifndef FOVPT_V_WAVES
#define FOVPT_V_WAVES 8
#endif

namespace {

// ------------------------------------------------------------------------------------------
// fp32 vector helpers with the semantics of sutil/vec_math.h
// ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ inline V3 v3(float s) { return v3(s, s, s); }
__device__ inline V3 v3(const float4& a) { return v3(a.x, a.y, a.z); }
__device__ inline V3 v3(const fovpt_float3& a) { return v3(a.x, a.y, a.z); }
__device__ inline V3 neg(const V3& a) { return v3(-a.x, -a.y, -a.z); }
__device__ inline V3 operator+(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline V3 operator-(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline V3 operator*(const V3& a, const V3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ inline V3 operator*(const V3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ inline V3 operator*(float s, const V3& a) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ inline V3 sub_sv(float a, const V3& b) { return v3(a - b.x, a - b.y, a - b.z); }          // float - float3
__device__ inline V3 add_vs(const V3& a, float b) { return v3(a.x + b, a.y + b, a.z + b); }          // float3 + float
__device__ inline V3 div_vs(const V3& a, float s) { float inv = 1.0f / s; return a * inv; }          // vec_math.h:487
__device__ inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(const V3& a, const V3& b)
{ return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline V3 normalize(const V3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
__device__ inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ inline V3 clamp3(const V3& v, float a, float b) { return v3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); }
__device__ inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
__device__ inline V3 lerp3(const V3& a, const V3& b, float t) { return a + t * (b - a); }
__device__ inline float sqr(float a) { return a * a; }
__device__ inline float4 f4(const V3& a, float w) { return make_float4(a.x, a.y, a.z, w); }

#define kPi (3.141592653589793f)
#define k2Pi (3.141592653589793f * 2.0f)
#define kInvPi (1.0f / kPi)
#define kInv2Pi (1.0f / k2Pi)

// ------------------------------------------------------------------------------------------
// RNG (cuda/random.h:34-59,101-104; maths.h:170-227)
// ------------------------------------------------------------------------------------------
__device__ inline uint32_t tea4(uint32_t v0, uint32_t v1)
{
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
__device__ inline float rnd(uint32_t& prev)
{
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}
struct Rng {
    uint32_t s1, s2;
    __device__ inline uint32_t next()
    {
        s1 = (s2 ^ ((s1 << 5) | (s1 >> 27))) ^ (s1 * s2);
        s2 = s1 ^ ((s2 << 12) | (s2 >> 20));
        return s1;
    }
    __device__ inline float randf()
    {
        // maths.h:199-210: value * (1/float(0xffffffff)), clamp to [0, 0.999999]
        return clampf((float)next() * (1.0f / 4294967296.0f), 0.f, 0.999999f);
    }
    __device__ inline float randf01()      // Randf(0,1), maths.h:213-217
    {
        float t = randf();
        return (1.0f - t) * 0.0f + t * 1.0f;
    }
};

// ------------------------------------------------------------------------------------------
// Probe (Probe.cuh)
// ------------------------------------------------------------------------------------------
__device__ inline void probe_dir_to_uv(const V3& dir, float& u, float& v)     // :38-46
{
    float theta = fovpt_dm_acosf(clampf(dir.y, -1.0f, 1.0f));
    float phi = (dir.x == 0.0f && dir.z == 0.0f) ? 0.0f : fovpt_dm_atan2f(dir.z, dir.x);
    u = (kPi + phi) * kInvPi * 0.5f;
    v = theta * kInvPi;
}
// row_mul is 1, or 0 when every row of the probe (texels and row tables) is bit-identical to row 0 -- the
// reference's shipped lighting is such a probe (loadColor, main.cpp:175-187) -- so that all lookups land in
// one L1-resident row instead of 8-33 MB of HBM/L2.  Same values either way.
__device__ inline float4 probe_eval(const fovpt_probe& pr, int row_mul, float u, float v)  // :61-67
{
    int px = max(0, min((int)(u * pr.width), pr.width - 1));
    int py = max(0, min((int)(v * pr.height), pr.height - 1));
    return ((const float4*)pr.data)[py * row_mul * pr.width + px];
}
__device__ inline int lower_bound(const float* __restrict__ a, int lower, int upper, float value)   // :119-136
{
    while (lower < upper) {
        int mid = lower + (upper - lower) / 2;
        if (a[mid] < value) lower = mid + 1;
        else upper = mid;
    }
    return lower;
}
// lower_bound through a guide table: G[m] = lower_bound(a, w_m), w_m = fl(m * fl(1/n)), m = 0..n+1 (relative to the
// start of the segment).  w_m is non-decreasing in m and lower_bound is monotone in its value, so for the m with
// w_m <= r < w_(m+1) the answer lies in [G[m], G[m+1]]: one or two candidates on a typical environment map, where the
// reference's search does log2(n) dependent loads.  m = int(r n) is off by at most one from that bracket (the products
// carry relative errors of 1e-7); it is corrected with two multiplications, and if the bracket still does not hold the
// whole segment is searched.  On a sorted array lower_bound is unique, so the result is the one the reference's full
// search returns (Probe.cuh:119-136); fovpt_set_probe only enables the tables after checking that the CDFs are
// non-decreasing.  (r1 bracketed with [G[m-1], G[m+2]]: three candidates on average, and more than four -- a divergent
// dependent search for the whole wave -- in 27 % of the row and 11 % of the column lookups on an HDR map: the shading
// launches took 0.315 instead of 0.229 ms per frame with a 1920x1080 HDR sky.)
// STRIDE: distance of consecutive array elements in floats (1: a plain CDF array; 8: the cdf member of the packed records)
template <int STRIDE>
__device__ inline int lower_bound_guided(const float* __restrict__ a, const uint32_t* __restrict__ guide, int base, int n, float value)
{
    const float fn = (float)n, inv = 1.0f / fn;
    int k = (int)(value * fn);
    k = max(0, min(k, n - 1));
    if ((float)k * inv > value) k = max(k - 1, 0);
    else if ((float)(k + 1) * inv <= value) k = min(k + 1, n - 1);
    const bool bracket = (float)k * inv <= value && value < (float)(k + 1) * inv;
    int lower = base + (bracket ? (int)guide[k] : 0);
    int upper = base + (bracket ? (int)guide[k + 1] : n);
    // A handful of candidates (the usual case): fetch them side by side instead of one after the other.
    // On a non-decreasing array the elements below `value` are a prefix of the range, so their number is
    // the offset the binary search would find.
    const int m = upper - lower;
    if (m <= 4) {
        if (m <= 0) return lower;
        const int last = upper - 1;
        const float v0 = a[(size_t)lower * STRIDE], v1 = a[(size_t)min(lower + 1, last) * STRIDE], v2 = a[(size_t)min(lower + 2, last) * STRIDE],
                    v3 = a[(size_t)min(lower + 3, last) * STRIDE];
        return lower + (int)(v0 < value) + (int)(m > 1 && v1 < value) + (int)(m > 2 && v2 < value) + (int)(m > 3 && v3 < value);
    }
    while (lower < upper) {
        int mid = lower + (upper - lower) / 2;
        if (a[(size_t)mid * STRIDE] < value) lower = mid + 1;
        else upper = mid;
    }
    return lower;
}
// rec: per texel one 32-byte record {cdfX, pdfX, r, g, b, -, -, -} (built at setProbe next to the guide tables, or null).
// The column search, the pdf and the colour of a sample then come from one or two cache lines instead of four arrays: on a
// 1920x1080 HDR map (58 MB of tables, nothing of it in L1) every lookup of the split layout is its own miss.
__device__ inline void probe_sample(const fovpt_probe& pr, const uint32_t* __restrict__ guide_x, const uint32_t* __restrict__ guide_y, const float4* __restrict__ rec,
                                    int row_mul, V3& dir, V3& color, float& pdf, Rng& rng)   // :138-169
{
    float r1 = rng.randf01();
    float r2 = rng.randf01();
    int row, col;
    if (guide_x) {
        row = lower_bound_guided<1>(pr.cdfValuesY, guide_y, 0, pr.height, r1);
        const int rx = row * row_mul;
        if (rec) col = lower_bound_guided<8>((const float*)rec, guide_x + (size_t)rx * (pr.width + 2), rx * pr.width, pr.width, r2) - rx * pr.width;
        else col = lower_bound_guided<1>(pr.cdfValuesX, guide_x + (size_t)rx * (pr.width + 2), rx * pr.width, pr.width, r2) - rx * pr.width;
    } else {
        row = lower_bound(pr.cdfValuesY, 0, pr.height, r1);
        const int rx = row * row_mul;
        col = lower_bound(pr.cdfValuesX, rx * pr.width, (rx + 1) * pr.width, r2) - rx * pr.width;
    }
    const int rowx = row * row_mul;
    if (guide_x && rec) {
        const float4 ra = rec[2 * (size_t)(rowx * pr.width + col)], rb = rec[2 * (size_t)(rowx * pr.width + col) + 1];
        color = v3(ra.z, ra.w, rb.x);
        pdf = ra.y * pr.pdfValuesY[row];
    } else {
        color = v3(((const float4*)pr.data)[rowx * pr.width + col]);
        pdf = pr.pdfValuesX[rowx * pr.width + col] * pr.pdfValuesY[row];
    }
    float u = col / float(pr.width);
    float v = row / float(pr.height);
    float sinTheta, cosTheta;
    fovpt_dm_sincos(v * kPi, &sinTheta, &cosTheta);
    if (sinTheta == 0.0f) pdf = 0.0f;
    else pdf *= pr.width * pr.height / (2.0f * kPi * kPi * sinTheta);
    // ProbeUVToDir :48-58 (theta = v*kPi is the same value as above)
    float sinPhi, cosPhi;
    fovpt_dm_sincos(u * 2.0f * kPi, &sinPhi, &cosPhi);
    dir = v3(-sinTheta * cosPhi, cosTheta, -sinTheta * sinPhi);
}

// ------------------------------------------------------------------------------------------
// Disney BSDF (Disney.cuh)
// ------------------------------------------------------------------------------------------
typedef fovpt_material Mat;

// The reference writes 1.0 / sqrtf(x) and 0.5 + y with binary64 literals: a binary64 operation on binary32 values, rounded back to
// binary32.  That equals the correctly rounded binary32 operation (rounding twice is innocuous for + - * / sqrt when the wide format
// has at least 2 * 24 + 2 significant bits), which is how the compiler emits them: no binary64 instruction comes from these lines
// (the 231 of k_shade are the polynomial cores of include/fovpt_detmath.h).  FOVPT_OP_RSQRTD / _HALFPLUS check the device's result
// against the oracle's binary64 expression over every binade (tests/test_gpu_parity.py).
__device__ inline float rcp_of_sqrt_as_the_reference(float x) { return (float)(1.0 / (double)sqrtf(x)); }
__device__ inline void basis_from_vector(const V3& w, V3& u, V3& v)          // maths.h:94-108
{
    if (fabsf(w.x) > fabsf(w.y)) {
        float invLen = rcp_of_sqrt_as_the_reference(w.x * w.x + w.z * w.z);
        u = v3(-w.z * invLen, 0.0f, w.x * invLen);
    } else {
        float invLen = rcp_of_sqrt_as_the_reference(w.y * w.y + w.z * w.z);
        u = v3(0.0f, w.z * invLen, -w.y * invLen);
    }
    v = cross(w, u);
}
__device__ inline V3 safe_normalize(const V3& a)                              // maths.h:144-156
{
    float m = dot(a, a);
    if ((double)m > 0.0) return a * rcp_of_sqrt_as_the_reference(m);
    return v3(0.0f);
}
__device__ inline float half_plus_as_the_reference(float y) { return (float)(0.5 + (double)y); }      // (see rcp_of_sqrt_as_the_reference)
__device__ inline float schlick(float u)                                      // Disney.cuh:51-56
{
    float m = clampf(1 - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
__device__ inline float gtr1_pre(float NDotH, float a2, float log_a2)         // GTR1 :58-64 with a*a and log(a*a) given (a2 < 0: a >= 1)
{
    if (a2 < 0.0f) return kInvPi;
    float t = 1 + (a2 - 1) * NDotH * NDotH;
    return (a2 - 1) / (kPi * log_a2 * t);
}
__device__ inline float gtr2(float NDotH, float a)                            // :66-71
{
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (kPi * t * t);
}
__device__ inline float smith_ggx(float NDotv, float alphaG)                  // :73-78
{
    float a = alphaG * alphaG;
    float b = NDotv * NDotv;
    return 1 / (NDotv + sqrtf(a + b - a * b));
}
__device__ inline float fresnel(float VDotN, float etaI, float etaT)          // Fr, :81-98
{
    float SinThetaT2 = sqr(etaI / etaT) * (1.0f - VDotN * VDotN);
    if (SinThetaT2 > 1.0f) return 1.0f;
    float LDotN = sqrtf(1.0f - SinThetaT2);
    float eta = etaT / etaI;
    float r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN);
    float r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN);
    return 0.5f * (sqr(r1) + sqr(r2));
}
// Terms of BSDFPdf / BSDFSample / BSDFEval that depend on the hit and the view direction only.  A hit
// evaluates the BSDF for two light directions (the probe sample and the BSDF sample) and its pdf for
// both: the reference recomputes these terms every time, here they are computed once -- the same
// expressions on the same operands, hence the same bits.
struct BsdfView {
    float etaI, etaO;
    float NDotV;          // dot(N, V)
    float FrV;            // Fr(dot(N, V), etaI, etaO)                       :154, :199, :340
    float a;              // max(0.001, roughness)
    float GV_a, GV_q;     // SmithGGX(NDotV, a), SmithGGX(NDotV, 0.25)       :350, :375, :383
    float FV;             // SchlickFresnel(NDotV)                           :362, :378
    V3 Cspec0;            // :330-334
    float cc_a2, cc_log;  // clearcoat GTR1: a = mix(.1, .001, clearcoatGloss); a*a and log(a*a)    :381, :58-64
};
__device__ inline BsdfView bsdf_view(const Mat& mat, const V3& albedo, float etaI, float etaO, const V3& N, const V3& V)
{
    BsdfView w;
    w.etaI = etaI; w.etaO = etaO;
    w.NDotV = dot(N, V);
    w.FrV = fresnel(w.NDotV, etaI, etaO);
    w.a = fmaxf(0.001f, mat.roughness);
    w.GV_a = smith_ggx(w.NDotV, w.a);
    w.GV_q = smith_ggx(w.NDotV, .25f);
    w.FV = schlick(w.NDotV);
    const V3 Cdlin = albedo;
    const float Cdlum = (float)(.3 * (double)Cdlin.x + .6 * (double)Cdlin.y + .1 * (double)Cdlin.z);
    const V3 Ctint = Cdlum > 0.0f ? div_vs(Cdlin, Cdlum) : v3(1.0f);
    w.Cspec0 = lerp3((float)((double)mat.specular * .08) * lerp3(v3(1.0f), Ctint, mat.specularTint), Cdlin, mat.metallic);
    const float cc_a = lerpf(.1f, .001f, mat.clearcoatGloss);
    w.cc_a2 = cc_a >= 1 ? -1.0f : cc_a * cc_a;                  // -1: GTR1 returns 1/pi (a >= 1)
    w.cc_log = cc_a >= 1 ? 0.0f : fovpt_dm_logf(w.cc_a2);
    return w;
}
__device__ float bsdf_pdf(const Mat& mat, const BsdfView& w, const V3& n, const V3& V, const V3& L)   // :152-193
{
    if (dot(L, n) <= 0.0f) {
        float bsdfPdf = 0.0f;
        float brdfPdf = kInv2Pi * mat.subsurface * 0.5f;
        return lerpf(brdfPdf, bsdfPdf, mat.transmission);
    }
    const float F = w.FrV;
    const float a = w.a;
    const V3 half = safe_normalize(L + V);
    const float cosThetaHalf = fabsf(dot(half, n));
    const float pdfHalf = gtr2(cosThetaHalf, a) * cosThetaHalf;
    float pdfSpec = 0.25f * pdfHalf / fmaxf(1.e-6f, dot(L, half));
    float pdfDiff = fabsf(dot(L, n)) * kInvPi * (1.0f - mat.subsurface);
    float bsdfPdf = pdfSpec * F;
    float brdfPdf = lerpf(pdfDiff, pdfSpec, 0.5f);
    return lerpf(brdfPdf, bsdfPdf, mat.transmission);
}
// returns pdf; light = sampled direction.
// The lanes of a wave take different branches of BSDFSample, and three of the four end in the same
// work: one sincos and a change of basis.  The random numbers are drawn in the reference's order and
// the branch is remembered; the sincos and (for both "sample specular" branches, Disney.cuh:211-226
// and :287-307) the GGX half vector are then evaluated once for all lanes.
__device__ float bsdf_sample(const Mat& mat, const BsdfView& w, const V3& U, const V3& V, const V3& N,
                             const V3& view, V3& light, Rng& rng)             // :197-315
{
    enum { SPECULAR, UNIFORM, COSINE };
    int kind;
    float r1, r2, z = 0.0f, angle;
    if (rng.randf() < mat.transmission) {
        const float F = w.FrV;
        if (rng.randf() < F) {
            r1 = rng.randf01();
            r2 = rng.randf01();
            kind = SPECULAR;
        } else {
            // Refract, :36-49
            float eta = w.etaI / w.etaO;
            float cosThetaI = w.NDotV;
            float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
            float sin2ThetaT = eta * eta * sin2ThetaI;
            if (sin2ThetaT >= 1) return 0.0f;
            float cosThetaT = sqrtf(1.0f - sin2ThetaT);
            light = eta * neg(view) + (eta * cosThetaI - cosThetaT) * N;
            return (1.0f - F) * mat.transmission;
        }
    } else {
        r1 = rng.randf01();
        r2 = rng.randf01();
        if (rng.randf() < 0.5f) {
            if (rng.randf() < mat.subsurface) { kind = UNIFORM; z = rng.randf01(); }
            else kind = COSINE;
        } else {
            kind = SPECULAR;
        }
    }
    if (kind == SPECULAR) angle = r1 * k2Pi;                   // phiHalf
    else if (kind == COSINE) angle = k2Pi * r2;                // theta, maths.h:262
    else angle = k2Pi * rng.randf01();                         // phi, maths.h:248
    float sn, cs;
    fovpt_dm_sincos(angle, &sn, &cs);
    if (kind == SPECULAR) {
        const float a = w.a;
        const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (sqr(a) - 1.0f) * r2));
        const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - sqr(cosThetaHalf)));
        V3 half = U * (sinThetaHalf * cs) + V * (sinThetaHalf * sn) + N * cosThetaHalf;
        if (dot(half, view) <= 0.0f) half = half * -1.0f;
        light = 2.0f * dot(view, half) * half - view;
    } else if (kind == UNIFORM) {
        // UniformSampleHemisphere, maths.h:243-254
        const float ww = sqrtf(1.0f - z * z);
        const float x = cs * ww, y = sn * ww;
        light = U * x + V * y - N * z;
    } else {
        // CosineSampleHemisphere, maths.h:256-277
        const float r = sqrtf(r1);
        const float sx = r * cs, sy = r * sn;
        const float zz = sqrtf(fmaxf(0.0f, 1.0f - sx * sx - sy * sy));
        light = U * sx + V * sy + N * zz;
    }
    return bsdf_pdf(mat, w, N, view, light);
}
__device__ V3 bsdf_eval(const Mat& mat, const V3& albedo, const BsdfView& w, const V3& N, const V3& V, const V3& L)   // :318-427
{
    float NDotL = dot(N, L);
    const float NDotV = w.NDotV;
    V3 H = normalize(L + V);
    float NDotH = dot(N, H);
    float LDotH = dot(L, H);
    const V3 Cdlin = albedo;
    const V3 Cspec0 = w.Cspec0;
    V3 bsdf = v3(0.0f);
    V3 brdf = v3(0.0f);
    // both lobes use the same D and G terms on the upper hemisphere (:346-351 and :371-376)
    float Ds = 0.0f, Gs = 0.0f;
    if (NDotL > 0) {
        Ds = gtr2(NDotH, w.a);
        Gs = w.GV_a * smith_ggx(NDotL, w.a);
    }
    if (mat.transmission > 0.0f) {
        if (NDotL <= 0) {
            const float F = w.FrV;
            bsdf = v3(mat.transmission * (1.0f - F) / fabsf(NDotL) * (1.0f - mat.metallic));
        } else {
            float FH = fresnel(LDotH, w.etaI, w.etaO);
            V3 Fs = lerp3(Cspec0, v3(1.0f), FH);
            bsdf = Gs * Fs * Ds;
        }
    }
    if (mat.transmission < 1.0f) {
        if (NDotL <= 0) {
            if (mat.subsurface > 0.0f) {
                V3 s = v3(sqrtf(mat.color.x), sqrtf(mat.color.y), sqrtf(mat.color.z));
                float FL = schlick(fabsf(NDotL)), FV = w.FV;
                float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
                brdf = kInvPi * s * mat.subsurface * Fd * (1.0f - mat.metallic);
            }
        } else {
            float FH = schlick(LDotH);
            V3 Fs = lerp3(Cspec0, v3(1.f), FH);
            float FL = schlick(NDotL), FV = w.FV;
            float Fd90 = half_plus_as_the_reference(2.0f * LDotH * LDotH * mat.roughness);     // Disney.cuh: 0.5 + (binary32 product), in binary64
            float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
            float Dr = gtr1_pre(NDotH, w.cc_a2, w.cc_log);
            float Fc = lerpf(.04f, 1.0f, FH);
            float Gr = smith_ggx(NDotL, .25f) * w.GV_q;
            brdf = add_vs(kInvPi * Fd * Cdlin * (1.0f - mat.metallic) * (1.0f - mat.subsurface) + Gs * Fs * Ds,
                          mat.clearcoat * Gr * Fc * Dr);
        }
    }
    (void)NDotV;
    return lerp3(brdf, bsdf, mat.transmission);
}

// ------------------------------------------------------------------------------------------
// wavefront plumbing
// ------------------------------------------------------------------------------------------
// Ballot compaction into a sharded queue.  Lanes with pred get distinct positions inside shard
// blockIdx % 8: wave ballot + popcount prefix, the four wave totals meet in LDS, and ONE atomic per
// block-iteration reserves the range.  Must be called by all 256 threads of the block (two barriers).
// Shard selection of a launch: 0 = all eight shards; 1 / 2 = the first / second four -- the two CHAINS of a frame that is rendered
// as two independent halves (fovpt_config.chains_per_frame = 2, fovpt_api.hip): each chain's kernels read and append only inside
// their own four shards of the same queue buffers.
__device__ inline uint32_t sel_first(uint32_t sel) { return sel == 2u ? 4u : 0u; }
__device__ inline uint32_t sel_mask(uint32_t sel) { return sel == 0u ? (uint32_t)FOVPT_SHARDS - 1u : 3u; }

__device__ inline uint32_t block_append(Counters* cnt, int word, uint32_t cap, bool pred, uint32_t* s_scratch /* [6] */, uint32_t sel = 0u)
{
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_scratch[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t c0 = s_scratch[0], c1 = s_scratch[1], c2 = s_scratch[2], c3 = s_scratch[3];
    const uint32_t shard = sel_first(sel) + (blockIdx.x & sel_mask(sel));
    if (threadIdx.x == 0) {
        const uint32_t total = c0 + c1 + c2 + c3;
        s_scratch[4] = total ? atomicAdd(&cnt->shard[shard][word], total) : 0u;
    }
    __syncthreads();
    const uint32_t before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
    const uint32_t pos = shard * cap + s_scratch[4] + before + prefix;
    __syncthreads();                 // s_scratch is reused by the next call
    return pos;
}

// Two appends at once (shadow queue and next radiance queue) behind ONE pair of barriers.
__device__ inline void block_append2(Counters* cnt, int word_a, bool pred_a, int word_b, bool pred_b, uint32_t cap,
                                     uint32_t* s_scratch /* [10] */, uint32_t& pos_a, uint32_t& pos_b, uint32_t sel = 0u)
{
    const unsigned long long ma = __ballot(pred_a), mb = __ballot(pred_b);
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t pa = __popcll(ma & below), pb = __popcll(mb & below);
    if (lane == 0) { s_scratch[wave] = (uint32_t)__popcll(ma); s_scratch[4 + wave] = (uint32_t)__popcll(mb); }
    __syncthreads();
    const uint32_t a0 = s_scratch[0], a1 = s_scratch[1], a2 = s_scratch[2], a3 = s_scratch[3];
    const uint32_t b0 = s_scratch[4], b1 = s_scratch[5], b2 = s_scratch[6], b3 = s_scratch[7];
    const uint32_t shard = sel_first(sel) + (blockIdx.x & sel_mask(sel));
    if (threadIdx.x == 0) {
        const uint32_t ta = a0 + a1 + a2 + a3;
        s_scratch[8] = ta ? atomicAdd(&cnt->shard[shard][word_a], ta) : 0u;
    }
    if (threadIdx.x == 64) {
        const uint32_t tb = b0 + b1 + b2 + b3;
        s_scratch[9] = tb ? atomicAdd(&cnt->shard[shard][word_b], tb) : 0u;
    }
    __syncthreads();
    pos_a = shard * cap + s_scratch[8] + (wave > 0 ? a0 : 0u) + (wave > 1 ? a1 : 0u) + (wave > 2 ? a2 : 0u) + pa;
    pos_b = shard * cap + s_scratch[9] + (wave > 0 ? b0 : 0u) + (wave > 1 ? b1 : 0u) + (wave > 2 ? b2 : 0u) + pb;
    __syncthreads();                 // s_scratch is reused by the next iteration
}

// The same with the SECOND append partitioned inside the block's range by one bit of the entry: class 0 first, then class 1.
// k_shade uses it for the next bounce's rays with the sign of dir.y as the class: the 16 consecutive rays of a traversal round then
// point more nearly the same way and need more nearly the same number of steps.  It is the block-local form of a direction-sorted
// queue -- no extra memory, no extra atomics, ~12 instructions per 256 rays -- and queue order never changes a result (every ray
// writes to its own slot).  Measured: foveated frames -1 % (C3, street), the uniform 1-spp frame +1 %: on for foveated frames only.
__device__ inline void block_append2d(Counters* cnt, int word_a, bool pred_a, int word_b, bool pred_b, bool cls_b, uint32_t cap,
                                      uint32_t* s_scratch /* [14] */, uint32_t& pos_a, uint32_t& pos_b, uint32_t sel = 0u)
{
    const unsigned long long ma = __ballot(pred_a), mb0 = __ballot(pred_b & !cls_b), mb1 = __ballot(pred_b & cls_b);
    const uint32_t lane = __lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t pa = __popcll(ma & below), pb = __popcll((cls_b ? mb1 : mb0) & below);
    if (lane == 0) { s_scratch[wave] = (uint32_t)__popcll(ma); s_scratch[4 + wave] = (uint32_t)__popcll(mb0); s_scratch[8 + wave] = (uint32_t)__popcll(mb1); }
    __syncthreads();
    const uint32_t a0 = s_scratch[0], a1 = s_scratch[1], a2 = s_scratch[2], a3 = s_scratch[3];
    uint32_t before_b = 0u, tb0 = 0u, tb1 = 0u;
#pragma unroll
    for (uint32_t w = 0; w < 4u; w++) {
        const uint32_t b0 = s_scratch[4 + w], b1 = s_scratch[8 + w];
        if (w < wave) before_b += cls_b ? b1 : b0;
        tb0 += b0; tb1 += b1;
    }
    const uint32_t shard = sel_first(sel) + (blockIdx.x & sel_mask(sel));
    if (threadIdx.x == 0) {
        const uint32_t ta = a0 + a1 + a2 + a3;
        s_scratch[12] = ta ? atomicAdd(&cnt->shard[shard][word_a], ta) : 0u;
    }
    if (threadIdx.x == 64) s_scratch[13] = (tb0 + tb1) ? atomicAdd(&cnt->shard[shard][word_b], tb0 + tb1) : 0u;
    __syncthreads();
    pos_a = shard * cap + s_scratch[12] + (wave > 0 ? a0 : 0u) + (wave > 1 ? a1 : 0u) + (wave > 2 ? a2 : 0u) + pa;
    pos_b = shard * cap + s_scratch[13] + (cls_b ? tb0 : 0u) + before_b + pb;
    __syncthreads();                 // s_scratch is reused by the next iteration
}

// logical index -> physical index of a sharded queue (all in scalar registers, no indexing)
struct ShardMap {
    uint32_t p1, p2, p3, p4, p5, p6, p7, p8;     // exclusive prefix sums of the shard counts (p0 = 0)
    uint32_t first_cap;                          // physical offset of the first shard of the selection (0, or 4 * cap for the second chain)
    __device__ inline void load(const Counters* cnt, int word, uint32_t sel = 0u, uint32_t cap = 0u)
    {
        const uint32_t f = sel_first(sel);
        first_cap = f * cap;
        p1 = cnt->shard[f][word]; p2 = p1 + cnt->shard[f + 1][word]; p3 = p2 + cnt->shard[f + 2][word]; p4 = p3 + cnt->shard[f + 3][word];
        if (sel == 0u) { p5 = p4 + cnt->shard[4][word]; p6 = p5 + cnt->shard[5][word]; p7 = p6 + cnt->shard[6][word]; p8 = p7 + cnt->shard[7][word]; }
        else p5 = p6 = p7 = p8 = p4;             // a chain's four shards: no index reaches the other four
    }
    __device__ inline uint32_t total() const { return p8; }
    __device__ inline uint32_t phys(uint32_t i, uint32_t cap) const
    {
        uint32_t s = 0, base = 0;
        if (i >= p1) { s = 1; base = p1; }
        if (i >= p2) { s = 2; base = p2; }
        if (i >= p3) { s = 3; base = p3; }
        if (i >= p4) { s = 4; base = p4; }
        if (i >= p5) { s = 5; base = p5; }
        if (i >= p6) { s = 6; base = p6; }
        if (i >= p7) { s = 7; base = p7; }
        return first_cap + s * cap + (i - base);
    }
    // the same for lane index i of 16 consecutive indices starting at the wave-uniform i0: the shard is
    // found with scalar instructions unless the 16 straddle a shard boundary
    __device__ inline uint32_t phys16(uint32_t i, uint32_t i0, uint32_t cap) const
    {
        uint32_t s = 0, base = 0, next = p1;
        if (i0 >= p1) { s = 1; base = p1; next = p2; }
        if (i0 >= p2) { s = 2; base = p2; next = p3; }
        if (i0 >= p3) { s = 3; base = p3; next = p4; }
        if (i0 >= p4) { s = 4; base = p4; next = p5; }
        if (i0 >= p5) { s = 5; base = p5; next = p6; }
        if (i0 >= p6) { s = 6; base = p6; next = p7; }
        if (i0 >= p7) { s = 7; base = p7; next = 0xffffffffu; }
        if (i0 + 15u < next) return first_cap + s * cap - base + i;
        return phys(i, cap);
    }
};

// p: index of the pass within this JOB; the rotation uses its index within the caller's FRAME (a chunked frame runs every
// pass as jobs of its own, and the gather plan is made for the frame)
__device__ inline bool launch_owned(const FrameDev& fd, int p, uint32_t lx, uint32_t ly)
{
    if (fd.world <= 1) return true;
    uint32_t tx = lx / (uint32_t)fd.tile_w, ty = ly / (uint32_t)fd.tile_h;
    return (int)((tx + 3u * ty + fd.pass[p].frame_pass) % (uint32_t)fd.world) == fd.rank;
}

// ring test of deviceProgram.cu:433-440 on the block's top-left pixel (uint arithmetic wraps)
__device__ inline bool ring_alive(const FrameDev& fd, const PassDev& P, uint32_t lx, uint32_t ly, uint32_t& ix, uint32_t& iy)
{
    ix = lx * P.fx + P.offx;
    iy = ly * P.fy + P.offy;
    const float dx = (float)ix - (float)fd.cx, dy = (float)iy - (float)fd.cy, dz = 0.0f - 0.0f;
    const float range = sqrtf(dx * dx + dy * dy + dz * dz);
    return !(range < P.r_inner || range > P.r_outer);
}

// ---- generate ----------------------------------------------------------------------------
// A rank's share of pass P in the tile-sharded frame, as an index space of its own: tile rows x the rank's tiles of a row
// (every world-th tile, launch_owned) x the tile's launch indices x samples.  k_generate<true> walks it instead of all sample
// slots -- 1 / world of the work; the sample slot of a launch index, and so everything downstream, is the same.
struct OwnedSpace {
    uint32_t ty0, tile_rows, per_row, tile_lis, count;      // first tile row, tile rows, the rank's tiles per tile row (at most), launch indices per tile
    __host__ __device__ inline void init(const FrameDev& fd, const PassDev& P)
    {
        const uint32_t tw = (uint32_t)fd.tile_w, th = (uint32_t)fd.tile_h, tiles_x = (P.gw + tw - 1u) / tw;
        ty0 = P.row0 / th;
        tile_rows = P.row1 > P.row0 ? (P.row1 - 1u) / th - ty0 + 1u : 0u;
        per_row = (tiles_x + (uint32_t)fd.world - 1u) / (uint32_t)fd.world;
        tile_lis = tw * th;
        count = tile_rows * per_row * tile_lis * P.spp;
    }
    // entry v -> launch index (lx, ly) and sample s; false: the entry is padding (a tile beyond the row's end, a launch index
    // beyond the grid or outside the job's rows)
    __device__ inline bool at(const FrameDev& fd, const PassDev& P, uint32_t v, uint32_t& lx, uint32_t& ly, uint32_t& s) const
    {
        const uint32_t tw = (uint32_t)fd.tile_w, th = (uint32_t)fd.tile_h, world = (uint32_t)fd.world;
        const uint32_t t = v / P.spp;
        s = v - t * P.spp;
        const uint32_t tile = t / tile_lis, in = t - tile * tile_lis;
        const uint32_t row = tile / per_row, k = tile - row * per_row;
        const uint32_t ty = ty0 + row;
        const uint32_t first = ((uint32_t)fd.rank + world - (3u * ty + P.frame_pass) % world) % world;    // (tx + 3 ty + pass) % world == rank
        const uint32_t tx = first + k * world;
        const uint32_t iy = in / tw;
        lx = tx * tw + (in - iy * tw);
        ly = ty * th + iy;
        return lx < P.gw && ly >= P.row0 && ly < P.row1;
    }
};

// One block iteration of the generate kernels: the camera rays of its live threads (launch index (lx, ly) of pass P, sample s,
// sample slot `slot`, pixel (ix, iy) from ring_alive) and their append to queue 0.  Called by every thread of the block.
__device__ inline void generate_rays(const FrameDev& fd, const PathState& ps, const RayQueue& queue0, uint32_t cap, Counters* __restrict__ cnt,
                                     uint32_t* s_scratch, uint32_t sel, const PassDev& P, bool live, uint32_t slot, uint32_t lx, uint32_t ly,
                                     uint32_t s, uint32_t ix, uint32_t iy)
{
    V3 ray_dir = v3(0.f);
    if (live) {
        uint32_t seed = tea4(ly * (uint32_t)fd.w + lx, P.subframe);        // :411
        for (uint32_t k = 0; k < s; k++) { (void)rnd(seed); (void)rnd(seed); }   // earlier samples' jitter draws
        Rng rng;                                                            // Random(seed), maths.h:176-180
        rng.s1 = 315645664u + seed;
        rng.s2 = rng.s1 ^ 0x13ab45feu;
        const float jx = rnd(seed);                                         // :479, x first
        const float jy = rnd(seed);
        const float dx = 2.0f * (((float)ix + jx) / (float)fd.w) - 1.0f;    // :483-486
        const float dy = 2.0f * (((float)iy + jy) / (float)fd.h) - 1.0f;
        const V3 U = v3(fd.U[0], fd.U[1], fd.U[2]), V = v3(fd.V[0], fd.V[1], fd.V[2]), W = v3(fd.W[0], fd.W[1], fd.W[2]);
        const V3 dir = normalize(dx * U + dy * V + W);                      // :491
        ray_dir = dir;
        ps.rng[slot] = make_uint4(rng.s1, rng.s2, 0u, 0u);                  // stateFlags 0, depth 0
        // Nothing else is initialised: pathThroughput / rayEta are (1,1,1) / 1 until the first shaded hit
        // (k_shade), the radiance cells [0, depth) and alpha are written exactly once before resolve
        // reads them (depth and FLAG_ALPHA_SET tell it which), :450-451
        if (ps.guide_n) { ps.guide_n[slot] = make_float4(0.f, 0.f, 0.f, 0.f); ps.guide_a[slot] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (s == P.spp - 1) {                                               // backplate of the last sample, :495
            float u, v;
            probe_dir_to_uv(dir, u, v);
            ps.backplate[P.launch_base + (ly - P.row0) * P.gw + lx] = probe_eval(fd.probe, fd.probe_row_mul, u, v);
        }
    }
    const uint32_t pos = block_append(cnt, FOVPT_CNT_Q(0), cap, live, s_scratch, sel);
    if (live) {
        queue0.o[pos] = make_float4(fd.eye[0], fd.eye[1], fd.eye[2], __uint_as_float(slot));
        queue0.d[pos] = f4(ray_dir, 0.f);
    }
}

__global__ __launch_bounds__(FOVPT_BLOCK) void k_generate(const FrameDev fd, PathState ps, RayQueue queue0, uint32_t cap,
                                                          Counters* __restrict__ cnt, uint32_t slot_begin, uint32_t total_slots, uint32_t sel)
{
    // slots [slot_begin, total_slots) into the shards of `sel` (a whole job: 0 .. all slots, all shards; a chain: its half)
    __shared__ uint32_t s_scratch[6];
    for (uint32_t base = slot_begin + blockIdx.x * FOVPT_BLOCK; base < total_slots; base += gridDim.x * FOVPT_BLOCK) {
        const uint32_t slot = base + threadIdx.x;
        bool live = slot < total_slots;
        int p = 0;
        uint32_t lx = 0, ly = 0, s = 0, ix = 0, iy = 0;
        if (live) {
            while (p + 1 < fd.npass && slot >= fd.pass[p + 1].slot_base) p++;
            const PassDev& P = fd.pass[p];
            const uint32_t rel = slot - P.slot_base;
            const uint32_t li = rel / P.spp;
            s = rel - li * P.spp;
            ly = li / P.gw;
            lx = li - ly * P.gw;
            ly += P.row0;                                      // slots and launch records are relative to the chunk
            live = ring_alive(fd, P, lx, ly, ix, iy) && launch_owned(fd, p, lx, ly);
        }
        generate_rays(fd, ps, queue0, cap, cnt, s_scratch, sel, fd.pass[p], live, slot, lx, ly, s, ix, iy);
    }
}

// The same for a rank of a tile-sharded frame (world > 1, one chain): the grid walks the rank's own tiles, see OwnedSpace -- every
// pass padded to whole block iterations, so that a block iteration serves ONE pass and reads its record with scalar loads.
__global__ __launch_bounds__(FOVPT_BLOCK) void k_generate_owned(const FrameDev fd, PathState ps, RayQueue queue0, uint32_t cap,
                                                                Counters* __restrict__ cnt)
{
    __shared__ uint32_t s_scratch[6];
    static_assert(FOVPT_MAX_PASSES == 3, "the pass of a block iteration is selected by hand below");
    OwnedSpace o0, o1, o2;
    o0.count = o1.count = o2.count = 0u;
    if (fd.npass > 0) o0.init(fd, fd.pass[0]);
    if (fd.npass > 1) o1.init(fd, fd.pass[1]);
    if (fd.npass > 2) o2.init(fd, fd.pass[2]);
    const uint32_t b0 = (o0.count + FOVPT_BLOCK - 1u) / FOVPT_BLOCK, b1 = (o1.count + FOVPT_BLOCK - 1u) / FOVPT_BLOCK,
                   b2 = (o2.count + FOVPT_BLOCK - 1u) / FOVPT_BLOCK;
    for (uint32_t it = blockIdx.x; it < b0 + b1 + b2; it += gridDim.x) {
        const int p = it < b0 ? 0 : it < b0 + b1 ? 1 : 2;
        const OwnedSpace own = p == 0 ? o0 : p == 1 ? o1 : o2;
        const PassDev& P = fd.pass[p];
        const uint32_t v = (it - (p == 0 ? 0u : p == 1 ? b0 : b0 + b1)) * FOVPT_BLOCK + threadIdx.x;
        uint32_t lx = 0, ly = 0, s = 0, ix = 0, iy = 0;
        bool live = v < own.count && own.at(fd, P, v, lx, ly, s);
        const uint32_t slot = P.slot_base + ((ly - P.row0) * P.gw + lx) * P.spp + s;
        live = live && ring_alive(fd, P, lx, ly, ix, iy) && launch_owned(fd, p, lx, ly);
        generate_rays(fd, ps, queue0, cap, cnt, s_scratch, 0u, P, live, slot, lx, ly, s, ix, iy);
    }
}

// ---- traversal ---------------------------------------------------------------------------
// Quad-cooperative traversal: FOUR adjacent lanes share one ray.  On a wide node lane j tests child j
// (so a node visit is one box test deep instead of four), on a leaf lane j tests triangle j; the four
// results meet through the wave ballot and DPP quad permutes (register to register).  A wave thus
// carries 16 rays, each step is ~3x shorter than with one lane per ray, and divergence is between 16
// rays instead of 64.  A step is one link of a serial chain per wave -- ~1000 cycles, ~60 % of them the wait for the two node
// loads of the wave's slowest quad, the rest the wave's own ~45 instructions (measured with s_memtime stamps,
// profiles/r03_step_cycles.txt) -- so the steps are written instruction by instruction.
struct RayT {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;          // safe reciprocal direction for the box test
    float nox, noy, noz;       // -origin * reciprocal
};

__device__ inline float safe_rcp(float d)
{
    // v_rcp_f32 (1 ulp) is enough: the reciprocal only feeds the conservative box test, whose boxes are
    // padded by >= 1e-5 of the coordinate magnitude at build time (bvh_build.hip, k_tri_bounds)
    const float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}
__device__ inline void ray_setup(RayT& r, const float4& o, const float4& d)
{
    r.ox = o.x; r.oy = o.y; r.oz = o.z; r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = safe_rcp(d.x); r.iy = safe_rcp(d.y); r.iz = safe_rcp(d.z);
    r.nox = -o.x * r.ix; r.noy = -o.y * r.iy; r.noz = -o.z * r.iz;
}
// conservative slab test (boxes are padded at build time); returns entry distance in tn.
// (Six scalar fmas: the v_pk_fma_f32 form on (lo, hi) pairs was measured 12 % slower.)
__device__ inline bool box_hit(const RayT& r, float lx, float ly, float lz, float hx, float hy, float hz, float tmin, float tmax, float& tn)
{
    const float ax = __builtin_fmaf(lx, r.ix, r.nox), bx = __builtin_fmaf(hx, r.ix, r.nox);
    const float ay = __builtin_fmaf(ly, r.iy, r.noy), by = __builtin_fmaf(hy, r.iy, r.noy);
    const float az = __builtin_fmaf(lz, r.iz, r.noz), bz = __builtin_fmaf(hz, r.iz, r.noz);
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    // tmax > 0: as signed integers the bit patterns order like the values whenever one is positive,
    // so the clamp is one v_min_i32 (no canonicalisation of the loop-carried tmax)
    const float far = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float t1 = __int_as_float(min(__float_as_int(far), __float_as_int(tmax)));
    tn = t0;
    // Relative slack: per axis the fma, the rounded -o/d and the 1-ulp reciprocal are off by up to ~2.4e-7
    // of t, independently for the near and the far plane; the build-time padding of the boxes does not
    // cover that for small geometry seen from far away
    return t0 <= t1 * 1.000001f;
}

// triangle records: Moeller-Trumbore on them is in leaf_step, in the operation order of the parity contract
// (oracle intersect_tri)
__device__ inline TriRec load_tri_off(const TriRec* __restrict__ tris, uint32_t byte_off)
{
    const float4* p = (const float4*)((const char*)tris + byte_off);
    const float4 a = p[0], b = p[1], c = p[2];
    TriRec T;
    T.v0x = a.x; T.v0y = a.y; T.v0z = a.z; T.e1x = a.w;
    T.e1y = b.x; T.e1z = b.y; T.e2x = b.z; T.e2y = b.w;
    T.e2z = c.x; T.prim = __float_as_uint(c.y); T.mesh = __float_as_uint(c.z); T.pad = 0;
    return T;
}

#define TMIN 0.01f     // deviceProgram.cu:41
#define TMAX 1e16f     // deviceProgram.cu:42

#define TRAV_DONE ((int)0x80000000)     // cur: traversal finished (never a valid leaf code: first_tri < 2^28)

// One ray per quad.  Everything that steers control flow (cur, sp, the quad-wide best distance) is
// identical in the four lanes; each lane keeps the best hit among the triangles IT tested and the
// four are merged once, at the end, by (t, primitive id) -- the same total order as a sequential scan.
//
// A node step sits on the wave's dependent chain (load -> test -> rank -> stack -> pop -> load), so it is kept short: the hit mask of the quad comes out of the wave
// ballot (one shift), every hit lane stores its child at stack[sp + H-1-rank] and the next node is
// simply popped -- descending and backtracking are the same code, no cross-lane selects.
__device__ inline uint32_t quad_rot1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x39, 0xf, 0xf, true); }   // [1,2,3,0]
__device__ inline uint32_t quad_rot2(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); }   // [2,3,0,1]
__device__ inline uint32_t quad_rot3(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x93, 0xf, 0xf, true); }   // [3,0,1,2]

enum { ROWB = FOVPT_TQUADS * 4,                 // byte distance of two stack rows
       ROWSHIFT = FOVPT_TQUADS == 16 ? 6 : FOVPT_TQUADS == 64 ? 8 : 10 };
static_assert(FOVPT_LEAF_MAX <= 4, "a leaf is tested in one quad step");
static_assert(FOVPT_TQUADS == 16 || FOVPT_TQUADS == 64 || FOVPT_TQUADS == 256, "row stride of the stack is 64, 256 or 1024 bytes");

// The traversal stack lives in LDS and is addressed through pointers that SAY so (address space 3, 32 bits): every access is a
// ds_read / ds_write by construction.  With a generic `char*` that held only as long as the compiler could infer the address
// space; a build in which it cannot (the stack pointer loaded back from memory, as with a run-time-indexed private array next
// to it -- round 2's faulting diagnostic build) would fall back to flat accesses on a 64-bit pointer.
typedef __attribute__((address_space(3))) char LdsChar;
typedef __attribute__((address_space(3))) int LdsInt;

struct QuadLane {                                   // per-lane constants of the quad traversal
    uint32_t j, qshift, from_me, j32, j3;
    int miss_rows;
    __device__ inline void init()
    {
        j = threadIdx.x & 3u;
        qshift = threadIdx.x & 60u;                 // first lane of this quad within its wave
        from_me = 15u & ~((1u << j) - 1u);          // lanes j..3 of the quad
        j32 = 32u * j; j3 = 3u * j;
        miss_rows = (int)(j + 1u) * ROWB;
    }
};
struct QuadTrav {                                   // state of one ray's traversal (identical in the 4 lanes except the best hit)
    int cur;                                        // node >= 0, leaf < 0, TRAV_DONE
    LdsChar* top;                                   // LDS byte address of the first free stack row
    float lim;                                      // closest: prunes boxes beyond the quad-wide best hit
    float bt, bu, bv; uint32_t bpos, bprim;         // best hit among the triangles THIS lane tested
#if FOVPT_V_STEPSTAT
    uint32_t steps;                                 // diagnostics: node steps | leaf steps << 16 of this ray
    unsigned long long tr_lo, tr_hi;                // node steps of the first 15 node phases, one byte each (tools/raysim.py)
    uint32_t n0;                                    // node steps in which no child was hit (top byte of the trace)
#endif
    // Row 0 holds the end marker, so "pop" needs no emptiness test; all row arithmetic stays in bytes.
    __device__ inline void start(int* stack, const QuadLane& q)
    {
        if (q.j == 0) stack[0] = TRAV_DONE;
        top = (LdsChar*)(LdsInt*)stack + ROWB;
        cur = 0; lim = TMAX;
        bt = INFINITY; bu = 0.f; bv = 0.f; bpos = 0xffffffffu; bprim = 0xffffffffu;
    }
};

#if FOVPT_V_STEPSTAT
__device__ inline void stepstat(unsigned long long* diag)
{
    const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(ex)) { atomicAdd(diag + 0, 1ull); atomicAdd(diag + 1, (unsigned long long)(__builtin_popcountll(ex) >> 2)); }
}
#define STEPSTAT(d) stepstat(d)
#else
#define STEPSTAT(d)
#endif

#if FOVPT_V_CYCLES
// Diagnostic build: s_memtime stamps (shader cycles) inside the steps of a SAMPLE of the waves -- wave 0 of the first 256
// blocks, about one wave per CU.  (Stamping every wave makes the launch 4-8 x slower: 8192 waves x 4 stamps per step is
// more than the timestamp path serves, and the time lands in whatever segment a wave happens to wait in.)  A stamp is tied
// to the registers whose arrival it marks (asm operands), so the compiler's own waits sit in front of it.  The sums are
// WAVE-level: a variable carried through the divergent step loops is per lane, so only the FIRST ACTIVE LANE of a step adds
// the step's times to its own registers (a select, no branch) and the lanes' sums are added up at the end; the stamp that
// ends a step travels to the next one through one LDS word.  FOVPT_V_CYCLES=2 adds histograms.
struct Cyc {
    bool on;
    uint32_t n_node, gap, load, alu, lds, n_leaf, lgap, lload, lrest;     // per LANE; only the first active lane of a step adds to its own
    __device__ inline void init(bool sampled) { on = sampled; n_node = gap = load = alu = lds = n_leaf = lgap = lload = lrest = 0u; }
};
__shared__ uint32_t s_cyc_last;                       // stamp at the end of the sampled wave's previous step (every active lane writes the same value)
__shared__ uint32_t s_cyc_hist[3 * 64];               // node load wait /16 | node step /32 | leaf step /32
__device__ inline uint32_t cyc_stamp()
{ unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return (uint32_t)t; }
template <typename A> __device__ inline uint32_t cyc_stamp(A& a)
{ unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(a) : : "memory"); return (uint32_t)t; }
template <typename A, typename B> __device__ inline uint32_t cyc_stamp(A& a, B& b)
{ unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(a), "+v"(b) : : "memory"); return (uint32_t)t; }
__device__ inline bool cyc_first_lane()
{ return (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)); }
__device__ inline void cyc_hist(uint32_t* h, uint32_t bin) { if (FOVPT_V_CYCLES >= 2) atomicAdd(h + min(bin, 63u), 1u); }
#define CYC_P , Cyc& C
#define CYC_A , C
#else
#define CYC_P
#define CYC_A
#endif

// The root and its children -- the first FOVPT_TOPN nodes of the breadth-first array, 640 bytes -- live in LDS beside a closest-hit
// wave's stack (4352 + 640 bytes: still eight waves per SIMD), and the first two steps of every ray of a round read them from
// there instead of through the CU's address path: all sixteen rays of a round start at the root together and most go on to one of
// its children, so the two steps are peeled in front of the loop (no per-step choice between the two memories).  Round 4,
// closest-hit launches -1.2 %, frames -1 % (atrium 0.676 -> 0.669, street 1.456 -> 1.439).  Occlusion rays
// start one by one (pool refills): for them a peeled step would be a pass of its own, and they keep the global path.
#ifndef FOVPT_V_TOPLDS
#define FOVPT_V_TOPLDS 1
#endif
#define FOVPT_TOPN 5
#if FOVPT_V_TOPLDS
__shared__ float4 s_top[FOVPT_TOPN * 8];
#endif

template <bool ANY_HIT>
__device__ inline void node_finish(const RayT& r, const QuadLane& q, QuadTrav& T, const float4& a, const float4& b CYC_P
#if FOVPT_V_CYCLES
                                   , uint32_t c0, uint32_t c1
#endif
);

// wide internal node: lane j owns child j
template <bool ANY_HIT, bool TOP = false>
__device__ inline void node_step(const SceneView& sc, const RayT& r, const QuadLane& q, QuadTrav& T CYC_P)
{
#if FOVPT_V_CYCLES
    uint32_t c0 = 0, c1 = 0;
    if (C.on) c0 = cyc_stamp(T.cur);
#endif
    // uniform base + 32-bit offset (fovpt_set_scene keeps nodes and triangles below 4 GB)
    const float4* np = (const float4*)((const char*)sc.nodes + (((uint32_t)T.cur << 7) | q.j32));
#if FOVPT_V_CYCLES
    float4 a = np[0], b = np[1];
    if (C.on) c1 = cyc_stamp(a.x, b.z);
#elif FOVPT_V_TOPLDS
    float4 a, b;
    if (TOP) {
        typedef __attribute__((address_space(3))) float LdsFloat;
        const LdsFloat* lp = (const LdsFloat*)((const LdsChar*)(LdsFloat*)(float*)s_top + (((uint32_t)T.cur << 7) | q.j32));
        a = make_float4(lp[0], lp[1], lp[2], lp[3]); b = make_float4(lp[4], lp[5], lp[6], lp[7]);
    } else { a = np[0]; b = np[1]; }
#else
    const float4 a = np[0], b = np[1];
#endif
    node_finish<ANY_HIT>(r, q, T, a, b CYC_A
#if FOVPT_V_CYCLES
                         , c0, c1
#endif
    );
}

// ... the rest of a node step, once the lane's child record (a, b) is there: box test, rank, push, pop
template <bool ANY_HIT>
__device__ inline void node_finish(const RayT& r, const QuadLane& q, QuadTrav& T, const float4& a, const float4& b CYC_P
#if FOVPT_V_CYCLES
                                   , uint32_t c0, uint32_t c1
#endif
)
{
#if FOVPT_V_CYCLES
    uint32_t c2 = 0;
#endif
    const int code = __float_as_int(b.z);
    float t;
    const bool h = box_hit(r, a.x, a.y, a.z, a.w, b.x, b.y, TMIN, T.lim, t);
    const uint32_t m4 = (uint32_t)(__builtin_amdgcn_ballot_w64(h) >> q.qshift) & 15u;
    int Hm1;                                    // H - 1 in one instruction (the compiler splits popcount - 1)
    asm("v_bcnt_u32_b32 %0, %1, -1" : "=v"(Hm1) : "v"(m4));
#if FOVPT_V_STEPSTAT
    if (m4 == 0u) T.n0++;
#endif
    // Every lane stores its child: the H hits land on rows top .. top+H-1 (the one to visit next
    // last), the misses on the free rows above them -- no branch, no select on the address.
    int row;                                    // in bytes, relative to top
    if (ANY_HIT && !FOVPT_V_ANYHIT_SORT) {
        // storage order (distance order was measured slower)
        row = ((__builtin_popcount(m4 & q.from_me) - 1) << ROWSHIFT) + (h ? 0 : q.miss_rows);
    } else {
        // front to back.  The key orders by entry distance (t >= TMIN > 0: the bit pattern is
        // monotonic) with the lane in the two lowest bits, so keys are distinct and below 2^31
        // (the sign of a difference is the comparison); misses sort first.  The visiting order
        // does not change the result, only the amount of pruning.
        uint32_t key;
        asm("v_and_or_b32 %0, %1, -4, %2" : "=v"(key) : "v"(h ? __float_as_uint(t) : 0u), "v"(__float_as_uint(b.w)));      // b.w: the child's tie rank 0..3 (BvhChild::pad)
        const int lt = ((int)(quad_rot1(key) - key) >> 31) + ((int)(quad_rot2(key) - key) >> 31) + ((int)(quad_rot3(key) - key) >> 31);
        row = (lt << ROWSHIFT) + 3 * ROWB;             // 3 - (number of keys below mine)
    }
#if FOVPT_V_CYCLES
    if (C.on) c2 = cyc_stamp(row, Hm1);
#endif
    *(LdsInt*)(T.top + row) = code;
    T.top += Hm1 * ROWB;                        // H pushed, one popped
    __builtin_amdgcn_wave_barrier();
    T.cur = *(const LdsInt*)T.top;
#if FOVPT_V_CYCLES
    if (C.on) {
        const uint32_t c3 = cyc_stamp(T.cur);
        const uint32_t last = s_cyc_last;
        const uint32_t f = cyc_first_lane() ? 0xffffffffu : 0u;
        C.n_node += f & 1u; C.gap += f & (c0 - last); C.load += f & (c1 - c0); C.alu += f & (c2 - c1); C.lds += f & (c3 - c2);
        if (FOVPT_V_CYCLES >= 2 && f) { cyc_hist(s_cyc_hist, (c1 - c0) >> 4); cyc_hist(s_cyc_hist + 64, (c3 - last) >> 5); }
        s_cyc_last = cyc_stamp();                      // (the bookkeeping itself stays out of the next step's gap)
    }
#endif
}

template <bool ANY_HIT>
__device__ inline bool leaf_finish(const RayT& r, const QuadLane& q, QuadTrav& T, const TriRec& R, uint32_t tri16 CYC_P
#if FOVPT_V_CYCLES
                                   , uint32_t c0, uint32_t c1
#endif
);

// leaf: lane j owns triangle j (branch-free: the four lanes of 16 rays never agree on an early out).
// A lane beyond the leaf's count repeats triangle 0: the duplicate candidate changes nothing.
// Any-hit: returns true when a front-facing triangle was hit (the ray is occluded, nothing is popped).
template <bool ANY_HIT>
__device__ inline bool leaf_step(const SceneView& sc, const RayT& r, const QuadLane& q, QuadTrav& T CYC_P)
{
#if FOVPT_V_CYCLES
    uint32_t c0 = 0, c1 = 0;
    if (C.on) c0 = cyc_stamp(T.cur);
#endif
    const uint32_t lcode = (uint32_t)~T.cur;
    const uint32_t tri16 = (lcode >> 3) + (q.j <= (lcode & 7u) ? q.j3 : 0u);     // in 16-byte units
#if FOVPT_V_CYCLES
    TriRec R = load_tri_off(sc.tris, tri16 << 4);
    if (C.on) c1 = cyc_stamp(R.v0x, R.e2z);
#define CYC_LEAF_END(x) if (C.on) { const uint32_t c2 = cyc_stamp(x); const uint32_t last = s_cyc_last; const uint32_t f = cyc_first_lane() ? 0xffffffffu : 0u; \
        C.n_leaf += f & 1u; C.lgap += f & (c0 - last); C.lload += f & (c1 - c0); C.lrest += f & (c2 - c1); \
        if (FOVPT_V_CYCLES >= 2 && f) cyc_hist(s_cyc_hist + 128, (c2 - last) >> 5); s_cyc_last = cyc_stamp(); }
#else
    const TriRec R = load_tri_off(sc.tris, tri16 << 4);
#define CYC_LEAF_END(x)
#endif
    return leaf_finish<ANY_HIT>(r, q, T, R, tri16 CYC_A
#if FOVPT_V_CYCLES
                                , c0, c1
#endif
    );
}

// ... the rest of a leaf step, once the lane's triangle record is there: Moeller-Trumbore, merge, pop
template <bool ANY_HIT>
__device__ inline bool leaf_finish(const RayT& r, const QuadLane& q, QuadTrav& T, const TriRec& R, uint32_t tri16 CYC_P
#if FOVPT_V_CYCLES
                                   , uint32_t c0, uint32_t c1
#endif
)
{
    const V3 d = v3(r.dx, r.dy, r.dz);
    const V3 e1 = v3(R.e1x, R.e1y, R.e1z), e2 = v3(R.e2x, R.e2y, R.e2z);
    const V3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = 1.0f / det;
    const V3 s = v3(r.ox, r.oy, r.oz) - v3(R.v0x, R.v0y, R.v0z);
    const float u = dot(s, p) * inv;
    const V3 qq = cross(s, e1);
    const float v = dot(d, qq) * inv;
    const float t = dot(e2, qq) * inv;
    // the contract's tests; "det != 0" and "u <= 1" are implied: with det == 0 u is +-inf or NaN
    // and then u >= 0 or u + v <= 1 fails; v >= 0 and fl(u + v) <= 1 give u <= 1
    const bool ok = (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > TMIN) & (t < TMAX);
    if (ANY_HIT) {
        // front face: counter-clockwise seen from the origin
        if ((uint32_t)(__builtin_amdgcn_ballot_w64(ok & (det > 0.0f)) >> q.qshift) & 15u) { CYC_LEAF_END(T.cur); return true; }
    } else {
        const bool better = ok & ((t < T.bt) | ((t == T.bt) & (R.prim < T.bprim)));
        T.bt = better ? t : T.bt; T.bu = better ? u : T.bu; T.bv = better ? v : T.bv;
        T.bpos = better ? tri16 : T.bpos; T.bprim = better ? R.prim : T.bprim;
        // quad-wide best: bt > 0 or +inf, so the bit patterns order like the values
        uint32_t m = __float_as_uint(T.bt);
        m = min(m, quad_rot2(m));
        m = min(m, quad_rot1(m));
        T.lim = fminf(TMAX, __uint_as_float(m) * 1.000001f);
    }
    T.top -= ROWB;
    T.cur = *(const LdsInt*)T.top;
    CYC_LEAF_END(T.cur);
    return false;
}

// A MIXED step (round 4): the pass the wave makes when its vote ends a node phase.  The rays waiting at a leaf test their
// triangles -- and the rays still at a node step theirs in the same pass, behind the SAME wait for memory: nodes and triangles
// live in one allocation, so every live lane fetches 48 bytes from one base register (a node lane's third 16 bytes are not
// used), then the node lanes and the leaf lanes finish their step one after the other.  The instructions are the ones two
// separate passes would issue; what goes is one memory round trip and the idling of the node lanes through a leaf step.
// Any-hit: returns true for a quad whose leaf step found an occluder.
template <bool ANY_HIT>
__device__ inline bool mixed_step(const SceneView& sc, const RayT& r, const QuadLane& q, QuadTrav& T, unsigned long long* diag)
{
    const bool at_node = T.cur >= 0;
    const uint32_t lcode = (uint32_t)~T.cur;
    const uint32_t tri16 = (lcode >> 3) + (q.j <= (lcode & 7u) ? q.j3 : 0u);
    const uint32_t off = at_node ? (((uint32_t)T.cur << 7) | q.j32) : sc.tri_off + (tri16 << 4);
    const float4* p = (const float4*)((const char*)sc.nodes + off);
    const float4 x0 = p[0], x1 = p[1], x2 = p[2];
    if (at_node) {
        STEPSTAT(diag);
#if FOVPT_V_CYCLES
        Cyc Cdummy; Cdummy.init(false);
        node_finish<ANY_HIT>(r, q, T, x0, x1, Cdummy, 0u, 0u);
#else
        node_finish<ANY_HIT>(r, q, T, x0, x1);
#endif
        return false;
    }
    STEPSTAT(diag + 2);
    TriRec R;
    R.v0x = x0.x; R.v0y = x0.y; R.v0z = x0.z; R.e1x = x0.w;
    R.e1y = x1.x; R.e1z = x1.y; R.e2x = x1.z; R.e2y = x1.w;
    R.e2z = x2.x; R.prim = __float_as_uint(x2.y); R.mesh = __float_as_uint(x2.z); R.pad = 0;
#if FOVPT_V_CYCLES
    Cyc Cdummy; Cdummy.init(false);
    return leaf_finish<ANY_HIT>(r, q, T, R, tri16, Cdummy, 0u, 0u);
#else
    return leaf_finish<ANY_HIT>(r, q, T, R, tri16);
#endif
}

// closest: store the hit record of the quad's ray AT THE RAY'S QUEUE POSITION (the shading kernel reads ray and
// hit side by side).  Lowest t (bit patterns of t > 0 order like the
// values), then lowest primitive id; lanes that tie on both hold the same triangle, hence the same
// record: they all store it.
__device__ inline void store_hit(const PathState& ps, uint32_t ph, const QuadTrav& T)
{
    uint32_t mt = __float_as_uint(T.bt);
    mt = min(mt, quad_rot2(mt));
    mt = min(mt, quad_rot1(mt));
    const bool cand = __float_as_uint(T.bt) == mt;
    uint32_t mp = cand ? T.bprim : 0xffffffffu;
    mp = min(mp, quad_rot2(mp));
    mp = min(mp, quad_rot1(mp));
    if (cand && T.bprim == mp) ps.hit[ph] = make_float4(T.bt, T.bu, T.bv, __uint_as_float(T.bpos));
}
// any-hit: the deferred NEE add of SampleLights / SampleShadow (deviceProgram.cu:323-341,367-385).
// Every (slot, depth) cell has exactly one writer, so this is a plain store and the shadow rays of a
// bounce may run at any time before resolve (own stream, see fovpt_api.hip)
__device__ inline void store_shadow(const PathState& ps, const ShadowQueue& sq, uint32_t ph, uint32_t slot, uint32_t target, bool occluded)
{
    const float4 val = occluded ? sq.val_occ[ph] : sq.val_vis[ph];
    float4* cell = target == 0xffffffffu ? ps.alpha + slot : ps.rad + ((size_t)slot * ps.stride + target);
    *cell = make_float4(val.x, val.y, val.z, 0.f);
}

// One ray per quad.  Everything that steers control flow (cur, top, the quad-wide best distance) is
// identical in the four lanes; each lane keeps the best hit among the triangles IT tested and the
// four are merged once, at the end, by (t, primitive id) -- the same total order as a sequential scan.
//
// A node step sits on the wave's dependent chain (load -> test -> rank -> stack -> pop -> load), so it is kept short: the hit mask of the quad comes out of the wave
// ballot (one shift), every lane stores its child at a row derived from its rank and the next node is
// simply popped -- descending and backtracking are the same code, no cross-lane selects.
// na live lanes of the wave, nn of them at a node: does the node phase end here?  (FOVPT_VOTE_C = 64: never -- the phases of rounds 1-3)
#ifndef FOVPT_VOTE_A
#define FOVPT_VOTE_A 2
#endif
#ifndef FOVPT_VOTE_B
#define FOVPT_VOTE_B 1
#endif
#ifndef FOVPT_VOTE_C
#define FOVPT_VOTE_C 1
#endif
__device__ inline bool vote_leaf(uint32_t na, uint32_t nn)
{
    return (na - nn) * (uint32_t)FOVPT_VOTE_A >= nn * (uint32_t)FOVPT_VOTE_B + 4u * (uint32_t)FOVPT_VOTE_C;
}
// (occlusion rays: the same rule with its own constants, for A/B)
#ifndef FOVPT_VOTE_AH_A
#define FOVPT_VOTE_AH_A FOVPT_VOTE_A
#endif
#ifndef FOVPT_VOTE_AH_B
#define FOVPT_VOTE_AH_B FOVPT_VOTE_B
#endif
#ifndef FOVPT_VOTE_AH_C
#define FOVPT_VOTE_AH_C FOVPT_VOTE_C
#endif
__device__ inline bool vote_leaf_anyhit(uint32_t na, uint32_t nn)
{
    return (na - nn) * (uint32_t)FOVPT_VOTE_AH_A >= nn * (uint32_t)FOVPT_VOTE_AH_B + 4u * (uint32_t)FOVPT_VOTE_AH_C;
}

__device__ inline void traverse_quad(const SceneView& sc, const RayT& r, int* __restrict__ stack /* [e * QUADS_PER_BLOCK] */, const QuadLane& q,
                              QuadTrav& T, unsigned long long* diag CYC_P)
{
    T.start(stack, q);
#if FOVPT_V_CYCLES
    if (C.on) s_cyc_last = cyc_stamp();
#endif
    // The wave VOTES when its node phase ends (round 4).  Rounds 1-3 ran node steps until EVERY ray of the wave had reached a leaf (or
    // was through): a ray at a leaf waited for the longest run of node steps among the other fifteen, and only 8.3 (atrium) / 5.8
    // (street) of the 16 rays took part in an average node step.  Now the phase ends as soon as the rays waiting at a leaf
    // outnumber those still stepping (vote_leaf: waiting x A >= stepping x B + C quads): a leaf step then serves the waiting rays,
    // the stepping ones sit it out and go on afterwards.  Leaf steps get emptier (9.3 -> 5.2 rays of 16, and 77 % more of them),
    // node steps fuller (8.3 -> 10.6, 22 % fewer): in node-step units (a leaf step costs 1.4) a wave's passes fall by 7 % on the
    // atrium and by 27 % on the street.  The vote rides on what the loop computes anyway -- the lanes that go on are the loop's
    // own exit mask, one s_bcnt1 and six scalar instructions -- after a first form with a ballot per state and a branch per pass
    // cost 12-15 % per pass and lost on the atrium (EXPERIMENTS.md).  Order of steps never changes a result.
#if FOVPT_V_STEPSTAT
    // (no arrays with a run-time index here: a diagnostic build whose traversal kernel used scratch memory faulted with
    // HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the closest-hit launch; the product kernels use none)
    uint32_t my_nodes = 0u, my_leaves = 0u;          // this ray's own steps (tools/stepcount.py, raystat.py); the per-phase trace is not kept any more
    T.n0 = 0u; T.tr_lo = T.tr_hi = 0ull;
#define RAYSTAT(x) (x)++
#else
#define RAYSTAT(x)
#endif
#if FOVPT_V_TOPLDS
    // every ray of a round starts at the root, and most go on to one of its children: those two steps out of LDS
    if (sc.num_nodes >= FOVPT_TOPN) {
        STEPSTAT(diag);
        node_step<false, true>(sc, r, q, T CYC_A);
        RAYSTAT(my_nodes);
        if (T.cur >= 0 && T.cur < FOVPT_TOPN) { STEPSTAT(diag); node_step<false, true>(sc, r, q, T CYC_A); RAYSTAT(my_nodes); }
    }
#endif
    while (T.cur != TRAV_DONE) {                                                                         // (rays that are through leave)
        const uint32_t na = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));            // live lanes of the wave
        while (T.cur >= 0) {
            STEPSTAT(diag);
            node_step<false>(sc, r, q, T CYC_A);
            RAYSTAT(my_nodes);
            const uint32_t nn = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(T.cur >= 0));   // of the lanes that stepped: who goes on
            if (vote_leaf(na, nn)) break;                                                                  // (wave-uniform)
        }
#if FOVPT_V_MIXED
        if (T.cur != TRAV_DONE) {
#if FOVPT_V_STEPSTAT
            if (T.cur >= 0) my_nodes++; else my_leaves++;
#endif
            mixed_step<false>(sc, r, q, T, diag);
        }
#else
        if (T.cur < 0 && T.cur != TRAV_DONE) { STEPSTAT(diag + 2); leaf_step<false>(sc, r, q, T CYC_A); RAYSTAT(my_leaves); }
#endif
    }
#if FOVPT_V_STEPSTAT
    T.steps = min(my_nodes, 4095u) | (min(my_leaves, 255u) << 12) | (min(T.n0, 63u) << 20);
#endif
#undef RAYSTAT
}

// Any-hit traversal over a POOL of shadow rays [first, end) owned by one wave: a quad that has finished
// its ray takes the next one of the pool as soon as FOVPT_REFILL quads of the wave are idle (all of them
// at the end), so the wave does not wait for its longest ray after every 16 -- occlusion rays end after
// very different numbers of steps (SIMD utilisation 36 % with static rounds).  The pool is private to
// the wave: no atomics.  (For closest-hit rays the same scheme was measured 10 % slower: the divergent
// refill -- merge and store the hit, fetch, set up -- costs more than their better lane use returns.)
#ifndef FOVPT_REFILL
#define FOVPT_REFILL 6
#endif
__device__ inline void traverse_shadow_pool(const SceneView& sc, const PathState& ps, const ShadowQueue& sq, const ShardMap& map, uint32_t cap,
                                            uint32_t first, uint32_t end, int* __restrict__ stack, const QuadLane& q, unsigned long long* diag CYC_P)
{
    QuadTrav T;
    T.start(stack, q);
    T.cur = TRAV_DONE;
#if FOVPT_V_CYCLES
    if (C.on) s_cyc_last = cyc_stamp();
#endif
    RayT r = {};
    uint32_t ph = 0;                              // physical index of the quad's shadow record
    bool pending = false, occluded = false;
    uint32_t next = first;
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(T.cur == TRAV_DONE);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle) >> 2;
        if (n_idle == 16u || (n_idle >= (uint32_t)FOVPT_REFILL && next < end)) {
            if (T.cur == TRAV_DONE) {
                if (pending && q.j == 0) store_shadow(ps, sq, ph, __float_as_uint(sq.o[ph].w), __float_as_uint(sq.d[ph].w), occluded);
                pending = false;
                // idle quads below mine (all four lanes of an idle quad are set in the mask)
                const uint32_t rank = (__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u)) - q.j) >> 2;
                const uint32_t idx = next + rank;
                if (idx < end) {
                    ph = map.phys16(idx, next, cap);
                    ray_setup(r, sq.o[ph], sq.d[ph]);
                    T.start(stack, q);
                    pending = true; occluded = false;
                }
            }
            next = min(end, next + n_idle);
#if FOVPT_V_CYCLES
            if (C.on) s_cyc_last = cyc_stamp();
#endif
            if (n_idle == 16u && __builtin_amdgcn_ballot_w64(T.cur != TRAV_DONE) == 0ull) return;      // pool exhausted, all results stored
        }
#if FOVPT_V_VOTE_ANYHIT
        // the same vote as for closest-hit rays (traverse_quad); the live lanes are the quads that hold a ray
        const uint32_t na = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(T.cur != TRAV_DONE));
        while (T.cur >= 0) {
            STEPSTAT(diag);
            node_step<true>(sc, r, q, T CYC_A);
            const uint32_t nn = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(T.cur >= 0));
            if (vote_leaf_anyhit(na, nn)) break;
        }
#if FOVPT_V_MIXED_ANYHIT
        if (T.cur != TRAV_DONE) {
            if (mixed_step<true>(sc, r, q, T, diag)) { occluded = true; T.cur = TRAV_DONE; }
        }
#else
        if (T.cur < 0 && T.cur != TRAV_DONE) {
            STEPSTAT(diag + 2);
            if (leaf_step<true>(sc, r, q, T CYC_A)) { occluded = true; T.cur = TRAV_DONE; }
        }
#endif
#else
        while (T.cur >= 0) { STEPSTAT(diag); node_step<true>(sc, r, q, T CYC_A); }
        if (T.cur != TRAV_DONE) {
            STEPSTAT(diag + 2);
            if (leaf_step<true>(sc, r, q, T CYC_A)) { occluded = true; T.cur = TRAV_DONE; }
        }
#endif
    }
}


// One traversal launch handles the occlusion rays of iteration it_shadow and/or the closest-hit rays of
// iteration it_closest, one ray per QUAD of lanes (16 rays per wave).
// (Dynamic work fetching with a global counter and per-lane replacement was measured and rejected: with
// so few rays per resident lane per launch a returning atomic per wave costs more than the imbalance.)
// (MODE 0: a closest-hit launch, 1: an occlusion launch, 2: both in one -- a build of the kernel for each, so that a wave of a
// closest-hit launch neither fetches the occlusion queue's shard sizes before it starts nor carries that loop's registers.)
template <int MODE>
__global__ __launch_bounds__(FOVPT_TBLOCK, FOVPT_V_WAVES) void k_traverse(SceneView sc, PathState ps, RayQueue queue, ShadowQueue sq,
                                                                         uint32_t cap, Counters* __restrict__ cnt, int it_closest, int it_shadow, uint32_t sel)
{
    __shared__ int s_stack[(FOVPT_STACK + 4) * FOVPT_TQUADS];  // + the end marker and three rows of slack above the top
    if (MODE == 0) it_shadow = -1;
    if (MODE == 1) it_closest = -1;
    ShardMap ms, mq;
    if (MODE != 0) ms.load(cnt, FOVPT_CNT_SQ(it_shadow >= 0 ? it_shadow : 0), sel, cap);
    if (MODE != 1) mq.load(cnt, FOVPT_CNT_Q(it_closest >= 0 ? it_closest : 0), sel, cap);
    const uint32_t n_sh = (MODE != 0 && it_shadow >= 0) ? ms.total() : 0u;
    const uint32_t n_cl = (MODE != 1 && it_closest >= 0) ? mq.total() : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_cl) atomicAdd(&cnt->stat_radiance, (unsigned long long)n_cl);
        if (n_sh) atomicAdd(&cnt->stat_shadow, (unsigned long long)n_sh);
        if (it_closest == 0) atomicAdd(&cnt->stat_paths, (unsigned long long)n_cl);
    }
    QuadLane q;
    q.init();
    int* stack = s_stack + (threadIdx.x >> 2);
#if FOVPT_V_TOPLDS
    if (MODE != 1 && sc.num_nodes >= FOVPT_TOPN) {
        for (uint32_t k = threadIdx.x; k < FOVPT_TOPN * 8u; k += FOVPT_TBLOCK) s_top[k] = ((const float4*)sc.nodes)[k];
        __syncthreads();
    }
#endif
#if FOVPT_V_CYCLES
    if (threadIdx.x < 192) s_cyc_hist[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_cyc_last = 0u;
    __syncthreads();
    Cyc C;
    // the sample: wave 0 of ~256 workgroups spread evenly over the grid (the first workgroups of a launch get the most rounds)
    const uint32_t cyc_every = gridDim.x >= 512u ? gridDim.x / 256u : 1u;
    C.init(FOVPT_V_CYCLES != 3 && __builtin_amdgcn_readfirstlane((blockIdx.x % cyc_every == 0u && threadIdx.x < 64u) ? 1 : 0) != 0);   // 3: wave start / end times only
    const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();      // every wave: when it starts and ends (100 MHz)
    const uint32_t life0 = cyc_stamp();
#endif
    // occlusion rays: every wave owns one contiguous pool
    if (n_sh) {
        const uint32_t nwaves = gridDim.x * (FOVPT_TBLOCK / 64), wave = blockIdx.x * (FOVPT_TBLOCK / 64) + (threadIdx.x >> 6);
        const uint32_t per = (n_sh + nwaves - 1u) / nwaves, first = min(n_sh, wave * per);
        traverse_shadow_pool(sc, ps, sq, ms, cap, first, min(n_sh, first + per), stack, q, cnt->diag[1] CYC_A);
    }
    // closest-hit rays: static grid-stride over quads, 16 consecutive rays per wave and round
    const uint32_t quads = gridDim.x * FOVPT_TQUADS;
    for (uint32_t i = blockIdx.x * FOVPT_TQUADS + (threadIdx.x >> 2); i < n_cl; i += quads) {
        const uint32_t i0 = __builtin_amdgcn_readfirstlane(i - ((threadIdx.x & 63u) >> 2));      // the wave's 16 rays: i0 .. i0+15
        RayT r;
        QuadTrav T;
        const uint32_t ph = mq.phys16(i, i0, cap);
        const float4 o = queue.o[ph], d = queue.d[ph];
        ray_setup(r, o, d);
        traverse_quad(sc, r, stack, q, T, cnt->diag[0] CYC_A);
        store_hit(ps, ph, T);
#if FOVPT_V_STEPSTAT
        if (q.j == 0) {                                                   // tools/raystat.py, raytrace_dump.py
            ((uint32_t*)&queue.d[ph])[3] = T.steps;
            ps.trace[ph] = make_uint4((uint32_t)T.tr_lo, (uint32_t)(T.tr_lo >> 32), (uint32_t)T.tr_hi, (uint32_t)(T.tr_hi >> 32));
        }
#endif
    }
#if FOVPT_V_CYCLES
    {
        const uint32_t life1 = cyc_stamp();
        const unsigned long long real1 = __builtin_amdgcn_s_memrealtime();
        const int kind = it_shadow >= 0 ? 1 : 0, itn = (it_shadow >= 0 ? it_shadow : it_closest) & 3;
        const uint32_t wave = blockIdx.x * (FOVPT_TBLOCK / 64) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63u) == 0u && wave < 32768u) { cnt->wtime[kind * 4 + itn][wave][0] = real0; cnt->wtime[kind * 4 + itn][wave][1] = real1; }
        if (C.on) {
            unsigned long long* g = cnt->cyc[kind][itn];
            const uint32_t v[9] = {C.n_node, C.gap, C.load, C.alu, C.lds, C.n_leaf, C.lgap, C.lload, C.lrest};
#pragma unroll
            for (int k = 0; k < 9; k++) if (v[k]) atomicAdd(g + k, (unsigned long long)v[k]);
            if ((threadIdx.x & 63u) == 0u) {
                const uint32_t x = cyc_stamp(), y = cyc_stamp();      // two stamps back to back: what a stamp costs
                atomicAdd(g + 9, (unsigned long long)(y - x)); atomicAdd(g + 10, 1ull);
                atomicAdd(g + 11, (unsigned long long)(life1 - life0)); atomicAdd(g + 12, (unsigned long long)(real1 - real0)); atomicAdd(g + 13, 1ull);
            }
        }
        if (FOVPT_V_CYCLES >= 2 && C.on && threadIdx.x < 64u)
            for (uint32_t k = threadIdx.x; k < 192u; k += 64u) if (s_cyc_hist[k]) atomicAdd(&cnt->hist[kind][itn][k >> 6][k & 63u], s_cyc_hist[k]);
    }
#endif
}

// ---- shade -------------------------------------------------------------------------------
#define FLAG_DONE 1u
#define FLAG_SECONDARY 2u

// (float)c / 255.0f for c = 0..255 without the division sequence: one Newton step on q = c * fl(1/255)
// with fused multiply-adds gives the correctly rounded quotient for every one of the 256 inputs
// (checked exhaustively: test_unorm8_device_matches_division, FOVPT_OP_UNORM8)
__device__ inline float unorm8(uint32_t c)
{
    const float f = (float)c, r = 1.0f / 255.0f;
    const float q = f * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, f), r, q);
}
__device__ inline float4 tex_unpack(uint32_t p)
{
    return make_float4(unorm8(p & 255u), unorm8((p >> 8) & 255u), unorm8((p >> 16) & 255u), unorm8(p >> 24));
}
// floor-mod of a texel coordinate (wrap addressing); a power-of-two size needs no division
__device__ inline int tex_wrap(int x, int n)
{
    if ((n & (n - 1)) == 0) return x & (n - 1);
    x %= n;
    return x < 0 ? x + n : x;
}
// bilinear, wrap, normalized coordinates (the fp32 contract standing in for tex2D<float4>, :664)
__device__ inline float4 tex2d(const TexDev& T, float u, float v)
{
    const float x = u * (float)T.w - 0.5f, y = v * (float)T.h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float fx = x - fx0, fy = y - fy0;
    const int x0 = tex_wrap((int)fmaxf(-1.0e9f, fminf(1.0e9f, fx0)), T.w), y0 = tex_wrap((int)fmaxf(-1.0e9f, fminf(1.0e9f, fy0)), T.h);
    const int x1 = x0 + 1 == T.w ? 0 : x0 + 1, y1 = y0 + 1 == T.h ? 0 : y0 + 1;
    const uint32_t* r0 = T.px + (size_t)y0 * T.w;
    const uint32_t* r1 = T.px + (size_t)y1 * T.w;
    const float4 c00 = tex_unpack(r0[x0]), c10 = tex_unpack(r0[x1]), c01 = tex_unpack(r1[x0]), c11 = tex_unpack(r1[x1]);
    const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    return make_float4(w00 * c00.x + w10 * c10.x + w01 * c01.x + w11 * c11.x,
                       w00 * c00.y + w10 * c10.y + w01 * c01.y + w11 * c11.y,
                       w00 * c00.z + w10 * c10.z + w01 * c01.z + w11 * c11.z,
                       w00 * c00.w + w10 * c10.w + w01 * c01.w + w11 * c11.w);
}

#define FLAG_ALPHA_ONE 4u      // prd.alpha = make_float3(1) happened (deviceProgram.cu:689)
#define FLAG_ALPHA_SET 8u      // prd.alpha += ... happened on a shadow catcher (:693): ps.alpha[slot] holds it

#ifndef FOVPT_V_SHADEWAVES
#define FOVPT_V_SHADEWAVES 1
#endif
// EXTRA: the build of the kernel that also knows the opt-in extensions of fovpt_config.options (sky radiance for escaped
// secondary rays, Russian roulette); the default build carries none of their code or registers.
template <bool EXTRA>
__global__ __launch_bounds__(FOVPT_BLOCK, FOVPT_V_SHADEWAVES) void k_shade(const FrameDev fd, SceneView sc, PathState ps,
                                                       RayQueue queue_in, RayQueue queue_out,
                                                       ShadowQueue sq, uint32_t cap, Counters* __restrict__ cnt, int depth_iter, uint32_t sel)
{
    __shared__ uint32_t s_scratch[14];
    ShardMap mq;
    mq.load(cnt, FOVPT_CNT_Q(depth_iter), sel, cap);
    const uint32_t n = mq.total();
    const uint32_t nround = (n + FOVPT_BLOCK - 1) / FOVPT_BLOCK * FOVPT_BLOCK;
    for (uint32_t i = blockIdx.x * FOVPT_BLOCK + threadIdx.x; i < nround; i += gridDim.x * FOVPT_BLOCK) {
        bool want_shadow = false, want_next = false;
        uint32_t slot = 0;
        V3 next_o = v3(0.f), next_d = v3(0.f);
        float next_pdf = 0.f;          // prd.bsdfPdf of the sample that sends the next ray (read by FOVPT_OPT_SKY_MISS only)
        float4 sh_o, sh_d, sh_vis, sh_occ;
        sh_o = sh_d = sh_vis = sh_occ = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const uint32_t ph = mq.phys(i, cap);
            const float4 o4 = queue_in.o[ph], d4 = queue_in.d[ph];
            slot = __float_as_uint(o4.w);
            const float4 hit = ps.hit[ph];
            const uint32_t tpos = __float_as_uint(hit.w);
            uint4 rs = ps.rng[slot];
            uint32_t flags = rs.z & 0xffu;
            int depth = (int)(rs.z >> 8);
            if (tpos == 0xffffffffu) {
                // __miss__radiance :253-282: DONE; nothing is added for this segment (:515 breaks first)
                flags |= FLAG_DONE;
                if (EXTRA && (fd.options & FOVPT_OPT_SKY_MISS) && (flags & FLAG_SECONDARY) && depth < fd.max_depth) {
                    // the block the reference carries commented out (:259-269): MIS counterpart of SampleLights; the
                    // segment is counted (one more radiance cell), see include/fovpt.h
                    const V3 dir = v3(d4);
                    const float bsdfPdf = d4.w;                                            // pdf of the sample that sent this ray
                    const V3 thr = v3(ps.thr[slot]);
                    float u, v;
                    probe_dir_to_uv(dir, u, v);
                    const fovpt_probe& pr = fd.probe;                                      // ProbePdf, Probe.cuh:69-93
                    const int col = max(0, min((int)(u * pr.width), pr.width - 1));
                    const int row = max(0, min((int)(v * pr.height), pr.height - 1));
                    float skyPdf = pr.pdfValuesX[row * fd.probe_row_mul * pr.width + col] * pr.pdfValuesY[row];
                    const float sinTheta = fovpt_dm_sinf(v * kPi);
                    if (fabsf(sinTheta) < 0.0001f) skyPdf = 0.0f;
                    else skyPdf *= float(pr.width) * float(pr.height) / (2.0f * kPi * kPi * sinTheta);
                    const float weight = 0.5f * bsdfPdf / (0.5f * bsdfPdf + 0.5f * skyPdf);
                    const V3 rad = v3(0.f) + (weight * v3(probe_eval(pr, fd.probe_row_mul, u, v))) * thr;
                    ps.rad[(size_t)slot * ps.stride + depth] = f4(rad, 0.f);
                    depth += 1;
                }
            } else {
                const V3 ray_o = v3(o4), ray_dir = v3(d4);
                // ProbeSample is the first consumer of the path's random numbers on a shaded hit (:303-344) and
                // depends on nothing else: its chain of dependent loads (guide tables, CDFs, texel) is started
                // here, beside the chain hit -> triangle -> mesh -> texels.  Branches that do not shade drop it.
                Rng rng_probe; rng_probe.s1 = rs.x; rng_probe.s2 = rs.y;
                V3 wi, skyColor; float skyPdf;
                probe_sample(fd.probe, fd.guide_x, fd.guide_y, fd.probe_rec, fd.probe_row_mul, wi, skyColor, skyPdf, rng_probe);
                const float4 thr_in = ps.thr[slot];                            // (unused garbage until the first shaded hit wrote it)
                const TriRec T = load_tri_off(sc.tris, tpos << 4);            // tpos: offset in 16-byte units
                const MeshDev M = sc.meshes[T.mesh];
                const Mat& mat = M.material;
                const V3 e1 = v3(T.e1x, T.e1y, T.e1z), e2 = v3(T.e2x, T.e2y, T.e2z);
                const V3 N_0 = normalize(cross(e1, e2));                                   // :632
                const V3 wo = neg(ray_dir);
                const V3 N = N_0 * copysignf(1.0f, dot(wo, N_0));                          // faceforward :634
                const V3 P = ray_o + hit.x * ray_dir;                                      // :638
                const bool catcher = (mat.flags & FOVPT_MATERIAL_FLAG_SHADOW_CATCHER) != 0;
                if (catcher && (flags & FLAG_SECONDARY)) {
                    // :646-651: pass straight through, depth unchanged after the loop's ++depth;
                    // the loop adds prd.radiance == 0 to direct/indirect, which changes nothing
                    next_o = P; next_d = ray_dir; next_pdf = d4.w;
                    want_next = true;
                } else if (depth >= fd.max_depth) {
                    // the reference's discarded last segment (:515).  It is only traced here when the
                    // scene holds a shadow catcher; a catcher hit is always a pass-through at this
                    // depth (SECONDARY is set), so this is a plain hit whose one lasting effect is :689
                    flags |= FLAG_ALPHA_ONE | FLAG_DONE;
                } else {
                    // pathThroughput (1,1,1) and rayEta 1 until the first shaded hit (:447-449)
                    const float4 t4 = (flags & FLAG_SECONDARY) ? thr_in : make_float4(1.f, 1.f, 1.f, 1.0f);
                    V3 thr = v3(t4);
                    float rayEta = t4.w;
                    Rng rng = rng_probe;
                    V3 albedo = v3(mat.color);
                    if (M.texture_id >= 0 && M.has_texcoord) {                             // :655-670
                        const float2* tc = sc.tri_tc + (size_t)T.prim * 3;
                        const float2 t0 = tc[0], t1 = tc[1], t2 = tc[2];
                        const float w0 = 1.f - hit.y - hit.z;
                        const float tcx = (w0 * t0.x + hit.y * t1.x) + hit.z * t2.x;
                        const float tcy = (w0 * t0.y + hit.y * t1.y) + hit.z * t2.y;
                        albedo = v3(tex2d(M.tex, tcx, tcy));
                    }
                    if (ps.guide_n && depth == 0 && (flags & FLAG_SECONDARY) == 0) {       // :509-512, :653-654
                        ps.guide_n[slot] = f4(N, 0.f);
                        ps.guide_a[slot] = f4(albedo, 0.f);
                    }
                    float outEta;
                    if (rayEta == 1.0f)                                                    // :673-683
                        outEta = (mat.eta == 0.0f) ? 2.0f / (1.0f - sqrtf(0.08f * mat.specular)) - 1.0f : mat.eta;
                    else
                        outEta = 1.0f;
                    // ---- SampleLights / SampleShadow :303-387 with the occlusion test deferred
                    const BsdfView view = bsdf_view(mat, albedo, rayEta, outEta, N, wo);
                    V3 sum_hit = v3(0.0f);        // value of `sum` on the branch that evaluates the BSDF
                    {
                        const float bsdfPdf = bsdf_pdf(mat, view, N, wo, wi);
                        const V3 f = bsdf_eval(mat, albedo, view, N, wo, wi);
                        if (bsdfPdf > 0.0f) {
                            const float weight = 0.5f * skyPdf / (0.5f * bsdfPdf + 0.5f * skyPdf);
                            if (weight > 0.0f) {
                                const V3 val = div_vs(weight * skyColor * f * fabsf(dot(wi, N)), skyPdf) * (1.0f / 1.f);
                                sum_hit = sum_hit + val;
                            }
                        }
                    }
                    const V3 sum_zero = v3(0.0f);
                    V3 rad_vis, rad_occ;          // prd.radiance after this hit if the shadow ray is un/occluded
                    V3 alpha_vis = v3(0.f), alpha_occ = v3(0.f);
                    bool alpha_set_one = false;
                    if (!catcher) {                                                        // :686-690
                        rad_vis = v3(0.f) + thr * sum_hit;
                        rad_occ = v3(0.f) + thr * sum_zero;
                        alpha_set_one = true;
                    } else {                                                               // :691-694 SampleShadow
                        rad_vis = v3(0.f); rad_occ = v3(0.f);
                        alpha_vis = thr * sum_zero;
                        alpha_occ = thr * sum_hit;
                    }
                    if ((flags & FLAG_SECONDARY) == 0) {                                   // :696-698
                        rad_vis = rad_vis + v3(mat.emission);
                        rad_occ = rad_occ + v3(mat.emission);
                    }
                    V3 bu, bv;
                    basis_from_vector(N, bu, bv);
                    V3 bsdfDir = v3(0.f);
                    const float bsdfPdf = bsdf_sample(mat, view, bu, bv, N, wo, bsdfDir, rng);            // :706
                    if (alpha_set_one) flags |= FLAG_ALPHA_ONE;                            // :689 (kept even when DONE)
                    const bool same = rad_vis.x == rad_occ.x && rad_vis.y == rad_occ.y && rad_vis.z == rad_occ.z
                                   && alpha_vis.x == alpha_occ.x && alpha_vis.y == alpha_occ.y && alpha_vis.z == alpha_occ.z;
                    if (catcher) {
                        // alpha += thr * shadowSample happens regardless of what follows (:693); alpha is still
                        // (0,0,0) here because only a primary hit gets this far on a catcher
                        flags |= FLAG_ALPHA_SET;
                        if (same) ps.alpha[slot] = f4(v3(0.f) + alpha_occ, 0.f);
                        else {
                            want_shadow = true;
                            sh_o = f4(P, __uint_as_float(slot)); sh_d = f4(wi, __uint_as_float(0xffffffffu));
                            sh_vis = f4(v3(0.f) + alpha_vis, 0.f); sh_occ = f4(v3(0.f) + alpha_occ, 0.f);
                        }
                    }
                    if (bsdfPdf <= 0.0f) {                                                 // :708-711
                        flags |= FLAG_DONE;       // radiance of this hit is dropped by the break at :515
                    } else {
                        // the segment counts: prd.radiance becomes the depth-th term of directLight (depth 0) /
                        // indirectLight (:522-527).  One cell per (slot, depth), summed in order by resolve.
                        float4* cell = ps.rad + ((size_t)slot * ps.stride + depth);
                        if (!catcher && !same) {
                            want_shadow = true;
                            sh_o = f4(P, __uint_as_float(slot)); sh_d = f4(wi, __uint_as_float((uint32_t)depth));
                            sh_vis = f4(rad_vis, 0.f); sh_occ = f4(rad_occ, 0.f);
                        } else {
                            *cell = f4(rad_occ, 0.f);
                        }
                        flags |= FLAG_SECONDARY;
                        depth += 1;                                                        // :529
                        // The reference traces once more at depth == max_depth and throws the result
                        // away (:515).  Without a shadow catcher in the scene that segment cannot
                        // change anything (alpha is already 1), so it is not traced -- and then nothing reads the
                        // throughput, the medium or the random numbers of this path any more (:714-724 skipped).
                        if (depth < fd.max_depth || sc.any_catcher) {
                            const V3 f = bsdf_eval(mat, albedo, view, N, wo, bsdfDir);              // :714
                            if (dot(bsdfDir, N) <= 0.0f) rayEta = outEta;                  // :717-721
                            thr = thr * div_vs(f * fabsf(dot(N, bsdfDir)), bsdfPdf);       // :724
                            bool rr_kill = false;
                            if (EXTRA && (fd.options & FOVPT_OPT_RUSSIAN_ROULETTE) && depth >= 2) {   // the //!TODO of :518-520, see include/fovpt.h
                                const float q = fmaxf(0.05f, fminf(1.0f, fmaxf(thr.x, fmaxf(thr.y, thr.z))));
                                if (rng.randf() >= q) rr_kill = true;
                                else thr = thr * (1.0f / q);
                            }
                            if (!rr_kill) {
                                next_o = P; next_d = bsdfDir; next_pdf = bsdfPdf;
                                ps.thr[slot] = f4(thr, rayEta);
                                want_next = true;
                            }
                        }
                    }
                    rs.x = rng.s1; rs.y = rng.s2;
                }
            }
            rs.z = flags | ((uint32_t)depth << 8);
            ps.rng[slot] = rs;
        }
        // ---- wavefront-ballot compaction into the next queues
        uint32_t spos, qpos;
        if (fd.partition)                                                          // (block-uniform)
            block_append2d(cnt, FOVPT_CNT_SQ(depth_iter), want_shadow, FOVPT_CNT_Q(depth_iter + 1), want_next, next_d.y > 0.0f, cap, s_scratch, spos, qpos, sel);
        else
            block_append2(cnt, FOVPT_CNT_SQ(depth_iter), want_shadow, FOVPT_CNT_Q(depth_iter + 1), want_next, cap, s_scratch, spos, qpos, sel);
        if (want_shadow) { sq.o[spos] = sh_o; sq.d[spos] = sh_d; sq.val_vis[spos] = sh_vis; sq.val_occ[spos] = sh_occ; }
        if (want_next) { queue_out.o[qpos] = f4(next_o, __uint_as_float(slot)); queue_out.d[qpos] = f4(next_d, next_pdf); }
    }
}

// ---- resolve -----------------------------------------------------------------------------
__device__ inline V3 reinhard(const V3& color, float white)                    // :126-131
{
    const float luminance = 0.2126f * color.x + 0.7152f * color.y + 0.0722f * color.z;
    return div_vs(color * 1.0f, 1.0f + luminance / white);
}
__device__ inline float srgb1(float c)                                          // cuda/helpers.h:35-43
{
    const float invGamma = 1.0f / 2.4f;
    const float powed = fovpt_dm_powf(c, invGamma);
    return c < 0.0031308f ? 12.92f * c : 1.055f * powed - 0.055f;
}
__device__ inline uint32_t quant8(float x)                                      // cuda/helpers.h:50-55
{
    x = clampf(x, 0.0f, 1.0f);
    return min((uint32_t)(x * 256.0f), 255u);
}
__device__ inline uint32_t make_color(const V3& c)                              // cuda/helpers.h:57-62
{
    const V3 cc = clamp3(c, 0.0f, 1.0f);
    return quant8(srgb1(cc.x)) | (quant8(srgb1(cc.y)) << 8) | (quant8(srgb1(cc.z)) << 16) | (255u << 24);
}

// candidate launch-index range along one axis for pixel coordinate x (resolve)
// `wrap` (>= 0 only on the clamped edge): launch indices 0 .. wrap have a "negative" pixel index, which in the
// reference's unsigned arithmetic is a huge one and is clamped onto this edge too (deviceProgram.cu:433, :554).
// They only ever pass the ring test when the gaze point itself is such a wrapped coordinate.
__device__ inline void writer_range(uint32_t x, uint32_t frame_dim, uint32_t factor, int fill, uint32_t off, uint32_t grid, long long& lo, long long& hi,
                                    long long& wrap)
{
    // r = x - (int32)off, a = max(0, ceil((r - (fill-1)) / f)), b = floor(r / f) (or grid-1 on the clamped edge).
    // Integer division is a long software routine on the GPU: power-of-two factors (1, 2, 4 in every
    // pass the reference launches) shift, everything else that fits uses 32-bit division.
    const long long r = (long long)x - (long long)(int32_t)off;
    const uint32_t f = factor ? factor : 1u;
    const bool pow2 = (f & (f - 1u)) == 0u;
    const int sh = 31 - __clz((int)f);
    const long long num = r - (long long)(fill - 1);          // a = ceil(num / f)
    long long a;
    if (num <= 0) a = 0;
    else if (num < 0x7fffffffll) a = pow2 ? (long long)(((uint32_t)num + f - 1u) >> sh) : (long long)(((uint32_t)num + f - 1u) / f);
    else a = (num + f - 1) / (long long)f;
    long long b;
    if (x + 1 == frame_dim) b = (long long)grid - 1;  // clamp at :554 folds everything beyond the edge onto it
    else {
        if (r < 0) b = -1;
        else if (r < 0x7fffffffll) b = pow2 ? (long long)((uint32_t)r >> sh) : (long long)((uint32_t)r / f);
        else b = r / (long long)f;
        if (b > (long long)grid - 1) b = (long long)grid - 1;
    }
    lo = a; hi = b;
    wrap = -1;
    if (x + 1 == frame_dim && (int32_t)off < 0) {
        const long long neg = -(long long)(int32_t)off;                       // index of launch l is l*f - neg
        long long cnt = (neg + (long long)f - 1) / (long long)f;              // l*f < neg
        if (cnt > (long long)grid) cnt = grid;
        wrap = cnt - 1;
        if (wrap >= lo) wrap = lo - 1;                                        // (already part of [lo, hi])
    }
}

// The last writer of pixel (x, y) in the reference's launch order: the highest pass that covers it (F over M over P),
// within a pass the launch index that comes last (ascending y, then x) among those whose clamped block fill reaches the
// pixel (deviceProgram.cu:546-554) and that pass the ring test (:433-440).  Only launch rows [row0, row1) of a pass count
// (a chunk of a large launch).  Used by the resolve and by the multi-GPU gather plan.
__device__ inline bool find_last_writer(const FrameDev& fd, uint32_t x, uint32_t y, int& wp, uint32_t& wlx, uint32_t& wly)
{
#pragma unroll
    for (int p = FOVPT_MAX_PASSES - 1; p >= 0; p--) {          // static indices: the pass records stay in SGPRs
        if (p >= fd.npass) continue;
        const PassDev& P = fd.pass[p];
        if (P.fill <= 0) continue;
        // The common case first (round 4; the general search below is ~800 instructions per pixel, most of a resolve's time):
        // blocks that tile the plane -- fill == factor, a power of two, as in every pass the reference launches -- have, away
        // from the frame's last column and row (where the clamp at :554 folds launches onto the edge), exactly ONE candidate
        // per axis: a = ceil((r - f + 1) / f) = floor(r / f) = b.
        const uint32_t f = P.fx;
        if ((uint32_t)P.fill == f && P.fy == f && (f & (f - 1u)) == 0u && x + 1u != (uint32_t)fd.w && y + 1u != (uint32_t)fd.h) {
            const int sh = 31 - __clz((int)f);
            const long long rx = (long long)x - (long long)(int32_t)P.offx, ry = (long long)y - (long long)(int32_t)P.offy;
            if (rx < 0 || ry < 0 || rx >= ((long long)P.gw << sh) || ry >= ((long long)P.row1 << sh)) continue;
            const uint32_t lx = (uint32_t)rx >> sh, ly = (uint32_t)ry >> sh;
            uint32_t ix, iy;
            if (ly < P.row0 || !ring_alive(fd, P, lx, ly, ix, iy)) continue;
            wp = p; wlx = lx; wly = ly;
            return true;
        }
        long long xa, xb, ya, yb, xw, yw;
        writer_range(x, (uint32_t)fd.w, P.fx, P.fill, P.offx, P.gw, xa, xb, xw);
        writer_range(y, (uint32_t)fd.h, P.fy, P.fill, P.offy, P.gh, ya, yb, yw);
        if (ya < (long long)P.row0) ya = P.row0;           // only this chunk's launch rows
        if (yb > (long long)P.row1 - 1) yb = (long long)P.row1 - 1;
        if (yw > (long long)P.row1 - 1) yw = (long long)P.row1 - 1;
        const long long y_end = yw >= (long long)P.row0 ? (long long)P.row0 : ya, x_end = xw >= 0 ? 0 : xa;
        // candidates in descending launch order: [ya, yb] then the wrapped rows [row0, yw]; same along x
        for (long long ly = yb; ly >= y_end; ly--) {
            if (ly < ya && ly > yw) { ly = yw + 1; continue; }
            for (long long lx = xb; lx >= x_end; lx--) {
                if (lx < xa && lx > xw) { lx = xw + 1; continue; }
                uint32_t ix, iy;
                if (!ring_alive(fd, P, (uint32_t)lx, (uint32_t)ly, ix, iy)) continue;
                wp = p; wlx = (uint32_t)lx; wly = (uint32_t)ly;
                return true;
            }
        }
    }
    return false;
}

// Resolve as a tiled LDS reduction.  A block owns a 64 x 4 pixel tile.
//   A. every pixel thread finds its last writer in the reference's launch order (gather)
//   B. runs of pixels with the same writer elect a leader; leaders are ballot-compacted into a tile-local
//      work list in LDS (a 4x4 periphery block is 16 pixels but ONE colour)
//   C. the first n threads of the block each reduce one list entry: ordered sum over the launch index's
//      sample slots and bounce cells, alpha, backplate mix, exposure, Reinhard, sRGB -> LDS
//   D. every pixel thread fetches its colour from LDS and writes float4 accum + rgba8, coalesced
//   E. (block 0) the job's queue counters are zeroed for the next job that uses this state set: resolve
//      is the last kernel of a job and nothing in it reads them
__global__ __launch_bounds__(FOVPT_BLOCK) void k_resolve(const FrameDev fd, PathState ps, Counters* __restrict__ cnt)
{
    if (!FOVPT_V_STEPSTAT && blockIdx.x == 0 && blockIdx.y == 0) {     // (a diagnostic build keeps them for tools/raystat.py)
        uint32_t* w = &cnt->shard[0][0];
        for (uint32_t i = threadIdx.x; i < FOVPT_SHARDS * FOVPT_SHARD_STRIDE; i += FOVPT_BLOCK) w[i] = 0u;
    }
    __shared__ uint32_t s_key[FOVPT_BLOCK];        // launch record id of the pixel's writer
    __shared__ uint32_t s_idx[FOVPT_BLOCK];        // leader thread -> tile list position
    __shared__ uint32_t s_list_li[FOVPT_BLOCK];    // tile list: launch index ...
    __shared__ uint32_t s_list_p[FOVPT_BLOCK];     // ... and pass
    __shared__ float4 s_accum[FOVPT_BLOCK];        // result per list entry: accum_color (pre-blend)
    __shared__ uint32_t s_rgba[FOVPT_BLOCK];       // result per list entry: tone-mapped pixel
    __shared__ float4 s_gn[FOVPT_BLOCK], s_ga[FOVPT_BLOCK];   // denoiser guides per list entry
    __shared__ uint32_t s_wave[4];

    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * 64 + tx;
    const uint32_t y = blockIdx.y * 4 + ty;
    const bool inside = x < (uint32_t)fd.w && y < (uint32_t)fd.h;
    const uint32_t image_index = y * (uint32_t)fd.w + x;

    // ---- A: last writer of this pixel
    int state = 0;                 // 0 nobody, 1 a launch index this rank owns, 2 another rank's
    int wp = 0;
    uint32_t wli = 0, key = 0xffffffffu;
    if (inside) {
        uint32_t wlx, wly;
        if (find_last_writer(fd, x, y, wp, wlx, wly)) {
            const PassDev& P = fd.pass[wp];
            state = launch_owned(fd, wp, wlx, wly) ? 1 : 2;
            wli = (wly - P.row0) * P.gw + wlx;
            key = P.launch_base + wli;
        }
    }
    if (state != 1) key = 0xffffffffu;
    s_key[threadIdx.x] = key;
    __syncthreads();

    // ---- B: run leaders -> tile list
    const bool leader = state == 1 && (tx == 0 || s_key[threadIdx.x - 1] != key);
    const unsigned long long lm = __ballot(leader);
    if (tx == 0) s_wave[ty] = (uint32_t)__popcll(lm);
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < ty; w++) base += s_wave[w];
    const uint32_t nlist = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (leader) {
        const uint32_t pos = base + (uint32_t)__popcll(lm & ((1ull << tx) - 1ull));
        s_idx[threadIdx.x] = pos;
        s_list_li[pos] = wli;
        s_list_p[pos] = (uint32_t)wp;
    }
    __syncthreads();

    // ---- C: one thread per distinct writer in the tile
    if (threadIdx.x < nlist) {
        const PassDev& P = fd.pass[s_list_p[threadIdx.x]];
        const uint32_t li = s_list_li[threadIdx.x];
        const uint32_t s0 = P.slot_base + li * P.spp;
        V3 result = v3(0.0f), alpha = v3(0.0f), gnorm = v3(0.0f), galb = v3(0.0f);
        for (uint32_t s = 0; s < P.spp; s++) {                                     // :536-537, in sample order
            const uint32_t slot = s0 + s;
            if (ps.guide_n) { gnorm = gnorm + v3(ps.guide_n[slot]); galb = galb + v3(ps.guide_a[slot]); }   // :510-511
            const float4* cells = ps.rad + (size_t)slot * ps.stride;                 // one 16*D-byte record per slot
            // the path's `depth` segments counted, cell dd holds prd.radiance of segment dd; the reference
            // adds nothing for a segment it never reached
            const uint32_t st = ps.rng[slot].z;
            const int nseg = min((int)(st >> 8), fd.max_depth);
            const V3 direct = v3(0.0f) + (nseg > 0 ? v3(cells[0]) : v3(0.0f));          // :523
            V3 indirect = v3(0.0f);
            for (int dd = 1; dd < nseg; dd++)                                          // :526, in bounce order
                indirect = indirect + v3(cells[dd]);
            result = result + (direct + indirect);
            alpha = alpha + ((st & FLAG_ALPHA_ONE) ? v3(1.0f) : (st & FLAG_ALPHA_SET) ? v3(ps.alpha[slot]) : v3(0.0f));
        }
        const float sppf = (float)P.spp;
        { const float inv = 1.0f / sppf; alpha = alpha * inv; gnorm = gnorm * inv; galb = galb * inv; }   // :541-543
        const V3 backplate = v3(ps.backplate[P.launch_base + li]);
        const V3 color = (backplate * sppf) * sub_sv(1.0f, alpha) + result;        // :558
        const V3 accum_color = div_vs(color, sppf);                                // :560
        s_gn[threadIdx.x] = f4(gnorm, 1.0f); s_ga[threadIdx.x] = f4(galb, 1.0f);
        s_accum[threadIdx.x] = f4(accum_color, 1.0f);
        s_rgba[threadIdx.x] = make_color(reinhard(accum_color * 16.0f, 1.0f));     // :586,597
    }
    __syncthreads();

    // ---- D: write the tile
    if (!inside) return;
    if (state == 1) {
        uint32_t t = threadIdx.x;
        while (!(t == ty * 64 || s_key[t - 1] != key)) t--;                        // start of this pixel's run
        const uint32_t e = s_idx[t];
        const PassDev& P = fd.pass[wp];
        float4 a = s_accum[e];
        uint32_t rgba = s_rgba[e];
        if (fd.accumulate && P.subframe > 0 && !P.redraw) {
            // PT_sv4_vmv2/deviceProgram.cu:545-553: clamp + running mean against THIS pixel's history
            V3 accum_color = clamp3(v3(a), 0.0f, 10.0f);
            const float alpha_value = 1.0f / (float)(P.subframe + 1);
            const fovpt_float4 pv = fd.accum_prev[image_index];
            accum_color = lerp3(v3(pv.x, pv.y, pv.z), accum_color, alpha_value);
            a = f4(accum_color, 1.0f);
            rgba = make_color(reinhard(accum_color * 16.0f, 1.0f));
        }
        fd.accum[image_index] = fovpt_float4{a.x, a.y, a.z, 1.0f};                 // :582
        fd.frame[image_index] = rgba;
        if (ps.guide_n) {                                                          // :612-614
            const float4 gn = s_gn[e], ga = s_ga[e];
            if (fd.g_normal) fd.g_normal[image_index] = fovpt_float4{gn.x, gn.y, gn.z, 1.0f};
            if (fd.g_color) fd.g_color[image_index] = fovpt_float4{a.x, a.y, a.z, 1.0f};
            if (fd.g_albedo) fd.g_albedo[image_index] = fovpt_float4{ga.x, ga.y, ga.z, 1.0f};
        }
    } else if (state == 2 || fd.zero_holes) {
        // another rank's pixel (or, for a whole frame on a rank other than 0, nobody's): zero keeps the sum-gather exact
        fd.accum[image_index] = fovpt_float4{0.f, 0.f, 0.f, 0.f};
        fd.frame[image_index] = 0u;
    }
}

// ---- multi-GPU: packed gather of the owned pixels -------------------------------------------------
// Every pixel of a frame has one last writer (find_last_writer) and that launch index has one owning rank
// (launch_owned: interleaved 8x4 launch-index tiles, the scheme of sutil/WorkDistribution.h:47-84), so the pixels of a
// frame partition by owner.  The PLAN lists, rank by rank, the pixel indices a rank owns in ascending order; every rank
// builds the same plan from the same launch parameters.  Per frame a rank packs its owned rgba8 words into a contiguous
// buffer (k_gather_pack), RCCL gathers the buffers onto rank 0, and rank 0 scatters them into the frame
// (k_gather_unpack): 1/N of the bytes of a full-frame reduce per rank.
#define FOVPT_PLAN_NOBODY 255u
__device__ inline uint32_t launch_owner_rank(const FrameDev& fd, int p, uint32_t lx, uint32_t ly)
{
    if (fd.world <= 1) return 0u;
    const uint32_t tx = lx / (uint32_t)fd.tile_w, ty = ly / (uint32_t)fd.tile_h;
    return (tx + 3u * ty + fd.pass[p].frame_pass) % (uint32_t)fd.world;
}
// owner[pixel] and, per block of 256 consecutive pixels, the number of pixels every rank owns
__global__ __launch_bounds__(FOVPT_BLOCK) void k_plan_owner(const FrameDev fd, uint8_t* __restrict__ owner, uint32_t* __restrict__ block_count)
{
    __shared__ uint32_t s_cnt[64];
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t npix = (uint32_t)fd.w * (uint32_t)fd.h;
    const uint32_t i = blockIdx.x * FOVPT_BLOCK + threadIdx.x;
    uint32_t o = FOVPT_PLAN_NOBODY;
    if (i < npix) {
        const uint32_t y = i / (uint32_t)fd.w, x = i - y * (uint32_t)fd.w;
        int wp; uint32_t lx, ly;
        if (find_last_writer(fd, x, y, wp, lx, ly)) o = launch_owner_rank(fd, wp, lx, ly);
        owner[i] = (uint8_t)o;
    }
    if (o != FOVPT_PLAN_NOBODY) atomicAdd(&s_cnt[o], 1u);
    __syncthreads();
    if ((int)threadIdx.x < fd.world) block_count[(size_t)blockIdx.x * fd.world + threadIdx.x] = s_cnt[threadIdx.x];
}
// exclusive scan of the block counts, rank by rank: ONE BLOCK per rank, every thread scans a contiguous chunk of the counts,
// the chunk totals are scanned in LDS, the chunks are written back with their offsets.  (The plan is rebuilt whenever the gaze
// moves -- every frame with an eye tracker -- so this is not a one-off: a single thread per rank walking 8-18 k blocks was.)
__global__ __launch_bounds__(1024) void k_plan_scan(uint32_t nblocks, int world, uint32_t* __restrict__ block_count, uint32_t* __restrict__ total)
{
    __shared__ uint32_t s_part[1024];
    const int r = (int)blockIdx.x;
    if (r >= world) return;
    const uint32_t per = (nblocks + blockDim.x - 1u) / blockDim.x;
    const uint32_t b0 = min(nblocks, threadIdx.x * per), b1 = min(nblocks, b0 + per);
    uint32_t sum = 0u;
    for (uint32_t b = b0; b < b1; b++) sum += block_count[(size_t)b * world + r];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1u; off < blockDim.x; off <<= 1) {                    // inclusive Hillis-Steele scan of the chunk totals
        const uint32_t v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;                                   // exclusive offset of my chunk
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t c = block_count[(size_t)b * world + r];
        block_count[(size_t)b * world + r] = run;
        run += c;
    }
    if (threadIdx.x == blockDim.x - 1u) total[r] = s_part[threadIdx.x];
}
// idx[rank_base[o] + block offset + position among the block's pixels of the same owner] = pixel
__global__ __launch_bounds__(FOVPT_BLOCK) void k_plan_fill(uint32_t npix, int world, const uint8_t* __restrict__ owner, const uint32_t* __restrict__ block_off,
                                                          const uint32_t* __restrict__ rank_base, uint32_t* __restrict__ idx)
{
    // position of a pixel among the block's pixels of the same owner, in ascending pixel order: within a wave one ballot per
    // DISTINCT owner present (64 consecutive pixels span a handful of 8-pixel-wide tiles), the waves' counts meet in LDS.
    // (Was: every thread scanning the up to 255 owners before it.)
    __shared__ uint32_t s_cnt[FOVPT_BLOCK / 64][64];
    const uint32_t i = blockIdx.x * FOVPT_BLOCK + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t o = i < npix ? owner[i] : FOVPT_PLAN_NOBODY;
    for (uint32_t k = threadIdx.x; k < (FOVPT_BLOCK / 64) * 64; k += FOVPT_BLOCK) (&s_cnt[0][0])[k] = 0u;
    __syncthreads();
    uint32_t in_wave = 0u;
    unsigned long long todo = __ballot(o != FOVPT_PLAN_NOBODY);
    while (todo) {                                                             // wave-uniform loop over the owners present
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)o, (int)leader);
        const unsigned long long same = __ballot(o == lo);
        if (o == lo) in_wave = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (lane == leader) s_cnt[wave][lo] = (uint32_t)__popcll(same);
        todo &= ~same;
    }
    __syncthreads();
    if (o == FOVPT_PLAN_NOBODY) return;
    uint32_t before = in_wave;
    for (uint32_t w = 0; w < wave; w++) before += s_cnt[w][o];
    idx[rank_base[o] + block_off[(size_t)blockIdx.x * world + o] + before] = i;
}
__global__ void k_gather_pack(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ frame, uint32_t* __restrict__ packed)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) packed[i] = frame[idx[i]];
}
// gathered: `world` buffers of `stride` words; rank r's pixels are idx[base[r] .. base[r+1])
__global__ void k_gather_unpack(int world, uint32_t stride, const uint32_t* __restrict__ base, const uint32_t* __restrict__ idx,
                                const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame)
{
    const uint32_t total = base[world];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int r = 0;
        while (r + 1 < world && i >= base[r + 1]) r++;
        frame[idx[i]] = gathered[(size_t)r * stride + (i - base[r])];
    }
}

// ---- probe guide tables (built once per setProbe) ------------------------------------------
// guide[seg * (n+2) + m] = lower_bound(cdf[seg*n .. seg*n+n), fl(m * fl(1/n))) - seg*n, m = 0..n+1
__global__ void k_build_guide(const float* __restrict__ cdf, int n, int segments, uint32_t* __restrict__ guide)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)segments * (n + 2);
    if (i >= total) return;
    const int seg = (int)(i / (n + 2)), m = (int)(i - (size_t)seg * (n + 2));
    const float value = (float)m * (1.0f / (float)n);                 // w_m of lower_bound_guided
    guide[i] = (uint32_t)(lower_bound(cdf, seg * n, (seg + 1) * n, value) - seg * n);
}

// packed per-texel records for probe_sample: {cdfX, pdfX, r, g, b, 0, 0, 0}
__global__ void k_probe_records(size_t n, const float* __restrict__ cdfX, const float* __restrict__ pdfX, const float4* __restrict__ data, float4* __restrict__ rec)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = data[i];
    rec[2 * i] = make_float4(cdfX[i], pdfX[i], c.x, c.y);
    rec[2 * i + 1] = make_float4(c.z, 0.f, 0.f, 0.f);
}

// ---- ProbeData::BuildCDF on the device (Probe.h:29-77) -----------------------------------------
// fp32 sums are not associative and the reference accumulates strictly left to right, so the
// parallelism is across rows only: one thread sums one row in order (out of LDS tiles that the
// wave loaded coalesced); a single thread then walks the row totals.  Same bits as the host helper
// fovpt_probe_build_cdf.
#define CDF_ROWS 8          // rows per block: a 2048-row probe gives 256 blocks, one per CU
#define CDF_COLS 256        // columns per tile = threads per block
#define CDF_LD (CDF_COLS + 4)   // row pitch in LDS: 16-byte reads of CDF_ROWS different rows fall into different banks
__global__ __launch_bounds__(CDF_COLS) void k_cdf_rows(int w, int h, const float4* __restrict__ data, float* __restrict__ pdfX, float* __restrict__ cdfX,
                                                      float* __restrict__ row_total)
{
    // A block = CDF_ROWS rows, walked in tiles of CDF_COLS columns.  All four waves read the tile's texels COALESCED (a wave
    // = 64 consecutive texels of one row, 1 KB) and park their luminances in LDS; lane r of wave 0 then adds up row r's
    // CDF_COLS values in order -- the fp32 sum of a row stays strictly left to right, which is what fixes its bits
    // (Probe.h:41-51) -- and all waves write weights and running sums back out, coalesced.  The next tile's texels are in
    // flight while the sums run.  (Was: one thread per row reading its row with a stride of w texels, every load a cache line
    // of its own, 32 waves on the whole chip: 1.8 ms for a 4096 x 2048 probe.)
    __shared__ __attribute__((aligned(16))) float s_w[CDF_ROWS][CDF_LD], s_c[CDF_ROWS][CDF_LD];
    const int j0 = blockIdx.x * CDF_ROWS, t = threadIdx.x;
    float run = 0.0f;                                                           // totalWeightX of row j0 + t so far (t < CDF_ROWS)
    float4 c[CDF_ROWS];
#pragma unroll
    for (int r = 0; r < CDF_ROWS; r++) c[r] = (j0 + r < h && t < w) ? data[(size_t)(j0 + r) * w + t] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0; i0 < w; i0 += CDF_COLS) {
#pragma unroll
        for (int r = 0; r < CDF_ROWS; r++) s_w[r][t] = c[r].x * 0.3f + c[r].y * 0.6f + c[r].z * 0.1f;      // Luminance, maths.h:165-168
        __syncthreads();
        const int in = i0 + CDF_COLS + t;
#pragma unroll
        for (int r = 0; r < CDF_ROWS; r++) c[r] = (j0 + r < h && in < w) ? data[(size_t)(j0 + r) * w + in] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < CDF_ROWS && j0 + t < h) {
            const int n = min(CDF_COLS, w - i0);
            int k = 0;
            for (; k + 4 <= n; k += 4) {
                const float4 v = *(const float4*)&s_w[t][k];
                float4 o;
                run += v.x; o.x = run; run += v.y; o.y = run; run += v.z; o.z = run; run += v.w; o.w = run;
                *(float4*)&s_c[t][k] = o;
            }
            for (; k < n; k++) { run += s_w[t][k]; s_c[t][k] = run; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CDF_ROWS; r++) {
            const int j = j0 + r, i = i0 + t;
            if (j < h && i < w) { pdfX[(size_t)j * w + i] = s_w[r][t]; cdfX[(size_t)j * w + i] = s_c[r][t]; }
        }
        __syncthreads();
    }
    if (t < CDF_ROWS && j0 + t < h) row_total[j0 + t] = run;
}
// second pass: pdf and cdf of every row times 1 / its total (the reference multiplies by the reciprocal, Probe.h:49-55)
__global__ void k_cdf_rows_scale(int w, int h, float* __restrict__ pdfX, float* __restrict__ cdfX, const float* __restrict__ row_total)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (size_t)w * h) return;
    const float invTotalWeightX = 1.0f / row_total[k / (size_t)w];
    pdfX[k] *= invTotalWeightX;
    cdfX[k] *= invTotalWeightX;
}
// (the running sum over the rows is sequential by contract -- one thread; the two divisions per row are not)
__global__ __launch_bounds__(256) void k_cdf_cols(int h, const float* __restrict__ row_total, float* __restrict__ pdfY, float* __restrict__ cdfY)
{
    // chunks of the row totals go through LDS: the block loads a chunk, thread 0 adds it up in order, the block writes the
    // running sums out.  (The one sequential thread used to read and write global memory value by value: 60 ns per row.)
    __shared__ __attribute__((aligned(16))) float s_v[2048], s_c[2048];
    __shared__ float s_total;
    if (blockIdx.x != 0) return;
    float totalWeightY = 0.0f;                                                  // thread 0's
    for (int j0 = 0; j0 < h; j0 += 2048) {
        const int n = min(2048, h - j0);
        for (int k = (int)threadIdx.x; k < n; k += (int)blockDim.x) s_v[k] = row_total[j0 + k];
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = 0;
            for (; k + 4 <= n; k += 4) {
                const float4 v = *(const float4*)&s_v[k];
                float4 o;
                totalWeightY += v.x; o.x = totalWeightY; totalWeightY += v.y; o.y = totalWeightY;
                totalWeightY += v.z; o.z = totalWeightY; totalWeightY += v.w; o.w = totalWeightY;
                *(float4*)&s_c[k] = o;
            }
            for (; k < n; k++) { totalWeightY += s_v[k]; s_c[k] = totalWeightY; }
        }
        __syncthreads();
        for (int k = (int)threadIdx.x; k < n; k += (int)blockDim.x) { pdfY[j0 + k] = s_v[k]; cdfY[j0 + k] = s_c[k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) s_total = totalWeightY;
    __syncthreads();
    const float total = s_total;
    for (int j = (int)threadIdx.x; j < h; j += (int)blockDim.x) {              // divisions here, not reciprocals (Probe.h:68-72)
        cdfY[j] /= total;
        pdfY[j] /= total;
    }
}

// ---- device self-test --------------------------------------------------------------------
__global__ void k_math(int op, const float* a, const float* b, float* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.f;
    switch (op) {
    case FOVPT_OP_SIN: r = fovpt_dm_sinf(a[i]); break;
    case FOVPT_OP_COS: r = fovpt_dm_cosf(a[i]); break;
    case FOVPT_OP_ACOS: r = fovpt_dm_acosf(a[i]); break;
    case FOVPT_OP_ATAN2: r = fovpt_dm_atan2f(a[i], b[i]); break;
    case FOVPT_OP_LOG: r = fovpt_dm_logf(a[i]); break;
    case FOVPT_OP_POW: r = fovpt_dm_powf(a[i], b[i]); break;
    case FOVPT_OP_SQRT: r = sqrtf(a[i]); break;
    case FOVPT_OP_DIV: r = a[i] / b[i]; break;
    case FOVPT_OP_RSQRTD: r = rcp_of_sqrt_as_the_reference(a[i]); break;
    case FOVPT_OP_HALFPLUS: r = half_plus_as_the_reference(a[i]); break;
    case FOVPT_OP_UNORM8: r = unorm8((uint32_t)a[i] & 255u); break;
    }
    out[i] = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------
void fovpt_launch_generate(hipStream_t st, const FrameDev& fd, PathState ps, RayQueue queue0, uint32_t cap, Counters* cnt, uint32_t slot_begin,
                           uint32_t slot_end, int grid, uint32_t sel)
{
    // a rank of a tile-sharded frame walks its own tiles only (one chain: a half-frame chain is a range of sample slots).
    // Its index space is padded (tiles beyond a row's end, launch indices beyond the grid's edge): a queue shard holds
    // slots / 8 + 512 entries and receives every eighth block iteration, so the space must not be larger than the slots
    // themselves -- it is not, except for grids a few launch indices wide, which go the other way.
    unsigned long long space = 0;
    for (int p = 0; p < fd.npass && p < FOVPT_MAX_PASSES; p++) {
        OwnedSpace o;
        o.init(fd, fd.pass[p]);
        space += ((unsigned long long)o.tile_rows * o.per_row * o.tile_lis * fd.pass[p].spp + FOVPT_BLOCK - 1u) / FOVPT_BLOCK * FOVPT_BLOCK;   // (64-bit: o.count may have wrapped)
    }
    if (FOVPT_V_GEN_OWNED && fd.world > 1 && sel == 0u && slot_begin == 0u && slot_end == fd.total_slots && space <= (unsigned long long)fd.total_slots)
        hipLaunchKernelGGL(k_generate_owned, dim3(grid), dim3(FOVPT_BLOCK), 0, st, fd, ps, queue0, cap, cnt);
    else
        hipLaunchKernelGGL(k_generate, dim3(grid), dim3(FOVPT_BLOCK), 0, st, fd, ps, queue0, cap, cnt, slot_begin, slot_end, sel);
}
void fovpt_launch_traverse(hipStream_t st, SceneView sc, PathState ps, RayQueue queue, ShadowQueue sq, uint32_t cap,
                           Counters* cnt, int it_closest, int it_shadow, int grid, hipEvent_t done, uint32_t sel)
{
    // `done` rides on the kernel's own completion signal (no separate marker packet in the queue)
    // (`grid` counts blocks of 256 threads, as for the other kernels; the traversal block may be larger)
    int blocks = grid * FOVPT_BLOCK / FOVPT_TBLOCK > 0 ? grid * FOVPT_BLOCK / FOVPT_TBLOCK : 1;
    if (it_shadow < 0) {
        // a closest-hit launch reads at most 8 * cap rays (the shards' capacities): no more workgroups than it can have rounds of
        // FOVPT_TQUADS rays (what matters for the small frames of a 1/N shard)
        const unsigned long long rounds = ((sel ? 4ull : 8ull) * cap + FOVPT_TQUADS - 1) / FOVPT_TQUADS;
        if ((unsigned long long)blocks > rounds) blocks = (int)((rounds + FOVPT_SHARDS - 1) / FOVPT_SHARDS * FOVPT_SHARDS);
    }
    auto kernel = it_shadow < 0 ? k_traverse<0> : it_closest < 0 ? k_traverse<1> : k_traverse<2>;
    if (done) hipExtLaunchKernelGGL(kernel, dim3(blocks), dim3(FOVPT_TBLOCK), 0, st, nullptr, done, 0, sc, ps, queue, sq, cap, cnt, it_closest, it_shadow, sel);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(FOVPT_TBLOCK), 0, st, sc, ps, queue, sq, cap, cnt, it_closest, it_shadow, sel);
}
void fovpt_launch_shade(hipStream_t st, const FrameDev& fd, SceneView sc, PathState ps, RayQueue queue_in, RayQueue queue_out,
                        ShadowQueue sq, uint32_t cap, Counters* cnt, int depth, int grid, hipEvent_t done, uint32_t sel)
{
    if (fd.options) {
        if (done) hipExtLaunchKernelGGL(k_shade<true>, dim3(grid), dim3(FOVPT_BLOCK), 0, st, nullptr, done, 0, fd, sc, ps, queue_in, queue_out, sq, cap, cnt, depth, sel);
        else hipLaunchKernelGGL(k_shade<true>, dim3(grid), dim3(FOVPT_BLOCK), 0, st, fd, sc, ps, queue_in, queue_out, sq, cap, cnt, depth, sel);
    } else {
        if (done) hipExtLaunchKernelGGL(k_shade<false>, dim3(grid), dim3(FOVPT_BLOCK), 0, st, nullptr, done, 0, fd, sc, ps, queue_in, queue_out, sq, cap, cnt, depth, sel);
        else hipLaunchKernelGGL(k_shade<false>, dim3(grid), dim3(FOVPT_BLOCK), 0, st, fd, sc, ps, queue_in, queue_out, sq, cap, cnt, depth, sel);
    }
}
void fovpt_launch_resolve(hipStream_t st, const FrameDev& fd, PathState ps, Counters* cnt, hipEvent_t done)
{
    dim3 grid((fd.w + 63) / 64, (fd.h + 3) / 4);
    if (done) hipExtLaunchKernelGGL(k_resolve, grid, dim3(FOVPT_BLOCK), 0, st, nullptr, done, 0, fd, ps, cnt);
    else hipLaunchKernelGGL(k_resolve, grid, dim3(FOVPT_BLOCK), 0, st, fd, ps, cnt);
}
void fovpt_launch_plan_owner(hipStream_t st, const FrameDev& fd, uint8_t* owner, uint32_t* block_count, uint32_t nblocks)
{
    hipLaunchKernelGGL(k_plan_owner, dim3(nblocks), dim3(FOVPT_BLOCK), 0, st, fd, owner, block_count);
}
void fovpt_launch_plan_scan_fill(hipStream_t st, uint32_t npix, uint32_t nblocks, int world, const uint8_t* owner, uint32_t* block_count,
                                 uint32_t* total, const uint32_t* rank_base, uint32_t* idx, int phase)
{
    if (phase == 0) hipLaunchKernelGGL(k_plan_scan, dim3(world), dim3(1024), 0, st, nblocks, world, block_count, total);
    else hipLaunchKernelGGL(k_plan_fill, dim3(nblocks), dim3(FOVPT_BLOCK), 0, st, npix, world, owner, block_count, rank_base, idx);
}
void fovpt_launch_gather_pack(hipStream_t st, uint32_t n, const uint32_t* idx, const uint32_t* frame, uint32_t* packed)
{
    if (n) hipLaunchKernelGGL(k_gather_pack, dim3((n + 255u) / 256u < 2048u ? (n + 255u) / 256u : 2048u), dim3(256), 0, st, n, idx, frame, packed);
}
void fovpt_launch_gather_unpack(hipStream_t st, int world, uint32_t stride, uint32_t total, const uint32_t* base, const uint32_t* idx,
                                const uint32_t* gathered, uint32_t* frame)
{
    if (total) hipLaunchKernelGGL(k_gather_unpack, dim3((total + 255u) / 256u < 2048u ? (total + 255u) / 256u : 2048u), dim3(256), 0, st, world, stride, base, idx, gathered, frame);
}
void fovpt_launch_build_guide(hipStream_t st, const float* cdf, int n, int segments, uint32_t* guide)
{
    const size_t total = (size_t)segments * (n + 2);
    hipLaunchKernelGGL(k_build_guide, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cdf, n, segments, guide);
}
void fovpt_launch_probe_records(hipStream_t st, size_t n, const float* cdfX, const float* pdfX, const float4* data, float4* rec)
{
    hipLaunchKernelGGL(k_probe_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, cdfX, pdfX, data, rec);
}
void fovpt_launch_build_cdf(hipStream_t st, int w, int h, const float4* data, float* pdfX, float* cdfX, float* pdfY, float* cdfY, float* row_total)
{
    hipLaunchKernelGGL(k_cdf_rows, dim3((h + CDF_ROWS - 1) / CDF_ROWS), dim3(CDF_COLS), 0, st, w, h, data, pdfX, cdfX, row_total);
    hipLaunchKernelGGL(k_cdf_rows_scale, dim3((unsigned)(((size_t)w * h + 255) / 256)), dim3(256), 0, st, w, h, pdfX, cdfX, row_total);
    hipLaunchKernelGGL(k_cdf_cols, dim3(1), dim3(256), 0, st, h, row_total, pdfY, cdfY);
}
void fovpt_launch_math(hipStream_t st, int op, const float* a, const float* b, float* out, size_t n)
{
    hipLaunchKernelGGL(k_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, a, b, out, n);
}
