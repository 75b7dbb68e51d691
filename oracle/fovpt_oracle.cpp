/*
 * fovpt_oracle.cpp -- CPU restatement of the reference's foveated path-tracing launch.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libfovpt, the python package, the C++
 * shim) links, imports or executes this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and there only as the checker / the timed CPU baseline.
 *
 * PINNED WHERE THE REFERENCE COMPILES, UNPINNED WHERE IT DOES NOT.  The reference has no tests, golden images or
 * fixtures for this path.  Its host-callable half -- PT_sv5_/maths.h, sample.h, Probe.cuh,
 * Material.h, Model.h/.cpp, cuda/random.h, cuda/helpers.h, sutil/vec_math.h, sutil/Camera.cpp -- compiles verbatim
 * against the real CUDA vector headers in the image (oracle/ref_shim.cpp, `make -C oracle ref` -> oracle/_ref/), and
 * tests/test_ref_pin_cpu.py holds THIS file to its outputs bit for bit (tests/golden/ref_vectors.npz): tea<4>, lcg,
 * rnd, Random, Sample2D, BasisFromVector, SafeNormalize, the hemisphere samplers, Luminance, ProbeDirToUV,
 * ProbeUVToDir, ProbeEval, LowerBound, ProbeSample, toSRGB, quantizeUnsigned8Bits, make_color, the vec_math
 * operators, Camera::UVWFrame, Material defaults.  PARITY UNPINNED for the rest: Disney.cuh (BSDFSample / Pdf /
 * Eval), Probe.h (BuildCDF) and deviceProgram.cu (raygen, closest-hit, SampleLights) include <optix.h> /
 * <optix_device.h>, which the image lacks and for which no stand-ins may be written.  Those parts are a line-by-line
 * restatement from reading the reference source, each function citing the file:line it follows; tests/mini_pt.py and
 * tests/disney_f64.py are a second, independent statement of them (scalar Python, binary64) that
 * tests/test_oracle_cpu.py holds this file against.  The arithmetic the reference delegates to
 * third-party code that is not under /root/reference is defined HERE as the parity contract:
 *   - NVIDIA OptiX 7/8 triangle intersection + traversal ("OptiX SDK 8.0", README.md:2;
 *     call sites PT_sv5_/deviceProgram.cu:209,234): Moeller-Trumbore in fp32, see
 *     intersect_tri(); closest hit = min t, ties -> lowest global primitive id; occlusion =
 *     any front-facing candidate with tmin < t < tmax.
 *   - CUDA texture unit bilinear filtering (deviceProgram.cu:664): fp32 weights, see tex2d().
 *   - CUDA libm (sinf, cosf, acosf, atan2f, logf, powf): either the host libm
 *     (math_mode 0) or include/fovpt_detmath.h (math_mode 1, bit-reproducible on the GPU).
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off; no fast-math).
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../include/fovpt.h"
#include "../include/fovpt_detmath.h"

namespace {

// ---------------------------------------------------------------------------------------
// vector helpers: the semantics of sutil/vec_math.h on plain structs
// ---------------------------------------------------------------------------------------
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
inline f3 mk3(float s) { return mk3(s, s, s); }
inline f3 mk3(const fovpt_float3& a) { return mk3(a.x, a.y, a.z); }
inline f3 mk3(const f4& a) { return mk3(a.x, a.y, a.z); }                    // vec_math.h make_float3(float4)
inline f3 operator-(const f3& a) { return mk3(-a.x, -a.y, -a.z); }           // vec_math.h:381
inline f3 operator+(const f3& a, const f3& b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }  // :414
inline f3 operator-(const f3& a, const f3& b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }  // :436
inline f3 operator-(float a, const f3& b) { return mk3(a - b.x, a - b.y, a - b.z); }            // :444
inline f3 operator*(const f3& a, const f3& b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }  // :458
inline f3 operator*(const f3& a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }            // :462
inline f3 operator*(float s, const f3& a) { return mk3(a.x * s, a.y * s, a.z * s); }            // :466
inline f3 operator/(const f3& a, const f3& b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }  // :483
inline f3 operator/(const f3& a, float s) { float inv = 1.0f / s; return a * inv; }             // :487
inline void operator+=(f3& a, const f3& b) { a.x += b.x; a.y += b.y; a.z += b.z; }              // :426
inline void operator*=(f3& a, const f3& s) { a.x *= s.x; a.y *= s.y; a.z *= s.z; }              // :470
inline void operator*=(f3& a, float s) { a.x *= s; a.y *= s; a.z *= s; }                        // :474
inline void operator/=(f3& a, float s) { float inv = 1.0f / s; a *= inv; }                      // :496
inline float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // :540
inline f3 cross(const f3& a, const f3& b)                                                        // :546
{ return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(const f3& v) { return sqrtf(dot(v, v)); }                                    // :552
inline f3 normalize(const f3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; } // :558
inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }                 // :119
inline f3 clamp3(const f3& v, float a, float b) { return mk3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); } // :528
inline float lerpf(float a, float b, float t) { return a + t * (b - a); }                        // :98
inline f3 lerp3(const f3& a, const f3& b, float t) { return a + t * (b - a); }                   // :507
inline f3 faceforward(const f3& n, const f3& i, const f3& nref) { return n * copysignf(1.0f, dot(i, nref)); } // :580
inline int clampi(int f, int a, int b) { return std::max(a, std::min(f, b)); }                   // :841
inline uint32_t clampu(uint32_t f, uint32_t a, uint32_t b) { return std::max(a, std::min(f, b)); } // :1277

// ---------------------------------------------------------------------------------------
// libm switch
// ---------------------------------------------------------------------------------------
int g_detmath = 1;
inline float m_sinf(float x) { return g_detmath ? fovpt_dm_sinf(x) : sinf(x); }
inline float m_cosf(float x) { return g_detmath ? fovpt_dm_cosf(x) : cosf(x); }
inline float m_acosf(float x) { return g_detmath ? fovpt_dm_acosf(x) : acosf(x); }
inline float m_atan2f(float y, float x) { return g_detmath ? fovpt_dm_atan2f(y, x) : atan2f(y, x); }
inline float m_logf(float x) { return g_detmath ? fovpt_dm_logf(x) : logf(x); }
inline float m_powf(float x, float y) { return g_detmath ? fovpt_dm_powf(x, y) : powf(x, y); }

// maths.h:29-32
const float kPi = 3.141592653589793f;
const float k2Pi = 3.141592653589793f * 2.0f;
const float kInvPi = 1.0f / kPi;
const float kInv2Pi = 1.0f / k2Pi;

inline float sqr(float a) { return a * a; }                                   // maths.h:78

// ---------------------------------------------------------------------------------------
// RNG: cuda/random.h:34-59,101-104 and maths.h:170-227
// ---------------------------------------------------------------------------------------
inline uint32_t tea4(uint32_t val0, uint32_t val1)                            // random.h:34-49, N = 4
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
inline uint32_t lcg(uint32_t& prev)                                           // random.h:53-59
{
    prev = 1664525u * prev + 1013904223u;
    return prev & 0x00FFFFFFu;
}
inline float rnd(uint32_t& prev) { return (float)lcg(prev) / (float)0x01000000; }   // random.h:101-104

struct Random {                                                               // maths.h:170-227
    uint32_t seed1, seed2;
    explicit Random(int seed = 0)
    {
        seed1 = 315645664u + (uint32_t)seed;
        seed2 = seed1 ^ 0x13ab45feu;
    }
    uint32_t Rand()
    {
        seed1 = (seed2 ^ ((seed1 << 5) | (seed1 >> 27))) ^ (seed1 * seed2);
        seed2 = seed1 ^ ((seed2 << 12) | (seed2 >> 20));
        return seed1;
    }
    float Randf()
    {
        uint32_t value = Rand();
        uint32_t limit = 0xffffffffu;
        return clampf((float)value * (1.0f / (float)limit), 0.f, 0.999999f);
    }
    float Randf(float mn, float mx)
    {
        float t = Randf();
        return (1.0f - t) * mn + t * mx;
    }
};
inline void Sample2D(Random& rand, float& u1, float& u2)                      // sample.h:253-259 (USE_RANDOM 1)
{
    u1 = rand.Randf(0.0f, 1.0f);
    u2 = rand.Randf(0.0f, 1.0f);
}

// maths.h:94-108.  sqrt(float) is the float overload; 1.0/x is a double division narrowed to float.
inline void BasisFromVector(const f3& w, f3* u, f3* v)
{
    if (fabsf(w.x) > fabsf(w.y)) {
        float invLen = (float)(1.0 / (double)sqrtf(w.x * w.x + w.z * w.z));
        *u = mk3(-w.z * invLen, 0.0f, w.x * invLen);
    } else {
        float invLen = (float)(1.0 / (double)sqrtf(w.y * w.y + w.z * w.z));
        *u = mk3(0.0f, w.z * invLen, -w.y * invLen);
    }
    *v = cross(w, *u);
}
inline f3 SafeNormalize(const f3& a)                                          // maths.h:144-156
{
    float m = dot(a, a);
    if ((double)m > 0.0) return a * (float)(1.0 / (double)sqrtf(m));
    return mk3(0.0f);
}
inline f3 UniformSampleHemisphere(Random& rand)                               // maths.h:243-254
{
    float z = rand.Randf(0.0f, 1.0f);
    float w = sqrtf(1.0f - z * z);
    float phi = k2Pi * rand.Randf(0.0f, 1.0f);
    float x = m_cosf(phi) * w;
    float y = m_sinf(phi) * w;
    return mk3(x, y, z);
}
inline f3 CosineSampleHemisphere(float u1, float u2)                          // maths.h:256-277
{
    float r = sqrtf(u1);
    float theta = k2Pi * u2;
    float sx = r * m_cosf(theta), sy = r * m_sinf(theta);
    float z = sqrtf(fmaxf(0.0f, 1.0f - sx * sx - sy * sy));   // maxf(a,b) = a > b ? a : b; same value for non-NaN
    return mk3(sx, sy, z);
}

// ---------------------------------------------------------------------------------------
// Probe: PT_sv5_/Probe.cuh and Probe.h
// ---------------------------------------------------------------------------------------
struct ProbeH {           // host view of fovpt_probe
    int width, height;
    const f4* data;
    const float *pdfX, *cdfX, *pdfY, *cdfY;
};
inline f2 ProbeDirToUV(const f3& dir)                                         // Probe.cuh:38-46
{
    float theta = m_acosf(clampf(dir.y, -1.0f, 1.0f));
    float phi = (dir.x == 0.0f && dir.z == 0.0f) ? 0.0f : m_atan2f(dir.z, dir.x);
    float u = (kPi + phi) * kInvPi * 0.5f;
    float v = theta * kInvPi;
    f2 r = {u, v};
    return r;
}
inline f3 ProbeUVToDir(const f2& uv)                                          // Probe.cuh:48-58
{
    float theta = uv.y * kPi;
    float phi = uv.x * 2.0f * kPi;
    float x = -m_sinf(theta) * m_cosf(phi);
    float y = m_cosf(theta);
    float z = -m_sinf(theta) * m_sinf(phi);
    return mk3(x, y, z);
}
inline f4 ProbeEval(const ProbeH& image, const f2& uv)                        // Probe.cuh:61-67
{
    int px = clampi(int(uv.x * image.width), 0, image.width - 1);
    int py = clampi(int(uv.y * image.height), 0, image.height - 1);
    return image.data[py * image.width + px];
}
inline int LowerBound(const float* array, int lower, int upper, const float value)   // Probe.cuh:119-136
{
    while (lower < upper) {
        int mid = lower + (upper - lower) / 2;
        if (array[mid] < value) lower = mid + 1;
        else upper = mid;
    }
    return lower;
}
inline void ProbeSample(const ProbeH& image, f3& dir, f3& color, float& pdf, Random& rand)   // Probe.cuh:138-169
{
    float r1, r2;
    Sample2D(rand, r1, r2);
    int row = LowerBound(image.cdfY, 0, image.height, r1);
    int col = LowerBound(image.cdfX, row * image.width, (row + 1) * image.width, r2) - row * image.width;
    color = mk3(image.data[row * image.width + col]);
    pdf = image.pdfX[row * image.width + col] * image.pdfY[row];
    float u = col / float(image.width);
    float v = row / float(image.height);
    float sinTheta = m_sinf(v * kPi);
    if (sinTheta == 0.0f) pdf = 0.0f;
    else pdf *= image.width * image.height / (2.0f * kPi * kPi * sinTheta);
    f2 uv = {u, v};
    dir = ProbeUVToDir(uv);
}
inline float ProbePdf(const ProbeH& image, const f3& d)                       // Probe.cuh:69-93 (used by FOVPT_OPT_SKY_MISS only)
{
    f2 uv = ProbeDirToUV(d);
    int col = clampi(int(uv.x * image.width), 0, image.width - 1);
    int row = clampi(int(uv.y * image.height), 0, image.height - 1);
    float pdf = image.pdfX[row * image.width + col] * image.pdfY[row];
    float sinTheta = m_sinf(uv.y * kPi);
    if (fabsf(sinTheta) < 0.0001f) pdf = 0.0f;
    else pdf *= float(image.width) * float(image.height) / (2.0f * kPi * kPi * sinTheta);
    return pdf;
}
inline float Luminance(const f4& c) { return c.x * 0.3f + c.y * 0.6f + c.z * 0.1f; }   // maths.h:165-168

void BuildCDF(int width, int height, const f4* data, float* pdfX, float* cdfX, float* pdfY, float* cdfY)  // Probe.h:29-77
{
    float totalWeightY = 0.0f;
    for (int j = 0; j < height; ++j) {
        float totalWeightX = 0.0f;
        for (int i = 0; i < width; ++i) {
            float weight = Luminance(data[j * width + i]);
            totalWeightX += weight;
            pdfX[j * width + i] = weight;
            cdfX[j * width + i] = totalWeightX;
        }
        float invTotalWeightX = 1.0f / totalWeightX;
        for (int i = 0; i < width; ++i) {
            pdfX[j * width + i] *= invTotalWeightX;
            cdfX[j * width + i] *= invTotalWeightX;
        }
        totalWeightY += totalWeightX;
        pdfY[j] = totalWeightX;
        cdfY[j] = totalWeightY;
    }
    for (int j = 0; j < height; ++j) {
        cdfY[j] /= float(totalWeightY);
        pdfY[j] /= float(totalWeightY);
    }
}

// ---------------------------------------------------------------------------------------
// Disney BSDF: PT_sv5_/Disney.cuh (USE_SIMPLE_BSDF 0, USE_UNIFORM_SAMPLING 0)
// ---------------------------------------------------------------------------------------
typedef fovpt_material Material;
enum BSDFType { eReflected, eTransmitted, eSpecular };

inline float GetIndexOfRefraction(const Material& m)                          // Material.h:40-46
{
    if (m.eta == 0.0f) return 2.0f / (1.0f - sqrtf(0.08f * m.specular)) - 1.0f;
    return m.eta;
}
inline bool Refract(const f3& wi, const f3& n, float eta, f3& wt)             // Disney.cuh:36-49
{
    float cosThetaI = dot(n, wi);
    float sin2ThetaI = fmaxf(0.0f, float(1.0f - cosThetaI * cosThetaI));
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1.0f - sin2ThetaT);
    wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
inline float SchlickFresnel(float u)                                          // Disney.cuh:51-56
{
    float m = clampf(1 - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
inline float GTR1(float NDotH, float a)                                       // Disney.cuh:58-64
{
    if (a >= 1) return kInvPi;
    float a2 = a * a;
    float t = 1 + (a2 - 1) * NDotH * NDotH;
    return (a2 - 1) / (kPi * m_logf(a2) * t);
}
inline float GTR2(float NDotH, float a)                                       // Disney.cuh:66-71
{
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (kPi * t * t);
}
inline float SmithGGX(float NDotv, float alphaG)                              // Disney.cuh:73-78
{
    float a = alphaG * alphaG;
    float b = NDotv * NDotv;
    return 1 / (NDotv + sqrtf(a + b - a * b));
}
inline float Fr(float VDotN, float etaI, float etaT)                          // Disney.cuh:81-98
{
    float SinThetaT2 = sqr(etaI / etaT) * (1.0f - VDotN * VDotN);
    if (SinThetaT2 > 1.0f) return 1.0f;
    float LDotN = sqrtf(1.0f - SinThetaT2);
    float eta = etaT / etaI;
    float r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN);
    float r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN);
    return 0.5f * (sqr(r1) + sqr(r2));
}
float BSDFPdf(const Material& mat, float etaI, float etaO, const f3& n, const f3& V, const f3& L)   // Disney.cuh:152-193
{
    if (dot(L, n) <= 0.0f) {
        float bsdfPdf = 0.0f;
        float brdfPdf = kInv2Pi * mat.subsurface * 0.5f;
        return lerpf(brdfPdf, bsdfPdf, mat.transmission);
    }
    float F = Fr(dot(n, V), etaI, etaO);
    const float a = fmaxf(0.001f, mat.roughness);
    const f3 half = SafeNormalize(L + V);
    const float cosThetaHalf = fabsf(dot(half, n));
    const float pdfHalf = GTR2(cosThetaHalf, a) * cosThetaHalf;
    float pdfSpec = 0.25f * pdfHalf / fmaxf(1.e-6f, dot(L, half));
    float pdfDiff = fabsf(dot(L, n)) * kInvPi * (1.0f - mat.subsurface);
    float bsdfPdf = pdfSpec * F;
    float brdfPdf = lerpf(pdfDiff, pdfSpec, 0.5f);
    return lerpf(brdfPdf, bsdfPdf, mat.transmission);
}
void BSDFSample(const Material& mat, float etaI, float etaO, const f3& U, const f3& V, const f3& N,
                const f3& view, f3& light, float& pdf, BSDFType& type, Random& rand)                  // Disney.cuh:197-315
{
    if (rand.Randf() < mat.transmission) {
        float F = Fr(dot(N, view), etaI, etaO);
        if (rand.Randf() < F) {
            float r1, r2;
            Sample2D(rand, r1, r2);
            const float a = fmaxf(0.001f, mat.roughness);
            const float phiHalf = r1 * k2Pi;
            const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (sqr(a) - 1.0f) * r2));
            const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - sqr(cosThetaHalf)));
            const float sinPhiHalf = m_sinf(phiHalf);
            const float cosPhiHalf = m_cosf(phiHalf);
            f3 half = U * (sinThetaHalf * cosPhiHalf) + V * (sinThetaHalf * sinPhiHalf) + N * cosThetaHalf;
            if (dot(half, view) <= 0.0f) half *= -1.0f;
            type = eReflected;
            light = 2.0f * dot(view, half) * half - view;
        } else {
            float eta = etaI / etaO;
            if (Refract(view, N, eta, light)) {
                type = eSpecular;
                pdf = (1.0f - F) * mat.transmission;
                return;
            } else {
                pdf = 0.0f;
                return;
            }
        }
    } else {
        float r1, r2;
        Sample2D(rand, r1, r2);
        if (rand.Randf() < 0.5f) {
            if (rand.Randf() < mat.subsurface) {
                const f3 d = UniformSampleHemisphere(rand);
                light = U * d.x + V * d.y - N * d.z;
                type = eTransmitted;
            } else {
                const f3 d = CosineSampleHemisphere(r1, r2);
                light = U * d.x + V * d.y + N * d.z;
                type = eReflected;
            }
        } else {
            const float a = fmaxf(0.001f, mat.roughness);
            const float phiHalf = r1 * k2Pi;
            const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (sqr(a) - 1.0f) * r2));
            const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - sqr(cosThetaHalf)));
            const float sinPhiHalf = m_sinf(phiHalf);
            const float cosPhiHalf = m_cosf(phiHalf);
            f3 half = U * (sinThetaHalf * cosPhiHalf) + V * (sinThetaHalf * sinPhiHalf) + N * cosThetaHalf;
            if (dot(half, view) <= 0.0f) half *= -1.0f;
            light = 2.0f * dot(view, half) * half - view;
            type = eReflected;
        }
    }
    pdf = BSDFPdf(mat, etaI, etaO, N, view, light);
}
f3 BSDFEval(const Material& mat, f3 albedo, float etaI, float etaO, const f3& N, const f3& V, const f3& L)   // Disney.cuh:318-427
{
    float NDotL = dot(N, L);
    float NDotV = dot(N, V);
    f3 H = normalize(L + V);
    float NDotH = dot(N, H);
    float LDotH = dot(L, H);
    f3 Cdlin = albedo;
    float Cdlum = (float)(.3 * (double)Cdlin.x + .6 * (double)Cdlin.y + .1 * (double)Cdlin.z);   // :329 double literals
    // :331 -- "Cdlin / Cdlum" is operator/(float3, float) = multiply by 1.0f/Cdlum (vec_math.h:487)
    f3 Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : mk3(1.0f);
    f3 Cspec0 = lerp3((float)((double)mat.specular * .08) * lerp3(mk3(1.0f), Ctint, mat.specularTint), Cdlin, mat.metallic);  // :332
    f3 bsdf = mk3(0.0f);
    f3 brdf = mk3(0.0f);
    if (mat.transmission > 0.0f) {
        if (NDotL <= 0) {
            float F = Fr(NDotV, etaI, etaO);
            bsdf = mk3(mat.transmission * (1.0f - F) / fabsf(NDotL) * (1.0f - mat.metallic));
        } else {
            float a = fmaxf(0.001f, mat.roughness);
            float Ds = GTR2(NDotH, a);
            float FH = Fr(LDotH, etaI, etaO);
            f3 Fs = lerp3(Cspec0, mk3(1.0f), FH);
            float roughg = a;
            float Gs = SmithGGX(NDotV, roughg) * SmithGGX(NDotL, roughg);
            bsdf = Gs * Fs * Ds;
        }
    }
    if (mat.transmission < 1.0f) {
        if (NDotL <= 0) {
            if (mat.subsurface > 0.0f) {
                f3 s = mk3(sqrtf(mat.color.x), sqrtf(mat.color.y), sqrtf(mat.color.z));
                float FL = SchlickFresnel(fabsf(NDotL)), FV = SchlickFresnel(NDotV);
                float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
                brdf = kInvPi * s * mat.subsurface * Fd * (1.0f - mat.metallic);
            }
        } else {
            float a = fmaxf(0.001f, mat.roughness);
            float Ds = GTR2(NDotH, a);
            float FH = SchlickFresnel(LDotH);
            f3 Fs = lerp3(Cspec0, mk3(1.f), FH);
            float roughg = a;
            float Gs = SmithGGX(NDotV, roughg) * SmithGGX(NDotL, roughg);
            float FL = SchlickFresnel(NDotL), FV = SchlickFresnel(NDotV);
            float Fd90 = (float)(0.5 + (double)(2.0f * LDotH * LDotH * mat.roughness));            // :398 double 0.5
            float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
            float Dr = GTR1(NDotH, lerpf(.1f, .001f, mat.clearcoatGloss));
            float Fc = lerpf(.04f, 1.0f, FH);
            float Gr = SmithGGX(NDotL, .25f) * SmithGGX(NDotV, .25f);
            brdf = kInvPi * Fd * Cdlin * (1.0f - mat.metallic) * (1.0f - mat.subsurface) + Gs * Fs * Ds
                   + mk3(mat.clearcoat * Gr * Fc * Dr);
        }
    }
    return lerp3(brdf, bsdf, mat.transmission);
}

// ---------------------------------------------------------------------------------------
// Scene: flattened triangles + a CPU BVH (binned SAH).  Traversal order never changes results:
// see intersect contract at the top of this file.
// ---------------------------------------------------------------------------------------
struct Tri {
    f3 v0, v1, v2;
    uint32_t mesh;
    uint32_t prim_in_mesh;
};
struct MeshInfo {
    Material material;
    int texture_id;            // <0 none
    bool has_texcoord;
    uint32_t first_tri;
    uint32_t first_vertex;
};
struct Tex { int w, h; std::vector<uint32_t> px; };
struct BNode {
    float lo[3], hi[3];
    int left, right;           // children (internal) or -1
    uint32_t first, count;     // leaf range into order[]
};
struct Scene {
    std::vector<Tri> tris;                 // global primitive id = index
    std::vector<MeshInfo> meshes;
    std::vector<f2> texcoord;              // per global vertex
    std::vector<uint32_t> index;           // 3 per triangle, global vertex ids
    std::vector<Tex> textures;
    std::vector<BNode> nodes;
    std::vector<uint32_t> order;
    bool any_catcher = false;
};

struct Hit { float t, u, v; uint32_t prim; };
const uint32_t NO_HIT = 0xffffffffu;

// Moeller-Trumbore, fp32, fixed operation order (the parity contract for OptiX's intersector).
// Returns true and (t,u,v) when the ray crosses the triangle's plane inside it; *det_out is
// dot(e1, cross(dir, e2)) whose sign gives the facing (det > 0 <=> dot(cross(e1,e2), dir) < 0,
// i.e. the triangle is wound counter-clockwise as seen from the ray origin = OptiX front face).
inline bool intersect_tri(const f3& o, const f3& d, const Tri& T, float& t, float& u, float& v, float& det_out)
{
    const f3 e1 = T.v1 - T.v0;
    const f3 e2 = T.v2 - T.v0;
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const f3 s = o - T.v0;
    u = dot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 q = cross(s, e1);
    v = dot(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot(e2, q) * inv;
    det_out = det;
    return true;
}

inline bool slab(const BNode& n, const f3& o, const f3& inv, float tmin, float tmax)
{
    // conservative: the node boxes are padded at build time, and the interval is widened here
    float t0 = tmin, t1 = tmax;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; a++) {
        float ta = (n.lo[a] - oo[a]) * ii[a];
        float tb = (n.hi[a] - oo[a]) * ii[a];
        if (ta != ta || tb != tb) continue;     // 0 * inf: origin on a slab plane, axis does not constrain
        if (ta > tb) std::swap(ta, tb);
        tb = tb * 1.0000004f + 1e-30f;
        ta = ta - fabsf(ta) * 4e-7f;
        t0 = std::max(t0, ta);
        t1 = std::min(t1, tb);
    }
    return t0 <= t1;
}

inline void closest_consider(const Scene& S, uint32_t prim, const f3& o, const f3& d, float tmin, float tmax, Hit& best)
{
    float t, u, v, det;
    if (!intersect_tri(o, d, S.tris[prim], t, u, v, det)) return;
    if (!(t > tmin && t < tmax)) return;
    if (t < best.t || (t == best.t && prim < best.prim)) {
        best.t = t; best.u = u; best.v = v; best.prim = prim;
    }
}

Hit trace_closest(const Scene& S, const f3& o, const f3& d, float tmin, float tmax, bool brute)
{
    Hit best; best.t = INFINITY; best.u = best.v = 0; best.prim = NO_HIT;
    if (brute || S.nodes.empty()) {
        for (uint32_t i = 0; i < S.tris.size(); i++) closest_consider(S, i, o, d, tmin, tmax, best);
        return best;
    }
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const BNode& n = S.nodes[stack[--sp]];
        float lim = best.prim == NO_HIT ? tmax : std::min(tmax, best.t * 1.000001f);
        if (!slab(n, o, inv, tmin, lim)) continue;
        if (n.left < 0) {
            for (uint32_t k = 0; k < n.count; k++) closest_consider(S, S.order[n.first + k], o, d, tmin, tmax, best);
        } else {
            stack[sp++] = n.left; stack[sp++] = n.right;
        }
    }
    return best;
}

// deviceProgram.cu:224-248,284-300: any candidate on a front-facing triangle occludes
bool trace_occluded(const Scene& S, const f3& o, const f3& d, float tmin, float tmax, bool brute)
{
    float t, u, v, det;
    if (brute || S.nodes.empty()) {
        for (uint32_t i = 0; i < S.tris.size(); i++)
            if (intersect_tri(o, d, S.tris[i], t, u, v, det) && det > 0.0f && t > tmin && t < tmax) return true;
        return false;
    }
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const BNode& n = S.nodes[stack[--sp]];
        if (!slab(n, o, inv, tmin, tmax)) continue;
        if (n.left < 0) {
            for (uint32_t k = 0; k < n.count; k++)
                if (intersect_tri(o, d, S.tris[S.order[n.first + k]], t, u, v, det) && det > 0.0f && t > tmin && t < tmax) return true;
        } else {
            stack[sp++] = n.left; stack[sp++] = n.right;
        }
    }
    return false;
}

// ---- BVH build (binned SAH, 16 bins, leaves <= 4) -----------------------------------------
struct BuildPrim { float lo[3], hi[3], c[3]; uint32_t id; };

int build_rec(Scene& S, std::vector<BuildPrim>& P, uint32_t begin, uint32_t end, int depth)
{
    BNode n;
    for (int a = 0; a < 3; a++) { n.lo[a] = INFINITY; n.hi[a] = -INFINITY; }
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = begin; i < end; i++)
        for (int a = 0; a < 3; a++) {
            n.lo[a] = std::min(n.lo[a], P[i].lo[a]); n.hi[a] = std::max(n.hi[a], P[i].hi[a]);
            clo[a] = std::min(clo[a], P[i].c[a]); chi[a] = std::max(chi[a], P[i].c[a]);
        }
    n.left = n.right = -1; n.first = begin; n.count = end - begin;
    int me = (int)S.nodes.size();
    S.nodes.push_back(n);
    if (end - begin <= 4 || depth > 100) return me;
    const int NB = 16;
    int best_axis = -1, best_split = -1; float best_cost = INFINITY;
    for (int a = 0; a < 3; a++) {
        float ext = chi[a] - clo[a];
        if (!(ext > 0)) continue;
        int cnt[NB] = {0}; float blo[NB][3], bhi[NB][3];
        for (int b = 0; b < NB; b++) for (int k = 0; k < 3; k++) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; }
        for (uint32_t i = begin; i < end; i++) {
            int b = std::min(NB - 1, (int)((P[i].c[a] - clo[a]) / ext * NB));
            cnt[b]++;
            for (int k = 0; k < 3; k++) { blo[b][k] = std::min(blo[b][k], P[i].lo[k]); bhi[b][k] = std::max(bhi[b][k], P[i].hi[k]); }
        }
        float la[NB], ra[NB]; int lc[NB], rc[NB];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}; int c = 0;
        for (int b = 0; b < NB; b++) {
            c += cnt[b];
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[b][k]); hi[k] = std::max(hi[k], bhi[b][k]); }
            float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            la[b] = c ? (dx * dy + dy * dz + dz * dx) : 0; lc[b] = c;
        }
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; } c = 0;
        for (int b = NB - 1; b >= 0; b--) {
            c += cnt[b];
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], blo[b][k]); hi[k] = std::max(hi[k], bhi[b][k]); }
            float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            ra[b] = c ? (dx * dy + dy * dz + dz * dx) : 0; rc[b] = c;
        }
        for (int b = 0; b < NB - 1; b++) {
            if (!lc[b] || !rc[b + 1]) continue;
            float cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
        }
    }
    uint32_t mid;
    if (best_axis < 0) {
        mid = (begin + end) / 2;
    } else {
        float ext = chi[best_axis] - clo[best_axis];
        float lo0 = clo[best_axis];
        int a = best_axis, sb = best_split;
        auto it = std::partition(P.begin() + begin, P.begin() + end, [&](const BuildPrim& p) {
            int b = std::min(NB - 1, (int)((p.c[a] - lo0) / ext * NB));
            return b <= sb;
        });
        mid = (uint32_t)(it - P.begin());
        if (mid == begin || mid == end) mid = (begin + end) / 2;
    }
    int l = build_rec(S, P, begin, mid, depth + 1);
    int r = build_rec(S, P, mid, end, depth + 1);
    S.nodes[me].left = l; S.nodes[me].right = r;
    return me;
}

void build_bvh(Scene& S)
{
    std::vector<BuildPrim> P(S.tris.size());
    for (uint32_t i = 0; i < S.tris.size(); i++) {
        const Tri& T = S.tris[i];
        const float vx[3][3] = {{T.v0.x, T.v0.y, T.v0.z}, {T.v1.x, T.v1.y, T.v1.z}, {T.v2.x, T.v2.y, T.v2.z}};
        float ext = 0, mag = 0;
        for (int a = 0; a < 3; a++) {
            float lo = std::min(vx[0][a], std::min(vx[1][a], vx[2][a]));
            float hi = std::max(vx[0][a], std::max(vx[1][a], vx[2][a]));
            P[i].lo[a] = lo; P[i].hi[a] = hi;
            ext = std::max(ext, hi - lo);
            mag = std::max(mag, std::max(fabsf(lo), fabsf(hi)));
        }
        // pad so that every Moeller-Trumbore-accepted hit point lies well inside the box
        float pad = 1e-4f * ext + 1e-5f * mag + 1e-20f;
        for (int a = 0; a < 3; a++) {
            P[i].lo[a] -= pad; P[i].hi[a] += pad;
            P[i].c[a] = 0.5f * (P[i].lo[a] + P[i].hi[a]);
        }
        P[i].id = i;
    }
    S.nodes.clear();
    S.nodes.reserve(P.size());
    if (!P.empty()) build_rec(S, P, 0, (uint32_t)P.size(), 0);
    S.order.resize(P.size());
    for (uint32_t i = 0; i < P.size(); i++) S.order[i] = P[i].id;
}

// bilinear RGBA8 fetch, wrap addressing, normalized coordinates, texel centres at +0.5
// (the parity contract for cudaFilterModeLinear / cudaAddressModeWrap / cudaReadModeNormalizedFloat,
//  SimplePathtracer.cpp:781-790).
inline f4 texel(const Tex& T, int x, int y)
{
    x %= T.w; if (x < 0) x += T.w;
    y %= T.h; if (y < 0) y += T.h;
    uint32_t p = T.px[(size_t)y * T.w + x];
    f4 r = {(float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f, (float)(p >> 24) / 255.0f};
    return r;
}
inline f4 tex2d(const Tex& T, float u, float v)
{
    float x = u * (float)T.w - 0.5f, y = v * (float)T.h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    // keep integer conversion in range for wild coordinates
    int x0 = (int)fmaxf(-1.0e9f, fminf(1.0e9f, fx0)), y0 = (int)fmaxf(-1.0e9f, fminf(1.0e9f, fy0));
    f4 c00 = texel(T, x0, y0), c10 = texel(T, x0 + 1, y0), c01 = texel(T, x0, y0 + 1), c11 = texel(T, x0 + 1, y0 + 1);
    float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    f4 r = {
        w00 * c00.x + w10 * c10.x + w01 * c01.x + w11 * c11.x,
        w00 * c00.y + w10 * c10.y + w01 * c01.y + w11 * c11.y,
        w00 * c00.z + w10 * c10.z + w01 * c01.z + w11 * c11.z,
        w00 * c00.w + w10 * c10.w + w01 * c01.w + w11 * c11.w};
    return r;
}

// ---------------------------------------------------------------------------------------
// deviceProgram.cu
// ---------------------------------------------------------------------------------------
const int RAY_STATE_FLAGS_DONE = 1 << 0;            // :55
const int RAY_STATE_FLAGS_SECONDARY_RAY = 1 << 1;   // :56
const float kTmin = 0.01f;                          // :41
const float kTmax = 1e16f;                          // :42

struct RadiancePRD {                                // :60-89
    f3 radiance, alpha, origin, direction, normal, albedo;
    float bsdfPdf;
    f3 pathThroughput;
    float rayEta;
    f3 rayAbsorption;
    int depth;
    int stateFlags;
    Random rand;
    bool sky_added = false;    // FOVPT_OPT_SKY_MISS: this (escaped, secondary) segment carries sky radiance to be counted
    bool rr_kill = false;      // FOVPT_OPT_RUSSIAN_ROULETTE: the path ends after this segment
};

struct Opts {
    int max_depth;       // deviceProgram.cu:515 hard-codes 4
    int accumulate;      // PT_sv4_vmv2/deviceProgram.cu:545-553
    int brute;
    int nthreads;
    int write_guides;    // PT_sv/deviceProgram.cu:555-557 (commented out in PT_sv5_/deviceProgram.cu:612-614)
    int options = 0;     // fovpt_config.options: FOVPT_OPT_SKY_MISS | FOVPT_OPT_RUSSIAN_ROULETTE (not in the reference)
};
struct Counters {
    std::atomic<uint64_t> radiance_rays{0}, shadow_rays{0}, paths{0};
};
// Rays the LIBRARY traces for the same frame.  The reference (and this restatement) trace two kinds of rays whose
// results cannot reach the image; libfovpt skips them and tests assert that its device counters
// equal these numbers exactly:
//   (i)  the segment at depth == max_depth, discarded by the break at deviceProgram.cu:515 -- unless the scene holds a
//        shadow catcher (then :689 makes its hit matter) -- and with it the shadow ray of its hit;
//   (ii) the shadow ray of a hit whose prd.radiance / prd.alpha come out bit-identical whether it is occluded or not,
//        or (not on a catcher) whose radiance is dropped because BSDFSample returned pdf <= 0 (:708-711 + :515).
std::atomic<uint64_t> g_lib_radiance{0}, g_lib_shadow{0}, g_occluded{0};      // g_occluded: diagnostics, occluded shadow rays (all)
int g_options = 0;         // fovpt_config.options for the launches that follow (orc_render sets it from its cfg)
int g_count_lib = 1;       // 0: skip the extra BSDF evaluation the counting needs on occluded rays (timed CPU baseline)

struct Ctx {
    const Scene* S;
    const fovpt_launch_params* lp;
    ProbeH probe;
    Opts opt;
    Counters* cnt;
    const fovpt_float4* accum_before = nullptr;     // accum_buffer as it was when the launch began (see launch())
};

// deviceProgram.cu:303-344 (SampleLights) and :347-387 (SampleShadow, want_occluded = true)
f3 SampleLightsOrShadow(const Ctx& C, const Material& material, f3 albedo, float etaI, float etaO,
                        const f3& surfacePos, const f3& surfaceNormal, const f3& wo, Random& rand,
                        bool want_occluded, uint64_t& nshadow, f3* sum_if_taken = nullptr)
{
    f3 sum = mk3(0.0f);
    f3 skyColor; float skyPdf; f3 wi;
    ProbeSample(C.probe, wi, skyColor, skyPdf, rand);
    nshadow++;
    const bool occluded = trace_occluded(*C.S, surfacePos, wi, kTmin, kTmax, C.opt.brute);
    if (occluded) g_occluded++;
    // the branch below is a pure function of the hit and wi (no random numbers): evaluated once, used if taken
    f3 taken = mk3(0.0f);
    if (g_count_lib || occluded == want_occluded) {
        float bsdfPdf = BSDFPdf(material, etaI, etaO, surfaceNormal, wo, wi);
        f3 f = BSDFEval(material, albedo, etaI, etaO, surfaceNormal, wo, wi);
        if (bsdfPdf > 0.0f) {
            int N = (int)(1.f + 1.f);                   // kProbeSamples + kBsdfSamples
            float cbsdf = 1.f / N;
            float csky = float(1.f) / N;
            float weight = csky * skyPdf / (cbsdf * bsdfPdf + csky * skyPdf);
            if (weight > 0.0f) {
                f3 val = weight * skyColor * f * fabsf(dot(wi, surfaceNormal)) / skyPdf * (1.0f / 1.f);
                taken += val;
            }
        }
    }
    if (occluded == want_occluded) sum = taken;
    if (sum_if_taken) *sum_if_taken = taken;
    return sum;
}

// optixTrace(RAY_TYPE_RADIANCE) + __closesthit__radiance (:619-732) / __miss__radiance (:253-282)
void traceRadiance(const Ctx& C, const f3& ray_origin, const f3& ray_dir, RadiancePRD* prd, uint64_t& nrad, uint64_t& nshadow)
{
    const Scene& S = *C.S;
    nrad++;
    Hit h = trace_closest(S, ray_origin, ray_dir, kTmin, kTmax, C.opt.brute);
    prd->sky_added = false; prd->rr_kill = false;
    if (h.prim == NO_HIT) {                                         // __miss__radiance
        if ((C.opt.options & FOVPT_OPT_SKY_MISS) && (prd->stateFlags & RAY_STATE_FLAGS_SECONDARY_RAY)) {
            // the block the reference carries commented out (:259-269), with the balance-heuristic constants of
            // SampleLights (:331-335; the commented code's integer 1 / 2 would make the weight zero)
            const float skyPdf = ProbePdf(C.probe, ray_dir);
            const float weight = 0.5f * prd->bsdfPdf / (0.5f * prd->bsdfPdf + 0.5f * skyPdf);
            prd->radiance += weight * mk3(ProbeEval(C.probe, ProbeDirToUV(ray_dir))) * prd->pathThroughput;
            prd->sky_added = true;
        }
        prd->albedo = mk3(0.f);
        prd->normal = mk3(0.f);
        prd->stateFlags |= RAY_STATE_FLAGS_DONE;
        return;
    }
    const Tri& T = S.tris[h.prim];
    const MeshInfo& M = S.meshes[T.mesh];
    const Material& mat = M.material;
    const f3 N_0 = normalize(cross(T.v1 - T.v0, T.v2 - T.v0));      // :632
    f3 N = faceforward(N_0, -ray_dir, N_0);                         // :634
    const float t = h.t;
    const f3 P = ray_origin + t * ray_dir;                          // :638
    float outEta; f3 outAbsorption;

    if ((mat.flags & FOVPT_MATERIAL_FLAG_SHADOW_CATCHER) != 0 && (prd->stateFlags & RAY_STATE_FLAGS_SECONDARY_RAY) != 0) {   // :646-651
        prd->origin = P;
        prd->direction = ray_dir;
        --prd->depth;
        return;
    }
    prd->normal = N;
    prd->albedo = mk3(mat.color);
    if (M.texture_id >= 0 && M.has_texcoord) {                      // :655-670
        const float u = h.u, v = h.v;
        const f2 t0 = S.texcoord[S.index[3 * h.prim + 0]], t1 = S.texcoord[S.index[3 * h.prim + 1]], t2 = S.texcoord[S.index[3 * h.prim + 2]];
        const float w0 = 1.f - u - v;
        // float * float2 + float * float2 + float * float2, left to right
        const float tcx = (w0 * t0.x + u * t1.x) + v * t2.x;
        const float tcy = (w0 * t0.y + u * t1.y) + v * t2.y;
        prd->albedo = mk3(tex2d(S.textures[M.texture_id], tcx, tcy));
    }
    if (prd->rayEta == 1.0f) {                                      // :673-683
        outEta = GetIndexOfRefraction(mat);
        outAbsorption = mk3(mat.absorption);
    } else {
        outEta = 1.0f;
        outAbsorption = mk3(0.0f);
    }
    const bool catcher = (mat.flags & FOVPT_MATERIAL_FLAG_SHADOW_CATCHER) != 0;
    const bool lib_shades = prd->depth < C.opt.max_depth;           // the library does not shade the discarded segment's hit
    f3 sum_taken;                                                   // what SampleLights/SampleShadow returns when its visibility test passes
    if (!catcher) {                                                 // :686-694
        f3 lightSample = SampleLightsOrShadow(C, mat, prd->albedo, prd->rayEta, outEta, P, N, -ray_dir, prd->rand, false, nshadow, &sum_taken);
        prd->radiance += prd->pathThroughput * lightSample;
        prd->alpha = mk3(1.0f);
    } else {
        f3 shadowSample = SampleLightsOrShadow(C, mat, prd->albedo, prd->rayEta, outEta, P, N, -ray_dir, prd->rand, true, nshadow, &sum_taken);
        prd->alpha += prd->pathThroughput * shadowSample;
    }
    // does the outcome of the occlusion test change a single bit?  (radiance / alpha were zero before this hit)
    bool shadow_matters;
    {
        const bool primary = (prd->stateFlags & RAY_STATE_FLAGS_SECONDARY_RAY) == 0;
        f3 a = mk3(0.f) + prd->pathThroughput * sum_taken, b = mk3(0.f) + prd->pathThroughput * mk3(0.0f);
        if (!catcher && primary) { a += mk3(mat.emission); b += mk3(mat.emission); }
        shadow_matters = !(a.x == b.x && a.y == b.y && a.z == b.z);
    }
    if ((prd->stateFlags & RAY_STATE_FLAGS_SECONDARY_RAY) == 0) prd->radiance += mk3(mat.emission);   // :696-698

    f3 u, v;
    BasisFromVector(N, &u, &v);
    f3 bsdfDir; BSDFType bsdfType;
    BSDFSample(mat, prd->rayEta, outEta, u, v, N, -ray_dir, bsdfDir, prd->bsdfPdf, bsdfType, prd->rand);   // :706
    if (lib_shades && shadow_matters && (catcher || prd->bsdfPdf > 0.0f)) g_lib_shadow++;
    if (prd->bsdfPdf <= 0.0f) {                                     // :708-711
        prd->stateFlags |= RAY_STATE_FLAGS_DONE;
        return;
    }
    f3 f = BSDFEval(mat, prd->albedo, prd->rayEta, outEta, N, -ray_dir, bsdfDir);   // :714
    if (dot(bsdfDir, N) <= 0.0f) {                                  // :717-721
        prd->rayEta = outEta;
        prd->rayAbsorption = outAbsorption;
    }
    prd->pathThroughput *= f * fabsf(dot(N, bsdfDir)) / prd->bsdfPdf;   // :724
    prd->direction = bsdfDir;
    prd->origin = P;
    prd->stateFlags |= RAY_STATE_FLAGS_SECONDARY_RAY;
    if ((C.opt.options & FOVPT_OPT_RUSSIAN_ROULETTE) && prd->depth + 1 >= 2) {     // the //!TODO of :518-520 (opt-in, see include/fovpt.h)
        const f3 t = prd->pathThroughput;
        const float q = fmaxf(0.05f, fminf(1.0f, fmaxf(t.x, fmaxf(t.y, t.z))));
        if (prd->rand.Randf() >= q) prd->rr_kill = true;
        else prd->pathThroughput = prd->pathThroughput * (1.0f / q);
    }
}

inline f3 reinhardToneMap(const f3& color, const float white)       // :126-131
{
    const float luminance = 0.2126f * color.x + 0.7152f * color.y + 0.0722f * color.z;
    return (color * 1.0f) / (1.0f + luminance / white);
}
inline f3 toSRGB(const f3& c)                                       // cuda/helpers.h:35-43
{
    float invGamma = 1.0f / 2.4f;
    f3 powed = mk3(m_powf(c.x, invGamma), m_powf(c.y, invGamma), m_powf(c.z, invGamma));
    return mk3(c.x < 0.0031308f ? 12.92f * c.x : 1.055f * powed.x - 0.055f,
               c.y < 0.0031308f ? 12.92f * c.y : 1.055f * powed.y - 0.055f,
               c.z < 0.0031308f ? 12.92f * c.z : 1.055f * powed.z - 0.055f);
}
inline uint32_t quantizeUnsigned8Bits(float x)                      // cuda/helpers.h:50-55
{
    x = clampf(x, 0.0f, 1.0f);
    return std::min((uint32_t)(x * 256.0f), 255u);
}
inline uint32_t make_color(const f3& c)                             // cuda/helpers.h:57-62; uchar4 packed little-endian
{
    f3 srgb = toSRGB(clamp3(c, 0.0f, 1.0f));
    return quantizeUnsigned8Bits(srgb.x) | (quantizeUnsigned8Bits(srgb.y) << 8) | (quantizeUnsigned8Bits(srgb.z) << 16) | (255u << 24);
}

// __raygen__renderFrame for one launch index (:392-617)
void raygen(const Ctx& C, uint32_t lx, uint32_t ly)
{
    const fovpt_launch_params& params = *C.lp;
    const int w = params.frame.size.x;
    const int h = params.frame.size.y;
    const f3 eye = mk3(params.camera.eye), U = mk3(params.camera.U), V = mk3(params.camera.V), W = mk3(params.camera.W);
    uint32_t idx[3] = {lx, ly, 0};
    const uint32_t subframe_index = params.frame.subframe_index;
    int samples_per_launch = (int)params.samples_per_launch;
    int i = samples_per_launch;
    uint32_t seed = tea4(idx[1] * (uint32_t)w + idx[0], subframe_index);            // :411
    f3 result = mk3(0.0f);
    idx[0] = idx[0] * params.frame.factor.x + params.frame.offset.x;               // :433 (uint, wraps)
    idx[1] = idx[1] * params.frame.factor.y + params.frame.offset.y;
    idx[2] = idx[2] * params.frame.factor.z + 0;
    float range = length(mk3((float)idx[0], (float)idx[1], (float)idx[2]) - mk3((float)params.frame.c.x, (float)params.frame.c.y, 0.0f));   // :435
    if (range < params.frame.r_inner || range > params.frame.r_outer) return;       // :437
    f3 normal = mk3(0.f);                                                           // :443-444 denoiser guides
    f3 albedo = mk3(0.f);
    f3 alpha = mk3(0.f);
    f3 backplate = mk3(0.f);
    uint64_t nrad = 0, nshadow = 0, npaths = 0, nlib = 0;
    do {
        f3 directLight = mk3(0.0f), indirectLight = mk3(0.0f);
        RadiancePRD prd;
        prd.radiance = mk3(0.f);
        prd.alpha = mk3(0.f);
        prd.rand = Random((int)seed);                                               // :464
        prd.rayEta = 1.0f;
        prd.pathThroughput = mk3(1.f);
        prd.rayAbsorption = mk3(0.f);
        prd.bsdfPdf = 1.0f;
        prd.normal = mk3(0.0f);
        prd.albedo = mk3(0.0f);
        prd.stateFlags = 0;
        prd.depth = 0;
        const float jx = rnd(seed);                                                 // :479, x drawn first
        const float jy = rnd(seed);
        const float dx = 2.0f * ((static_cast<float>(idx[0]) + jx) / static_cast<float>(w)) - 1.0f;   // :483-486
        const float dy = 2.0f * ((static_cast<float>(idx[1]) + jy) / static_cast<float>(h)) - 1.0f;
        f3 ray_direction = normalize(dx * U + dy * V + W);                          // :491
        f3 ray_origin = eye;
        backplate = mk3(ProbeEval(C.probe, ProbeDirToUV(ray_direction)));           // :495
        npaths++;
        for (;;) {
            prd.radiance = mk3(0.f);
            if (prd.depth < C.opt.max_depth || C.S->any_catcher) nlib++;            // (the library skips the discarded segment)
            traceRadiance(C, ray_origin, ray_direction, &prd, nrad, nshadow);       // :501
            if (prd.depth == 0.f) {                                                 // :509-512
                normal += prd.normal;
                albedo += prd.albedo;
            }
            if ((prd.stateFlags & RAY_STATE_FLAGS_DONE) || prd.depth >= C.opt.max_depth) {   // :515
                // FOVPT_OPT_SKY_MISS: an escaped secondary ray's sky radiance counts (depth >= 1 here: secondary)
                if (prd.sky_added && prd.depth < C.opt.max_depth) indirectLight += prd.radiance;
                break;
            }
            if (prd.depth == 0) directLight += prd.radiance;
            else indirectLight += prd.radiance;
            ++prd.depth;
            if (prd.rr_kill) break;                                                // FOVPT_OPT_RUSSIAN_ROULETTE
            ray_origin = prd.origin;
            ray_direction = prd.direction;
        }
        result += directLight + indirectLight;                                      // :536
        alpha += prd.alpha;
    } while (--i);
    C.cnt->radiance_rays += nrad; C.cnt->shadow_rays += nshadow; C.cnt->paths += npaths;
    g_lib_radiance += nlib;
    normal /= static_cast<float>(samples_per_launch);                               // :541-542
    albedo /= static_cast<float>(samples_per_launch);
    alpha /= static_cast<float>(samples_per_launch);                                // :543

    for (int fi = 0; fi < params.frame.fillSize; ++fi) {                            // :546-616
        for (int fj = 0; fj < params.frame.fillSize; ++fj) {
            uint32_t ix = lx * params.frame.factor.x + (uint32_t)fi + params.frame.offset.x;
            uint32_t iy = ly * params.frame.factor.y + (uint32_t)fj + params.frame.offset.y;
            ix = clampu(ix, 0u, (uint32_t)(w - 1));
            iy = clampu(iy, 0u, (uint32_t)(h - 1));
            const uint32_t image_index = iy * (uint32_t)w + ix;
            f3 color = (backplate * static_cast<float>(params.samples_per_launch)) * (1.0f - alpha) + result;   // :558
            f3 accum_color = color / static_cast<float>(params.samples_per_launch);                             // :560
            if (C.opt.accumulate && subframe_index > 0 && !params.frame.redraw) {
                // PT_sv4_vmv2/deviceProgram.cu:545-553 (commented out in PT_sv5_/deviceProgram.cu:565-581)
                accum_color = clamp3(accum_color, 0.0f, 10.0f);
                const float alpha_value = 1.0f / static_cast<float>(subframe_index + 1);
                const fovpt_float4 pv = C.accum_before ? C.accum_before[image_index] : params.frame.accum_buffer[image_index];
                const f3 accum_color_prev = mk3(pv.x, pv.y, pv.z);
                accum_color = lerp3(accum_color_prev, accum_color, alpha_value);
            }
            fovpt_float4 out = {accum_color.x, accum_color.y, accum_color.z, 1.0f};
            params.frame.accum_buffer[image_index] = out;                           // :582
            f3 exposed = accum_color * 16.0f;                                       // :586 pow(2.0f, 4.0f)
            params.frame.frame_buffer[image_index] = make_color(reinhardToneMap(exposed, 1.0f));   // :597
            if (C.opt.write_guides) {                                               // :612-614 (live in PT_sv/deviceProgram.cu:555-557)
                if (params.frame.normal_buffer) { fovpt_float4 v = {normal.x, normal.y, normal.z, 1.0f}; params.frame.normal_buffer[image_index] = v; }
                if (params.frame.color_buffer) { fovpt_float4 v = {accum_color.x, accum_color.y, accum_color.z, 1.0f}; params.frame.color_buffer[image_index] = v; }
                if (params.frame.albedo_buffer) { fovpt_float4 v = {albedo.x, albedo.y, albedo.z, 1.0f}; params.frame.albedo_buffer[image_index] = v; }
            }
        }
    }
}

// optixLaunch over a width x height grid.  Rows are processed in ascending launch order within a
// thread and write races between launch indices (only possible through the clamp at :554) are
// resolved in ascending (y, x) order by running such launches single-threaded.
void launch(const Scene& S, const fovpt_launch_params& lp, uint32_t width, uint32_t height, const Opts& opt, Counters& cnt)
{
    Ctx C;
    C.S = &S; C.lp = &lp; C.opt = opt; C.cnt = &cnt;
    C.probe.width = lp.probe.width; C.probe.height = lp.probe.height;
    C.probe.data = (const f4*)lp.probe.data;
    C.probe.pdfX = lp.probe.pdfValuesX; C.probe.cdfX = lp.probe.cdfValuesX;
    C.probe.pdfY = lp.probe.pdfValuesY; C.probe.cdfY = lp.probe.cdfValuesY;
    // does any launch index write outside the frame (and hence get clamped onto an edge pixel)?
    bool clamps = false;
    {
        uint64_t maxx = (uint64_t)(width ? width - 1 : 0) * lp.frame.factor.x + (uint64_t)std::max(lp.frame.fillSize - 1, 0) + lp.frame.offset.x;
        uint64_t maxy = (uint64_t)(height ? height - 1 : 0) * lp.frame.factor.y + (uint64_t)std::max(lp.frame.fillSize - 1, 0) + lp.frame.offset.y;
        if (maxx >= (uint64_t)lp.frame.size.x || maxy >= (uint64_t)lp.frame.size.y) clamps = true;
        if ((uint32_t)lp.frame.fillSize > lp.frame.factor.x || (uint32_t)lp.frame.fillSize > lp.frame.factor.y) clamps = true;   // overlapping fills
    }
    int nt = std::max(1, opt.nthreads);
    if (clamps) nt = 1;
    // Accumulate mode (PT_sv4_vmv2 :545-553) reads accum_buffer[pixel] and writes it back.  When several launch
    // indices of ONE launch write the same pixel (clamped or overlapping fills) that is a data race in the
    // reference.  It is resolved here the way a GPU resolves it when the reads precede the conflicting writes:
    // every writer blends with the value from BEFORE the launch, and the last one in launch order stays.
    std::vector<fovpt_float4> before;
    if (clamps && opt.accumulate && lp.frame.subframe_index > 0 && !lp.frame.redraw) {
        before.assign(lp.frame.accum_buffer, lp.frame.accum_buffer + (size_t)lp.frame.size.x * lp.frame.size.y);
        C.accum_before = before.data();
    }
    if (nt == 1) {
        for (uint32_t y = 0; y < height; y++)
            for (uint32_t x = 0; x < width; x++) raygen(C, x, y);
        return;
    }
    std::atomic<uint32_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&]() {
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= height) break;
                for (uint32_t x = 0; x < width; x++) raygen(C, x, y);
            }
        });
    for (auto& t : th) t.join();
}

fovpt_uint3 mk_u3(uint32_t a, uint32_t b, uint32_t c) { fovpt_uint3 r = {a, b, c}; return r; }
fovpt_uint2 mk_u2(uint32_t a, uint32_t b) { fovpt_uint2 r = {a, b}; return r; }

}  // namespace

// =========================================================================================
// C ABI for the tests (ctypes)
// =========================================================================================
extern "C" {

void orc_set_math_mode(int detmath) { g_detmath = detmath ? 1 : 0; }

void* orc_scene_create(const fovpt_mesh_desc* meshes, int num_meshes, const fovpt_texture_desc* textures, int num_textures)
{
    Scene* S = new Scene;
    uint32_t vbase = 0;
    for (int m = 0; m < num_meshes; m++) {
        const fovpt_mesh_desc& D = meshes[m];
        MeshInfo mi;
        mi.material = D.material;
        mi.texture_id = (D.texture_id >= 0 && D.texture_id < num_textures) ? D.texture_id : -1;
        mi.has_texcoord = D.texcoord != nullptr;
        mi.first_tri = (uint32_t)S->tris.size();
        mi.first_vertex = vbase;
        if (D.material.flags & FOVPT_MATERIAL_FLAG_SHADOW_CATCHER) S->any_catcher = true;
        for (uint32_t v = 0; v < D.num_vertices; v++) {
            f2 tc = {0, 0};
            if (D.texcoord) { tc.x = D.texcoord[2 * v]; tc.y = D.texcoord[2 * v + 1]; }
            S->texcoord.push_back(tc);
        }
        for (uint32_t t = 0; t < D.num_triangles; t++) {
            Tri T;
            const uint32_t a = D.index[3 * t], b = D.index[3 * t + 1], c = D.index[3 * t + 2];
            T.v0 = mk3(D.vertex[3 * a], D.vertex[3 * a + 1], D.vertex[3 * a + 2]);
            T.v1 = mk3(D.vertex[3 * b], D.vertex[3 * b + 1], D.vertex[3 * b + 2]);
            T.v2 = mk3(D.vertex[3 * c], D.vertex[3 * c + 1], D.vertex[3 * c + 2]);
            T.mesh = (uint32_t)m; T.prim_in_mesh = t;
            S->tris.push_back(T);
            S->index.push_back(vbase + a); S->index.push_back(vbase + b); S->index.push_back(vbase + c);
        }
        vbase += D.num_vertices;
        S->meshes.push_back(mi);
    }
    for (int t = 0; t < num_textures; t++) {
        Tex X; X.w = textures[t].width; X.h = textures[t].height;
        X.px.assign(textures[t].pixel, textures[t].pixel + (size_t)X.w * X.h);
        S->textures.push_back(X);
    }
    build_bvh(*S);
    return S;
}
void orc_scene_destroy(void* s) { delete (Scene*)s; }
uint64_t orc_scene_num_triangles(void* s) { return ((Scene*)s)->tris.size(); }

/* optixLaunch equivalent; lp carries HOST pointers (frame buffers, probe arrays). counters[3] += {radiance, shadow, paths} */
int orc_launch2(void* scene, const fovpt_launch_params* lp, uint32_t width, uint32_t height,
                int max_depth, int accumulate, int brute, int nthreads, int write_guides, uint64_t* counters)
{
    if (!scene || !lp || !lp->frame.accum_buffer || !lp->frame.frame_buffer || !lp->probe.data) return FOVPT_E_INVALID;
    if (lp->samples_per_launch == 0) return FOVPT_E_INVALID;     // do{}while(--i) needs spp >= 1 (:448,539)
    Opts o; o.max_depth = max_depth; o.accumulate = accumulate; o.brute = brute; o.nthreads = nthreads; o.write_guides = write_guides;
    o.options = g_options;
    Counters c;
    launch(*(Scene*)scene, *lp, width, height, o, c);
    if (counters) { counters[0] += c.radiance_rays; counters[1] += c.shadow_rays; counters[2] += c.paths; }
    return 0;
}

int orc_launch(void* scene, const fovpt_launch_params* lp, uint32_t width, uint32_t height,
               int max_depth, int accumulate, int brute, int nthreads, uint64_t* counters)
{ return orc_launch2(scene, lp, width, height, max_depth, accumulate, brute, nthreads, 0, counters); }

/* SampleRenderer::render(), SimplePathtracer.cpp:77-214, with the #defines turned into cfg fields */
int orc_render(void* scene, fovpt_launch_params* lp, const fovpt_config* cfg, int brute, int nthreads, uint64_t* counters)
{
    if (!scene || !lp || !cfg) return FOVPT_E_INVALID;
    if (lp->frame.size.x == 0) return 0;                                            // :81-82
    fovpt_launch_params& L = *lp;
    int rc;
    struct OptGuard { int saved; OptGuard(int v) : saved(g_options) { g_options = v; } ~OptGuard() { g_options = saved; } } opt_guard(cfg->options);
    if (cfg->uniform) {                                                             // :85-131 (FOV_OFF)
        L.frame.subframe_index = 0;
        L.frame.factor = mk_u3(1, 1, 1);
        L.frame.fillSize = 1;
        L.frame.r_outer = 1000000000;
        L.frame.r_inner = 0;
        L.samples_per_launch = (uint32_t)cfg->spp_uniform;
        L.frame.offset = mk_u2(0, 0);
        L.frame.redraw = 0;
        L.viewportSize.x = L.frame.size.x; L.viewportSize.y = L.frame.size.y;
        int temp_frame = (int)L.frame.subframe_index;
        rc = orc_launch2(scene, &L, (uint32_t)L.frame.size.x, (uint32_t)L.frame.size.y, cfg->max_depth, cfg->accumulate, brute, nthreads, cfg->write_guides, counters);
        L.frame.subframe_index = (uint32_t)temp_frame;
        L.frame.subframe_index++;
        return rc;
    }
    const int inner_radius = cfg->r_inner, outer_radius = cfg->r_outer;
    // periphery :137-157
    L.frame.factor = mk_u3(4, 4, 1);
    L.frame.fillSize = 4;
    L.frame.r_outer = 1000000000;
    L.frame.r_inner = (float)outer_radius;
    L.samples_per_launch = (uint32_t)cfg->spp_periphery;
    L.frame.offset = mk_u2(0, 0);
    L.frame.redraw = 0;
    rc = orc_launch2(scene, &L, (uint32_t)(L.frame.size.x / 4), (uint32_t)(L.frame.size.y / 4), cfg->max_depth, cfg->accumulate, brute, nthreads, cfg->write_guides, counters);
    if (rc) return rc;
    // intermediate :160-187
    int temp_frame = (int)L.frame.subframe_index;
    L.frame.subframe_index = 0;
    L.frame.factor = mk_u3(2, 2, 1);
    L.frame.fillSize = 2;
    L.frame.r_outer = (float)(outer_radius + 2);
    L.frame.r_inner = (float)inner_radius;
    L.samples_per_launch = (uint32_t)cfg->spp_middle;
    L.frame.offset = mk_u2(L.frame.c.x - (uint32_t)(outer_radius + 2), L.frame.c.y - (uint32_t)(outer_radius + 2));
    L.frame.redraw = 1;
    rc = orc_launch2(scene, &L, (uint32_t)L.frame.r_outer, (uint32_t)L.frame.r_outer, cfg->max_depth, cfg->accumulate, brute, nthreads, cfg->write_guides, counters);
    if (rc) return rc;
    // fovea :189-209
    L.frame.factor = mk_u3(1, 1, 1);
    L.frame.fillSize = 1;
    L.frame.r_outer = (float)(inner_radius + 1);
    L.frame.r_inner = 0;
    L.samples_per_launch = (uint32_t)cfg->spp_fovea;
    L.frame.offset = mk_u2(L.frame.c.x - (uint32_t)(inner_radius + 1), L.frame.c.y - (uint32_t)(inner_radius + 1));
    L.frame.redraw = 1;
    rc = orc_launch2(scene, &L, (uint32_t)(L.frame.r_outer * 2), (uint32_t)(L.frame.r_outer * 2), cfg->max_depth, cfg->accumulate, brute, nthreads, cfg->write_guides, counters);
    L.frame.subframe_index = (uint32_t)temp_frame;
    L.frame.subframe_index++;
    return rc;
}

/* rays libfovpt traces for the frames rendered since the last reset (see g_lib_radiance above) */
void orc_lib_counts(uint64_t* out2, int reset)
{
    if (out2) { out2[0] = g_lib_radiance; out2[1] = g_lib_shadow; }
    if (reset) { g_lib_radiance = 0; g_lib_shadow = 0; }
}
void orc_set_options(int options) { g_options = options; }
void orc_set_lib_counting(int on) { g_count_lib = on ? 1 : 0; }
uint64_t orc_occluded_count(int reset)
{
    const uint64_t v = g_occluded;
    if (reset) g_occluded = 0;
    return v;
}

/* ---- unit-level entry points --------------------------------------------------------- */
uint32_t orc_tea4(uint32_t a, uint32_t b) { return tea4(a, b); }
void orc_lcg_stream(uint32_t seed, int n, uint32_t* out_lcg, float* out_rnd)
{
    uint32_t s = seed;
    for (int i = 0; i < n; i++) { uint32_t t = s; out_lcg[i] = lcg(t); out_rnd[i] = rnd(s); }
}
void orc_random_stream(int seed, int n, uint32_t* out_u, float* out_f)
{
    Random a(seed), b(seed);
    for (int i = 0; i < n; i++) { out_u[i] = a.Rand(); out_f[i] = b.Randf(); }
}
void orc_build_cdf(int w, int h, const fovpt_float4* data, float* pdfX, float* cdfX, float* pdfY, float* cdfY)
{ BuildCDF(w, h, (const f4*)data, pdfX, cdfX, pdfY, cdfY); }

void orc_probe_sample(const fovpt_probe* p, int seed, int n, float* dir3, float* color3, float* pdf)
{
    ProbeH P; P.width = p->width; P.height = p->height; P.data = (const f4*)p->data;
    P.pdfX = p->pdfValuesX; P.cdfX = p->cdfValuesX; P.pdfY = p->pdfValuesY; P.cdfY = p->cdfValuesY;
    Random r(seed);
    for (int i = 0; i < n; i++) {
        f3 d, c; float pd;
        ProbeSample(P, d, c, pd, r);
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
        color3[3 * i] = c.x; color3[3 * i + 1] = c.y; color3[3 * i + 2] = c.z;
        pdf[i] = pd;
    }
}
void orc_probe_dir_to_uv(int n, const float* dir3, float* uv2)
{
    for (int i = 0; i < n; i++) {
        f2 uv = ProbeDirToUV(mk3(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]));
        uv2[2 * i] = uv.x; uv2[2 * i + 1] = uv.y;
    }
}
/* one BSDFSample + BSDFPdf + BSDFEval per row: inputs N (unit), view (unit), etaI, etaO, seed */
void orc_bsdf_table(const fovpt_material* mat, int n, const float* N3, const float* view3, const float* albedo3,
                    const float* etaI, const float* etaO, const int* seeds,
                    float* light3, float* pdf, int* type, float* eval3, float* pdf_again, uint32_t* rng_after)
{
    for (int i = 0; i < n; i++) {
        f3 N = mk3(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2]);
        f3 view = mk3(view3[3 * i], view3[3 * i + 1], view3[3 * i + 2]);
        f3 alb = mk3(albedo3[3 * i], albedo3[3 * i + 1], albedo3[3 * i + 2]);
        f3 u, v;
        BasisFromVector(N, &u, &v);
        Random r(seeds[i]);
        f3 L = mk3(0.f); float p = 0.f; BSDFType ty = eReflected;
        BSDFSample(*mat, etaI[i], etaO[i], u, v, N, view, L, p, ty, r);
        f3 f = mk3(0.f); float p2 = 0.f;
        if (p > 0.0f) {
            f = BSDFEval(*mat, alb, etaI[i], etaO[i], N, view, L);
            p2 = BSDFPdf(*mat, etaI[i], etaO[i], N, view, L);
        }
        light3[3 * i] = L.x; light3[3 * i + 1] = L.y; light3[3 * i + 2] = L.z;
        pdf[i] = p; type[i] = (int)ty;
        eval3[3 * i] = f.x; eval3[3 * i + 1] = f.y; eval3[3 * i + 2] = f.z;
        pdf_again[i] = p2;
        rng_after[2 * i] = r.seed1; rng_after[2 * i + 1] = r.seed2;
    }
}
void orc_make_color(int n, const float* rgb3, uint32_t* out)
{
    for (int i = 0; i < n; i++) out[i] = make_color(reinhardToneMap(mk3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]) * 16.0f, 1.0f));
}
/* sutil::Camera::UVWFrame, sutil/Camera.cpp:32-44 */
void orc_camera_uvw(const float* eye, const float* lookat, const float* up, float fovY, float aspect, float* U, float* V, float* W)
{
    f3 w = mk3(lookat[0], lookat[1], lookat[2]) - mk3(eye[0], eye[1], eye[2]);
    float wlen = length(w);
    f3 u = normalize(cross(w, mk3(up[0], up[1], up[2])));
    f3 v = normalize(cross(u, w));
    float vlen = wlen * tanf(0.5f * fovY * 3.14159265358979323846f / 180.0f);
    v *= vlen;
    float ulen = vlen * aspect;
    u *= ulen;
    U[0] = u.x; U[1] = u.y; U[2] = u.z; V[0] = v.x; V[1] = v.y; V[2] = v.z; W[0] = w.x; W[1] = w.y; W[2] = w.z;
}
/* closest-hit / occlusion queries for intersection-contract tests */
void orc_trace(void* scene, int n, const float* o3, const float* d3, int brute, uint32_t* prim, float* tuv3, uint8_t* occluded)
{
    Scene& S = *(Scene*)scene;
    for (int i = 0; i < n; i++) {
        f3 o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), d = mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        Hit h = trace_closest(S, o, d, kTmin, kTmax, brute != 0);
        prim[i] = h.prim; tuv3[3 * i] = h.t; tuv3[3 * i + 1] = h.u; tuv3[3 * i + 2] = h.v;
        occluded[i] = trace_occluded(S, o, d, kTmin, kTmax, brute != 0) ? 1 : 0;
    }
}
/* scalar math under the current math mode (for detmath-vs-libm and GPU bit-compat tests) */
void orc_math(int op, size_t n, const float* a, const float* b, float* out)
{
    for (size_t i = 0; i < n; i++) {
        switch (op) {
        case FOVPT_OP_SIN: out[i] = m_sinf(a[i]); break;
        case FOVPT_OP_COS: out[i] = m_cosf(a[i]); break;
        case FOVPT_OP_ACOS: out[i] = m_acosf(a[i]); break;
        case FOVPT_OP_ATAN2: out[i] = m_atan2f(a[i], b[i]); break;
        case FOVPT_OP_LOG: out[i] = m_logf(a[i]); break;
        case FOVPT_OP_POW: out[i] = m_powf(a[i], b[i]); break;
        case FOVPT_OP_SQRT: out[i] = sqrtf(a[i]); break;
        case FOVPT_OP_DIV: out[i] = a[i] / b[i]; break;
        case FOVPT_OP_RSQRTD: out[i] = (float)(1.0 / (double)sqrtf(a[i])); break;
        case FOVPT_OP_HALFPLUS: out[i] = (float)(0.5 + (double)a[i]); break;
        default: out[i] = 0.0f;
        }
    }
}
/* ---- the same flat entry points as oracle/ref_shim.cpp offers over the reference's own headers (tests/test_ref_pin_cpu.py) ---- */
void orc_sample2d_stream(int seed, int n, float* out2, uint32_t* state_after2)
{
    Random r(seed);
    for (int i = 0; i < n; i++) Sample2D(r, out2[2 * i], out2[2 * i + 1]);
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}
void orc_basis_from_vector(int n, const float* w3, float* u3, float* v3)
{
    for (int i = 0; i < n; i++) {
        f3 u, v;
        BasisFromVector(mk3(w3[3 * i], w3[3 * i + 1], w3[3 * i + 2]), &u, &v);
        u3[3 * i] = u.x; u3[3 * i + 1] = u.y; u3[3 * i + 2] = u.z;
        v3[3 * i] = v.x; v3[3 * i + 1] = v.y; v3[3 * i + 2] = v.z;
    }
}
void orc_safe_normalize(int n, const float* a3, float* out3)
{
    for (int i = 0; i < n; i++) {
        const f3 r = SafeNormalize(mk3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]));
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
void orc_luminance(int n, const float* rgba4, float* out)
{
    for (int i = 0; i < n; i++) { f4 c = {rgba4[4 * i], rgba4[4 * i + 1], rgba4[4 * i + 2], rgba4[4 * i + 3]}; out[i] = Luminance(c); }
}
void orc_uniform_hemisphere(int seed, int n, float* out3, uint32_t* state_after2)
{
    Random r(seed);
    for (int i = 0; i < n; i++) {
        const f3 d = UniformSampleHemisphere(r);
        out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
    }
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}
void orc_cosine_hemisphere(int n, const float* u2, float* out3)
{
    for (int i = 0; i < n; i++) {
        const f3 d = CosineSampleHemisphere(u2[2 * i], u2[2 * i + 1]);
        out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
    }
}
void orc_probe_sample2(int w, int h, const float* data4, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY,
                       int seed, int n, float* dir3, float* color3, float* pdf, uint32_t* state_after2)
{
    ProbeH P; P.width = w; P.height = h; P.data = (const f4*)data4;
    P.pdfX = pdfX; P.cdfX = cdfX; P.pdfY = pdfY; P.cdfY = cdfY;
    Random r(seed);
    for (int i = 0; i < n; i++) {
        f3 d, c; float pd;
        ProbeSample(P, d, c, pd, r);
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
        color3[3 * i] = c.x; color3[3 * i + 1] = c.y; color3[3 * i + 2] = c.z;
        pdf[i] = pd;
    }
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}
void orc_probe_uv_to_dir(int n, const float* uv2, float* dir3)
{
    for (int i = 0; i < n; i++) {
        f2 uv = {uv2[2 * i], uv2[2 * i + 1]};
        const f3 d = ProbeUVToDir(uv);
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
    }
}
void orc_probe_eval(int w, int h, const float* data4, int n, const float* uv2, float* rgba4)
{
    ProbeH P; P.width = w; P.height = h; P.data = (const f4*)data4; P.pdfX = P.cdfX = P.pdfY = P.cdfY = nullptr;
    for (int i = 0; i < n; i++) {
        f2 uv = {uv2[2 * i], uv2[2 * i + 1]};
        const f4 c = ProbeEval(P, uv);
        rgba4[4 * i] = c.x; rgba4[4 * i + 1] = c.y; rgba4[4 * i + 2] = c.z; rgba4[4 * i + 3] = c.w;
    }
}
void orc_lower_bound(const float* array, int lower, int upper, int n, const float* values, int* out)
{
    for (int i = 0; i < n; i++) out[i] = LowerBound(array, lower, upper, values[i]);
}
void orc_make_color_raw(int n, const float* rgb3, uint32_t* out)           /* make_color alone, no tone mapping */
{
    for (int i = 0; i < n; i++) out[i] = make_color(mk3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
}
void orc_to_srgb(int n, const float* rgb3, float* out3)
{
    for (int i = 0; i < n; i++) {
        const f3 c = toSRGB(mk3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
        out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z;
    }
}
void orc_quantize8(int n, const float* x, uint8_t* out)
{
    for (int i = 0; i < n; i++) out[i] = (uint8_t)quantizeUnsigned8Bits(x[i]);
}
/* op: 0 normalize(a), 1 cross(a,b), 2 a/s, 3 lerp(a,b,s), 4 faceforward(a,b,a), 5 clamp(a,0,10), 6 a*b, 7 s-a, 8 a/b */
void orc_vec3_op(int op, int n, const float* a3, const float* b3, const float* s, float* out3)
{
    for (int i = 0; i < n; i++) {
        const f3 a = mk3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
        const f3 b = b3 ? mk3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]) : mk3(0.f);
        const float t = s ? s[i] : 0.f;
        f3 r = mk3(0.f);
        switch (op) {
        case 0: r = normalize(a); break;
        case 1: r = cross(a, b); break;
        case 2: r = a / t; break;
        case 3: r = lerp3(a, b, t); break;
        case 4: r = faceforward(a, b, a); break;
        case 5: r = clamp3(a, 0.0f, 10.0f); break;
        case 6: r = a * b; break;
        case 7: r = t - a; break;
        case 8: r = a / b; break;
        }
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
void orc_vec3_dot_length(int n, const float* a3, const float* b3, float* dot_out, float* len_out)
{
    for (int i = 0; i < n; i++) {
        const f3 a = mk3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b = mk3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        dot_out[i] = dot(a, b); len_out[i] = length(a);
    }
}
float orc_material_ior(float eta, float specular)
{
    Material m; memset(&m, 0, sizeof(m)); m.eta = eta; m.specular = specular;
    return GetIndexOfRefraction(m);
}
void orc_tex2d(const fovpt_texture_desc* t, int n, const float* uv2, float* rgba4)
{
    Tex X; X.w = t->width; X.h = t->height; X.px.assign(t->pixel, t->pixel + (size_t)X.w * X.h);
    for (int i = 0; i < n; i++) {
        f4 c = tex2d(X, uv2[2 * i], uv2[2 * i + 1]);
        rgba4[4 * i] = c.x; rgba4[4 * i + 1] = c.y; rgba4[4 * i + 2] = c.z; rgba4[4 * i + 3] = c.w;
    }
}

}  // extern "C"
