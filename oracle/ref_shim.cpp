/*
 * ref_shim.cpp -- the REFERENCE ITSELF, compiled on the host, behind flat C entry points.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  This file contains no reference
 * code: it #includes the reference's own host-callable headers from where they lie under
 * /root/reference and forwards arrays to the functions they define.  It is built by
 * `make -C oracle ref` into oracle/_ref/libfovpt_ref.so (git-ignored, never shipped) and only
 * exists to PIN oracle/fovpt_oracle.cpp: tests/golden/make_ref_golden.py calls it to write
 * tests/golden/ref_vectors.npz, and tests/test_ref_pin_cpu.py holds the oracle to those vectors
 * (and, when the .so is present, to the live reference on wider random sweeps).
 *
 * What compiles here, verbatim, with the real CUDA vector headers that ship in the image
 * (triton/backends/nvidia/include: vector_types.h, vector_functions.h, cuda_runtime.h -- no
 * stand-in headers are written):
 *     PT_sv5_/maths.h          Random, BasisFromVector, SafeNormalize, Luminance, hemisphere samplers
 *     PT_sv5_/sample.h         Sample2D
 *     PT_sv5_/Probe.cuh        Probe, ProbeDirToUV, ProbeUVToDir, ProbeEval, ProbePdf, LowerBound, ProbeSample
 *     PT_sv5_/Material.h       Material (ctor defaults, layout), GetIndexOfRefraction
 *     PT_sv5_/Model.h/.cpp     TriangleMesh / Texture / Model, loadOBJ, addBox (tinyobjloader + stb_image
 *                              from the reference's support/ directory)
 *     cuda/random.h            tea<4>, lcg, rnd
 *     cuda/helpers.h           toSRGB, quantizeUnsigned8Bits, make_color
 *     sutil/vec_math.h         float2/3/4 operators, normalize, cross, dot, lerp, clamp, faceforward
 *     sutil/Camera.{h,cpp}     Camera::UVWFrame
 *     support/stb/stb_image.h  stbi_loadf (loadProbe, main.cpp:161-171), stbi_load (loadTexture)
 * `Material.h:3` includes "Maths.h": the reference was written on a case-insensitive file system where
 * that is PT_sv5_/maths.h; the recipe reproduces that with a clang VFS overlay (-ivfsoverlay, a redirect
 * to the reference's own file, not a substitute).
 *
 * What does NOT compile here and therefore stays unpinned by the reference:
 *     PT_sv5_/Disney.cuh       includes LaunchParams.h -> <optix.h>  (OptiX SDK, absent)
 *     PT_sv5_/Probe.h          includes CUDABuffer.h   -> <optix.h>, <optix_stubs.h>
 *     PT_sv5_/deviceProgram.cu includes <optix_device.h>
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "PT_sv5_/maths.h"
#include "PT_sv5_/sample.h"
#include "PT_sv5_/Probe.cuh"
#include "PT_sv5_/Material.h"
#include "PT_sv5_/Model.h"
#include "cuda/random.h"
#include "cuda/helpers.h"
#include <sutil/Camera.h>

// loadProbe / loadTexture decode through stb_image; Model.cpp includes the header without the
// implementation (the application provides it in another translation unit, sutil.cpp).
#define STB_IMAGE_IMPLEMENTATION
#include "support/stb/stb_image.h"

extern "C" {

/* ---- cuda/random.h, maths.h:170-227, sample.h:253-259 ---------------------------------- */
uint32_t ref_tea4(uint32_t a, uint32_t b) { return tea<4>(a, b); }

void ref_lcg_stream(uint32_t seed, int n, uint32_t* out_lcg, float* out_rnd)
{
    unsigned int s = seed;
    for (int i = 0; i < n; i++) { unsigned int t = s; out_lcg[i] = lcg(t); out_rnd[i] = rnd(s); }
}

void ref_random_stream(int seed, int n, uint32_t* out_u, float* out_f)
{
    Random a(seed), b(seed);
    for (int i = 0; i < n; i++) { out_u[i] = a.Rand(); out_f[i] = b.Randf(); }
}

void ref_sample2d_stream(int seed, int n, float* out2, uint32_t* state_after2)
{
    Random r(seed);
    for (int i = 0; i < n; i++) Sample2D(r, out2[2 * i], out2[2 * i + 1]);
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}

/* ---- maths.h:94-108,144-156,165-168,243-277 -------------------------------------------- */
void ref_basis_from_vector(int n, const float* w3, float* u3, float* v3)
{
    for (int i = 0; i < n; i++) {
        float3 u, v;
        BasisFromVector(make_float3(w3[3 * i], w3[3 * i + 1], w3[3 * i + 2]), &u, &v);
        u3[3 * i] = u.x; u3[3 * i + 1] = u.y; u3[3 * i + 2] = u.z;
        v3[3 * i] = v.x; v3[3 * i + 1] = v.y; v3[3 * i + 2] = v.z;
    }
}

void ref_safe_normalize(int n, const float* a3, float* out3)
{
    for (int i = 0; i < n; i++) {
        const float3 r = SafeNormalize(make_float3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]));
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}

void ref_luminance(int n, const float* rgba4, float* out)
{
    for (int i = 0; i < n; i++) out[i] = Luminance(make_float4(rgba4[4 * i], rgba4[4 * i + 1], rgba4[4 * i + 2], rgba4[4 * i + 3]));
}

void ref_uniform_hemisphere(int seed, int n, float* out3, uint32_t* state_after2)
{
    Random r(seed);
    for (int i = 0; i < n; i++) {
        const float3 d = UniformSampleHemisphere(r);
        out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
    }
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}

void ref_cosine_hemisphere(int n, const float* u2, float* out3)
{
    for (int i = 0; i < n; i++) {
        const float3 d = CosineSampleHemisphere(u2[2 * i], u2[2 * i + 1]);
        out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
    }
}

/* ---- Probe.cuh ---------------------------------------------------------------------------- */
static Probe make_probe(int w, int h, const float* data4, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY)
{
    Probe p;
    p.width = w; p.height = h;
    p.data = (Color*)data4;
    p.offset = make_float3(0.f, 0.f, 0.f);
    p.pdfValuesX = (float*)pdfX; p.cdfValuesX = (float*)cdfX;
    p.pdfValuesY = (float*)pdfY; p.cdfValuesY = (float*)cdfY;
    return p;
}

void ref_probe_sample(int w, int h, const float* data4, const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY,
                      int seed, int n, float* dir3, float* color3, float* pdf, uint32_t* state_after2)
{
    const Probe P = make_probe(w, h, data4, pdfX, cdfX, pdfY, cdfY);
    Random r(seed);
    for (int i = 0; i < n; i++) {
        float3 d, c; float pd;
        ProbeSample(P, d, c, pd, r);
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
        color3[3 * i] = c.x; color3[3 * i + 1] = c.y; color3[3 * i + 2] = c.z;
        pdf[i] = pd;
    }
    state_after2[0] = r.seed1; state_after2[1] = r.seed2;
}

void ref_probe_dir_to_uv(int n, const float* dir3, float* uv2)
{
    for (int i = 0; i < n; i++) {
        const float2 uv = ProbeDirToUV(make_float3(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]));
        uv2[2 * i] = uv.x; uv2[2 * i + 1] = uv.y;
    }
}

void ref_probe_uv_to_dir(int n, const float* uv2, float* dir3)
{
    for (int i = 0; i < n; i++) {
        const float3 d = ProbeUVToDir(make_float2(uv2[2 * i], uv2[2 * i + 1]));
        dir3[3 * i] = d.x; dir3[3 * i + 1] = d.y; dir3[3 * i + 2] = d.z;
    }
}

void ref_probe_eval(int w, int h, const float* data4, int n, const float* uv2, float* rgba4)
{
    const Probe P = make_probe(w, h, data4, nullptr, nullptr, nullptr, nullptr);
    for (int i = 0; i < n; i++) {
        const float4 c = ProbeEval(P, make_float2(uv2[2 * i], uv2[2 * i + 1]));
        rgba4[4 * i] = c.x; rgba4[4 * i + 1] = c.y; rgba4[4 * i + 2] = c.z; rgba4[4 * i + 3] = c.w;
    }
}

void ref_lower_bound(const float* array, int lower, int upper, int n, const float* values, int* out)
{
    for (int i = 0; i < n; i++) out[i] = LowerBound(array, lower, upper, values[i]);
}

/* ---- cuda/helpers.h:35-62 ---------------------------------------------------------------- */
void ref_make_color(int n, const float* rgb3, uint32_t* out)
{
    for (int i = 0; i < n; i++) {
        const uchar4 c = make_color(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
        out[i] = (uint32_t)c.x | ((uint32_t)c.y << 8) | ((uint32_t)c.z << 16) | ((uint32_t)c.w << 24);
    }
}

void ref_to_srgb(int n, const float* rgb3, float* out3)
{
    for (int i = 0; i < n; i++) {
        const float3 c = toSRGB(make_float3(rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]));
        out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z;
    }
}

void ref_quantize8(int n, const float* x, uint8_t* out)
{
    for (int i = 0; i < n; i++) out[i] = quantizeUnsigned8Bits(x[i]);
}

/* ---- sutil/vec_math.h (the operators the path uses) ----------------------------------------- */
/* op: 0 normalize(a), 1 cross(a,b), 2 a/s (float3 / float, :487), 3 lerp(a,b,s), 4 faceforward(a, b, a),
 *     5 clamp(a, 0, 10), 6 a*b, 7 s - a (:444), 8 a / b (component-wise :483) */
void ref_vec3_op(int op, int n, const float* a3, const float* b3, const float* s, float* out3)
{
    for (int i = 0; i < n; i++) {
        const float3 a = make_float3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
        const float3 b = b3 ? make_float3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]) : make_float3(0.f);
        const float t = s ? s[i] : 0.f;
        float3 r = make_float3(0.f);
        switch (op) {
        case 0: r = normalize(a); break;
        case 1: r = cross(a, b); break;
        case 2: r = a / t; break;
        case 3: r = lerp(a, b, t); break;
        case 4: r = faceforward(a, b, a); break;
        case 5: r = clamp(a, 0.0f, 10.0f); break;
        case 6: r = a * b; break;
        case 7: r = t - a; break;
        case 8: r = a / b; break;
        }
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
void ref_vec3_dot_length(int n, const float* a3, const float* b3, float* dot_out, float* len_out)
{
    for (int i = 0; i < n; i++) {
        const float3 a = make_float3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
        const float3 b = make_float3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
        dot_out[i] = dot(a, b); len_out[i] = length(a);
    }
}

/* ---- sutil/Camera.cpp:32-44 ---------------------------------------------------------------- */
void ref_camera_uvw(const float* eye, const float* lookat, const float* up, float fovY, float aspect, float* U, float* V, float* W)
{
    sutil::Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                      make_float3(up[0], up[1], up[2]), fovY, aspect);
    float3 u, v, w;
    cam.UVWFrame(u, v, w);
    U[0] = u.x; U[1] = u.y; U[2] = u.z; V[0] = v.x; V[1] = v.y; V[2] = v.z; W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

/* ---- Material.h, Model.h: layout and constructor defaults ----------------------------------- */
/* out: sizeof(Material), then the bytes of a default-constructed Material */
int ref_material_default(uint8_t* out, int cap)
{
    Material m;
    if (cap < (int)sizeof(Material)) return -(int)sizeof(Material);
    memcpy(out, &m, sizeof(Material));
    return (int)sizeof(Material);
}
float ref_material_ior(float eta, float specular)
{
    Material m; m.eta = eta; m.specular = specular;
    return m.GetIndexOfRefraction();
}
/* offsets of every Material member, in declaration order (19 entries) */
int ref_material_offsets(int* out, int cap)
{
    const int o[] = {
        (int)offsetof(Material, emission), (int)offsetof(Material, color), (int)offsetof(Material, absorption),
        (int)offsetof(Material, eta), (int)offsetof(Material, metallic), (int)offsetof(Material, subsurface),
        (int)offsetof(Material, specular), (int)offsetof(Material, roughness), (int)offsetof(Material, specularTint),
        (int)offsetof(Material, anisotropic), (int)offsetof(Material, sheen), (int)offsetof(Material, sheenTint),
        (int)offsetof(Material, clearcoat), (int)offsetof(Material, clearcoatGloss), (int)offsetof(Material, transmission),
        (int)offsetof(Material, bump), (int)offsetof(Material, bumpTile), (int)offsetof(Material, flags),
    };
    const int n = (int)(sizeof(o) / sizeof(o[0]));
    for (int i = 0; i < n && i < cap; i++) out[i] = o[i];
    return n;
}
int ref_probe_struct(int* out /* sizeof, offsets of width,height,data,offset,pdfX,cdfX,pdfY,cdfY */)
{
    out[0] = (int)sizeof(Probe);
    out[1] = (int)offsetof(Probe, width); out[2] = (int)offsetof(Probe, height); out[3] = (int)offsetof(Probe, data);
    out[4] = (int)offsetof(Probe, offset); out[5] = (int)offsetof(Probe, pdfValuesX); out[6] = (int)offsetof(Probe, cdfValuesX);
    out[7] = (int)offsetof(Probe, pdfValuesY); out[8] = (int)offsetof(Probe, cdfValuesY);
    return 9;
}
int ref_trianglemesh_default_texture_id() { TriangleMesh m; return m.diffuseTextureID; }

/* ---- Model.cpp: loadOBJ / addBox ---------------------------------------------------------- */
void* ref_load_obj(const char* path)
{
    try { return loadOBJ(std::string(path)); } catch (...) { return nullptr; }
}
void* ref_box_model(const float* center, const float* half_size)
{
    Model* m = new Model;
    Material mat;
    addBox(m, mat, make_float3(center[0], center[1], center[2]), make_float3(half_size[0], half_size[1], half_size[2]));
    return m;
}
void ref_model_free(void* h)
{
    Model* m = (Model*)h;
    if (!m) return;
    // the texels come from stbi_load (malloc), ~Texture would delete[] them
    for (auto* t : m->textures) { if (t->pixel) stbi_image_free(t->pixel); t->pixel = nullptr; }
    delete m;
}
int ref_model_num_meshes(void* h) { return (int)((Model*)h)->meshes.size(); }
int ref_model_num_textures(void* h) { return (int)((Model*)h)->textures.size(); }
/* counts: vertices, normals, texcoords, triangles, diffuseTextureID */
void ref_model_mesh_info(void* h, int k, int* counts5, uint8_t* material104)
{
    const TriangleMesh* t = ((Model*)h)->meshes[k];
    counts5[0] = (int)t->vertex.size(); counts5[1] = (int)t->normal.size(); counts5[2] = (int)t->texcoord.size();
    counts5[3] = (int)t->index.size(); counts5[4] = t->diffuseTextureID;
    memcpy(material104, &t->material, sizeof(Material));
}
void ref_model_mesh_data(void* h, int k, float* vertex3, float* normal3, float* texcoord2, uint32_t* index3)
{
    const TriangleMesh* t = ((Model*)h)->meshes[k];
    for (size_t i = 0; i < t->vertex.size(); i++) { vertex3[3 * i] = t->vertex[i].x; vertex3[3 * i + 1] = t->vertex[i].y; vertex3[3 * i + 2] = t->vertex[i].z; }
    for (size_t i = 0; i < t->normal.size(); i++) { normal3[3 * i] = t->normal[i].x; normal3[3 * i + 1] = t->normal[i].y; normal3[3 * i + 2] = t->normal[i].z; }
    for (size_t i = 0; i < t->texcoord.size(); i++) { texcoord2[2 * i] = t->texcoord[i].x; texcoord2[2 * i + 1] = t->texcoord[i].y; }
    for (size_t i = 0; i < t->index.size(); i++) { index3[3 * i] = t->index[i].x; index3[3 * i + 1] = t->index[i].y; index3[3 * i + 2] = t->index[i].z; }
}
void ref_model_texture_info(void* h, int k, int* wh2)
{
    const Texture* t = ((Model*)h)->textures[k];
    wh2[0] = t->resolution.x; wh2[1] = t->resolution.y;
}
void ref_model_texture_data(void* h, int k, uint32_t* pixels)
{
    const Texture* t = ((Model*)h)->textures[k];
    memcpy(pixels, t->pixel, (size_t)t->resolution.x * t->resolution.y * 4);
}

/* ---- stb_image as loadProbe (main.cpp:161-171) and loadTexture (Model.cpp:87-136) call it ----- */
/* returns 0 on failure; otherwise fills w,h and (if out is non-null and cap suffices) w*h*4 floats */
int ref_stbi_loadf(const char* path, int* w, int* h, float* out, size_t cap_floats)
{
    int n = 0;
    float* p = stbi_loadf(path, w, h, &n, 4);
    if (!p) return 0;
    const size_t need = (size_t)(*w) * (*h) * 4;
    if (out && cap_floats >= need) memcpy(out, p, need * sizeof(float));
    stbi_image_free(p);
    return 1;
}
int ref_stbi_load(const char* path, int* w, int* h, uint8_t* out, size_t cap_bytes)
{
    int n = 0;
    unsigned char* p = stbi_load(path, w, h, &n, STBI_rgb_alpha);
    if (!p) return 0;
    const size_t need = (size_t)(*w) * (*h) * 4;
    if (out && cap_bytes >= need) memcpy(out, p, need);
    stbi_image_free(p);
    return 1;
}

}  // extern "C"
