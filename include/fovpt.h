/*
 * fovpt.h -- C ABI of libfovpt, the MI355X (gfx950) foveated path-tracing launch.
 *
 * This is the drop-in boundary for ONE hot path of the reference renderer
 * (bipul-mohanto/fovPathTracing_optixCodeLatest): the per-frame foveated launch
 *   SampleRenderer::render()            PT_sv5_/SimplePathtracer.cpp:77-214
 *     -> 3 x optixLaunch                PT_sv5_/SimplePathtracer.cpp:148-209
 *       -> __raygen__renderFrame        PT_sv5_/deviceProgram.cu:392-617
 *       -> __closesthit__radiance       PT_sv5_/deviceProgram.cu:619-732
 *       -> miss / occlusion programs    PT_sv5_/deviceProgram.cu:253-300
 * Everything the reference obtained from OptiX for that path (accel build,
 * traversal, SBT, launch) lives behind these entry points.  Plain pointers and
 * sizes only; no C++ or torch types.  Every function returns 0 on success and
 * a negative FOVPT_E_* code on failure; fovpt_last_error() gives the text
 * (the reference throws sutil::Exception, sutil/Exception.h:93-193 -- the C++
 * shim in SimplePathtracer.h converts codes back into std::runtime_error).
 *
 * All structs below are bit-compatible with the reference's device-visible
 * structs as built by nvcc (8-byte aligned int2/uint2/float2, 16-byte float4):
 *   fovpt_material        == Material              PT_sv5_/Material.h:11-70      (104 B)
 *   fovpt_probe           == Probe                 PT_sv5_/Probe.cuh:6-21        ( 64 B)
 *   fovpt_launch_params   == LaunchParams          PT_sv5_/LaunchParams.h:49-91  (248 B)
 */
#ifndef FOVPT_H
#define FOVPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOVPT_OK              0
#define FOVPT_E_INVALID      -1  /* bad argument / inconsistent sizes            */
#define FOVPT_E_DEVICE       -2  /* a HIP runtime call failed                     */
#define FOVPT_E_NO_SCENE     -3  /* launch before fovpt_set_scene                 */
#define FOVPT_E_NO_PROBE     -4  /* launch with a null probe                      */
#define FOVPT_E_NO_FRAME     -5  /* launch with null frame buffers                */
#define FOVPT_E_BVH_DEPTH    -6  /* built hierarchy deeper than the traversal stack */
#define FOVPT_E_NOMEM        -7

/* ---- POD vectors (layout of CUDA vector_types.h) --------------------------- */
typedef struct { float x, y, z; } fovpt_float3;
typedef struct { float x, y, z, w; } fovpt_float4;
typedef struct { int32_t x, y; } fovpt_int2;
typedef struct { uint32_t x, y; } fovpt_uint2;
typedef struct { uint32_t x, y, z; } fovpt_uint3;

/* ---- Material: PT_sv5_/Material.h:48-69 ------------------------------------ */
#define FOVPT_MATERIAL_FLAG_SHADOW_CATCHER 1 /* Material.h:9 */
typedef struct fovpt_material {
    fovpt_float3 emission;      /*   0 */
    fovpt_float3 color;         /*  12 */
    fovpt_float3 absorption;    /*  24 */
    float eta;                  /*  36 */
    float metallic;             /*  40 */
    float subsurface;           /*  44 */
    float specular;             /*  48 */
    float roughness;            /*  52 */
    float specularTint;         /*  56 */
    float anisotropic;          /*  60 */
    float sheen;                /*  64 */
    float sheenTint;            /*  68 */
    float clearcoat;            /*  72 */
    float clearcoatGloss;       /*  76 */
    float transmission;         /*  80 */
    float bump;                 /*  84 */
    fovpt_float3 bumpTile;      /*  88 */
    int32_t flags;              /* 100 */
} fovpt_material;               /* 104 */

/* ---- Probe: PT_sv5_/Probe.cuh:6-21 (device pointers) ----------------------- */
typedef struct fovpt_probe {
    int32_t width;              /*  0 */
    int32_t height;             /*  4 */
    fovpt_float4* data;         /*  8 */
    fovpt_float3 offset;        /* 16 */
    uint32_t _pad0;             /* 28 */
    float* pdfValuesX;          /* 32 */
    float* cdfValuesX;          /* 40 */
    float* pdfValuesY;          /* 48 */
    float* cdfValuesY;          /* 56 */
} fovpt_probe;                  /* 64 */

/* ---- LaunchParams: PT_sv5_/LaunchParams.h:49-91 ----------------------------- */
typedef struct fovpt_launch_params {
    struct {
        fovpt_float4* accum_buffer;  /*   0  float4 per pixel, device            */
        uint32_t* frame_buffer;      /*   8  rgba8 per pixel (uchar4), device    */
        fovpt_float4* color_buffer;  /*  16  dead in sv5                         */
        fovpt_float4* normal_buffer; /*  24  dead in sv5                         */
        fovpt_float4* albedo_buffer; /*  32  dead in sv5                         */
        fovpt_int2 size;             /*  40                                      */
        uint32_t subframe_index;     /*  48                                      */
        fovpt_uint3 factor;          /*  52                                      */
        int32_t fillSize;            /*  64                                      */
        uint32_t _pad0;              /*  68                                      */
        fovpt_uint2 c;               /*  72  gaze centre, pixels                 */
        float r_inner;               /*  80                                      */
        float r_outer;               /*  84                                      */
        fovpt_uint2 offset;          /*  88                                      */
        uint32_t redraw;             /*  96                                      */
        uint32_t _pad1;              /* 100                                      */
    } frame;
    struct {
        fovpt_float3 eye;            /* 104 */
        fovpt_float3 U;              /* 116 */
        fovpt_float3 V;              /* 128 */
        fovpt_float3 W;              /* 140 */
    } camera;
    uint32_t samples_per_launch;     /* 152 */
    uint32_t _pad2;                  /* 156 */
    uint64_t traversable;            /* 160  scene handle from fovpt_set_scene   */
    fovpt_probe probe;               /* 168 */
    fovpt_int2 viewportSize;         /* 232  dead in sv5                         */
    float white;                     /* 240  dead in sv5                         */
    uint32_t _pad3;                  /* 244 */
} fovpt_launch_params;               /* 248 */

/* ---- scene upload: replaces buildAccel + createTextures + buildSBT ----------
 * (PT_sv5_/SimplePathtracer.cpp:602-746, 748-799, 534-599).  Host pointers;
 * everything is copied before the call returns.                                */
typedef struct fovpt_mesh_desc {
    const float* vertex;        /* xyz triples, TriangleMesh::vertex  (Model.h:12) */
    const float* normal;        /* xyz triples or NULL; the path never reads them
                                   (deviceProgram.cu:632-634 uses the face normal) */
    const float* texcoord;      /* uv pairs per vertex or NULL        (Model.h:14) */
    const uint32_t* index;      /* 3 per triangle                     (Model.h:15) */
    uint32_t num_vertices;
    uint32_t num_triangles;
    int32_t texture_id;         /* TriangleMesh::diffuseTextureID; <0 = untextured */
    fovpt_material material;
} fovpt_mesh_desc;

typedef struct fovpt_texture_desc {
    const uint32_t* pixel;      /* RGBA8, row-major, Texture::pixel (Model.h:27)   */
    int32_t width, height;      /* Texture::resolution                             */
} fovpt_texture_desc;

/* ---- run-time configuration (the reference's compile-time #defines) ---------
 * Defaults reproduce PT_sv5_ as shipped: FOV_ON, radii 74/241
 * (SimplePathtracer.cpp:20-23), spp 8/16/32 (:142,170,193), uniform spp 4 (:95),
 * depth cap 4 (deviceProgram.cu:515), accumulate off (:565-581).               */
typedef struct fovpt_config {
    int32_t uniform;            /* 1 = FOV_OFF branch (SimplePathtracer.cpp:85-131) */
    int32_t r_inner;            /* inner_radius                                    */
    int32_t r_outer;            /* outer_radius                                    */
    int32_t spp_periphery;      /* samples_per_launch of pass P                    */
    int32_t spp_middle;         /* ... pass M                                      */
    int32_t spp_fovea;          /* ... pass F                                      */
    int32_t spp_uniform;        /* ... FOV_OFF                                     */
    int32_t max_depth;          /* depth cap, deviceProgram.cu:515                 */
    int32_t accumulate;         /* 1 = PT_sv4_vmv2 clamp(0,10)+running mean
                                   (OtherProjects_02_latest/PT_sv4_vmv2/deviceProgram.cu:545-553) */
    int32_t rank;               /* multi-GPU tile shard: this handle renders the   */
    int32_t world;              /* launch-index tiles t with owner(t) == rank      */
    int32_t tile_w, tile_h;     /* launch-index tile, default 8 x 4
                                   (sutil/WorkDistribution.h:47-84 scheme)         */
    int32_t profile;            /* 1 = time each kernel with hipEvents; 2 = the same with every kernel
                                   run ALONE (host synchronisation around it: serialised times)     */
    int32_t write_guides;       /* 1 = also write normal/color/albedo_buffer, the denoiser guides of
                                   PT_sv/deviceProgram.cu:555-557 (commented out in PT_sv5_, :612-614);
                                   not available with shadow-catcher materials         */
    int32_t options;            /* opt-in extensions beyond the reference's behaviour (0 = PT_sv5_ as shipped):
                                   FOVPT_OPT_SKY_MISS, FOVPT_OPT_RUSSIAN_ROULETTE                     */
    int32_t frames_in_flight;   /* jobs (frames issued without fovpt_synchronize between them) whose main chains --
                                   generate, closest hit, shade, one after the other -- may run BESIDE each other:
                                   0 = the library's default (2), 1 = one frame at a time (lowest latency per frame),
                                   2 = two (highest throughput: each chain fills the other's gaps; a frame then takes
                                   about twice as long from first to last kernel); 3 and 4 are accepted and measured slower
                                   than 2 (a context has FOVPT_LANES = 2 stream pairs unless the environment says more; a
                                   larger value means all of them).  Results do not depend on it: resolves
                                   run in issue order, and fovpt_stream() is ordered behind every finished frame.
                                   (Path state and queues exist once per state set; jobs of up to 16 Mi sample slots rotate
                                   through twice as many sets as stream pairs -- FOVPT_SETS -- so that a frame's head does
                                   not wait for the resolve of the frame two before it.)                                 */
    int32_t chains_per_frame;   /* 0 / 1 = a frame is one chain of dependent launches; 2 = every frame is rendered as TWO
                                   independent chains over halves of its sample slots (each with four of the eight queue
                                   shards, on its own stream pair) and resolved once: frames issued back to back then
                                   follow each other as closely as with frames_in_flight = 2 while each is finished in
                                   about half the time after its issue (one frame in flight instead of two).  It does not
                                   shorten a frame for a caller that synchronises after every frame (measured), and costs
                                   2-5 % of the throughput on multi-million-triangle scenes.  With 2, frames are issued one
                                   at a time (frames_in_flight is ignored).  Results do not depend on it.              */
} fovpt_config;

/* fovpt_config.options.  Both are NON-PARITY modes with respect to the reference (it has neither); the CPU oracle
 * implements them identically, so the GPU is still checked bit for bit against it.
 *  SKY_MISS          the multiple-importance-sampling counterpart of SampleLights that PT_sv5_ carries commented out
 *                    in __miss__radiance (deviceProgram.cu:259-269, :273-278): a SECONDARY ray that escapes adds
 *                    w * ProbeEval(dir) * pathThroughput, w = bsdfPdf / (bsdfPdf + ProbePdf(dir)) (Probe.cuh:69-93), and the
 *                    path loop counts that segment although it is DONE (the reference's break at :515 would drop it).
 *  RUSSIAN_ROULETTE  the //!TODO of deviceProgram.cu:518-520: from the second bounce on a path survives a shaded hit
 *                    with probability q = clamp(max component of pathThroughput, 0.05, 1), decided by one more
 *                    Random::Randf() after BSDFSample's draws; survivors carry pathThroughput / q.                */
#define FOVPT_OPT_SKY_MISS 1
#define FOVPT_OPT_RUSSIAN_ROULETTE 2

typedef struct fovpt_stats {
    uint64_t radiance_rays;     /* closest-hit rays traced since last reset        */
    uint64_t shadow_rays;       /* occlusion rays traced                           */
    uint64_t paths;             /* camera paths started                            */
    uint64_t frames;            /* fovpt_render / fovpt_launch calls               */
    /* per-kernel device time, only filled when config.profile = 1 or 2           */
    double ms_generate, ms_trace, ms_shade, ms_shadow, ms_resolve;
    uint64_t n_trace_launches;  /* closest-hit kernel launches behind ms_trace     */
    uint64_t n_shadow_launches;
    /* scene facts */
    uint64_t num_triangles, num_bvh_nodes, bvh_max_depth, bvh_bytes, tri_bytes;
    double ms_bvh_build;
} fovpt_stats;

typedef struct fovpt_frame_ptrs {   /* what resize() allocates, SimplePathtracer.cpp:242-260 */
    uint32_t* frame_buffer;
    fovpt_float4* accum_buffer;
    fovpt_float4* color_buffer;
    fovpt_float4* normal_buffer;
    fovpt_float4* albedo_buffer;
} fovpt_frame_ptrs;

typedef struct fovpt_ctx fovpt_ctx;

/* SampleRenderer ctor minus the scene: initOptix/createContext (SimplePathtracer.cpp:310-340). */
int fovpt_create(fovpt_ctx** out, int device);
void fovpt_destroy(fovpt_ctx* ctx);
const char* fovpt_last_error(const fovpt_ctx* ctx);  /* ctx may be NULL: last create error */

/* buildAccel + createTextures + buildSBT.  *traversable_out is what the reference
 * stores in launchParams.traversable (SimplePathtracer.cpp:61).  At most 2^26
 * triangles in total (FOVPT_E_INVALID beyond: the traversal addresses the 48-byte
 * triangle records and 128-byte nodes with 32-bit byte offsets); a hierarchy deeper
 * than 21 four-wide levels is refused with FOVPT_E_BVH_DEPTH.                      */
int fovpt_set_scene(fovpt_ctx* ctx, const fovpt_mesh_desc* meshes, int num_meshes,
                    const fovpt_texture_desc* textures, int num_textures,
                    uint64_t* traversable_out);

/* CUDAProbeData::createBuffer (Probe.h:102-124): uploads the 5 arrays, fills *probe_out
 * with device pointers exactly as setProbe does (SimplePathtracer.cpp:292-308).   */
int fovpt_set_probe(fovpt_ctx* ctx, int width, int height, const fovpt_float4* data,
                    const float* pdfValuesX, const float* cdfValuesX,
                    const float* pdfValuesY, const float* cdfValuesY,
                    const fovpt_float3* offset, fovpt_probe* probe_out);

/* The same, with ProbeData::BuildCDF (Probe.h:29-77) done on the device: uploads only the texels and
 * builds pdf/cdf tables in HBM with the reference's left-to-right fp32 summation order (rows in
 * parallel).  Bit-identical to fovpt_probe_build_cdf + fovpt_set_probe.                          */
int fovpt_set_probe_data(fovpt_ctx* ctx, int width, int height, const fovpt_float4* data,
                         const fovpt_float3* offset, fovpt_probe* probe_out);

/* SampleRenderer::resize (SimplePathtracer.cpp:228-274): (re)allocates the five
 * full-frame buffers.  No-op returning FOVPT_OK with *out untouched when w or h is 0. */
int fovpt_resize(fovpt_ctx* ctx, int width, int height, fovpt_frame_ptrs* out);

int fovpt_get_config(const fovpt_ctx* ctx, fovpt_config* out);
int fovpt_set_config(fovpt_ctx* ctx, const fovpt_config* cfg);

/* One optixLaunch of the raygen program over a width x height grid with the given
 * parameters (SimplePathtracer.cpp:148-157).  Asynchronous on fovpt_stream().
 * With world > 1: a pixel whose last writer in THIS launch is another rank's launch index is
 * written as zero, a pixel this launch does not write is left untouched on every rank -- so a
 * frame composed of several launches (P, M, F) sums over the ranks to the single-GPU frame,
 * provided the ranks other than 0 start the frame from a cleared target (fovpt_render does
 * that clearing itself).                                                            */
int fovpt_launch(fovpt_ctx* ctx, const fovpt_launch_params* lp, uint32_t width, uint32_t height);

/* SampleRenderer::render() (SimplePathtracer.cpp:77-214): fills in the per-pass
 * fields of *lp exactly as the reference mutates its public launchParams (factor,
 * fillSize, radii, offset, redraw, samples_per_launch; ++subframe_index), and runs
 * the three passes (or the FOV_OFF pass) as ONE fused wavefront job.  Silently
 * returns FOVPT_OK when lp->frame.size.x == 0 (:81-82).  Asynchronous.            */
int fovpt_render(fovpt_ctx* ctx, fovpt_launch_params* lp);

/* ---- multi-GPU: packed gather of the final framebuffer ----------------------------------
 * New with this library: the reference is single-GPU (SimplePathtracer.cpp:331-340).  With
 * fovpt_config.rank/world every handle renders the launch-index tiles it owns -- interleaved
 * 8 x 4 tiles dealt round-robin, the scheme of the SDK's unused sutil/WorkDistribution.h:47-84
 * -- so the pixels of a frame partition by the owner of their last writer.
 *   fovpt_gather_plan    builds (or reuses) the partition for the frame fovpt_render would
 *                        draw with the current config and lp (frame size, gaze): per rank the
 *                        ascending list of pixel indices it owns; counts_out[r] = their number.
 *                        Every rank computes the same plan.  Synchronises when it rebuilds.
 *   fovpt_gather_pack    copies THIS rank's owned rgba8 words of `frame` into `packed`
 *                        (counts[rank] words), in plan order; asynchronous on fovpt_stream().
 *   fovpt_gather_unpack  scatters `world` packed buffers (rank r's at gathered + r * stride)
 *                        into `frame`; pixels nobody owns are left untouched; asynchronous.
 * Between pack and unpack the caller moves the buffers with RCCL (gather to the root over xGMI:
 * 1/world of the frame per rank instead of a full-frame reduce).  All pointers are device
 * pointers.                                                                                */
int fovpt_gather_plan(fovpt_ctx* ctx, const fovpt_launch_params* lp, uint32_t* counts_out, int counts_len);
int fovpt_gather_pack(fovpt_ctx* ctx, const uint32_t* frame, uint32_t* packed);
int fovpt_gather_unpack(fovpt_ctx* ctx, const uint32_t* gathered, uint32_t stride, uint32_t* frame);

/* ---- multi-GPU: the RCCL transport of that gather, for C / C++ hosts ----------------------
 * One fovpt_ctx per GPU (one process or thread each), fovpt_config.rank / world set on each.
 *   fovpt_comm_get_unique_id  ncclGetUniqueId: one rank creates the 128-byte id and hands it to the others out of band
 *                             (MPI_Bcast, a file, a socket: the host application's business)
 *   fovpt_comm_init           ncclCommInitRank on the context's device; collective over all ranks
 *   fovpt_gather_frame        plan (cached) -> pack -> ncclGroupStart / ncclSend to `root` / on the root ncclRecv from every
 *                             rank / ncclGroupEnd -> on the root unpack into full_frame.  Everything is enqueued on
 *                             fovpt_stream(): asynchronous, ordered behind the frame just rendered, and running beside the
 *                             next frame's rendering.  `frame` is this rank's rgba8 frame (what fovpt_render wrote);
 *                             full_frame (root only; may be `frame` itself) receives the whole image.
 *   fovpt_comm_destroy        ncclCommDestroy (also done by fovpt_destroy)
 * librccl is loaded at run time (dlopen: $FOVPT_RCCL_LIB, librccl.so.1, librccl.so); without it these return
 * FOVPT_E_DEVICE and everything else in this header works.  Replaces nothing in the reference (single-GPU,
 * SimplePathtracer.cpp:331-340); the partition is the scheme of sutil/WorkDistribution.h:47-84.                        */
#define FOVPT_COMM_ID_BYTES 128
int fovpt_comm_get_unique_id(void* id);
int fovpt_comm_init(fovpt_ctx* ctx, const void* id, int rank, int world);
int fovpt_comm_destroy(fovpt_ctx* ctx);
int fovpt_gather_frame(fovpt_ctx* ctx, const fovpt_launch_params* lp, int root, const uint32_t* frame, uint32_t* full_frame);

/* CUDA_SYNC_CHECK() (SimplePathtracer.cpp:212). */
int fovpt_synchronize(fovpt_ctx* ctx);

/* CUDABuffer::download (CUDABuffer.h:82-88): device -> host copy of n_bytes.      */
int fovpt_download(fovpt_ctx* ctx, const void* device_src, void* host_dst, size_t n_bytes);

int fovpt_get_stats(fovpt_ctx* ctx, fovpt_stats* out);   /* synchronises first */
int fovpt_reset_stats(fovpt_ctx* ctx);
/* hipStream_t; SampleRenderer::stream.  Frames complete on it in submission order: work
 * queued on it after fovpt_render / fovpt_launch sees the finished frame and is ordered
 * before the next frame's writes to the render target.                              */
void* fovpt_stream(fovpt_ctx* ctx);

/* ---- host-side helpers that the reference runs on the CPU too ---------------- */
/* ProbeData::BuildCDF (Probe.h:29-77): sequential fp32 accumulation, order preserved. */
int fovpt_probe_build_cdf(int width, int height, const fovpt_float4* data,
                          float* pdfValuesX, float* cdfValuesX,
                          float* pdfValuesY, float* cdfValuesY);
/* sutil::Camera::UVWFrame (sutil/Camera.cpp:32-44). */
int fovpt_camera_uvw(const fovpt_float3* eye, const fovpt_float3* lookat, const fovpt_float3* up,
                     float fovY_degrees, float aspect,
                     fovpt_float3* U, fovpt_float3* V, fovpt_float3* W);

/* ---- Scene ingestion on the host: what loadOBJ returns, PT_sv5_/Model.cpp:138-217 --------------------
 * (with addVertex :49-82 and loadTexture :84-136, i.e. the vendored tinyobjloader with triangulate = true and
 * stbi_load(..., STBI_rgb_alpha) mirrored along y).  Plain host code, no GPU needed.  One mesh per (shape, material id);
 * PNG, JPEG, Truevision TGA and binary PPM textures are decoded, any other format counts as "could not load"
 * (texture id -1, as :129-131).
 * The arrays stay owned by the model; include/Model.h wraps this as `Model* loadOBJ(const std::string&)`.
 * Errors: FOVPT_E_INVALID, text from fovpt_last_error(NULL) ("Could not read OBJ model from ...", :160-162).          */
typedef struct fovpt_model fovpt_model;
typedef struct fovpt_model_mesh {
    const fovpt_float3* vertex;      /* TriangleMesh::vertex                                   */
    const fovpt_float3* normal;      /* TriangleMesh::normal, NULL when the mesh has none      */
    const float* texcoord;           /* TriangleMesh::texcoord as (u, v) pairs, NULL when none */
    const fovpt_uint3* index;        /* TriangleMesh::index                                    */
    uint32_t num_vertices, num_normals, num_texcoords, num_triangles;
    fovpt_material material;         /* reference defaults + Kd -> color, Ke -> emission (:190-191) */
    int32_t diffuse_texture_id;      /* index into the model's textures, or -1                 */
} fovpt_model_mesh;
int fovpt_model_load_obj(const char* obj_file, fovpt_model** out);
/* glTF 2.0 (.gltf with external or data-URI buffers, or a .glb container) -> the same model structure, one mesh per triangle
 * primitive in WORLD space, with the node rules of sutil::Scene (sutil/Scene.cpp:109-442: roots = nodes without a parent,
 * parent * matrix * T * R * S in binary32, nothing below a mesh or camera node, base colour / roughness / metallic factors,
 * emissiveFactor -> emission, base colour texture from a PNG / TGA / PPM file).  Replaces tinygltf + sutil::loadScene for C++
 * callers; include/Model.h wraps it as `Model* loadGLTF(const std::string&)`.                                            */
int fovpt_model_load_gltf(const char* gltf_file, fovpt_model** out);
void fovpt_model_destroy(fovpt_model* model);
int fovpt_model_counts(const fovpt_model* model, int* num_meshes, int* num_textures);
int fovpt_model_get_mesh(const fovpt_model* model, int i, fovpt_model_mesh* out);
int fovpt_model_get_texture(const fovpt_model* model, int i, const uint32_t** pixels, int* width, int* height);

/* The float4 texels loadProbe hands to ProbeData::BuildCDF (PT_sv5_/main.cpp:160-171): what
 * stbi_loadf(file, &w, &h, &n, 4) returns -- Radiance .hdr as it is (RLE and flat scanlines, alpha 1), 8-bit PNG / TGA /
 * PPM through stb's gamma-2.2 conversion.  *texels is malloc'ed, width * height entries, row 0 first; release it with
 * fovpt_image_free.  Errors: FOVPT_E_INVALID with fovpt_last_error(NULL) (the reference does not check stbi_loadf's
 * result and would build the CDF over a null pointer).                                                               */
int fovpt_image_load_float4(const char* file, int* width, int* height, fovpt_float4** texels);
/* stbi_load(file, &w, &h, &n, STBI_rgb_alpha) as loadTexture calls it (PT_sv5_/Model.cpp:106-107) for the formats this library
 * reads -- PNG, JPEG (baseline, extended, progressive; gray, YCbCr, RGB, CMYK, YCCK), Truevision TGA, binary PPM: rgba8, row 0
 * first, bit for bit what the reference's vendored stb_image returns.  *pixels is malloc'ed: fovpt_image_free_rgba8.           */
int fovpt_image_load_rgba8(const char* file, int* width, int* height, uint32_t** pixels);
void fovpt_image_free_rgba8(uint32_t* pixels);
void fovpt_image_free(fovpt_float4* texels);

/* ---- device self-test hook (tests only): evaluates one scalar function on the GPU
 * for n inputs; op codes FOVPT_OP_*.  a,b host arrays (b may be NULL), out host.  */
#define FOVPT_OP_SIN    1
#define FOVPT_OP_COS    2
#define FOVPT_OP_ACOS   3
#define FOVPT_OP_ATAN2  4
#define FOVPT_OP_LOG    5
#define FOVPT_OP_POW    6
#define FOVPT_OP_SQRT   7
#define FOVPT_OP_DIV    8
#define FOVPT_OP_RSQRTD 9   /* (float)(1.0 / (double)sqrtf(a)), maths.h:98 */
#define FOVPT_OP_UNORM8 10  /* texel channel (uint8)a / 255.0f as the shading kernel computes it */
#define FOVPT_OP_HALFPLUS 11 /* (float)(0.5 + (double)a), Disney.cuh Fd90 */
int fovpt_debug_math(fovpt_ctx* ctx, int op, const float* a, const float* b, float* out, size_t n);
/* tests only: the production traversal kernel on a batch of n rays (host arrays, 3 floats per origin / direction): closest
 * hit -> global primitive id (0xffffffff = miss) and (t, u, v); occlusion ray (deviceProgram.cu:224-248) -> 0 / 1.
 * Any output may be NULL.  Synchronises.                                                                              */
int fovpt_debug_trace(fovpt_ctx* ctx, int n, const float* origins3, const float* dirs3, uint32_t* prim_out, float* tuv_out3, uint8_t* occluded_out);
/* tests/diagnostics only: device address and size of an internal buffer ("sq_occ", "counters", "hit", "bvh_nodes", ...) */
int fovpt_debug_buffer(fovpt_ctx* ctx, const char* name, void** ptr, size_t* bytes);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
static_assert(sizeof(fovpt_material) == 104, "Material ABI");
static_assert(sizeof(fovpt_probe) == 64, "Probe ABI");
static_assert(sizeof(fovpt_launch_params) == 248, "LaunchParams ABI");
static_assert(offsetof(fovpt_launch_params, camera) == 104, "LaunchParams ABI");
static_assert(offsetof(fovpt_launch_params, traversable) == 160, "LaunchParams ABI");
static_assert(offsetof(fovpt_launch_params, probe) == 168, "LaunchParams ABI");
#endif

#endif /* FOVPT_H */
