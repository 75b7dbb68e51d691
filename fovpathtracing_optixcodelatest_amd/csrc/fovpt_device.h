// fovpt_device.h -- structures shared by the host API (fovpt_api.hip), the LBVH builder
// (bvh_build.hip) and the wavefront kernels (wavefront.hip).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fovpt.h"

#define FOVPT_WAVE 64
#define FOVPT_BLOCK 256
#ifndef FOVPT_V_STEPSTAT
#define FOVPT_V_STEPSTAT 0         // 1: diagnostic build (tools/stepstat.py, tools/raystat.py)
#endif
#ifndef FOVPT_V_CYCLES
#define FOVPT_V_CYCLES 0           // 1: diagnostic build with s_memtime stamps inside the traversal steps (tools/stepcycles.py)
#endif
#ifndef FOVPT_V_VOTE_ANYHIT
#define FOVPT_V_VOTE_ANYHIT 1      // occlusion rays end their node phases by the same vote as closest-hit rays (wavefront.hip, vote_leaf); 0: when
                                   // every ray of the wave has reached a leaf, as rounds 1-3 (A/B: street occlusion launches 0.755 -> 0.600 ms, atrium 0.250 -> 0.244)
#endif
#ifndef FOVPT_V_MIXED
#define FOVPT_V_MIXED 1            // the pass after a vote is a MIXED step (leaf lanes test triangles, node lanes step, behind one wait: wavefront.hip
                                   // mixed_step); 0: a leaf step in which the node lanes idle (A/B: atrium 0.667 -> 0.663, street 1.444 -> 1.409 with both on)
#endif
#ifndef FOVPT_V_MIXED_ANYHIT
#define FOVPT_V_MIXED_ANYHIT 1     // the same for occlusion rays
#endif
#ifndef FOVPT_V_GEN_OWNED
#define FOVPT_V_GEN_OWNED 1         // a rank of a tile-sharded frame generates over its own tiles only (k_generate<true>)
#endif
#ifndef FOVPT_V_ANYHIT_SORT
#define FOVPT_V_ANYHIT_SORT 0      // 1: occlusion rays visit a node's children nearest first, like closest-hit rays (A/B; the product uses storage order)
#endif
#ifndef FOVPT_LEAF_MAX
#define FOVPT_LEAF_MAX 4          // triangles per BVH leaf (<= 8: three bits in the leaf code)
#endif
#define FOVPT_STACK 64            // traversal stack entries per ray, all in LDS (a wide node leaves <= 3 behind)
#define FOVPT_QUADS_PER_BLOCK (FOVPT_BLOCK / 4)
#ifndef FOVPT_TBLOCK
#define FOVPT_TBLOCK 64           // threads per block of k_traverse (64, 256 or 1024: the rays of a block share one LDS stack array).
                                  // 64 = a workgroup is ONE wave: it gives its slot, registers and 4.3 KB of LDS back the moment it ends (a
                                  // 256-thread block holds all four slots until its slowest wave is through), and with more workgroups than
                                  // slots the hardware's dispatcher is the run-time scheduler that atomics were too slow to be (EXPERIMENTS.md)
#endif
#define FOVPT_TQUADS (FOVPT_TBLOCK / 4)
#ifndef FOVPT_GRID_PER_CU
#define FOVPT_GRID_PER_CU 24        // closest-hit launches: 256-thread units per CU (8 = every wave slot once; 24 = three times as many
                                    // one-wave workgroups as slots, measured best of 8 / 16 / 24 / 32 / 64)
#endif
#ifndef FOVPT_GRID_SHADOW_PER_CU
#define FOVPT_GRID_SHADOW_PER_CU 6
#endif
#define FOVPT_MAX_PASSES 3
#define FOVPT_MAX_ITERS 63         // wavefront iterations per frame (max_depth + catcher pass-throughs)
#define FOVPT_SHARDS 8            // queue shards: one append counter per blockIdx % 8 (~ per XCD)

// One triangle in BVH leaf order, pre-subtracted edges (exactly v1-v0 and v2-v0 in fp32,
// so Moeller-Trumbore gives the same bits as the contract in include/fovpt.h / oracle).
struct alignas(16) TriRec {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z;
    uint32_t prim;                // global primitive id (tie-break key)
    uint32_t mesh;
    uint32_t pad;
};                                // 48 B
static_assert(sizeof(TriRec) == 48, "TriRec");

// 4-wide BVH node: four 32-byte child records.  The traversal gives one ray to four adjacent lanes
// (a quad); lane j of the quad owns child j and fetches exactly its record (two 16-byte loads, the
// quad's eight loads cover the 128-byte node).  code >= 0: index of a wide node; code < 0: leaf,
// ~code = (offset of the first triangle in 16-byte units << 3) | (count-1).  An unused slot holds the degenerate box lo = hi = +inf,
// which no ray passes.
struct alignas(16) BvhChild {
    float lox, loy, loz, hix;
    float hiy, hiz;
    int32_t code;
    uint32_t pad;                 // 0..3, distinct within a node: where closest-hit rays rank this child among children they enter at the
                                  // same distance (the two lowest bits of the traversal's sort key; equal keys would collide on the stack)
};
struct alignas(16) BvhNode4 {
    BvhChild c[4];
};                                // 128 B
static_assert(sizeof(BvhNode4) == 128, "BvhNode4");

struct TexDev {
    const uint32_t* px;
    int32_t w, h;
};

struct MeshDev {                  // the SBT record of the reference (LaunchParams.h:38-47), minus geometry
    fovpt_material material;      // 104 B
    int32_t texture_id;           // <0: none
    int32_t has_texcoord;
    TexDev tex;                   // a copy of textures[texture_id]: one dependent load less on the way to the texels
};                                // 128 B

struct SceneView {
    const BvhNode4* nodes;
    const TriRec* tris;           // leaf order; lives in the nodes' allocation, tri_off bytes behind `nodes`
    const float2* tri_tc;         // 3 per global primitive id (or null)
    const MeshDev* meshes;
    const TexDev* textures;
    uint32_t num_tris;
    uint32_t any_catcher;
    uint32_t tri_off;             // (const char*)tris - (const char*)nodes: one base register reaches both
    uint32_t num_nodes;           // wide nodes of the hierarchy
};


struct PassDev {                  // one optixLaunch worth of parameters
    uint32_t gw, gh;              // launch grid
    uint32_t fx, fy, fz;          // frame.factor
    int32_t fill;                 // frame.fillSize
    uint32_t offx, offy;          // frame.offset
    float r_inner, r_outer;
    uint32_t spp;                 // samples_per_launch
    uint32_t subframe;            // frame.subframe_index
    uint32_t redraw;              // frame.redraw
    uint32_t slot_base;           // first sample slot of this pass
    uint32_t launch_base;         // first launch record of this pass
    uint32_t row0, row1;          // launch rows [row0, row1) handled by this job (a chunk of a large launch)
    uint32_t frame_pass;          // index of this launch within the caller's frame (P 0, M 1, F 2): the tile ownership of
                                  // launch_owned() rotates with it, also when the launch runs as a job of its own (chunks)
};

struct FrameDev {
    PassDev pass[FOVPT_MAX_PASSES];
    int32_t npass;
    int32_t w, h;                 // frame.size
    uint32_t cx, cy;              // frame.c
    float eye[3], U[3], V[3], W[3];
    fovpt_probe probe;            // device pointers
    const uint32_t* guide_x;      // lower_bound guide tables for probe.cdfValuesX / Y, or null
    const uint32_t* guide_y;
    const float4* probe_rec;      // packed {cdfX, pdfX, r, g, b} records per texel (two float4), or null
    int32_t probe_row_mul;        // 0 when all probe rows are identical (constant ambient probe), else 1
    fovpt_float4* accum;
    const fovpt_float4* accum_prev;   // what accumulate mode blends with: accum itself, or its copy from before a chunked launch
    uint32_t* frame;
    fovpt_float4 *g_normal, *g_color, *g_albedo;   // denoiser guide targets (null unless write_guides)
    uint32_t total_slots;
    int32_t max_depth;
    int32_t accumulate;
    int32_t rank, world, tile_w, tile_h;
    int32_t chunked;              // 1: this job is one of several over the same frame
    int32_t zero_holes;           // 1: clear the pixels no launch index of this job writes (a whole frame on a rank other than 0)
    int32_t options;              // fovpt_config.options (FOVPT_OPT_*): opt-in extensions, 0 = the reference's behaviour
    int32_t partition;            // 1: k_shade appends a block's next rays in two direction classes (foveated frames), see block_append2d
};

// Per-sample-slot path state (SoA, 16-B vectors so every access is one dwordx4).
// A radiance-ray queue holds the RAYS, not slot numbers: entry = origin.xyz + sample slot (bits of .w),
// direction.xyz.  Sharded like every queue (FOVPT_SHARDS regions of `cap` entries).  A traversal or shading
// wave reads 16 / 64 consecutive entries with coalesced loads and no indirection.
struct RayQueue {
    float4* o;
    float4* d;
};

struct PathState {
    float4* thr;        // pathThroughput.xyz, rayEta
    uint4* rng;         // Random.seed1, Random.seed2, stateFlags (DONE 1, SECONDARY 2, ALPHA_ONE 4) | depth << 8, unused
    float4* hit;        // t, u, v, tri position in leaf order as bits (0xffffffff = miss)
    float4* rad;        // [slot][depth]: prd.radiance of the segment at that depth (directLight = term 0,
                        // indirectLight = terms 1.. summed in order); one writer per cell
    size_t stride;      // cells per slot (= max_depth)
    float4* alpha;      // prd.alpha contribution of a shadow-catcher primary hit
    float4* guide_n;    // write_guides: prd.normal of the primary hit (deviceProgram.cu:509-512), else null
    float4* guide_a;    // write_guides: prd.albedo of the primary hit
    float4* backplate;  // per launch record: backplate of the last sample (deviceProgram.cu:495)
#if FOVPT_V_STEPSTAT
    uint4* trace;       // diagnostics: node steps of each of the ray's first 16 node phases, one byte each (tools/raysim.py)
#endif
};

// Shadow (occlusion) ray queue, indexed by queue position.
struct ShadowQueue {
    float4* o;          // origin.xyz, slot as bits
    float4* d;          // direction.xyz, target as bits (depth cell of rad, or 0xffffffff = alpha)
    float4* val_vis;    // value added when NOT occluded
    float4* val_occ;    // value added when occluded
};

// Queues are sharded: shard s of a queue with capacity `cap` lives at [s*cap, s*cap + count[s]).
// One returning atomic on ONE word saturates at ~88/us on MI355X, so every producer block appends
// to the counter of shard blockIdx % 8 with one atomic per block-iteration.
// Queue sizes live per SHARD, each shard's block FOVPT_SHARD_STRIDE words away from the next: the blocks
// of one XCD append to one shard, and returning atomics on words of the same memory channel serialise
// (~88 per microsecond), so the eight shards must not share one.
#ifndef FOVPT_SHARD_STRIDE
#define FOVPT_SHARD_STRIDE 128    // uint32 words: 512 B (measured: 0.940 ms/frame; 4 KB: 0.957; all eight in one line: 0.977)
#endif
struct Counters {       // device-resident, zeroed per frame except the stats block
    uint32_t shard[FOVPT_SHARDS][FOVPT_SHARD_STRIDE];   // [s][it]: radiance queue size of iteration it (0 = camera rays);
                                                        // [s][FOVPT_MAX_ITERS + 1 + it]: shadow queue size
    unsigned long long stat_radiance, stat_shadow, stat_paths;
    // diagnostics of a -DFOVPT_V_STEPSTAT=1 build (tools/stepstat.py): per ray kind [closest, any-hit]
    // wave-level node steps, active quads in them, wave-level leaf steps, active quads in them
    unsigned long long diag[2][4];
#if FOVPT_V_CYCLES
    // diagnostics of a -DFOVPT_V_CYCLES=1 build (tools/stepcycles.py): s_memtime ticks (shader cycles) summed per wave
    unsigned long long cyc[2][8][16];   // [ray kind][iteration & 7][field], fields: see struct Cyc in wavefront.hip
    uint32_t hist[2][8][3][64];         // [kind][iteration & 7][node load wait /16 | node step /32 | leaf step /32][bin]
    unsigned long long wtime[8][32768][2];   // [kind * 4 + iteration & 3][wave]: s_memrealtime (100 MHz) at the wave's start and end
#endif
};
static_assert(2 * (FOVPT_MAX_ITERS + 1) <= FOVPT_SHARD_STRIDE, "shard block holds both queues' sizes");
#define FOVPT_CNT_Q(it) (it)                               // word index inside a shard's block
#define FOVPT_CNT_SQ(it) (FOVPT_MAX_ITERS + 1 + (it))

// ---- launchers implemented in wavefront.hip / bvh_build.hip -------------------------------
struct BvhBuildResult {
    BvhNode4* nodes;              // ONE allocation: the emitted nodes, then (256-byte aligned) the triangles; free `nodes` only
    TriRec* tris;
    uint32_t num_nodes;           // wide nodes emitted (breadth-first order, root = 0)
    uint32_t num_refs;            // triangle records behind the nodes: the triangles, or more when triangles were split into references
    uint32_t max_depth;
    uint32_t reinserted;          // 1: reinsertion rounds changed the PLOC tree
    size_t node_bytes, tri_bytes;
};

// flat: 9 floats per triangle (v0,v1,v2), mesh_of_prim: mesh id per triangle.  All device pointers.
// split_budget: references added by spatial splits as a fraction of the triangles (0 = none), see bvh_build.hip.
// reinsert: rounds of reinsertion on the PLOC tree (0 = none, -1 = FOVPT_REINSERT or the build's default).
hipError_t fovpt_build_lbvh(hipStream_t st, const float* flat, const uint32_t* mesh_of_prim, uint32_t n, int use_ploc, float split_budget,
                            int reinsert, BvhBuildResult* out, char* err, size_t errlen);

// cap = shard capacity (in items) of the radiance queues and of the shadow queue.
// sel: 0 = all eight queue shards (a whole job); 1 / 2 = the first / second four (one of the two chains of a frame)
void fovpt_launch_generate(hipStream_t st, const FrameDev& fd, PathState ps, RayQueue queue0, uint32_t cap, Counters* cnt, uint32_t slot_begin,
                           uint32_t slot_end, int grid, uint32_t sel = 0);
// One launch that traces the shadow queue of iteration it_shadow (if >= 0) and the radiance queue of
// iteration it_closest (if >= 0).
void fovpt_launch_traverse(hipStream_t st, SceneView sc, PathState ps, RayQueue queue, ShadowQueue sq, uint32_t cap,
                           Counters* cnt, int it_closest, int it_shadow, int grid, hipEvent_t done = nullptr, uint32_t sel = 0);
void fovpt_launch_shade(hipStream_t st, const FrameDev& fd, SceneView sc, PathState ps, RayQueue queue_in, RayQueue queue_out,
                        ShadowQueue sq, uint32_t cap, Counters* cnt, int depth, int grid, hipEvent_t done = nullptr, uint32_t sel = 0);
void fovpt_launch_resolve(hipStream_t st, const FrameDev& fd, PathState ps, Counters* cnt, hipEvent_t done = nullptr);
// multi-GPU gather plan (see wavefront.hip): owner map + per-block counts; scan (phase 0) / fill (phase 1); pack; unpack
void fovpt_launch_plan_owner(hipStream_t st, const FrameDev& fd, uint8_t* owner, uint32_t* block_count, uint32_t nblocks);
void fovpt_launch_plan_scan_fill(hipStream_t st, uint32_t npix, uint32_t nblocks, int world, const uint8_t* owner, uint32_t* block_count,
                                 uint32_t* total, const uint32_t* rank_base, uint32_t* idx, int phase);
void fovpt_launch_gather_pack(hipStream_t st, uint32_t n, const uint32_t* idx, const uint32_t* frame, uint32_t* packed);
void fovpt_launch_gather_unpack(hipStream_t st, int world, uint32_t stride, uint32_t total, const uint32_t* base, const uint32_t* idx,
                                const uint32_t* gathered, uint32_t* frame);
void fovpt_launch_build_guide(hipStream_t st, const float* cdf, int n, int segments, uint32_t* guide);
void fovpt_launch_probe_records(hipStream_t st, size_t n, const float* cdfX, const float* pdfX, const float4* data, float4* rec);
void fovpt_launch_build_cdf(hipStream_t st, int w, int h, const float4* data, float* pdfX, float* cdfX, float* pdfY, float* cdfY, float* row_total);
void fovpt_launch_math(hipStream_t st, int op, const float* a, const float* b, float* out, size_t n);
