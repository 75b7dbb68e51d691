// fovpt_api.hip -- host side of libfovpt: the C ABI of include/fovpt.h over HIP.
//
// One context = one device = one stream, like the reference's SampleRenderer
// (PT_sv5_/SimplePathtracer.cpp:331-340).  Calls on a context are not thread-safe.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fovpt_device.h"
#ifndef FOVPT_ASYNC_LAST_SHADE_DEFAULT
#define FOVPT_ASYNC_LAST_SHADE_DEFAULT 0
#endif
#ifndef FOVPT_SPLIT_BUDGET_DEFAULT
#define FOVPT_SPLIT_BUDGET_DEFAULT 0.0f
#endif
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is loaded at run time (fovpt_comm_*), libfovpt.so does not link it

namespace {

std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct EventPair { hipEvent_t a, b; int kind; };   // kind: 0 gen, 1 trace, 2 shade, 3 shadow, 4 resolve

}  // namespace

// for the context-free entry points of other translation units (model_loader.cpp): the text fovpt_last_error(NULL) returns
void fovpt_internal_set_error(const char* text) { g_create_error = text ? text : ""; }

// Shadow-queue buffers per state set: bounce it writes buffer it % FOVPT_NSQ, so with max_depth <= FOVPT_NSQ the
// main chain never has to wait for an occlusion launch inside a job.
#define FOVPT_NSQ 4

struct StateSet {
    DevBuf s_thr, s_rng, s_hit, s_rad, s_alpha, s_backplate, s_guide_n, s_guide_a, s_trace;
    DevBuf q_o[2], q_d[2], counters;       // q_*: the two radiance-ray queues (ping-pong)
    DevBuf sq_o[FOVPT_NSQ], sq_d[FOVPT_NSQ], sq_vis[FOVPT_NSQ], sq_occ[FOVPT_NSQ];   // shadow queues, one per bounce in flight
    hipEvent_t ev_shade[FOVPT_MAX_ITERS + 1] = {};
    hipEvent_t ev_shadow[FOVPT_MAX_ITERS + 1] = {};
    hipEvent_t ev_shade2[FOVPT_MAX_ITERS + 1] = {};     // the same for the second chain of a frame (fovpt_config.chains_per_frame = 2)
    hipEvent_t ev_shadow2[FOVPT_MAX_ITERS + 1] = {};
    hipEvent_t ev_last_closest2 = nullptr;
    hipEvent_t ev_done = nullptr;          // recorded after the resolve of the last job that used this set
    hipEvent_t ev_last_closest = nullptr;  // completion of the job's last closest-hit launch (the last shading launch may run on the shadow stream)
    bool used = false;
    std::vector<DevBuf*> all()
    {
        std::vector<DevBuf*> v = {&q_o[0], &q_d[0], &q_o[1], &q_d[1], &s_thr, &s_rng, &s_hit, &s_rad, &s_alpha, &s_backplate, &s_guide_n, &s_guide_a, &s_trace, &counters};
        for (int k = 0; k < FOVPT_NSQ; k++) { v.push_back(&sq_o[k]); v.push_back(&sq_d[k]); v.push_back(&sq_vis[k]); v.push_back(&sq_occ[k]); }
        return v;
    }
};

#ifndef FOVPT_LANES_DEFAULT
#define FOVPT_LANES_DEFAULT 2
#endif
#define FOVPT_MAX_LANES 4
#ifndef FOVPT_SETS_SLOT_LIMIT
#define FOVPT_SETS_SLOT_LIMIT (16ull << 20)    // (~330 B of state and queues per slot and set)
#endif

struct fovpt_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;          // main chain: generate, closest-hit traversal, shade
    hipStream_t shadow_stream = nullptr;   // occlusion rays of every bounce and the resolve: off the critical path
    // Second LANE (round 3): consecutive jobs alternate between two (main, shadow) stream pairs, so the main chain of job k+1
    // -- generate, closest-hit, shade, strictly one after the other -- runs BESIDE the main chain of job k instead of behind it:
    // the launch gaps, ramps and tails of one chain are filled by the other.  Resolves stay in job order (each waits for the
    // previous job's), and `shadow_stream` remains the one stream every finished frame is ordered on (fovpt_stream()).
    hipStream_t lane_main[FOVPT_MAX_LANES] = {}, lane_shadow[FOVPT_MAX_LANES] = {};   // [0] = stream / shadow_stream
    int lanes = FOVPT_LANES_DEFAULT;
    int chains_default = 1;                // what fovpt_config.chains_per_frame = 0 means (FOVPT_CHAINS)
    int partition = 1;                     // direction classes in k_shade's appends: 0 never, 1 foveated frames, 2 always (FOVPT_PARTITION)
    std::string err;
    fovpt_config cfg;
    // scene
    bool has_scene = false;
    uint64_t scene_id = 0;
    BvhNode4* nodes = nullptr;
    TriRec* tris = nullptr;
    DevBuf tri_tc, meshes, textures;
    std::vector<void*> tex_pixels;
    uint32_t num_tris = 0, any_catcher = 0;
    // probe
    DevBuf pr_data, pr_pdfx, pr_cdfx, pr_pdfy, pr_cdfy, pr_guidex, pr_guidey, pr_rec;
    bool guide_ok = false;
    int guide_w = 0, guide_h = 0;
    bool rows_identical = false;           // every row of data / pdfX / cdfX equals row 0 bit for bit
    // frame buffers (resize)
    DevBuf fb_frame, fb_accum, fb_color, fb_normal, fb_albedo;
    DevBuf accum_before;                   // accumulate mode, chunked launch: the accum buffer as it was before the launch
    // multi-GPU gather plan (fovpt_gather_plan): pixel indices grouped by owning rank
    DevBuf plan_owner, plan_blocks, plan_total, plan_base, plan_idx;
    std::vector<uint32_t> plan_off;        // host copy: rank r owns plan_idx[plan_off[r] .. plan_off[r + 1])
    std::string plan_key;                  // what the plan was built for
    bool use_accum_before = false;
    // RCCL transport of the packed gather (fovpt_comm_init / fovpt_gather_frame)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    DevBuf comm_packed, comm_gathered;
    // Wavefront state, several sets used in rotation by consecutive jobs: the tail of job k (its last occlusion
    // rays and its resolve, on the shadow stream) runs beside the head of job k+1 (generate, camera rays).
    // Round 4: TWICE as many sets as lanes, so that the job which follows job k on the same lane (job k + lanes) does not
    // wait for job k's resolve before it may overwrite the path state: its main chain starts as soon as job k's has ended,
    // and k's tail runs beside it.  (What a 1/N shard of a frame needs: its launches are short, and last occlusion launch +
    // resolve were a third of a lane's cycle.)
    StateSet set[2 * FOVPT_MAX_LANES];
    unsigned nsets = 4;                    // sets in rotation for ordinary jobs = 2 * lanes (FOVPT_SETS: 2 .. 2 * FOVPT_MAX_LANES)
    unsigned last_set = 0;                 // the set the most recent job used
    unsigned jobs = 0;                     // jobs issued so far; job j runs on lane j % lanes
    int grid = 2048, grid_trace = 2048, grid_shadow = 1024, grid_shade = 1024;
    int spread_occlusion = 1;              // sharded frames: one occlusion launch of a first-lane job runs on the second lane's shadow stream (FOVPT_SPREAD_OCCLUSION)
    int async_last_shade = FOVPT_ASYNC_LAST_SHADE_DEFAULT;   // 1: the last shading launch of a job runs on the shadow stream (see run_job)
    uint64_t slot_budget = 64ull << 20;    // sample slots per wavefront job (~330 B of state and queues each and per set; jobs above FOVPT_SETS_SLOT_LIMIT rotate through one set per lane)
    // stats
    fovpt_stats stats;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
};

namespace {

int fail(fovpt_ctx* c, int code, const char* fmt, ...);
int sync_all(fovpt_ctx* c);

int fail(fovpt_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail((c), FOVPT_E_DEVICE, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

int sync_all(fovpt_ctx* c)
{
    for (int l = 0; l < FOVPT_MAX_LANES; l++) {
        if (c->lane_main[l]) HIPCHK(c, hipStreamSynchronize(c->lane_main[l]));
        if (c->lane_shadow[l]) HIPCHK(c, hipStreamSynchronize(c->lane_shadow[l]));
    }
    if (c->shadow_stream) HIPCHK(c, hipStreamSynchronize(c->shadow_stream));      // last: the resolves wait for the lanes
    return FOVPT_OK;
}

fovpt_config default_config()
{
    fovpt_config c;
    memset(&c, 0, sizeof(c));
    c.uniform = 0;
    c.r_inner = 74; c.r_outer = 241;                    // SimplePathtracer.cpp:20-21
    c.spp_periphery = 8; c.spp_middle = 16; c.spp_fovea = 32;   // :142,170,193
    c.spp_uniform = 4;                                  // :95
    c.max_depth = 4;                                    // deviceProgram.cu:515
    c.accumulate = 0;
    c.rank = 0; c.world = 1; c.tile_w = 8; c.tile_h = 4;
    return c;
}

hipEvent_t get_event(fovpt_ctx* c)
{
    if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;      // (Timed skips the measurement)
    return e;
}

struct Timed {
    fovpt_ctx* c; int kind; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Timed(fovpt_ctx* c_, int k, hipStream_t s = nullptr) : c(c_), kind(k), st(s ? s : c_->stream)
    {
        if (c->cfg.profile == 2) (void)sync_all(c);   // the kernel runs alone
        if (c->cfg.profile) {
            a = get_event(c); b = get_event(c);
            if (a && b) (void)hipEventRecord(a, st);
            else { if (a) c->free_events.push_back(a); if (b) c->free_events.push_back(b); a = b = nullptr; }
        }
    }
    ~Timed()
    {
        if (a && b) { (void)hipEventRecord(b, st); EventPair p = {a, b, kind}; c->pending.push_back(p); }
        if (c->cfg.profile == 2) (void)hipStreamSynchronize(st);
    }
};

void drain_events(fovpt_ctx* c)
{
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            switch (p.kind) {
            case 0: c->stats.ms_generate += ms; break;
            case 1: c->stats.ms_trace += ms; c->stats.n_trace_launches++; break;
            case 2: c->stats.ms_shade += ms; break;
            case 3: c->stats.ms_shadow += ms; c->stats.n_shadow_launches++; break;
            case 4: c->stats.ms_resolve += ms; break;
            }
        }
        c->free_events.push_back(p.a); c->free_events.push_back(p.b);
    }
    c->pending.clear();
}

void free_scene(fovpt_ctx* c)
{
    if (c->nodes) (void)hipFree(c->nodes);          // (the triangles live in the same allocation, behind the nodes)
    c->nodes = nullptr; c->tris = nullptr;
    for (void* p : c->tex_pixels) (void)hipFree(p);
    c->tex_pixels.clear();
    c->has_scene = false;
}

// A queue shard receives the appends of the blocks whose index is congruent to it modulo FOVPT_SHARDS.  The
// producing kernels walk their index space in 256-wide block iterations m = 0, 1, 2, ... with a grid that is
// a multiple of FOVPT_SHARDS, so shard s gets the iterations m = s (mod 8): at most ceil(M / 8) + 1 of them,
// each appending at most 256 entries.  Nothing can overflow a shard of slots / 8 + 512 entries.
uint32_t shard_capacity(size_t slots) { return (uint32_t)(slots / FOVPT_SHARDS + 512); }

int ensure_state(fovpt_ctx* c, StateSet& S, size_t slots, size_t launches, hipStream_t st)
{
    const size_t v = 16;
    HIPCHK(c, S.s_thr.reserve(slots * v)); HIPCHK(c, S.s_rng.reserve(slots * v));
    HIPCHK(c, S.s_hit.reserve((size_t)shard_capacity(slots) * FOVPT_SHARDS * v));      // indexed like the ray queues
#if FOVPT_V_STEPSTAT
    HIPCHK(c, S.s_trace.reserve((size_t)shard_capacity(slots) * FOVPT_SHARDS * v));
#endif
    HIPCHK(c, S.s_alpha.reserve(slots * v));
    HIPCHK(c, S.s_rad.reserve(slots * v * (size_t)c->cfg.max_depth));
    HIPCHK(c, S.s_backplate.reserve(launches * v));
    if (c->cfg.write_guides) { HIPCHK(c, S.s_guide_n.reserve(slots * v)); HIPCHK(c, S.s_guide_a.reserve(slots * v)); }
    // sharded queues: FOVPT_SHARDS regions of shard_capacity(slots) entries each;
    // there is one shadow queue per bounce in flight (FOVPT_NSQ): bounce it's occlusion rays may still be running
    // on the shadow stream while later bounces are shaded
    const size_t qn = (size_t)shard_capacity(slots) * FOVPT_SHARDS;
    for (int k = 0; k < 2; k++) { HIPCHK(c, S.q_o[k].reserve(qn * v)); HIPCHK(c, S.q_d[k].reserve(qn * v)); }
    const int iters_max = c->cfg.max_depth + (c->any_catcher ? 25 : 0);
    for (int k = 0; k < FOVPT_NSQ && k < iters_max; k++) {
        HIPCHK(c, S.sq_o[k].reserve(qn * v)); HIPCHK(c, S.sq_d[k].reserve(qn * v));
        HIPCHK(c, S.sq_vis[k].reserve(qn * v)); HIPCHK(c, S.sq_occ[k].reserve(qn * v));
    }
    if (!S.counters.p) {
        HIPCHK(c, S.counters.reserve(sizeof(Counters)));
        HIPCHK(c, hipMemsetAsync(S.counters.p, 0, sizeof(Counters), st));      // (the main stream of the job that is about to use the set)
        HIPCHK(c, hipStreamSynchronize(st));                                   // once per set: a second chain starts on another stream
    }
    return FOVPT_OK;
}

// What the traversal and shading kernels see of the scene.
SceneView scene_view(const fovpt_ctx* c)
{
    SceneView sc;
    sc.nodes = c->nodes; sc.tris = c->tris; sc.tri_tc = (const float2*)c->tri_tc.p;
    sc.meshes = (const MeshDev*)c->meshes.p; sc.textures = (const TexDev*)c->textures.p;
    sc.num_tris = c->num_tris; sc.any_catcher = c->any_catcher;
    sc.tri_off = (uint32_t)((const char*)c->tris - (const char*)c->nodes);
    sc.num_nodes = c->stats.num_bvh_nodes;
    return sc;
}


int run_job(fovpt_ctx* c, const fovpt_launch_params* lp, const PassDev* passes_in, int npass, int chunked, int whole_frame);

// The engine: all passes of one frame as one wavefront job (or several, for very large launches).
// whole_frame: the passes are a complete frame (fovpt_render), so with world > 1 the pixels nobody writes are
// cleared on the ranks other than 0 (rank 0 keeps them, as the reference's frame buffer keeps what no launch
// overwrites): the sum over the ranks is then the single-GPU frame.  A single fovpt_launch leaves them alone on
// every rank -- it may be one of several launches that make up the caller's frame.
int run_passes(fovpt_ctx* c, const fovpt_launch_params* lp, const PassDev* passes_in, int npass, int whole_frame)
{
    if (!c->has_scene || lp->traversable != c->scene_id) return fail(c, FOVPT_E_NO_SCENE, "launch without a scene (traversable %llu, current %llu)",
                                                                     (unsigned long long)lp->traversable, (unsigned long long)c->scene_id);
    if (!lp->probe.data || !lp->probe.cdfValuesX || !lp->probe.cdfValuesY || !lp->probe.pdfValuesX || !lp->probe.pdfValuesY
        || lp->probe.width <= 0 || lp->probe.height <= 0)
        return fail(c, FOVPT_E_NO_PROBE, "launch with an incomplete probe");
    if (!lp->frame.accum_buffer || !lp->frame.frame_buffer) return fail(c, FOVPT_E_NO_FRAME, "launch with null frame buffers");
    if (lp->frame.size.x <= 0 || lp->frame.size.y <= 0) return fail(c, FOVPT_E_INVALID, "bad frame size");
    if (c->cfg.max_depth < 1 || c->cfg.max_depth > 32) return fail(c, FOVPT_E_INVALID, "max_depth out of range");

    // A launch whose sample slots would not fit the per-job budget is cut into chunks of launch rows and
    // run as several jobs in the reference's order (pass by pass, rows ascending): later jobs overwrite
    // earlier ones exactly as later launch indices overwrite earlier ones in the reference.
    uint64_t all_slots = 0;
    for (int p = 0; p < npass; p++) {
        if (passes_in[p].spp == 0) return fail(c, FOVPT_E_INVALID, "samples_per_launch must be >= 1 (do{}while(--i), deviceProgram.cu:448,539)");
        all_slots += (uint64_t)passes_in[p].gw * passes_in[p].gh * passes_in[p].spp;
    }
    const uint64_t budget = c->slot_budget;
    if (all_slots > budget) {
        c->stats.frames++;
        const size_t npix = (size_t)lp->frame.size.x * lp->frame.size.y;
        // Accumulate mode blends with the pixel's value from BEFORE the launch (a pass of render() = one
        // optixLaunch).  The chunks of a pass are separate jobs, so a pixel that two of them write (clamped
        // or overlapping fills) would otherwise be blended twice: keep a copy for the chunks to read.  The copy
        // of the FIRST pass is taken here, before anything of this frame touches the buffer (the clearing below
        // included); render() blends in pass P only, which is the first.
        auto blends = [&](const PassDev& P) { return c->cfg.accumulate && P.subframe > 0 && !P.redraw; };
        auto snapshot = [&]() -> int {
            HIPCHK(c, c->accum_before.reserve(npix * 16));
            HIPCHK(c, hipMemcpyAsync(c->accum_before.p, lp->frame.accum_buffer, npix * 16, hipMemcpyDeviceToDevice, c->shadow_stream));
            return FOVPT_OK;
        };
        if (npass > 0 && blends(passes_in[0])) { int rc = snapshot(); if (rc) return rc; }
        if (whole_frame && c->cfg.world > 1 && c->cfg.rank != 0) {
            // The chunk jobs zero the pixels whose last writer (within the chunk) belongs to another rank; pixels no
            // chunk writes must not keep this rank's stale values (rank 0 keeps its own, like the unchunked path)
            // (on the stream the resolves run on: after the previous frame's, before this frame's)
            HIPCHK(c, hipMemsetAsync(lp->frame.frame_buffer, 0, npix * 4, c->shadow_stream));
            HIPCHK(c, hipMemsetAsync(lp->frame.accum_buffer, 0, npix * 16, c->shadow_stream));
        }
        for (int p = 0; p < npass; p++) {
            const PassDev& P = passes_in[p];
            c->use_accum_before = blends(P);
            if (p > 0 && c->use_accum_before) { int rc = snapshot(); if (rc) return rc; }      // (no caller of this library gets here)
            const uint64_t per_row = (uint64_t)P.gw * P.spp;
            if (per_row == 0 || P.gh == 0) continue;
            if (per_row > budget) return fail(c, FOVPT_E_INVALID, "one launch row needs %llu sample slots (budget %llu)", (unsigned long long)per_row, (unsigned long long)budget);
            const uint32_t rows_per = (uint32_t)(budget / per_row);
            for (uint32_t y0 = 0; y0 < P.gh; y0 += rows_per) {
                PassDev Q = P;
                Q.frame_pass = (uint32_t)p;        // the job holds ONE pass; ownership still rotates with its place in the frame
                Q.row0 = y0; Q.row1 = y0 + rows_per < P.gh ? y0 + rows_per : P.gh;
                int rc = run_job(c, lp, &Q, 1, 1, 0);
                if (rc) return rc;
            }
        }
        return FOVPT_OK;
    }
    PassDev full[FOVPT_MAX_PASSES];
    for (int p = 0; p < npass; p++) { full[p] = passes_in[p]; full[p].row0 = 0; full[p].row1 = passes_in[p].gh; full[p].frame_pass = (uint32_t)p; }
    c->stats.frames++;
    return run_job(c, lp, full, npass, 0, whole_frame);
}

// One wavefront job: generate -> (closest, shade, occlusion) x depth -> resolve over the given passes / row ranges.
int run_job(fovpt_ctx* c, const fovpt_launch_params* lp, const PassDev* passes_in, int npass, int chunked, int whole_frame)
{
    FrameDev fd;
    memset(&fd, 0, sizeof(fd));
    fd.chunked = chunked;
    fd.zero_holes = (whole_frame && !chunked && c->cfg.world > 1 && c->cfg.rank != 0) ? 1 : 0;
    uint64_t slots = 0, launches = 0;
    for (int p = 0; p < npass; p++) {
        PassDev P = passes_in[p];
        const uint64_t rows = P.row1 - P.row0;
        P.slot_base = (uint32_t)slots; P.launch_base = (uint32_t)launches;
        slots += (uint64_t)P.gw * rows * P.spp;
        launches += (uint64_t)P.gw * rows;
        fd.pass[p] = P;
    }
    if (slots >= (1ull << 31)) return fail(c, FOVPT_E_INVALID, "launch too large: %llu sample slots", (unsigned long long)slots);
    fd.npass = npass;
    fd.w = lp->frame.size.x; fd.h = lp->frame.size.y;
    fd.cx = lp->frame.c.x; fd.cy = lp->frame.c.y;
    const fovpt_float3* cam[4] = {&lp->camera.eye, &lp->camera.U, &lp->camera.V, &lp->camera.W};
    float* dst[4] = {fd.eye, fd.U, fd.V, fd.W};
    for (int k = 0; k < 4; k++) { dst[k][0] = cam[k]->x; dst[k][1] = cam[k]->y; dst[k][2] = cam[k]->z; }
    fd.probe = lp->probe;
    // the guide tables belong to the probe this context uploaded; a caller-supplied foreign probe is searched plainly
    const bool own_probe = c->guide_ok && lp->probe.cdfValuesX == (float*)c->pr_cdfx.p && lp->probe.cdfValuesY == (float*)c->pr_cdfy.p
                           && lp->probe.width == c->guide_w && lp->probe.height == c->guide_h;
    fd.guide_x = own_probe ? (const uint32_t*)c->pr_guidex.p : nullptr;
    fd.guide_y = own_probe ? (const uint32_t*)c->pr_guidey.p : nullptr;
    // (a probe whose rows are all alike is served from ONE row of the split arrays, which stays in L1; its 32-byte records would not)
    fd.probe_rec = (own_probe && !c->rows_identical && lp->probe.data == (fovpt_float4*)c->pr_data.p && lp->probe.pdfValuesX == (float*)c->pr_pdfx.p)
                   ? (const float4*)c->pr_rec.p : nullptr;
    fd.probe_row_mul = (c->rows_identical && lp->probe.data == (fovpt_float4*)c->pr_data.p && lp->probe.pdfValuesX == (float*)c->pr_pdfx.p
                        && lp->probe.cdfValuesX == (float*)c->pr_cdfx.p && lp->probe.width == c->guide_w && lp->probe.height == c->guide_h) ? 0 : 1;
    fd.accum = lp->frame.accum_buffer;
    fd.accum_prev = (chunked && c->use_accum_before) ? (const fovpt_float4*)c->accum_before.p : lp->frame.accum_buffer;
    fd.frame = lp->frame.frame_buffer;
    if (c->cfg.write_guides) {
        if (c->any_catcher) return fail(c, FOVPT_E_INVALID, "write_guides is not available with shadow-catcher materials");
        fd.g_normal = lp->frame.normal_buffer; fd.g_color = lp->frame.color_buffer; fd.g_albedo = lp->frame.albedo_buffer;
    }
    fd.total_slots = (uint32_t)slots;
    fd.max_depth = c->cfg.max_depth;
    fd.accumulate = c->cfg.accumulate;
    fd.options = c->cfg.options;
    fd.partition = (c->partition == 2 || (c->partition == 1 && !c->cfg.uniform)) ? 1 : 0;
    fd.rank = c->cfg.rank; fd.world = c->cfg.world < 1 ? 1 : c->cfg.world;
    fd.tile_w = c->cfg.tile_w > 0 ? c->cfg.tile_w : 8; fd.tile_h = c->cfg.tile_h > 0 ? c->cfg.tile_h : 4;

    if (slots == 0) return FOVPT_OK;
    // the chunk jobs of an oversized launch all run on lane 0 (their memsets / snapshot copies are ordered on shadow_stream,
    // as before round 3) and, like every job of more than FOVPT_SETS_SLOT_LIMIT sample slots (~330 B of state each), rotate
    // through no more sets than before round 4
    const unsigned few = c->lanes < 2 ? 2u : (unsigned)c->lanes;
    const unsigned nrot = (chunked || (unsigned long long)slots > FOVPT_SETS_SLOT_LIMIT || c->nsets < few) ? few : c->nsets;
    const unsigned set_index = c->jobs == 0 ? 0u : (c->last_set + 1u) % nrot;
    StateSet& S = c->set[set_index];
    const int fif = c->cfg.frames_in_flight > 0 ? c->cfg.frames_in_flight : c->lanes;
    const unsigned lanes = (unsigned)(fif < c->lanes ? fif : c->lanes);
    // chains_per_frame = 2: the job is TWO chains -- the halves of its sample slots, each with four of the eight queue shards, on
    // the two lanes -- and one resolve behind both.  (Not for chunk jobs, nor for jobs too small to split.)
    const int chains = c->cfg.chains_per_frame > 0 ? c->cfg.chains_per_frame : c->chains_default;
    const bool two_chains = chains == 2 && c->lanes >= 2 && !chunked && slots >= 16384;
    const unsigned lane = two_chains ? 0u : (lanes > 1u && !chunked) ? c->jobs % lanes : 0u;
    hipStream_t st = c->lane_main[lane], ss = c->lane_shadow[lane];
    // the set is free once the resolve of the job that used it last has run (reserve() may also free and
    // reallocate its buffers, which the runtime orders after all device work)
    if (S.used) HIPCHK(c, hipStreamWaitEvent(st, S.ev_done, 0));
    if (S.used && two_chains) HIPCHK(c, hipStreamWaitEvent(c->lane_main[1], S.ev_done, 0));
    int rc = ensure_state(c, S, (size_t)slots, (size_t)launches, st);
    if (rc) return rc;
    // the other sets of the rotation now rather than one per job: an allocation waits for the device, and the first jobs of a
    // caller that keeps several frames in flight would each stop for one
    for (unsigned k = 0; k < nrot && !chunked; k++)
        if (k != set_index && (c->set[k].s_thr.bytes < (size_t)slots * 16 || c->set[k].s_rad.bytes < (size_t)slots * 16 * (size_t)c->cfg.max_depth)) {
            rc = ensure_state(c, c->set[k], (size_t)slots, (size_t)launches, st);
            if (rc) return rc;
        }
    c->jobs++;
    c->last_set = set_index;
    S.used = true;

    PathState ps;
    ps.thr = (float4*)S.s_thr.p;
    ps.rng = (uint4*)S.s_rng.p; ps.hit = (float4*)S.s_hit.p; ps.rad = (float4*)S.s_rad.p; ps.stride = (size_t)c->cfg.max_depth;
    ps.alpha = (float4*)S.s_alpha.p; ps.backplate = (float4*)S.s_backplate.p;
    ps.guide_n = c->cfg.write_guides ? (float4*)S.s_guide_n.p : nullptr;
    ps.guide_a = c->cfg.write_guides ? (float4*)S.s_guide_a.p : nullptr;
#if FOVPT_V_STEPSTAT
    ps.trace = (uint4*)S.s_trace.p;
#endif
    ShadowQueue sq[FOVPT_NSQ];
    for (int k = 0; k < FOVPT_NSQ; k++) {
        sq[k].o = (float4*)S.sq_o[k].p; sq[k].d = (float4*)S.sq_d[k].p;
        sq[k].val_vis = (float4*)S.sq_vis[k].p; sq[k].val_occ = (float4*)S.sq_occ[k].p;
    }
    const SceneView sc = scene_view(c);
    Counters* cnt = (Counters*)S.counters.p;
    // (the queue counters are zero: at allocation, and again by the resolve of the set's previous job)
#if FOVPT_V_STEPSTAT
    HIPCHK(c, hipMemsetAsync(cnt, 0, offsetof(Counters, stat_radiance), st));      // the diagnostic build keeps them after the job
#endif
    const uint32_t cap = shard_capacity((size_t)slots);
    // iterations: depth 0 .. max_depth-1, plus the reference's discarded segment and shadow-catcher
    // pass-throughs (which do not advance depth) when the scene holds a catcher
    int iters = c->cfg.max_depth + (c->any_catcher ? 1 + 24 : 0);
    if (iters > FOVPT_MAX_ITERS) iters = FOVPT_MAX_ITERS;
    const int nsq = iters < FOVPT_NSQ ? iters : FOVPT_NSQ;        // (only that many shadow queue buffers are allocated)
    // Main chain (stream `st`):    generate, closest(0), shade(0), closest(1), shade(1), ... shade(D-1)
    // Shadow chain (stream `ss`):  occlusion(it) as soon as shade(it) has queued its rays; then resolve.
    // Every radiance cell has one writer, so the only joins are: shade(it+2) reuses the shadow queue
    // buffer of bounce it, and resolve needs everything -- it runs on the shadow stream, behind the last
    // occlusion launch (which waited for the last shade), so the main chain is free for the next job.
    // The LAST shading launch of a job feeds nothing on the main chain (no closest-hit launch follows it): with
    // async_last_shade it runs on the shadow stream, in front of the last occlusion launch and the resolve, so the main stream
    // is free for the next job's generate and camera rays one shading launch earlier.
    const bool tail_async = c->async_last_shade != 0;
    // Sharded frames (world > 1): the completion stream -- the first lane's shadow stream -- carries every job's resolve (whose
    // writer search and clearing cover the whole frame on every rank) on top of that lane's occlusion launches and is then as
    // long as the main chains (kernel trace, round 4).  One occlusion launch of a first-lane job moves to the second lane's
    // shadow stream: occlusion launches depend on their shading launch only, and the resolve waits for all of them.
    int spread_it = (c->spread_occlusion && fd.world > 1 && lanes > 1u && lane == 0u && !two_chains && !chunked && !tail_async
                           && c->cfg.max_depth >= 2) ? (c->spread_occlusion == 2 ? 1 : -2) : -1;      // -2: the job's LAST one (below)
    // Which one: a launch is queued when the job is issued and holds the stream's later entries back until its own shading launch
    // has run.  The job's last occlusion launch is the smallest and the resolve waits for it anyway -- but behind it the other
    // lane's next job would find its occlusion launches held back for a whole chain, and with more bounces than shadow queue
    // buffers (iters > FOVPT_NSQ) that job's main chain waits for them (measured: C5, depth 8, 1/8 shard 0.76 -> 1.0 ms).  So:
    // the last launch when no main chain can depend on an occlusion launch, else the second (held back for one bounce only).
    if (spread_it == -2) spread_it = iters <= nsq ? iters - 1 : 1;
    // One chain: the sample slots [slot_begin, slot_end) through the queue shards of `sel` (0: all eight; 1 / 2: one half), with
    // 1 / div of the usual grids.
    auto issue_chain = [&](hipStream_t st, hipStream_t ss, uint32_t sel, uint32_t slot_begin, uint32_t slot_end, int div,
                           hipEvent_t* ev_shade, hipEvent_t* ev_shadow, hipEvent_t ev_last_closest) -> int {
        RayQueue qa, qb;
        qa.o = (float4*)S.q_o[0].p; qa.d = (float4*)S.q_d[0].p;
        qb.o = (float4*)S.q_o[1].p; qb.d = (float4*)S.q_d[1].p;
        const int g_gen = c->grid / div, g_trace = c->grid_trace / div, g_shadow = c->grid_shadow / div, g_shade = c->grid_shade / div;
        { Timed t(c, 0, st); fovpt_launch_generate(st, fd, ps, qa, cap, cnt, slot_begin, slot_end, g_gen, sel); }
        { Timed t(c, 1, st); fovpt_launch_traverse(st, sc, ps, qa, sq[0], cap, cnt, 0, -1, g_trace, (tail_async && iters == 1) ? ev_last_closest : nullptr, sel); }
        for (int it = 0; it < iters; it++) {
            const bool last = it + 1 == iters;
            if (last && tail_async) {
                HIPCHK(c, hipStreamWaitEvent(ss, ev_last_closest, 0));      // (the shadow queue it writes was read by occlusion(it - nsq): earlier on this stream)
                { Timed t(c, 2, ss); fovpt_launch_shade(ss, fd, sc, ps, qa, qb, sq[it % nsq], cap, cnt, it, g_shade, nullptr, sel); }
            } else {
                if (it >= nsq) HIPCHK(c, hipStreamWaitEvent(st, ev_shadow[it - nsq], 0));
                // the events ride on the kernels' own completion signals (hipExtLaunchKernel): a separate
                // hipEventRecord would put a marker packet between shade(it) and closest(it+1), ~6 us on the critical path
                { Timed t(c, 2, st); fovpt_launch_shade(st, fd, sc, ps, qa, qb, sq[it % nsq], cap, cnt, it, g_shade, ev_shade[it], sel); }
                HIPCHK(c, hipStreamWaitEvent(ss, ev_shade[it], 0));
            }
            if (it == spread_it) {
                // (a sharded frame on the first lane: this bounce's occlusion rays run on the OTHER lane's shadow stream -- the
                // completion stream carries every job's resolve as well; the resolve below waits for it)
                HIPCHK(c, hipStreamWaitEvent(c->lane_shadow[1], ev_shade[it], 0));
                { Timed t(c, 3, c->lane_shadow[1]); fovpt_launch_traverse(c->lane_shadow[1], sc, ps, qb, sq[it % nsq], cap, cnt, -1, it, g_shadow, ev_shadow[it], sel); }
            } else
            { Timed t(c, 3, ss); fovpt_launch_traverse(ss, sc, ps, qb, sq[it % nsq], cap, cnt, -1, it, g_shadow, ev_shadow[it], sel); }
            if (!last) { Timed t(c, 1, st); fovpt_launch_traverse(st, sc, ps, qb, sq[0], cap, cnt, it + 1, -1, g_trace, (tail_async && it + 2 == iters) ? ev_last_closest : nullptr, sel); }
            const RayQueue tmp = qa; qa = qb; qb = tmp;
        }
        return FOVPT_OK;
    };
    if (two_chains) {
        const uint32_t half = (uint32_t)(slots / 2 / FOVPT_BLOCK * FOVPT_BLOCK);      // (a whole number of 256-slot block iterations)
        rc = issue_chain(c->lane_main[0], c->lane_shadow[0], 1u, 0u, half, 2, S.ev_shade, S.ev_shadow, S.ev_last_closest);
        if (rc) return rc;
        rc = issue_chain(c->lane_main[1], c->lane_shadow[1], 2u, half, (uint32_t)slots, 2, S.ev_shade2, S.ev_shadow2, S.ev_last_closest2);
        if (rc) return rc;
        HIPCHK(c, hipStreamWaitEvent(c->shadow_stream, S.ev_shadow2[iters - 1], 0));      // (the first chain's last occlusion launch IS on shadow_stream)
    } else {
        rc = issue_chain(st, ss, 0u, 0u, (uint32_t)slots, 1, S.ev_shade, S.ev_shadow, S.ev_last_closest);
        if (rc) return rc;
        // Every resolve runs on shadow_stream, whatever the lane: in job order (a later job's pixels overwrite, or blend with, an
        // earlier one's), behind whatever the caller has queued on fovpt_stream() since the previous frame, and in front of what it
        // queues next.  A job of the second lane joins it behind its last occlusion launch (which waited for its last shade).
        if (ss != c->shadow_stream) HIPCHK(c, hipStreamWaitEvent(c->shadow_stream, S.ev_shadow[iters - 1], 0));
        if (spread_it >= 0 && spread_it < iters) HIPCHK(c, hipStreamWaitEvent(c->shadow_stream, S.ev_shadow[spread_it], 0));
    }
    { Timed t(c, 4, c->shadow_stream); fovpt_launch_resolve(c->shadow_stream, fd, ps, cnt, S.ev_done); }
    HIPCHK(c, hipGetLastError());
    return FOVPT_OK;
}

PassDev pass_from_lp(const fovpt_launch_params* lp, uint32_t gw, uint32_t gh)
{
    PassDev P;
    memset(&P, 0, sizeof(P));
    P.gw = gw; P.gh = gh;
    P.fx = lp->frame.factor.x; P.fy = lp->frame.factor.y; P.fz = lp->frame.factor.z;
    P.fill = lp->frame.fillSize;
    P.offx = lp->frame.offset.x; P.offy = lp->frame.offset.y;
    P.r_inner = lp->frame.r_inner; P.r_outer = lp->frame.r_outer;
    P.spp = lp->samples_per_launch;
    P.subframe = lp->frame.subframe_index;
    P.redraw = lp->frame.redraw;
    return P;
}

// The launches of one SampleRenderer::render() call (SimplePathtracer.cpp:77-214): sets the per-pass members of L the
// way the reference leaves them after its last launch and returns the passes.  (The caller restores subframe_index.)
int frame_passes(const fovpt_config& cfg, fovpt_launch_params& L, PassDev* P)
{
    if (cfg.uniform) {                                                                       // FOV_OFF :85-131
        L.frame.subframe_index = 0;
        L.frame.factor.x = L.frame.factor.y = L.frame.factor.z = 1;
        L.frame.fillSize = 1;
        L.frame.r_outer = 1000000000;
        L.frame.r_inner = 0;
        L.samples_per_launch = (uint32_t)cfg.spp_uniform;
        L.frame.offset.x = L.frame.offset.y = 0;
        L.frame.redraw = 0;
        L.viewportSize.x = L.frame.size.x; L.viewportSize.y = L.frame.size.y;
        P[0] = pass_from_lp(&L, (uint32_t)L.frame.size.x, (uint32_t)L.frame.size.y);
        return 1;
    }
    const int inner_radius = cfg.r_inner, outer_radius = cfg.r_outer;
    // periphery :137-157
    L.frame.factor.x = 4; L.frame.factor.y = 4; L.frame.factor.z = 1;
    L.frame.fillSize = 4;
    L.frame.r_outer = 1000000000;
    L.frame.r_inner = (float)outer_radius;
    L.samples_per_launch = (uint32_t)cfg.spp_periphery;
    L.frame.offset.x = L.frame.offset.y = 0;
    L.frame.redraw = 0;
    P[0] = pass_from_lp(&L, (uint32_t)(L.frame.size.x / 4), (uint32_t)(L.frame.size.y / 4));
    // intermediate :160-187
    L.frame.subframe_index = 0;
    L.frame.factor.x = 2; L.frame.factor.y = 2; L.frame.factor.z = 1;
    L.frame.fillSize = 2;
    L.frame.r_outer = (float)(outer_radius + 2);
    L.frame.r_inner = (float)inner_radius;
    L.samples_per_launch = (uint32_t)cfg.spp_middle;
    L.frame.offset.x = L.frame.c.x - (uint32_t)(outer_radius + 2);
    L.frame.offset.y = L.frame.c.y - (uint32_t)(outer_radius + 2);
    L.frame.redraw = 1;
    P[1] = pass_from_lp(&L, (uint32_t)L.frame.r_outer, (uint32_t)L.frame.r_outer);
    // fovea :189-209
    L.frame.factor.x = 1; L.frame.factor.y = 1; L.frame.factor.z = 1;
    L.frame.fillSize = 1;
    L.frame.r_outer = (float)(inner_radius + 1);
    L.frame.r_inner = 0;
    L.samples_per_launch = (uint32_t)cfg.spp_fovea;
    L.frame.offset.x = L.frame.c.x - (uint32_t)(inner_radius + 1);
    L.frame.offset.y = L.frame.c.y - (uint32_t)(inner_radius + 1);
    L.frame.redraw = 1;
    P[2] = pass_from_lp(&L, (uint32_t)(L.frame.r_outer * 2), (uint32_t)(L.frame.r_outer * 2));
    return 3;
}

}  // namespace

extern "C" {

int fovpt_create(fovpt_ctx** out, int device)
{
    if (!out) return fail(nullptr, FOVPT_E_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(nullptr, FOVPT_E_DEVICE, "no HIP device found (%s)", hipGetErrorString(e));   // initOptix :317-321
    if (device < 0 || device >= n) return fail(nullptr, FOVPT_E_INVALID, "device %d out of range (%d devices)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, FOVPT_E_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    fovpt_ctx* c = new fovpt_ctx;
    c->device = device;
    c->cfg = default_config();
    memset(&c->stats, 0, sizeof(c->stats));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    c->grid = c->num_cus * 8;                 // generate: 8 blocks of 256 = 32 waves per CU, grid-stride; a multiple of FOVPT_SHARDS
    c->grid_trace = c->num_cus * FOVPT_GRID_PER_CU;   // closest-hit launches, in units of 256 threads (k_traverse's own blocks are FOVPT_TBLOCK threads)
    c->grid_shadow = c->num_cus * FOVPT_GRID_SHADOW_PER_CU;   // occlusion launches (measured best of 2..8 with per-wave ray pools)
    if (const char* g = getenv("FOVPT_GRID_GEN")) { const int v = atoi(g); if (v > 0 && v <= 64) c->grid = c->num_cus * v; }             // tuning: blocks per CU
    if (const char* g = getenv("FOVPT_GRID")) { const int v = atoi(g); if (v > 0 && v <= 64) c->grid_trace = c->num_cus * v; }           // tuning: blocks per CU
    if (const char* g = getenv("FOVPT_GRID_SHADOW")) { const int v = atoi(g); if (v > 0 && v <= 16) c->grid_shadow = c->num_cus * v; }   // tuning: blocks per CU
    c->grid = (c->grid + FOVPT_SHARDS - 1) / FOVPT_SHARDS * FOVPT_SHARDS;      // shard_capacity() relies on it
    c->grid_trace = (c->grid_trace + FOVPT_SHARDS - 1) / FOVPT_SHARDS * FOVPT_SHARDS;
    c->grid_shadow = (c->grid_shadow + FOVPT_SHARDS - 1) / FOVPT_SHARDS * FOVPT_SHARDS;      // the work fetch of k_traverse relies on equal shard groups
    // the shading kernel holds 4 waves per SIMD (104 VGPRs) = 4 blocks per CU; twice the resident number of blocks is
    // measured best (blocks per CU 2 / 3 / 4 / 6 / 8: shading 0.307 / 0.274 / 0.261 / 0.263 / 0.244 ms per C3 frame)
    c->grid_shade = c->num_cus * 8;
    if (const char* g = getenv("FOVPT_GRID_SHADE")) { const int v = atoi(g); if (v > 0 && v <= 16) c->grid_shade = c->num_cus * v; }     // tuning: blocks per CU
    c->grid_shade = (c->grid_shade + FOVPT_SHARDS - 1) / FOVPT_SHARDS * FOVPT_SHARDS;
    if (const char* a = getenv("FOVPT_ASYNC_LAST_SHADE")) c->async_last_shade = atoi(a) != 0;
    if (const char* sb = getenv("FOVPT_SLOT_BUDGET")) { const long long v = atoll(sb); if (v > 0) c->slot_budget = (uint64_t)v; }   // tests: force chunking
    // the main chain is the critical path: give it the higher priority so occlusion waves only fill gaps
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    e = hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->shadow_stream, hipStreamDefault, prio_lo);
    if (const char* l = getenv("FOVPT_LANES")) { const int v = atoi(l); if (v >= 1 && v <= FOVPT_MAX_LANES) c->lanes = v; }
    if (const char* l = getenv("FOVPT_PARTITION")) { const int v = atoi(l); if (v >= 0 && v <= 2) c->partition = v; }
    if (const char* l = getenv("FOVPT_SPREAD_OCCLUSION")) c->spread_occlusion = atoi(l);      // 0 off, 1 the last occlusion launch, 2 the second (A/B)
    if (const char* l = getenv("FOVPT_CHAINS")) { const int v = atoi(l); if (v == 1 || v == 2) c->chains_default = v; }
    c->nsets = 2u * (unsigned)c->lanes;
    if (const char* l = getenv("FOVPT_SETS")) { const int v = atoi(l); if (v >= 2 && v <= 2 * FOVPT_MAX_LANES) c->nsets = (unsigned)v; }
    c->lane_main[0] = c->stream; c->lane_shadow[0] = c->shadow_stream;
    for (int l = 1; l < c->lanes; l++) {
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->lane_main[l], hipStreamDefault, prio_hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->lane_shadow[l], hipStreamDefault, prio_lo);
    }
    for (StateSet& S : c->set) {
        for (int k = 0; k <= FOVPT_MAX_ITERS && e == hipSuccess; k++) {
            e = hipEventCreateWithFlags(&S.ev_shade[k], hipEventDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_shadow[k], hipEventDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_shade2[k], hipEventDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_shadow2[k], hipEventDefault);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_done, hipEventDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_last_closest, hipEventDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S.ev_last_closest2, hipEventDefault);
    }
    if (e != hipSuccess) { fovpt_destroy(c); return fail(nullptr, FOVPT_E_DEVICE, "stream/event creation: %s", hipGetErrorString(e)); }
    *out = c;
    return FOVPT_OK;
}

void fovpt_destroy(fovpt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    drain_events(c);
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    for (StateSet& S : c->set) {
        for (int k = 0; k <= FOVPT_MAX_ITERS; k++) {
            if (S.ev_shade[k]) (void)hipEventDestroy(S.ev_shade[k]);
            if (S.ev_shadow[k]) (void)hipEventDestroy(S.ev_shadow[k]);
            if (S.ev_shade2[k]) (void)hipEventDestroy(S.ev_shade2[k]);
            if (S.ev_shadow2[k]) (void)hipEventDestroy(S.ev_shadow2[k]);
        }
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        if (S.ev_last_closest) (void)hipEventDestroy(S.ev_last_closest);
        if (S.ev_last_closest2) (void)hipEventDestroy(S.ev_last_closest2);
        for (DevBuf* b : S.all()) b->release();
    }
    (void)fovpt_comm_destroy(c);
    free_scene(c);
    DevBuf* bufs[] = {&c->tri_tc, &c->meshes, &c->textures, &c->pr_data, &c->pr_pdfx, &c->pr_cdfx, &c->pr_pdfy, &c->pr_cdfy, &c->pr_guidex, &c->pr_guidey, &c->pr_rec,
                      &c->fb_frame, &c->fb_accum, &c->fb_color, &c->fb_normal, &c->fb_albedo, &c->accum_before,
                      &c->plan_owner, &c->plan_blocks, &c->plan_total, &c->plan_base, &c->plan_idx,
                      &c->comm_packed, &c->comm_gathered};
    for (DevBuf* b : bufs) b->release();
    for (int l = 0; l < FOVPT_MAX_LANES; l++) {
        if (c->lane_main[l] && c->lane_main[l] != c->stream) (void)hipStreamDestroy(c->lane_main[l]);
        if (c->lane_shadow[l] && c->lane_shadow[l] != c->shadow_stream) (void)hipStreamDestroy(c->lane_shadow[l]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->shadow_stream) (void)hipStreamDestroy(c->shadow_stream);
    delete c;
}

const char* fovpt_last_error(const fovpt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int fovpt_set_scene(fovpt_ctx* c, const fovpt_mesh_desc* meshes, int num_meshes,
                    const fovpt_texture_desc* textures, int num_textures, uint64_t* traversable_out)
{
    if (!c) return FOVPT_E_INVALID;
    if (!meshes || num_meshes <= 0) return fail(c, FOVPT_E_INVALID, "scene needs at least one mesh");
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    free_scene(c);
    uint64_t ntri = 0;
    bool any_tc = false;
    for (int m = 0; m < num_meshes; m++) {
        const fovpt_mesh_desc& D = meshes[m];
        if (!D.vertex || !D.index) return fail(c, FOVPT_E_INVALID, "mesh %d has null vertex/index", m);
        ntri += D.num_triangles;
        if (D.texcoord && D.texture_id >= 0) any_tc = true;
        if (D.texture_id >= num_textures) return fail(c, FOVPT_E_INVALID, "mesh %d references texture %d of %d (SimplePathtracer.cpp:581-583 would index out of range)", m, D.texture_id, num_textures);
    }
    if (ntri == 0) return fail(c, FOVPT_E_INVALID, "scene has no triangles");
    // the traversal addresses 48-B triangle records and 128-B nodes with 32-bit byte offsets
    if (ntri > (1ull << 26)) return fail(c, FOVPT_E_INVALID, "too many triangles (%llu > 2^26)", (unsigned long long)ntri);
    // flatten on the host: 9 floats per triangle in global primitive order (mesh order, then index order)
    std::vector<float> flat((size_t)ntri * 9);
    std::vector<uint32_t> mesh_of((size_t)ntri);
    std::vector<float> tc(any_tc ? (size_t)ntri * 6 : 0, 0.0f);
    std::vector<MeshDev> md((size_t)num_meshes);
    c->any_catcher = 0;
    size_t t = 0;
    for (int m = 0; m < num_meshes; m++) {
        const fovpt_mesh_desc& D = meshes[m];
        md[m].material = D.material;
        md[m].texture_id = D.texture_id >= 0 ? D.texture_id : -1;
        md[m].has_texcoord = D.texcoord ? 1 : 0;
        if (D.material.flags & FOVPT_MATERIAL_FLAG_SHADOW_CATCHER) c->any_catcher = 1;
        for (uint32_t k = 0; k < D.num_triangles; k++, t++) {
            for (int v = 0; v < 3; v++) {
                const uint32_t idx = D.index[3 * (size_t)k + v];
                if (idx >= D.num_vertices) return fail(c, FOVPT_E_INVALID, "mesh %d triangle %u indexes vertex %u of %u", m, k, idx, D.num_vertices);
                flat[t * 9 + v * 3 + 0] = D.vertex[3 * (size_t)idx + 0];
                flat[t * 9 + v * 3 + 1] = D.vertex[3 * (size_t)idx + 1];
                flat[t * 9 + v * 3 + 2] = D.vertex[3 * (size_t)idx + 2];
                if (any_tc && D.texcoord) { tc[t * 6 + v * 2] = D.texcoord[2 * (size_t)idx]; tc[t * 6 + v * 2 + 1] = D.texcoord[2 * (size_t)idx + 1]; }
            }
            mesh_of[t] = (uint32_t)m;
        }
    }
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } t_flat, t_mesh_of;      // freed on every return path
    HIPCHK(c, hipMalloc(&t_flat.p, flat.size() * 4));
    HIPCHK(c, hipMalloc(&t_mesh_of.p, mesh_of.size() * 4));
    float* d_flat = (float*)t_flat.p; uint32_t* d_mesh_of = (uint32_t*)t_mesh_of.p;
    HIPCHK(c, hipMemcpy(d_flat, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_mesh_of, mesh_of.data(), mesh_of.size() * 4, hipMemcpyHostToDevice));
    if (any_tc) {
        HIPCHK(c, c->tri_tc.reserve(tc.size() * 4));
        HIPCHK(c, hipMemcpy(c->tri_tc.p, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
    }
    // textures: createTextures :748-799 (RGBA8, wrap, bilinear, normalized float)
    std::vector<TexDev> td((size_t)(num_textures > 0 ? num_textures : 1));
    for (int k = 0; k < num_textures; k++) {
        const fovpt_texture_desc& X = textures[k];
        if (!X.pixel || X.width <= 0 || X.height <= 0) return fail(c, FOVPT_E_INVALID, "texture %d is empty", k);
        void* px = nullptr;
        const size_t nb = (size_t)X.width * X.height * 4;
        HIPCHK(c, hipMalloc(&px, nb));
        c->tex_pixels.push_back(px);
        HIPCHK(c, hipMemcpy(px, X.pixel, nb, hipMemcpyHostToDevice));
        td[k].px = (const uint32_t*)px; td[k].w = X.width; td[k].h = X.height;
    }
    HIPCHK(c, c->textures.reserve(td.size() * sizeof(TexDev)));
    HIPCHK(c, hipMemcpy(c->textures.p, td.data(), td.size() * sizeof(TexDev), hipMemcpyHostToDevice));
    for (int m = 0; m < num_meshes; m++) {
        md[m].tex.px = nullptr; md[m].tex.w = md[m].tex.h = 0;
        if (md[m].texture_id >= 0) md[m].tex = td[md[m].texture_id];
    }
    HIPCHK(c, c->meshes.reserve(md.size() * sizeof(MeshDev)));
    HIPCHK(c, hipMemcpy(c->meshes.p, md.data(), md.size() * sizeof(MeshDev), hipMemcpyHostToDevice));

    struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } ev0, ev1;    // destroyed on every return path
    HIPCHK(c, hipEventCreate(&ev0.e)); HIPCHK(c, hipEventCreate(&ev1.e));
    const hipEvent_t e0 = ev0.e, e1 = ev1.e;
    HIPCHK(c, hipEventRecord(e0, c->stream));
    BvhBuildResult br;
    memset(&br, 0, sizeof(br));
    char errbuf[256];
    const char* bvh_env = getenv("FOVPT_BVH");          // "lbvh" = plain Karras tree (A/B testing); default PLOC
    const int use_ploc = !(bvh_env && strcmp(bvh_env, "lbvh") == 0);
    float split_budget = FOVPT_SPLIT_BUDGET_DEFAULT;    // references added by spatial splits, as a fraction of the triangles
    if (const char* sb = getenv("FOVPT_SPLIT")) split_budget = (float)atof(sb);
    if (!(split_budget >= 0.f) || split_budget > 2.f) split_budget = 0.f;
    hipError_t be = fovpt_build_lbvh(c->stream, d_flat, d_mesh_of, (uint32_t)ntri, use_ploc, split_budget, -1, &br, errbuf, sizeof(errbuf));
    if (be == hipSuccess && br.reinserted && 3 * br.max_depth + 1 > FOVPT_STACK) {
        // reinsertion lowers the tree's cost, not its depth: a hierarchy it made too deep for the traversal stack is built again without it
        (void)hipFree(br.nodes);
        memset(&br, 0, sizeof(br));
        be = fovpt_build_lbvh(c->stream, d_flat, d_mesh_of, (uint32_t)ntri, use_ploc, split_budget, 0, &br, errbuf, sizeof(errbuf));
    }
    (void)hipEventRecord(e1, c->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (be != hipSuccess) return fail(c, FOVPT_E_DEVICE, "LBVH build: %s", errbuf);
    if (3 * br.max_depth + 1 > FOVPT_STACK) {       // a wide node leaves at most 3 entries behind
        (void)hipFree(br.nodes);
        return fail(c, FOVPT_E_BVH_DEPTH, "hierarchy depth %u needs more than the %d traversal stack entries", br.max_depth, FOVPT_STACK);
    }
    if ((size_t)((const char*)br.tris - (const char*)br.nodes) + br.tri_bytes + 64 >= (1ull << 32)) {
        (void)hipFree(br.nodes);
        return fail(c, FOVPT_E_INVALID, "hierarchy of %llu bytes exceeds the 32-bit offsets of the traversal",
                    (unsigned long long)(br.node_bytes + br.tri_bytes));
    }
    c->nodes = br.nodes; c->tris = br.tris;
    c->num_tris = br.num_refs;                       // triangle RECORDS (>= the triangles when some were split into references)
    c->has_scene = true;
    c->scene_id = (c->scene_id & 0xffffffffull) + 1;
    c->scene_id |= 0x464f565000000000ull;          // 'FOVP' tag so a stale/foreign handle is recognisable
    c->stats.num_triangles = ntri; c->stats.num_bvh_nodes = br.num_nodes; c->stats.bvh_max_depth = br.max_depth;
    c->stats.bvh_bytes = br.node_bytes; c->stats.tri_bytes = br.tri_bytes; c->stats.ms_bvh_build = ms;
    if (traversable_out) *traversable_out = c->scene_id;
    return FOVPT_OK;
}

int fovpt_set_probe(fovpt_ctx* c, int width, int height, const fovpt_float4* data,
                    const float* pdfX, const float* cdfX, const float* pdfY, const float* cdfY,
                    const fovpt_float3* offset, fovpt_probe* out)
{
    if (!c) return FOVPT_E_INVALID;
    if (!data || !pdfX || !cdfX || !pdfY || !cdfY || width <= 0 || height <= 0 || !out)
        return fail(c, FOVPT_E_INVALID, "Probe Data is not valid");                         // Probe.h:104-105
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    const size_t n = (size_t)width * height;
    HIPCHK(c, c->pr_pdfx.reserve(n * 4)); HIPCHK(c, c->pr_cdfx.reserve(n * 4));
    HIPCHK(c, c->pr_pdfy.reserve((size_t)height * 4)); HIPCHK(c, c->pr_cdfy.reserve((size_t)height * 4));
    HIPCHK(c, c->pr_data.reserve(n * 16));
    HIPCHK(c, hipMemcpy(c->pr_pdfx.p, pdfX, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->pr_cdfx.p, cdfX, n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->pr_pdfy.p, pdfY, (size_t)height * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->pr_cdfy.p, cdfY, (size_t)height * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->pr_data.p, data, n * 16, hipMemcpyHostToDevice));
    // guide tables for the two CDF searches of ProbeSample: only valid on non-decreasing CDFs
    bool sorted = true;
    for (int row = 0; row < height && sorted; row++) {
        const float* cr = cdfX + (size_t)row * width;
        for (int k = 1; k < width; k++) if (!(cr[k] >= cr[k - 1])) { sorted = false; break; }
    }
    for (int k = 1; k < height && sorted; k++) if (!(cdfY[k] >= cdfY[k - 1])) sorted = false;
    c->guide_ok = false;
    if (sorted) {
        HIPCHK(c, c->pr_guidex.reserve((size_t)height * (width + 2) * 4));
        HIPCHK(c, c->pr_guidey.reserve((size_t)(height + 2) * 4));
        fovpt_launch_build_guide(c->stream, (const float*)c->pr_cdfx.p, width, height, (uint32_t*)c->pr_guidex.p);
        fovpt_launch_build_guide(c->stream, (const float*)c->pr_cdfy.p, height, 1, (uint32_t*)c->pr_guidey.p);
        HIPCHK(c, c->pr_rec.reserve(n * 32));
        fovpt_launch_probe_records(c->stream, n, (const float*)c->pr_cdfx.p, (const float*)c->pr_pdfx.p, (const float4*)c->pr_data.p, (float4*)c->pr_rec.p);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->guide_ok = true;
    }
    c->guide_w = width; c->guide_h = height;
    c->rows_identical = true;
    for (int row = 1; row < height && c->rows_identical; row++)
        if (memcmp(data + (size_t)row * width, data, (size_t)width * 16) || memcmp(pdfX + (size_t)row * width, pdfX, (size_t)width * 4)
            || memcmp(cdfX + (size_t)row * width, cdfX, (size_t)width * 4)) c->rows_identical = false;
    memset(out, 0, sizeof(*out));
    out->width = width; out->height = height;
    out->data = (fovpt_float4*)c->pr_data.p;
    out->pdfValuesX = (float*)c->pr_pdfx.p; out->cdfValuesX = (float*)c->pr_cdfx.p;
    out->pdfValuesY = (float*)c->pr_pdfy.p; out->cdfValuesY = (float*)c->pr_cdfy.p;
    if (offset) out->offset = *offset;
    return FOVPT_OK;
}

int fovpt_set_probe_data(fovpt_ctx* c, int width, int height, const fovpt_float4* data, const fovpt_float3* offset, fovpt_probe* out)
{
    if (!c) return FOVPT_E_INVALID;
    if (!data || width <= 0 || height <= 0 || !out) return fail(c, FOVPT_E_INVALID, "Probe Data is not valid");
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    const size_t n = (size_t)width * height;
    HIPCHK(c, c->pr_pdfx.reserve(n * 4)); HIPCHK(c, c->pr_cdfx.reserve(n * 4));
    HIPCHK(c, c->pr_pdfy.reserve((size_t)height * 4)); HIPCHK(c, c->pr_cdfy.reserve((size_t)height * 4));
    HIPCHK(c, c->pr_data.reserve(n * 16));
    HIPCHK(c, hipMemcpy(c->pr_data.p, data, n * 16, hipMemcpyHostToDevice));
    DevBuf row_total;
    HIPCHK(c, row_total.reserve((size_t)height * 4));
    fovpt_launch_build_cdf(c->stream, width, height, (const float4*)c->pr_data.p, (float*)c->pr_pdfx.p, (float*)c->pr_cdfx.p,
                           (float*)c->pr_pdfy.p, (float*)c->pr_cdfy.p, (float*)row_total.p);
    // monotonicity decides whether the guide tables may be used; check on the host copy of the result
    std::vector<float> hx(n), hy((size_t)height);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    row_total.release();
    HIPCHK(c, hipMemcpy(hx.data(), c->pr_cdfx.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hy.data(), c->pr_cdfy.p, (size_t)height * 4, hipMemcpyDeviceToHost));
    bool sorted = true;
    for (int row = 0; row < height && sorted; row++) {
        const float* cr = hx.data() + (size_t)row * width;
        for (int k = 1; k < width; k++) if (!(cr[k] >= cr[k - 1])) { sorted = false; break; }
    }
    for (int k = 1; k < height && sorted; k++) if (!(hy[k] >= hy[k - 1])) sorted = false;
    c->guide_ok = false;
    if (sorted) {
        HIPCHK(c, c->pr_guidex.reserve((size_t)height * (width + 2) * 4));
        HIPCHK(c, c->pr_guidey.reserve((size_t)(height + 2) * 4));
        fovpt_launch_build_guide(c->stream, (const float*)c->pr_cdfx.p, width, height, (uint32_t*)c->pr_guidex.p);
        fovpt_launch_build_guide(c->stream, (const float*)c->pr_cdfy.p, height, 1, (uint32_t*)c->pr_guidey.p);
        HIPCHK(c, c->pr_rec.reserve(n * 32));
        fovpt_launch_probe_records(c->stream, n, (const float*)c->pr_cdfx.p, (const float*)c->pr_pdfx.p, (const float4*)c->pr_data.p, (float4*)c->pr_rec.p);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->guide_ok = true;
    }
    c->guide_w = width; c->guide_h = height;
    // identical texel rows give identical pdfX/cdfX rows (same arithmetic on the same inputs)
    c->rows_identical = true;
    for (int row = 1; row < height && c->rows_identical; row++)
        if (memcmp(data + (size_t)row * width, data, (size_t)width * 16) || memcmp(hx.data() + (size_t)row * width, hx.data(), (size_t)width * 4))
            c->rows_identical = false;
    memset(out, 0, sizeof(*out));
    out->width = width; out->height = height;
    out->data = (fovpt_float4*)c->pr_data.p;
    out->pdfValuesX = (float*)c->pr_pdfx.p; out->cdfValuesX = (float*)c->pr_cdfx.p;
    out->pdfValuesY = (float*)c->pr_pdfy.p; out->cdfValuesY = (float*)c->pr_cdfy.p;
    if (offset) out->offset = *offset;
    return FOVPT_OK;
}

int fovpt_resize(fovpt_ctx* c, int width, int height, fovpt_frame_ptrs* out)
{
    if (!c) return FOVPT_E_INVALID;
    if (width == 0 || height == 0) return FOVPT_OK;                                          // :231
    if (width < 0 || height < 0 || !out) return fail(c, FOVPT_E_INVALID, "bad resize arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    const size_t n = (size_t)width * height;
    HIPCHK(c, c->fb_frame.reserve(n * 4)); HIPCHK(c, c->fb_accum.reserve(n * 16));
    HIPCHK(c, c->fb_color.reserve(n * 16)); HIPCHK(c, c->fb_normal.reserve(n * 16)); HIPCHK(c, c->fb_albedo.reserve(n * 16));
    HIPCHK(c, hipMemset(c->fb_frame.p, 0, n * 4));
    HIPCHK(c, hipMemset(c->fb_accum.p, 0, n * 16));
    HIPCHK(c, hipMemset(c->fb_color.p, 0, n * 16));
    HIPCHK(c, hipMemset(c->fb_normal.p, 0, n * 16));
    HIPCHK(c, hipMemset(c->fb_albedo.p, 0, n * 16));
    out->frame_buffer = (uint32_t*)c->fb_frame.p; out->accum_buffer = (fovpt_float4*)c->fb_accum.p;
    out->color_buffer = (fovpt_float4*)c->fb_color.p; out->normal_buffer = (fovpt_float4*)c->fb_normal.p;
    out->albedo_buffer = (fovpt_float4*)c->fb_albedo.p;
    return FOVPT_OK;
}

int fovpt_get_config(const fovpt_ctx* c, fovpt_config* out)
{
    if (!c || !out) return FOVPT_E_INVALID;
    *out = c->cfg;
    return FOVPT_OK;
}

int fovpt_set_config(fovpt_ctx* c, const fovpt_config* cfg)
{
    if (!c || !cfg) return FOVPT_E_INVALID;
    if (cfg->max_depth < 1 || cfg->max_depth > 32) return fail(c, FOVPT_E_INVALID, "max_depth must be in [1,32]");
    if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world) return fail(c, FOVPT_E_INVALID, "bad rank/world %d/%d", cfg->rank, cfg->world);
    if (cfg->spp_periphery < 1 || cfg->spp_middle < 1 || cfg->spp_fovea < 1 || cfg->spp_uniform < 1) return fail(c, FOVPT_E_INVALID, "spp must be >= 1");
    if (cfg->r_inner < 0 || cfg->r_outer < cfg->r_inner) return fail(c, FOVPT_E_INVALID, "bad radii");
    if (cfg->frames_in_flight < 0 || cfg->frames_in_flight > FOVPT_MAX_LANES) return fail(c, FOVPT_E_INVALID, "frames_in_flight must be 0 (the default) or 1 .. %d (more than the context's stream pairs, FOVPT_LANES, means all of them)", FOVPT_MAX_LANES);
    if (cfg->chains_per_frame < 0 || cfg->chains_per_frame > 2) return fail(c, FOVPT_E_INVALID, "chains_per_frame must be 0, 1 or 2");
    if (cfg->options & ~(FOVPT_OPT_SKY_MISS | FOVPT_OPT_RUSSIAN_ROULETTE)) return fail(c, FOVPT_E_INVALID, "unknown option bits %d", cfg->options);
    c->cfg = *cfg;
    return FOVPT_OK;
}

int fovpt_launch(fovpt_ctx* c, const fovpt_launch_params* lp, uint32_t width, uint32_t height)
{
    if (!c || !lp) return FOVPT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    PassDev P = pass_from_lp(lp, width, height);
    return run_passes(c, lp, &P, 1, 0);
}

int fovpt_render(fovpt_ctx* c, fovpt_launch_params* lp)
{
    if (!c || !lp) return FOVPT_E_INVALID;
    if (lp->frame.size.x == 0) return FOVPT_OK;                                              // :81-82
    HIPCHK(c, hipSetDevice(c->device));
    PassDev P[3];
    const uint32_t temp_frame = c->cfg.uniform ? 0u : lp->frame.subframe_index;              // :97 / :161 (FOV_OFF zeroes it first, :87)
    const int npass = frame_passes(c->cfg, *lp, P);
    const int rc = run_passes(c, lp, P, npass, 1);
    lp->frame.subframe_index = temp_frame;                                                   // :128-129 / :210-211
    lp->frame.subframe_index++;
    return rc;
}

// ---- multi-GPU: packed gather of the owned pixels ---------------------------------------------------
int fovpt_gather_plan(fovpt_ctx* c, const fovpt_launch_params* lp, uint32_t* counts_out, int counts_len)
{
    if (!c || !lp) return FOVPT_E_INVALID;
    const int world = c->cfg.world < 1 ? 1 : c->cfg.world;
    if (world > 64) return fail(c, FOVPT_E_INVALID, "gather plans support up to 64 ranks (world = %d)", world);
    if (lp->frame.size.x <= 0 || lp->frame.size.y <= 0) return fail(c, FOVPT_E_INVALID, "bad frame size");
    if (counts_out && counts_len < world) return fail(c, FOVPT_E_INVALID, "counts_out holds %d entries, world is %d", counts_len, world);
    HIPCHK(c, hipSetDevice(c->device));
    char key[256];
    snprintf(key, sizeof(key), "%d x %d u%d r%d/%d c%u,%u w%d t%dx%d", lp->frame.size.x, lp->frame.size.y, c->cfg.uniform, c->cfg.r_inner, c->cfg.r_outer,
             lp->frame.c.x, lp->frame.c.y, world, c->cfg.tile_w, c->cfg.tile_h);
    if (c->plan_key != key) {
        fovpt_launch_params L = *lp;
        PassDev P[3];
        FrameDev fd;
        memset(&fd, 0, sizeof(fd));
        fd.npass = frame_passes(c->cfg, L, P);
        for (int p = 0; p < fd.npass; p++) { fd.pass[p] = P[p]; fd.pass[p].row0 = 0; fd.pass[p].row1 = P[p].gh; fd.pass[p].frame_pass = (uint32_t)p; }
        fd.w = L.frame.size.x; fd.h = L.frame.size.y;
        fd.cx = L.frame.c.x; fd.cy = L.frame.c.y;
        fd.rank = c->cfg.rank; fd.world = world;
        fd.tile_w = c->cfg.tile_w > 0 ? c->cfg.tile_w : 8; fd.tile_h = c->cfg.tile_h > 0 ? c->cfg.tile_h : 4;
        const uint32_t npix = (uint32_t)fd.w * (uint32_t)fd.h, nblocks = (npix + FOVPT_BLOCK - 1) / FOVPT_BLOCK;
        HIPCHK(c, c->plan_owner.reserve(npix));
        HIPCHK(c, c->plan_blocks.reserve((size_t)nblocks * world * 4));
        HIPCHK(c, c->plan_total.reserve(64 * 4));
        HIPCHK(c, c->plan_base.reserve(65 * 4));
        HIPCHK(c, c->plan_idx.reserve((size_t)npix * 4));
        hipStream_t st = c->shadow_stream;                  // the stream frames complete on: pack / unpack run there too
        fovpt_launch_plan_owner(st, fd, (uint8_t*)c->plan_owner.p, (uint32_t*)c->plan_blocks.p, nblocks);
        fovpt_launch_plan_scan_fill(st, npix, nblocks, world, (const uint8_t*)c->plan_owner.p, (uint32_t*)c->plan_blocks.p,
                                    (uint32_t*)c->plan_total.p, nullptr, nullptr, 0);
        uint32_t total[64];
        HIPCHK(c, hipMemcpyAsync(total, c->plan_total.p, (size_t)world * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->plan_off.assign((size_t)world + 1, 0u);
        for (int r = 0; r < world; r++) c->plan_off[r + 1] = c->plan_off[r] + total[r];
        HIPCHK(c, hipMemcpyAsync(c->plan_base.p, c->plan_off.data(), (size_t)(world + 1) * 4, hipMemcpyHostToDevice, st));
        fovpt_launch_plan_scan_fill(st, npix, nblocks, world, (const uint8_t*)c->plan_owner.p, (uint32_t*)c->plan_blocks.p,
                                    nullptr, (const uint32_t*)c->plan_base.p, (uint32_t*)c->plan_idx.p, 1);
        HIPCHK(c, hipStreamSynchronize(st));                // (plan_off.data() must outlive the copy)
        HIPCHK(c, hipGetLastError());
        c->plan_key = key;
    }
    if (counts_out) for (int r = 0; r < world; r++) counts_out[r] = c->plan_off[r + 1] - c->plan_off[r];
    return FOVPT_OK;
}

int fovpt_gather_pack(fovpt_ctx* c, const uint32_t* frame, uint32_t* packed)
{
    if (!c || !frame || !packed) return FOVPT_E_INVALID;
    if (c->plan_key.empty()) return fail(c, FOVPT_E_INVALID, "fovpt_gather_pack without a plan (fovpt_gather_plan)");
    HIPCHK(c, hipSetDevice(c->device));
    const int r = c->cfg.rank;
    if (r < 0 || (size_t)r + 1 >= c->plan_off.size()) return fail(c, FOVPT_E_INVALID, "rank %d is not part of the plan", r);
    fovpt_launch_gather_pack(c->shadow_stream, c->plan_off[r + 1] - c->plan_off[r], (const uint32_t*)c->plan_idx.p + c->plan_off[r], frame, packed);
    HIPCHK(c, hipGetLastError());
    return FOVPT_OK;
}

int fovpt_gather_unpack(fovpt_ctx* c, const uint32_t* gathered, uint32_t stride, uint32_t* frame)
{
    if (!c || !gathered || !frame) return FOVPT_E_INVALID;
    if (c->plan_key.empty()) return fail(c, FOVPT_E_INVALID, "fovpt_gather_unpack without a plan (fovpt_gather_plan)");
    HIPCHK(c, hipSetDevice(c->device));
    const int world = (int)c->plan_off.size() - 1;
    for (int r = 0; r < world; r++)
        if (c->plan_off[r + 1] - c->plan_off[r] > stride) return fail(c, FOVPT_E_INVALID, "stride %u is smaller than rank %d's %u pixels", stride, r, c->plan_off[r + 1] - c->plan_off[r]);
    fovpt_launch_gather_unpack(c->shadow_stream, world, stride, c->plan_off[world], (const uint32_t*)c->plan_base.p, (const uint32_t*)c->plan_idx.p, gathered, frame);
    HIPCHK(c, hipGetLastError());
    return FOVPT_OK;
}

// ---- multi-GPU: the transport between pack and unpack, RCCL over xGMI, for C / C++ hosts ----------------------------
// One process (or thread) per GPU, each with its own fovpt_ctx; the gather of a frame is
//   plan -> pack (HIP) -> ncclGroupStart; ncclSend to the root; on the root ncclRecv from every rank; ncclGroupEnd -> unpack
// all enqueued on fovpt_stream(), the stream frames complete on: no host synchronisation, and the transport of frame k runs
// beside the rendering of frame k + 1.  librccl is loaded at run time so that libfovpt.so has no link-time dependency on it
// (a process that already holds an RCCL -- PyTorch's -- gets that one: same SONAME).
}  // extern "C"
namespace {
struct Rccl {
    void* lib = nullptr;
    bool tried = false;
    std::string why;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl()
{
    static Rccl R;
    if (R.tried) return R;
    R.tried = true;
    // A copy the process already holds comes first (RTLD_NOLOAD): a host that has PyTorch loaded has PyTorch's bundled
    // librccl.so -- another file than /opt/rocm's librccl.so.1, so asking for the latter by name would put a SECOND RCCL into the
    // process (fovpathtracing_optixcodelatest_amd/lib.py loads torch's copy first when torch is installed and not imported yet).
    const char* names[] = {getenv("FOVPT_RCCL_LIB"), "librccl.so", "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int k = 0; k < 6 && !R.lib; k++) {
        const char* n = names[k];
        if (!n || !*n) continue;
        const bool only_if_loaded = k == 1 || k == 2;
        // (RTLD_LOCAL: every entry point is looked up with dlsym, and RCCL brings librocm_smi64 with it, whose `amd::smi` globals
        // must not become the process's: /opt/rocm's libamd_smi.so -- which PyTorch's device queries load -- defines the same ones,
        // and two libraries then run their static destructors on one object)
        R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (only_if_loaded ? RTLD_NOLOAD : 0));
        if (!R.lib && !only_if_loaded) R.why = dlerror() ? dlerror() : "dlopen failed";
    }
    if (!R.lib) { if (R.why.empty()) R.why = "librccl not found"; return R; }
    struct { const char* n; void** f; } syms[] = {
        {"ncclGetUniqueId", (void**)&R.GetUniqueId}, {"ncclCommInitRank", (void**)&R.CommInitRank}, {"ncclCommDestroy", (void**)&R.CommDestroy},
        {"ncclGroupStart", (void**)&R.GroupStart}, {"ncclGroupEnd", (void**)&R.GroupEnd}, {"ncclSend", (void**)&R.Send}, {"ncclRecv", (void**)&R.Recv},
        {"ncclGetErrorString", (void**)&R.GetErrorString}};
    for (auto& sy : syms) {
        *sy.f = dlsym(R.lib, sy.n);
        if (!*sy.f) { R.why = std::string("librccl lacks ") + sy.n; dlclose(R.lib); R.lib = nullptr; return R; }
    }
    return R;
}
#define NCCLCHK(c, x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail((c), FOVPT_E_DEVICE, "%s: %s", #x, rccl().GetErrorString(r_)); } while (0)
}  // namespace
extern "C" {

int fovpt_comm_get_unique_id(void* id)
{
    if (!id) return fail(nullptr, FOVPT_E_INVALID, "fovpt_comm_get_unique_id: null argument");
    Rccl& R = rccl();
    if (!R.lib) return fail(nullptr, FOVPT_E_DEVICE, "RCCL is not available: %s", R.why.c_str());
    static_assert(FOVPT_COMM_ID_BYTES == sizeof(ncclUniqueId), "unique id size");
    ncclUniqueId u;
    NCCLCHK(nullptr, R.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return FOVPT_OK;
}

int fovpt_comm_init(fovpt_ctx* c, const void* id, int rank, int world)
{
    if (!c || !id) return FOVPT_E_INVALID;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, FOVPT_E_INVALID, "bad rank %d of %d (1 .. 64 ranks)", rank, world);
    Rccl& R = rccl();
    if (!R.lib) return fail(c, FOVPT_E_DEVICE, "RCCL is not available: %s", R.why.c_str());
    int rc = fovpt_comm_destroy(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCLCHK(c, R.CommInitRank(&c->comm, world, u, rank));      // collective: every rank calls it, each on its own device
    c->comm_rank = rank; c->comm_world = world;
    return FOVPT_OK;
}

int fovpt_comm_destroy(fovpt_ctx* c)
{
    if (!c) return FOVPT_E_INVALID;
    if (!c->comm) return FOVPT_OK;
    (void)hipSetDevice(c->device);
    if (c->shadow_stream) (void)hipStreamSynchronize(c->shadow_stream);
    ncclComm_t comm = c->comm;
    c->comm = nullptr; c->comm_world = 0;
    NCCLCHK(c, rccl().CommDestroy(comm));
    return FOVPT_OK;
}

int fovpt_gather_frame(fovpt_ctx* c, const fovpt_launch_params* lp, int root, const uint32_t* frame, uint32_t* full_frame)
{
    if (!c || !lp || !frame) return FOVPT_E_INVALID;
    if (!c->comm) return fail(c, FOVPT_E_INVALID, "fovpt_gather_frame without a communicator (fovpt_comm_init)");
    const int world = c->comm_world, rank = c->comm_rank;
    if (c->cfg.world != world || c->cfg.rank != rank)
        return fail(c, FOVPT_E_INVALID, "the communicator is rank %d of %d, fovpt_config says %d of %d", rank, world, c->cfg.rank, c->cfg.world);
    if (root < 0 || root >= world) return fail(c, FOVPT_E_INVALID, "bad root %d", root);
    if (rank == root && !full_frame) return fail(c, FOVPT_E_INVALID, "the root needs a frame to gather into");
    uint32_t counts[64];
    int rc = fovpt_gather_plan(c, lp, counts, 64);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t stride = 0;
    for (int r = 0; r < world; r++) stride = counts[r] > stride ? counts[r] : stride;
    stride = (stride + 63u) & ~63u;
    if (stride == 0) return FOVPT_OK;                               // no launch index writes any pixel
    HIPCHK(c, c->comm_packed.reserve((size_t)stride * 4));
    if (rank == root) HIPCHK(c, c->comm_gathered.reserve((size_t)stride * 4 * world));
    rc = fovpt_gather_pack(c, frame, (uint32_t*)c->comm_packed.p);
    if (rc) return rc;
    Rccl& R = rccl();
    hipStream_t st = c->shadow_stream;
    // A group that was opened is always closed: the first failing call is remembered, the remaining point-to-point calls are
    // skipped, ncclGroupEnd still runs (an open group would leave this rank's later collectives queued for ever and its peers
    // blocked in theirs), and only then does the call fail.
    NCCLCHK(c, R.GroupStart());
    ncclResult_t first_err = ncclSuccess;
    const char* first_what = "";
    if (rank == root)
        for (int r = 0; r < world && first_err == ncclSuccess; r++)
            if (counts[r]) {
                first_err = R.Recv((uint32_t*)c->comm_gathered.p + (size_t)r * stride, counts[r], ncclUint32, r, c->comm, st);
                first_what = "ncclRecv";
            }
    if (counts[rank] && first_err == ncclSuccess) { first_err = R.Send(c->comm_packed.p, counts[rank], ncclUint32, root, c->comm, st); first_what = "ncclSend"; }
    const ncclResult_t end_err = R.GroupEnd();
    if (first_err != ncclSuccess) return fail(c, FOVPT_E_DEVICE, "%s: %s", first_what, R.GetErrorString(first_err));
    if (end_err != ncclSuccess) return fail(c, FOVPT_E_DEVICE, "ncclGroupEnd: %s", R.GetErrorString(end_err));
    if (rank == root) {
        rc = fovpt_gather_unpack(c, (const uint32_t*)c->comm_gathered.p, stride, full_frame);
        if (rc) return rc;
    }
    return FOVPT_OK;
}

int fovpt_synchronize(fovpt_ctx* c)
{
    if (!c) return FOVPT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    return FOVPT_OK;
}

int fovpt_download(fovpt_ctx* c, const void* device_src, void* host_dst, size_t n_bytes)
{
    if (!c || !device_src || !host_dst) return FOVPT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    HIPCHK(c, hipMemcpy(host_dst, device_src, n_bytes, hipMemcpyDeviceToHost));
    return FOVPT_OK;
}

int fovpt_get_stats(fovpt_ctx* c, fovpt_stats* out)
{
    if (!c || !out) return FOVPT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    drain_events(c);
    c->stats.radiance_rays = c->stats.shadow_rays = c->stats.paths = 0;
    for (StateSet& S : c->set) {
        if (!S.counters.p) continue;
        unsigned long long h[3];                          // stat_radiance, stat_shadow, stat_paths
        HIPCHK(c, hipMemcpy(h, (const char*)S.counters.p + offsetof(Counters, stat_radiance), sizeof(h), hipMemcpyDeviceToHost));
        c->stats.radiance_rays += h[0]; c->stats.shadow_rays += h[1]; c->stats.paths += h[2];
    }
    *out = c->stats;
    return FOVPT_OK;
}

int fovpt_reset_stats(fovpt_ctx* c)
{
    if (!c) return FOVPT_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    drain_events(c);
    for (StateSet& S : c->set)
        if (S.counters.p) HIPCHK(c, hipMemset((char*)S.counters.p + offsetof(Counters, stat_radiance), 0, sizeof(Counters) - offsetof(Counters, stat_radiance)));
    c->stats.radiance_rays = c->stats.shadow_rays = c->stats.paths = c->stats.frames = 0;
    c->stats.ms_generate = c->stats.ms_trace = c->stats.ms_shade = c->stats.ms_shadow = c->stats.ms_resolve = 0.0;
    c->stats.n_trace_launches = c->stats.n_shadow_launches = 0;
    return FOVPT_OK;
}

// The stream on which frames COMPLETE, in order (occlusion rays and the resolve run on it): work a
// caller queues on it after fovpt_render sees the finished frame and runs before the next frame's
// resolve touches the render target.  (The head of a frame runs on an internal higher-priority stream.)
void* fovpt_stream(fovpt_ctx* c) { return c ? (void*)c->shadow_stream : nullptr; }

// ---- host helpers ------------------------------------------------------------------------
// ProbeData::BuildCDF, PT_sv5_/Probe.h:29-77.  Strictly sequential fp32 sums: the order is part of
// the result.
int fovpt_probe_build_cdf(int width, int height, const fovpt_float4* data,
                          float* pdfValuesX, float* cdfValuesX, float* pdfValuesY, float* cdfValuesY)
{
    if (width <= 0 || height <= 0 || !data || !pdfValuesX || !cdfValuesX || !pdfValuesY || !cdfValuesY) return FOVPT_E_INVALID;
    volatile float col_total = 0.0f;                  // volatile: keep every partial sum rounded to fp32
    for (int row = 0; row < height; ++row) {
        volatile float row_total = 0.0f;
        float* pdf_row = pdfValuesX + (size_t)row * width;
        float* cdf_row = cdfValuesX + (size_t)row * width;
        const fovpt_float4* px = data + (size_t)row * width;
        for (int k = 0; k < width; ++k) {
            const float lum = px[k].x * 0.3f + px[k].y * 0.6f + px[k].z * 0.1f;   // Luminance, maths.h:165-168
            row_total = row_total + lum;
            pdf_row[k] = lum;
            cdf_row[k] = row_total;
        }
        const float inv = 1.0f / row_total;
        for (int k = 0; k < width; ++k) { pdf_row[k] *= inv; cdf_row[k] *= inv; }
        col_total = col_total + row_total;
        pdfValuesY[row] = row_total;
        cdfValuesY[row] = col_total;
    }
    const float total = col_total;
    for (int row = 0; row < height; ++row) { cdfValuesY[row] /= total; pdfValuesY[row] /= total; }
    return FOVPT_OK;
}

// sutil::Camera::UVWFrame, sutil/Camera.cpp:32-44
int fovpt_camera_uvw(const fovpt_float3* eye, const fovpt_float3* lookat, const fovpt_float3* up,
                     float fovY, float aspect, fovpt_float3* U, fovpt_float3* V, fovpt_float3* W)
{
    if (!eye || !lookat || !up || !U || !V || !W) return FOVPT_E_INVALID;
    const float wx = lookat->x - eye->x, wy = lookat->y - eye->y, wz = lookat->z - eye->z;
    const float wlen = sqrtf(wx * wx + wy * wy + wz * wz);
    float ux = wy * up->z - wz * up->y, uy = wz * up->x - wx * up->z, uz = wx * up->y - wy * up->x;    // cross(W, up)
    float inv = 1.0f / sqrtf(ux * ux + uy * uy + uz * uz);
    ux *= inv; uy *= inv; uz *= inv;
    float vx = uy * wz - uz * wy, vy = uz * wx - ux * wz, vz = ux * wy - uy * wx;                      // cross(U, W)
    inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
    vx *= inv; vy *= inv; vz *= inv;
    const float vlen = wlen * tanf(0.5f * fovY * 3.14159265358979323846f / 180.0f);
    vx *= vlen; vy *= vlen; vz *= vlen;
    const float ulen = vlen * aspect;
    ux *= ulen; uy *= ulen; uz *= ulen;
    U->x = ux; U->y = uy; U->z = uz; V->x = vx; V->y = vy; V->z = vz; W->x = wx; W->y = wy; W->z = wz;
    return FOVPT_OK;
}

// test/diagnostic hook: device address and size of an internal buffer
int fovpt_debug_buffer(fovpt_ctx* c, const char* name, void** ptr, size_t* bytes)
{
    if (!c || !name || !ptr || !bytes) return FOVPT_E_INVALID;
    if (strcmp(name, "bvh_nodes") == 0 && c->has_scene) { *ptr = c->nodes; *bytes = (size_t)c->stats.bvh_bytes; return FOVPT_OK; }   // tools/bvhstat.py
    StateSet& S = c->set[c->last_set];                    // the set the most recent job used
    struct { const char* n; DevBuf* b; } tab[] = {
        {"sq_o", &S.sq_o[0]}, {"sq_d", &S.sq_d[0]}, {"sq_vis", &S.sq_vis[0]}, {"sq_occ", &S.sq_occ[0]}, {"counters", &S.counters},
        {"hit", &S.s_hit}, {"trace", &S.s_trace}, {"queue_a_o", &S.q_o[0]}, {"queue_a_d", &S.q_d[0]}, {"queue_b_o", &S.q_o[1]}, {"queue_b_d", &S.q_d[1]},
    };
    for (auto& t : tab)
        if (strcmp(t.n, name) == 0) { *ptr = t.b->p; *bytes = t.b->bytes; return FOVPT_OK; }
    return fail(c, FOVPT_E_INVALID, "unknown debug buffer %s", name);
}

// test hook: the PRODUCTION traversal kernel on a caller-supplied batch of rays -- closest hit (global primitive id, t, u, v)
// and the occlusion predicate of the shadow rays (any front-facing candidate in (0.01, 1e16), deviceProgram.cu:224-248,
// 284-300) -- so that optixTrace's two ray types can be compared with the oracle ray by ray, not only through frames.
int fovpt_debug_trace(fovpt_ctx* c, int n, const float* origins3, const float* dirs3, uint32_t* prim_out, float* tuv_out3, uint8_t* occluded_out)
{
    if (!c || n < 0 || (n && (!origins3 || !dirs3))) return FOVPT_E_INVALID;
    if (!c->has_scene) return fail(c, FOVPT_E_NO_SCENE, "fovpt_debug_trace without a scene");
    if (n == 0) return FOVPT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    StateSet& S = c->set[(c->last_set + 1u) % 2u];                  // (the device is idle: any set will do)
    const size_t slots = (size_t)n * FOVPT_SHARDS;                  // all rays go to shard 0: its capacity must hold them
    int rc = ensure_state(c, S, slots, 1, c->stream);
    if (rc) return rc;
    const uint32_t cap = shard_capacity(slots);
    hipStream_t st = c->stream;
    std::vector<float> o4((size_t)n * 4), d4((size_t)n * 4), d4s((size_t)n * 4), vis((size_t)n * 4, 0.f), occ((size_t)n * 4, 0.f);
    for (int i = 0; i < n; i++) {
        const uint32_t slot = (uint32_t)i, cell = 0u;
        for (int k = 0; k < 3; k++) { o4[4 * (size_t)i + k] = origins3[3 * (size_t)i + k]; d4[4 * (size_t)i + k] = d4s[4 * (size_t)i + k] = dirs3[3 * (size_t)i + k]; }
        memcpy(&o4[4 * (size_t)i + 3], &slot, 4);                   // origin.w = sample slot
        d4[4 * (size_t)i + 3] = 0.f;
        memcpy(&d4s[4 * (size_t)i + 3], &cell, 4);                  // shadow record: direction.w = target cell of the slot
        vis[4 * (size_t)i] = 1.f; occ[4 * (size_t)i] = 2.f;         // what store_shadow leaves in the cell: 1 visible, 2 occluded
    }
    const size_t bytes = (size_t)n * 16;
    HIPCHK(c, hipMemcpyAsync(S.q_o[0].p, o4.data(), bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(S.q_d[0].p, d4.data(), bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(S.sq_o[0].p, o4.data(), bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(S.sq_d[0].p, d4s.data(), bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(S.sq_vis[0].p, vis.data(), bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(S.sq_occ[0].p, occ.data(), bytes, hipMemcpyHostToDevice, st));
    // queue sizes: n entries in shard 0 of the radiance queue of iteration 0 and of the shadow queue of iteration 0
    HIPCHK(c, hipMemsetAsync(S.counters.p, 0, offsetof(Counters, stat_radiance), st));
    const uint32_t un = (uint32_t)n;
    Counters* cnt = (Counters*)S.counters.p;
    HIPCHK(c, hipMemcpyAsync(&cnt->shard[0][FOVPT_CNT_Q(0)], &un, 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(&cnt->shard[0][FOVPT_CNT_SQ(0)], &un, 4, hipMemcpyHostToDevice, st));
    PathState ps;
    memset(&ps, 0, sizeof(ps));
    ps.thr = (float4*)S.s_thr.p; ps.rng = (uint4*)S.s_rng.p; ps.hit = (float4*)S.s_hit.p; ps.rad = (float4*)S.s_rad.p;
    ps.stride = (size_t)c->cfg.max_depth; ps.alpha = (float4*)S.s_alpha.p; ps.backplate = (float4*)S.s_backplate.p;
#if FOVPT_V_STEPSTAT
    ps.trace = (uint4*)S.s_trace.p;
#endif
    RayQueue q; q.o = (float4*)S.q_o[0].p; q.d = (float4*)S.q_d[0].p;
    ShadowQueue sq; sq.o = (float4*)S.sq_o[0].p; sq.d = (float4*)S.sq_d[0].p; sq.val_vis = (float4*)S.sq_vis[0].p; sq.val_occ = (float4*)S.sq_occ[0].p;
    const SceneView sc = scene_view(c);
    fovpt_launch_traverse(st, sc, ps, q, sq, cap, cnt, 0, -1, c->grid_trace);        // closest hit, as run_job launches it
    fovpt_launch_traverse(st, sc, ps, q, sq, cap, cnt, -1, 0, c->grid_shadow);       // occlusion, as run_job launches it
    HIPCHK(c, hipGetLastError());
    std::vector<float> hit((size_t)n * 4), cell((size_t)n * 4 * (size_t)c->cfg.max_depth);
    HIPCHK(c, hipMemcpyAsync(hit.data(), S.s_hit.p, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(cell.data(), S.s_rad.p, bytes * (size_t)c->cfg.max_depth, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<TriRec> tris(c->num_tris);                           // leaf order -> global primitive id
    HIPCHK(c, hipMemcpy(tris.data(), c->tris, sizeof(TriRec) * (size_t)c->num_tris, hipMemcpyDeviceToHost));
    // leave the set as a job expects to find it (resolve zeroes the queue counters at the end of every job)
    HIPCHK(c, hipMemset(S.counters.p, 0, offsetof(Counters, stat_radiance)));
    for (int i = 0; i < n; i++) {
        uint32_t pos;
        memcpy(&pos, &hit[4 * (size_t)i + 3], 4);
        const bool miss = pos == 0xffffffffu;
        const size_t tri = miss ? 0 : (size_t)pos / 3;               // pos: offset in 16-byte units, a record is 48 bytes
        if (!miss && (pos % 3u != 0u || tri >= tris.size())) return fail(c, FOVPT_E_DEVICE, "hit record %d points at 16-byte unit %u", i, pos);
        if (prim_out) prim_out[i] = miss ? 0xffffffffu : tris[tri].prim;
        if (tuv_out3) for (int k = 0; k < 3; k++) tuv_out3[3 * (size_t)i + k] = hit[4 * (size_t)i + k];
        if (occluded_out) {
            const float v = cell[4 * (size_t)i * (size_t)c->cfg.max_depth];
            if (v != 1.f && v != 2.f) return fail(c, FOVPT_E_DEVICE, "shadow ray %d left %g in its cell", i, (double)v);
            occluded_out[i] = v == 2.f ? 1 : 0;
        }
    }
    return FOVPT_OK;
}

int fovpt_debug_math(fovpt_ctx* c, int op, const float* a, const float* b, float* out, size_t n)
{
    if (!c || !a || !out) return FOVPT_E_INVALID;
    if (n == 0) return FOVPT_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf ba, bb, bo;
    struct Rel { DevBuf &a, &b, &o; ~Rel() { a.release(); b.release(); o.release(); } } rel = {ba, bb, bo};
    HIPCHK(c, ba.reserve(n * 4)); HIPCHK(c, bb.reserve(n * 4)); HIPCHK(c, bo.reserve(n * 4));
    float *da = (float*)ba.p, *db = (float*)bb.p, *dout = (float*)bo.p;
    HIPCHK(c, hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    if (b) HIPCHK(c, hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice));
    else HIPCHK(c, hipMemset(db, 0, n * 4));
    fovpt_launch_math(c->stream, op, da, db, dout, n);
    { const int rc_ = sync_all(c); if (rc_) return rc_; }
    HIPCHK(c, hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    return FOVPT_OK;
}

}  // extern "C"
