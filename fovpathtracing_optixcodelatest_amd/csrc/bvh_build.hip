// bvh_build.hip -- LBVH construction on the GPU (gfx950).
//
// Replaces optixAccelBuild + optixAccelCompact (PT_sv5_/SimplePathtracer.cpp:677-735): one
// acceleration structure over all meshes.  Pipeline, all on the device:
//   1. per-triangle padded AABB + centroid, scene centroid bounds (wave-reduced atomics)
//   2. 63-bit Morton code of the centroid
//   3. rocPRIM radix sort of (code, primitive)
//   4. Karras 2012 binary radix tree (one thread per internal node)
//   5. bottom-up AABB refit with per-node arrival counters
//   6. collapse subtrees of <= FOVPT_LEAF_MAX triangles into leaves and emit 64-byte nodes that
//      carry both child boxes; emit 48-byte triangle records in leaf order
// Results never depend on the tree (see the intersection contract in include/fovpt.h), only
// speed does.
#include <cstdio>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fovpt_device.h"

namespace {

__device__ inline uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

struct Box { float lo[3], hi[3]; };

// 1. padded boxes + centroid bounds.  bounds[0..2] = min (ordered uint), bounds[3..5] = max
__global__ void k_tri_bounds(const float* __restrict__ flat, uint32_t n, Box* __restrict__ boxes, uint32_t* __restrict__ bounds)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    bool live = i < n;
    if (live) {
        const float* p = flat + (size_t)i * 9;
        float ext = 0.f, mag = 0.f;
        Box b;
        for (int a = 0; a < 3; a++) {
            float x0 = p[a], x1 = p[3 + a], x2 = p[6 + a];
            float lo = fminf(x0, fminf(x1, x2)), hi = fmaxf(x0, fmaxf(x1, x2));
            b.lo[a] = lo; b.hi[a] = hi;
            ext = fmaxf(ext, hi - lo);
            mag = fmaxf(mag, fmaxf(fabsf(lo), fabsf(hi)));
        }
        // pad so that every Moeller-Trumbore-accepted hit point lies well inside the box and the
        // fused-multiply-add slab test of the traversal stays conservative
        float pad = 1e-4f * ext + 1e-5f * mag + 1e-20f;
        for (int a = 0; a < 3; a++) {
            b.lo[a] -= pad; b.hi[a] += pad;
            c[a] = 0.5f * (b.lo[a] + b.hi[a]);
        }
        boxes[i] = b;
    }
    // wave-level min/max, one atomic per wave and component
    for (int a = 0; a < 3; a++) {
        float mn = live ? c[a] : INFINITY, mx = live ? c[a] : -INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bounds[a], f2ord(mn));
            atomicMax(&bounds[3 + a], f2ord(mx));
        }
    }
}

__device__ inline uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// 2. Morton codes
__global__ void k_morton(const Box* __restrict__ boxes, uint32_t n, const uint32_t* __restrict__ bounds,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        float mn = ord2f(bounds[a]), mx = ord2f(bounds[3 + a]);
        float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
        float ext = mx - mn;
        float t = ext > 0.f ? (c - mn) / ext : 0.f;
        t = fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
        q[a] = (uint32_t)t;
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[i] = i;
}

__device__ inline int delta(const uint64_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// 4. Karras: internal node i in [0, n-1).  Children coded: >=0 internal, <0 ~leaf position.
__global__ void k_hierarchy(const uint64_t* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                            int* __restrict__ parent_int, int* __restrict__ parent_leaf,
                            int* __restrict__ rfirst, int* __restrict__ rlast)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    int lc, rc;
    if (lo == gamma) { lc = ~gamma; parent_leaf[gamma] = i; } else { lc = gamma; parent_int[gamma] = i; }
    if (hi == gamma + 1) { rc = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { rc = gamma + 1; parent_int[gamma + 1] = i; }
    left[i] = lc; right[i] = rc;
    rfirst[i] = lo; rlast[i] = hi;
    if (i == 0) parent_int[0] = -1;
}

// 5. refit: one thread per leaf climbs; the second arrival at a node computes its box
__global__ void k_refit(const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, int n,
                        const int* __restrict__ left, const int* __restrict__ right,
                        const int* __restrict__ parent_int, const int* __restrict__ parent_leaf,
                        Box* __restrict__ ibox, uint32_t* __restrict__ arrive)
{
    int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    int node = parent_leaf[leaf];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrive[node], 1u) == 0u) return;   // first arrival: sibling not ready yet
        __threadfence();
        Box a, b;
        int lc = left[node], rc = right[node];
        // the sibling's box was stored before its fence + atomic; the fence above is our acquire
        const Box* pa = lc < 0 ? &boxes[vals[~lc]] : &ibox[lc];
        const Box* pb = rc < 0 ? &boxes[vals[~rc]] : &ibox[rc];
        a = *pa; b = *pb;
        Box u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
        ibox[node] = u;
        node = parent_int[node];
    }
}

__device__ inline int leaf_code(int first, int count) { return ~((first << 3) | (count - 1)); }

// 6a. emit traversal nodes (sparse: node i keeps its Karras index)
__global__ void k_emit_nodes(int n, const int* __restrict__ left, const int* __restrict__ right,
                             const int* __restrict__ rfirst, const int* __restrict__ rlast,
                             const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const Box* __restrict__ ibox,
                             const int* __restrict__ parent_int, BvhNode* __restrict__ nodes, uint32_t* __restrict__ stats)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int size = rlast[i] - rfirst[i] + 1;
    if (size <= FOVPT_LEAF_MAX && i != 0) return;        // collapsed into an ancestor's leaf reference
    int code[2];
    Box cb[2];
    int ch[2] = {left[i], right[i]};
    for (int k = 0; k < 2; k++) {
        int c = ch[k];
        if (c < 0) {
            int pos = ~c;
            code[k] = leaf_code(pos, 1);
            cb[k] = boxes[vals[pos]];
        } else {
            int sz = rlast[c] - rfirst[c] + 1;
            code[k] = sz <= FOVPT_LEAF_MAX ? leaf_code(rfirst[c], sz) : c;
            cb[k] = ibox[c];
        }
    }
    BvhNode nd;
    nd.lo0x = cb[0].lo[0]; nd.lo0y = cb[0].lo[1]; nd.lo0z = cb[0].lo[2];
    nd.hi0x = cb[0].hi[0]; nd.hi0y = cb[0].hi[1]; nd.hi0z = cb[0].hi[2];
    nd.lo1x = cb[1].lo[0]; nd.lo1y = cb[1].lo[1]; nd.lo1z = cb[1].lo[2];
    nd.hi1x = cb[1].hi[0]; nd.hi1y = cb[1].hi[1]; nd.hi1z = cb[1].hi[2];
    nd.c0 = code[0]; nd.c1 = code[1]; nd.pad0 = nd.pad1 = 0;
    nodes[i] = nd;
    // depth of this node = number of kept ancestors + 1
    int depth = 1;
    for (int p = parent_int[i]; p >= 0; p = parent_int[p]) depth++;
    atomicMax(&stats[0], (uint32_t)depth);
    atomicAdd(&stats[1], 1u);
}

// tiny scenes (n <= FOVPT_LEAF_MAX): a root whose first child is the only leaf
__global__ void k_emit_tiny(int n, const Box* __restrict__ boxes, BvhNode* __restrict__ nodes, uint32_t* __restrict__ stats)
{
    Box u = boxes[0];
    for (int i = 1; i < n; i++)
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(u.lo[k], boxes[i].lo[k]); u.hi[k] = fmaxf(u.hi[k], boxes[i].hi[k]); }
    BvhNode nd;
    nd.lo0x = u.lo[0]; nd.lo0y = u.lo[1]; nd.lo0z = u.lo[2];
    nd.hi0x = u.hi[0]; nd.hi0y = u.hi[1]; nd.hi0z = u.hi[2];
    nd.lo1x = nd.lo1y = nd.lo1z = INFINITY;      // empty child: a point at +inf never passes the slab test
    nd.hi1x = nd.hi1y = nd.hi1z = INFINITY;
    nd.c0 = leaf_code(0, n); nd.c1 = leaf_code(0, 1); nd.pad0 = nd.pad1 = 0;
    nodes[0] = nd;
    stats[0] = 1; stats[1] = 1;
}

// 6b. triangle records in leaf (sorted) order
__global__ void k_emit_tris(const float* __restrict__ flat, const uint32_t* __restrict__ mesh_of_prim,
                            const uint32_t* __restrict__ vals, uint32_t n, TriRec* __restrict__ tris)
{
    uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n) return;
    uint32_t prim = vals[pos];
    const float* p = flat + (size_t)prim * 9;
    TriRec t;
    t.v0x = p[0]; t.v0y = p[1]; t.v0z = p[2];
    t.e1x = p[3] - p[0]; t.e1y = p[4] - p[1]; t.e1z = p[5] - p[2];
    t.e2x = p[6] - p[0]; t.e2y = p[7] - p[1]; t.e2z = p[8] - p[2];
    t.prim = prim; t.mesh = mesh_of_prim[prim]; t.pad = 0;
    tris[pos] = t;
}

__global__ void k_iota(uint32_t* v, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errlen, "%s failed: %s", #x, hipGetErrorString(e_)); goto fail; } } while (0)

}  // namespace

hipError_t fovpt_build_lbvh(hipStream_t st, const float* flat, const uint32_t* mesh_of_prim, uint32_t n,
                            BvhBuildResult* out, char* err, size_t errlen)
{
    Box *boxes = nullptr, *ibox = nullptr;
    uint32_t *bounds = nullptr, *vals = nullptr, *vals_s = nullptr, *arrive = nullptr, *stats = nullptr;
    uint64_t *keys = nullptr, *keys_s = nullptr;
    int *left = nullptr, *right = nullptr, *parent_int = nullptr, *parent_leaf = nullptr, *rfirst = nullptr, *rlast = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    BvhNode* nodes = nullptr;
    TriRec* tris = nullptr;
    const uint32_t ni = n > 1 ? n - 1 : 1;
    const int B = 256;
    const uint32_t gn = (n + B - 1) / B, gi = (ni + B - 1) / B;
    uint32_t h_bounds[6];
    uint32_t h_stats[2] = {0, 0};
    hipError_t rc = hipSuccess;
    err[0] = 0;
    for (int a = 0; a < 3; a++) { h_bounds[a] = 0xffffffffu; h_bounds[3 + a] = 0u; }

    HC(hipMalloc(&boxes, sizeof(Box) * n));
    HC(hipMalloc(&ibox, sizeof(Box) * ni));
    HC(hipMalloc(&bounds, 6 * 4));
    HC(hipMalloc(&stats, 2 * 4));
    HC(hipMalloc(&keys, 8ull * n)); HC(hipMalloc(&keys_s, 8ull * n));
    HC(hipMalloc(&vals, 4ull * n)); HC(hipMalloc(&vals_s, 4ull * n));
    HC(hipMalloc(&arrive, 4ull * ni));
    HC(hipMalloc(&left, 4ull * ni)); HC(hipMalloc(&right, 4ull * ni));
    HC(hipMalloc(&parent_int, 4ull * ni)); HC(hipMalloc(&parent_leaf, 4ull * n));
    HC(hipMalloc(&rfirst, 4ull * ni)); HC(hipMalloc(&rlast, 4ull * ni));
    HC(hipMalloc(&nodes, sizeof(BvhNode) * ni));
    HC(hipMalloc(&tris, sizeof(TriRec) * n));
    HC(hipMemcpyAsync(bounds, h_bounds, sizeof(h_bounds), hipMemcpyHostToDevice, st));
    HC(hipMemsetAsync(stats, 0, 8, st));
    HC(hipMemsetAsync(arrive, 0, 4ull * ni, st));
    HC(hipMemsetAsync(nodes, 0, sizeof(BvhNode) * ni, st));

    hipLaunchKernelGGL(k_tri_bounds, dim3(gn), dim3(B), 0, st, flat, n, boxes, bounds);
    if (n <= FOVPT_LEAF_MAX) {
        hipLaunchKernelGGL(k_iota, dim3(1), dim3(64), 0, st, vals_s, n);
        hipLaunchKernelGGL(k_emit_tiny, dim3(1), dim3(1), 0, st, (int)n, boxes, nodes, stats);
    } else {
        hipLaunchKernelGGL(k_morton, dim3(gn), dim3(B), 0, st, boxes, n, bounds, keys, vals);
        HC(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_s, vals, vals_s, (size_t)n, 0, 63, st));
        HC(hipMalloc(&temp, temp_bytes));
        HC(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_s, vals, vals_s, (size_t)n, 0, 63, st));
        hipLaunchKernelGGL(k_hierarchy, dim3(gi), dim3(B), 0, st, keys_s, (int)n, left, right, parent_int, parent_leaf, rfirst, rlast);
        hipLaunchKernelGGL(k_refit, dim3(gn), dim3(B), 0, st, boxes, vals_s, (int)n, left, right, parent_int, parent_leaf, ibox, arrive);
        hipLaunchKernelGGL(k_emit_nodes, dim3(gi), dim3(B), 0, st, (int)n, left, right, rfirst, rlast, boxes, vals_s, ibox, parent_int, nodes, stats);
    }
    hipLaunchKernelGGL(k_emit_tris, dim3(gn), dim3(B), 0, st, flat, mesh_of_prim, vals_s, n, tris);
    HC(hipGetLastError());
    HC(hipMemcpyAsync(h_stats, stats, 8, hipMemcpyDeviceToHost, st));
    HC(hipStreamSynchronize(st));

    out->nodes = nodes; out->tris = tris;
    out->num_nodes = h_stats[1];
    out->max_depth = h_stats[0];
    out->node_bytes = sizeof(BvhNode) * (size_t)ni;
    out->tri_bytes = sizeof(TriRec) * (size_t)n;
    nodes = nullptr; tris = nullptr;
fail:
    if (err[0]) rc = hipErrorUnknown;
    (void)hipFree(boxes); (void)hipFree(ibox); (void)hipFree(bounds); (void)hipFree(stats); (void)hipFree(keys); (void)hipFree(keys_s);
    (void)hipFree(vals); (void)hipFree(vals_s); (void)hipFree(arrive); (void)hipFree(left); (void)hipFree(right); (void)hipFree(parent_int);
    (void)hipFree(parent_leaf); (void)hipFree(rfirst); (void)hipFree(rlast); (void)hipFree(temp); (void)hipFree(nodes); (void)hipFree(tris);
    return rc;
}
