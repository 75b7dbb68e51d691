// bvh_build.hip -- LBVH construction on the GPU (gfx950).
//
// Replaces optixAccelBuild + optixAccelCompact (PT_sv5_/SimplePathtracer.cpp:677-735): one
// acceleration structure over all meshes.  Pipeline, all on the device:
//   1. per-triangle padded AABB + centroid, scene centroid bounds (wave-reduced atomics)
//   2. 63-bit Morton code of the centroid
//   3. rocPRIM radix sort of (code, primitive)
//   4. Karras 2012 binary radix tree (one thread per internal node)
//   5. bottom-up AABB refit with per-node arrival counters
//      (4'/5': default instead of 4/5 -- PLOC, SAH-guided agglomeration along the Morton order)
//   6. cost-optimal collapse into 4-wide 128-byte nodes and leaves of <= FOVPT_LEAF_MAX triangles
//      (k_dp_collapse bottom-up, k_collapse4 top-down), breadth-first node order; 48-byte triangle
//      records in depth-first leaf order
// Results never depend on the tree (see the intersection contract in include/fovpt.h), only
// speed does.
#include <cstdio>
#include <cstring>
#include <vector>

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

// leaf: ~((first triangle's offset in 16-byte units) << 3 | (count - 1)); a TriRec is three such units
__device__ inline int leaf_code(int first, int count) { return ~(((first * 3) << 3) | (count - 1)); }

__global__ void k_iota(uint32_t* v, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}


// ------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) on the Morton-sorted
// primitives: SAH-guided bottom-up agglomeration -- the "SAH refit" of the LBVH order.  Each
// round every cluster looks FOVPT_PLOC_RADIUS neighbours left and right along the Morton curve for
// the partner that minimises the surface area of the merged box; mutual choices merge.
// ------------------------------------------------------------------------------------------
#ifndef FOVPT_PLOC_RADIUS
#define FOVPT_PLOC_RADIUS 16
#endif

__device__ inline float union_area(const Box& a, const Box& b)
{
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_ploc_init(int n, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, int* __restrict__ c_node, Box* __restrict__ c_box)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    c_node[i] = ~i;                       // leaf at sorted position i
    c_box[i] = boxes[vals[i]];
}

__global__ void k_ploc_nn(int n, int force, const Box* __restrict__ c_box, int* __restrict__ nn)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (force) { int j = i ^ 1; nn[i] = j < n ? j : i; return; }
    const Box me = c_box[i];
    float best = INFINITY;
    int bj = i;
    const int lo = max(0, i - FOVPT_PLOC_RADIUS), hi = min(n - 1, i + FOVPT_PLOC_RADIUS);
    for (int j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float a = union_area(me, c_box[j]);
        if (a < best) { best = a; bj = j; }
    }
    nn[i] = bj;
}

// children coded: >= 0 internal node id, < 0 ~sorted leaf position
__global__ void k_ploc_merge(int n, const int* __restrict__ c_node, const Box* __restrict__ c_box, const int* __restrict__ nn,
                             uint32_t* __restrict__ node_counter, int* __restrict__ left, int* __restrict__ right,
                             int* __restrict__ parent_int, int* __restrict__ parent_leaf, unsigned char* __restrict__ side_int,
                             unsigned char* __restrict__ side_leaf, Box* __restrict__ ibox, uint32_t* __restrict__ size_int,
                             uint32_t* __restrict__ valid, int* __restrict__ t_node, Box* __restrict__ t_box)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    if (!mutual) { valid[i] = 1u; t_node[i] = c_node[i]; t_box[i] = c_box[i]; return; }
    if (i > j) { valid[i] = 0u; return; }
    const int node = (int)atomicAdd(node_counter, 1u);
    const int a = c_node[i], b = c_node[j];
    const Box ba = c_box[i], bb = c_box[j];
    Box u;
    for (int k = 0; k < 3; k++) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
    left[node] = a; right[node] = b;
    ibox[node] = u;
    size_int[node] = (a < 0 ? 1u : size_int[a]) + (b < 0 ? 1u : size_int[b]);
    parent_int[node] = -1;
    if (a < 0) { parent_leaf[~a] = node; side_leaf[~a] = 0; } else { parent_int[a] = node; side_int[a] = 0; }
    if (b < 0) { parent_leaf[~b] = node; side_leaf[~b] = 1; } else { parent_int[b] = node; side_int[b] = 1; }
    valid[i] = 1u; t_node[i] = node; t_box[i] = u;
}

__global__ void k_ploc_compact(int n, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ pos,
                               const int* __restrict__ t_node, const Box* __restrict__ t_box, int* __restrict__ c_node, Box* __restrict__ c_box)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !valid[i]) return;
    c_node[pos[i]] = t_node[i];
    c_box[pos[i]] = t_box[i];
}

// depth-first position of every leaf and first position of every internal node: climb to the root,
// adding the size of the left sibling whenever we come up from a right child
__global__ void k_dfs_offsets(int n, const int* __restrict__ left, const int* __restrict__ parent_int, const int* __restrict__ parent_leaf,
                              const unsigned char* __restrict__ side_int, const unsigned char* __restrict__ side_leaf,
                              const uint32_t* __restrict__ size_int, uint32_t* __restrict__ leaf_pos, uint32_t* __restrict__ node_first,
                              uint32_t* __restrict__ node_depth)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n - 1) return;
    const bool is_leaf = i < n;
    int cur = is_leaf ? parent_leaf[i] : parent_int[i - n];
    unsigned char side = is_leaf ? side_leaf[i] : side_int[i - n];
    uint32_t off = 0, depth = 0;
    while (cur >= 0) {
        if (side) { const int l = left[cur]; off += l < 0 ? 1u : size_int[l]; }
        side = side_int[cur];
        cur = parent_int[cur];
        depth++;
    }
    if (is_leaf) leaf_pos[i] = off;
    else { node_first[i - n] = off; node_depth[i - n] = depth; }
}

// LBVH (Karras) trees in the generic form: subtree sizes and first positions come from the ranges
__global__ void k_lbvh_generic(int n, const int* __restrict__ rfirst, const int* __restrict__ rlast,
                               uint32_t* __restrict__ size_int, uint32_t* __restrict__ node_first, uint32_t* __restrict__ leaf_pos)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf_pos[i] = (uint32_t)i;
    if (i < n - 1) { size_int[i] = (uint32_t)(rlast[i] - rfirst[i] + 1); node_first[i] = (uint32_t)rfirst[i]; }
}

struct Work4 { int node; uint32_t out; uint32_t depth; };

__device__ inline float box_area(const Box& b);

// ------------------------------------------------------------------------------------------
// Reinsertion (after Meister & Bittner 2018, "Parallel reinsertion for bounding volume hierarchy optimization") on the binary
// tree PLOC leaves behind, before the 2 -> 4 collapse.  PLOC merges what lies close on the Morton curve; what it cannot see is
// that a subtree would sit better somewhere else entirely -- a large foliage card among the small triangles of a facade.  Every
// node x (leaf or internal, not the root or one of its children) looks for the position that lowers the tree's SAH cost (the sum
// of the internal nodes' surface areas) most: x and its parent p are taken out (p's other child s takes p's place, the
// ancestors above shrink), and p is put back as the parent of (x, y) for some node y, whose ancestors grow.  The search climbs
// from p to the root; at every level it descends into the subtree that hangs off the path there, branch and bound:
//     gain(y) = G_k - inc(y) - A(x u y),   G_k = A(p) + what the path's nodes below level k shrink by,
//     inc(y)  = what y's ancestors inside that subtree grow by,          and below y no gain exceeds G_k - inc'(y) - A(x).
// Moves conflict when they touch the same nodes; every move stamps (gain, x) with atomicMax on the six nodes whose links it
// rewrites (x, p, s, p's parent, y, y's parent), only a move that still owns all its stamps survives, and of two survivors that
// would move their subtrees into each other one yields (k_reinsert_check2).  Then the boxes are refitted and the next round
// starts.  The tree's shape changes, the set of triangles below the root does not, and no shape changes a
// result (include/fovpt.h) -- only the node steps per ray.
// ------------------------------------------------------------------------------------------
struct TreeView {
    int n;                                  // leaves; internal nodes 0 .. n-2 (root = n-2 for PLOC), leaf l is coded ~l
    int *left, *right, *parent_int, *parent_leaf;
    Box* ibox;
    const Box* boxes;
    const uint32_t* vals;
    __device__ inline Box box(int c) const { return c < 0 ? boxes[vals[~c]] : ibox[c]; }
    __device__ inline int parent(int c) const { return c < 0 ? parent_leaf[~c] : parent_int[c]; }
    __device__ inline uint32_t slot(int c) const { return c < 0 ? (uint32_t)(n - 1) + (uint32_t)~c : (uint32_t)c; }      // index into per-node arrays
};
__device__ inline float merged_area(const Box& a, const Box& b) { return union_area(a, b); }

struct Move { int out; int pivot; float gain; };

#define FOVPT_REINSERT_STACK 96
#ifndef FOVPT_REINSERT_DEFAULT
#define FOVPT_REINSERT_DEFAULT 12          // at most this many rounds; a round that lowers the cost by less than FOVPT_REINSERT_STOP ends them
#endif
#ifndef FOVPT_REINSERT_VISITS
#define FOVPT_REINSERT_VISITS 2048
#endif
#ifndef FOVPT_REINSERT_STOP
#define FOVPT_REINSERT_STOP 0.0015f
#endif
__global__ void k_reinsert_find(TreeView T, int root, Move* __restrict__ moves, uint32_t phase, uint32_t phases)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint32_t)(2 * T.n - 1)) return;
    Move mv; mv.out = 0; mv.pivot = -1; mv.gain = 0.f;
    const int c = x < (uint32_t)(T.n - 1) ? (int)x : ~(int)(x - (uint32_t)(T.n - 1));
    const int p = T.parent(c);
    if (c == root || p < 0 || p == root || (phases > 1u && (x % phases) != phase)) { moves[x] = mv; return; }
    const Box bin = T.box(c);
    const float a_in = box_area(bin);
    float G = box_area(T.ibox[p]);
    Box shrunk = bin;                                   // (overwritten at level 0)
    int path_child = c, pivot = p, level = 0;
    int st_node[FOVPT_REINSERT_STACK];
    float st_inc[FOVPT_REINSERT_STACK];
    // The bound prunes nothing among coincident boxes (merging costs no area there), and a search that opened every node below every
    // level of its path would be quadratic in a scene of duplicates: a node's search ends after FOVPT_REINSERT_VISITS opened nodes
    // and keeps the best position found so far (a normal search opens a few dozen).
    int visits = FOVPT_REINSERT_VISITS;
    while (pivot >= 0 && visits > 0) {
        const int l = T.left[pivot], r = T.right[pivot];
        const int sib = l == path_child ? r : l;
        const Box bs = T.box(sib);
        {
            // the subtree that hangs off the path at this level, root first
            int sp = 0;
            const float m = merged_area(bin, bs);
            if (level > 0) { const float g = G - m; if (g > mv.gain) { mv.gain = g; mv.out = sib; mv.pivot = pivot; } }      // (level 0: beside its own sibling = where it is)
            if (sib >= 0) { const float inc = m - box_area(bs); if (G - inc - a_in > mv.gain) { st_node[0] = sib; st_inc[0] = inc; sp = 1; } }
            while (sp > 0 && visits > 0) {
                const int y = st_node[--sp];
                const float inc = st_inc[sp];
                if (!(G - inc - a_in > mv.gain)) continue;               // (the bound may have tightened since the push)
                visits--;
                const int ch[2] = {T.left[y], T.right[y]};
                float mc[2], ic[2];
                for (int k = 0; k < 2; k++) {
                    const Box bc = T.box(ch[k]);
                    mc[k] = merged_area(bin, bc);
                    ic[k] = inc + mc[k] - box_area(bc);
                    const float g = G - inc - mc[k];
                    if (g > mv.gain) { mv.gain = g; mv.out = ch[k]; mv.pivot = pivot; }
                }
                // the more promising child is popped first
                const int first = mc[0] <= mc[1] ? 0 : 1;
                for (int q = 1; q >= 0; q--) {
                    const int k = q ? 1 - first : first;
                    if (ch[k] >= 0 && G - ic[k] - a_in > mv.gain && sp < FOVPT_REINSERT_STACK) { st_node[sp] = ch[k]; st_inc[sp] = ic[k]; sp++; }
                }
            }
        }
        // one level up: `pivot` becomes a node of the path, with the box of what stays below it
        if (level == 0) shrunk = bs;
        else {
            for (int k = 0; k < 3; k++) { shrunk.lo[k] = fminf(shrunk.lo[k], bs.lo[k]); shrunk.hi[k] = fmaxf(shrunk.hi[k], bs.hi[k]); }
            G += box_area(T.ibox[pivot]) - box_area(shrunk);
        }
        path_child = pivot; pivot = T.parent_int[pivot]; level++;
    }
    // (a gain that is noise against the areas involved is no gain)
    if (!(mv.gain > 1e-6f * G)) { mv.pivot = -1; mv.gain = 0.f; }
    moves[x] = mv;
}

// The six nodes whose links a move rewrites: x, p, s, q = parent(p), y, parent(y).  Two moves that share one of them conflict; the
// larger (gain, x) stamp wins all six or the move is dropped.
template <typename F> __device__ inline void reinsert_touch(const TreeView& T, int c, const Move& mv, F&& f)
{
    const int p = T.parent(c), q = T.parent_int[p];
    const int s = T.left[p] == c ? T.right[p] : T.left[p];
    f(T.slot(c)); f(T.slot(p)); f(T.slot(s)); f(T.slot(q)); f(T.slot(mv.out)); f(T.slot(T.parent(mv.out)));
}
// one atomic per wave for a counter that many lanes bump (a returning or non-returning atomic on ONE word drains at ~88 per
// microsecond on this chip: 300 k candidates one by one were 3.5 ms of every round)
__device__ inline void wave_count(uint32_t* counter)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
}
__device__ inline unsigned long long move_key(const Move& mv, uint32_t x) { return ((unsigned long long)__float_as_uint(mv.gain) << 32) | (unsigned long long)x; }
__global__ void k_reinsert_lock(TreeView T, const Move* __restrict__ moves, unsigned long long* __restrict__ lock, uint32_t* __restrict__ counts)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint32_t)(2 * T.n - 1)) return;
    const Move mv = moves[x];
    if (mv.pivot < 0) return;
    const int c = x < (uint32_t)(T.n - 1) ? (int)x : ~(int)(x - (uint32_t)(T.n - 1));
    const unsigned long long key = move_key(mv, x);
    reinsert_touch(T, c, mv, [&](uint32_t i) { atomicMax(&lock[i], key); });
    wave_count(&counts[2]);                            // candidates (diagnostics)
}
// first check: the move owns its six nodes.  A survivor marks the node it moves.
__global__ void k_reinsert_check(TreeView T, Move* __restrict__ moves, const unsigned long long* __restrict__ lock, unsigned char* __restrict__ moving)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint32_t)(2 * T.n - 1)) return;
    Move mv = moves[x];
    if (mv.pivot < 0) return;
    const int c = x < (uint32_t)(T.n - 1) ? (int)x : ~(int)(x - (uint32_t)(T.n - 1));
    const unsigned long long key = move_key(mv, x);
    bool mine = true;
    reinsert_touch(T, c, mv, [&](uint32_t i) { if (lock[i] != key) mine = false; });
    if (!mine) { mv.pivot = -1; moves[x] = mv; }
    else moving[x] = 1;
}
// second check: two subtrees must not be moved INTO each other (x_A below an ancestor that move B carries off into x_A's own
// subtree would close a cycle).  A survivor whose target y has, below the search level, an ancestor that another survivor moves
// yields when that move's stamp is the larger one: in any such ring the move in front of the largest stamp yields, so no ring
// survives.  (`moving` and `lock` are only read here; the verdict goes into its own array.)
__global__ void k_reinsert_check2(TreeView T, const Move* __restrict__ moves, const unsigned long long* __restrict__ lock,
                                  const unsigned char* __restrict__ moving, unsigned char* __restrict__ yield, uint32_t* __restrict__ counts)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint32_t)(2 * T.n - 1)) return;
    const Move mv = moves[x];
    if (mv.pivot < 0) return;
    const unsigned long long key = move_key(mv, x);
    bool y = false;
    for (int a = T.parent(mv.out); a >= 0 && a != mv.pivot; a = T.parent_int[a])
        if (moving[T.slot(a)] && lock[T.slot(a)] > key) { y = true; break; }      // (a survivor owns the stamp of the node it moves)
    yield[x] = y ? 1 : 0;
    if (!y) wave_count(&counts[0]);
}
__device__ inline void set_parent(const TreeView& T, int c, int p) { if (c < 0) T.parent_leaf[~c] = p; else T.parent_int[c] = p; }
__global__ void k_reinsert_apply(TreeView T, const Move* __restrict__ moves, const unsigned char* __restrict__ yield)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= (uint32_t)(2 * T.n - 1)) return;
    const Move mv = moves[x];
    if (mv.pivot < 0 || yield[x]) return;
    // every link written here belongs to a node this move owns (k_reinsert_check), and no other surviving move reads it
    const int c = x < (uint32_t)(T.n - 1) ? (int)x : ~(int)(x - (uint32_t)(T.n - 1));
    const int p = T.parent(c), q = T.parent_int[p];
    const int s = T.left[p] == c ? T.right[p] : T.left[p];
    // p leaves: s takes its place under q
    if (T.left[q] == p) T.left[q] = s; else T.right[q] = s;
    set_parent(T, s, q);
    // p returns as the parent of (x, y) where y was
    const int y = mv.out;
    const int py = T.parent(y);                       // (read after s moved: y may be s's child, never s itself)
    if (T.left[py] == y) T.left[py] = p; else T.right[py] = p;
    T.parent_int[p] = py;
    if (T.left[p] == c) T.right[p] = y; else T.left[p] = y;
    set_parent(T, y, p);
}
// Bottom-up passes over an ARBITRARY binary tree (the reinserted one has no rounds and no ranges to go by), level-synchronous:
// a sweep is one thread per internal node, and a node is computed in the first sweep that finds both its children finished by
// an EARLIER sweep (`done` holds the sweep that finished a node, + 1; a kernel boundary lies between writer and reader, so no
// fence is needed).  The climbing form of rounds 1-3 (k_refit / k_dp_collapse: the second thread to arrive at a node computes it,
// `__threadfence` + a returning atomic per level) costs 24 ms per pass over 3.8 M leaves on this chip -- 292 of the 414 ms of a
// build with twelve reinsertion rounds -- because an agent-scope fence writes back and invalidates an XCD's L2; a sweep reads one
// flag per finished node and costs ~20 us, and a tree of 3.8 M leaves is 60-90 sweeps high.
//   MODE 0: boxes, subtree sizes, child sides, and the tree's cost (sum of the internal nodes' areas)
//   MODE 1: the collapse's dynamic programme (dp_node)
struct DpCost;
__device__ inline void dp_node(int node, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const int* __restrict__ left,
                               const int* __restrict__ right, const uint32_t* __restrict__ size_int, const Box* __restrict__ ibox, DpCost* dp);
template <int MODE>
__global__ void k_sweep_up(TreeView T, uint32_t sweep, unsigned short* __restrict__ done, uint32_t* __restrict__ size_int,
                           unsigned char* __restrict__ side_int, unsigned char* __restrict__ side_leaf, float* __restrict__ cost, DpCost* __restrict__ dp)
{
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    float area = 0.f;
    if (node < T.n - 1 && done[node] == 0) {
        const int lc = T.left[node], rc = T.right[node];
        const uint32_t dl = lc < 0 ? 1u : done[lc], dr = rc < 0 ? 1u : done[rc];
        if (dl != 0u && dl <= sweep && dr != 0u && dr <= sweep) {           // both finished BEFORE this sweep (leaves always are)
            if (MODE == 0) {
                const Box a = T.box(lc), b = T.box(rc);
                Box u;
                for (int k = 0; k < 3; k++) { u.lo[k] = fminf(a.lo[k], b.lo[k]); u.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
                T.ibox[node] = u;
                size_int[node] = (lc < 0 ? 1u : size_int[lc]) + (rc < 0 ? 1u : size_int[rc]);
                if (lc < 0) side_leaf[~lc] = 0; else side_int[lc] = 0;
                if (rc < 0) side_leaf[~rc] = 1; else side_int[rc] = 1;
                area = box_area(u);
            } else dp_node(node, T.boxes, T.vals, T.left, T.right, size_int, T.ibox, dp);
            done[node] = (unsigned short)(sweep + 1u);
        }
    }
    if (MODE == 0 && cost) {                                                // one atomic per wave
        for (int off = 32; off > 0; off >>= 1) area += __shfl_xor(area, off);
        if ((threadIdx.x & 63) == 0 && area != 0.f) atomicAdd(cost, area);
    }
}

__device__ inline float box_area(const Box& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// ---- cost-optimal 2 -> 4 collapse (the dynamic programme of Ylitie et al. 2017 for wide BVHs) -------------------
// C(x, i) = cheapest way to present the binary subtree x to a wide parent in at most i of its child slots:
//   C(x, 1) = min( area(x) * COST_LEAF                                    if x holds <= FOVPT_LEAF_MAX triangles,
//                  area(x) * 1 + min_k C(left, k) + C(right, 4 - k) )     x becomes a wide node
//   C(x, i) = min( C(x, i-1), min_k C(left, k) + C(right, i - k) )        x dissolves into its parent
// in units of node steps (a leaf step costs the traversal kernel about 2.7 of them).  Computed bottom-up,
// the second thread to arrive at a node does the work (as in k_refit); the decisions steer k_collapse4.
#ifndef FOVPT_V_DPCOLLAPSE
#define FOVPT_V_DPCOLLAPSE 1
#endif
#ifndef FOVPT_COST_LEAF
#define FOVPT_COST_LEAF 2.7f
#endif
struct alignas(16) DpCost {
    float c1, c2, c3;
    uint32_t dec;       // bits 0-1: slots for the left child when x is a wide node; 2: x is a leaf; 3: two slots = (1, 1);
};                      // bits 4-5: three slots = 0 as for two, 1 (1, 2), 2 (2, 1)

__device__ inline void dp_child(int x, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const DpCost* dp,
                                float& a1, float& a2, float& a3)
{
    if (x < 0) { a1 = a2 = a3 = box_area(boxes[vals[~x]]) * FOVPT_COST_LEAF; }
    else { const DpCost c = dp[x]; a1 = c.c1; a2 = c.c2; a3 = c.c3; }
}

__device__ inline void dp_node(int node, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const int* __restrict__ left,
                               const int* __restrict__ right, const uint32_t* __restrict__ size_int, const Box* __restrict__ ibox, DpCost* dp)
{
    float l1, l2, l3, r1, r2, r3;
    dp_child(left[node], boxes, vals, dp, l1, l2, l3);
    dp_child(right[node], boxes, vals, dp, r1, r2, r3);
    float s = l1 + r3;
    uint32_t dec = 1u;
    if (l2 + r2 < s) { s = l2 + r2; dec = 2u; }
    if (l3 + r1 < s) { s = l3 + r1; dec = 3u; }
    const float area = box_area(ibox[node]);
    float c1 = area + s;
    if (size_int[node] <= FOVPT_LEAF_MAX && area * FOVPT_COST_LEAF <= c1) { c1 = area * FOVPT_COST_LEAF; dec |= 4u; }
    // (ties dissolve the node: with degenerate boxes, all costs zero, the wide tree must still get shallower)
    float c2 = c1;
    if (l1 + r1 <= c2) { c2 = l1 + r1; dec |= 8u; }
    float c3 = c2;
    uint32_t d3 = 0u;
    if (l1 + r2 <= c3) { c3 = l1 + r2; d3 = 1u; }
    if (l2 + r1 < c3) { c3 = l2 + r1; d3 = 2u; }
    DpCost out; out.c1 = c1; out.c2 = c2; out.c3 = c3; out.dec = dec | (d3 << 4);
    dp[node] = out;
}

// any binary tree: climb from the leaves, the second thread to arrive at a node computes it (fences as in k_refit)
__global__ void k_dp_collapse(int n, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const int* __restrict__ left,
                              const int* __restrict__ right, const int* __restrict__ parent_int, const int* __restrict__ parent_leaf,
                              const uint32_t* __restrict__ size_int, const Box* __restrict__ ibox, DpCost* dp, uint32_t* __restrict__ arrive)
{
    int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    int node = parent_leaf[leaf];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrive[node], 1u) == 0u) return;   // first arrival: the sibling's costs are not there yet
        __threadfence();
        dp_node(node, boxes, vals, left, right, size_int, ibox, dp);
        node = parent_int[node];
    }
}

// PLOC trees: the nodes of one merge round have all their children in earlier rounds, and their ids are one
// contiguous range -- one plain launch per round, no fences
__global__ void k_dp_range(int first, int count, const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const int* __restrict__ left,
                           const int* __restrict__ right, const uint32_t* __restrict__ size_int, const Box* __restrict__ ibox, DpCost* dp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dp_node(first + i, boxes, vals, left, right, size_int, ibox, dp);
}

// One level of the top-down 2 -> 4 collapse: every work item is a binary node that becomes a wide node.
// Its two children are split over the four slots as k_dp_collapse decided; children that stay internal are
// queued for the next level.
__global__ void k_collapse4(int nwork, const Work4* __restrict__ work_in, Work4* __restrict__ work_out, uint32_t* __restrict__ counters,
                            const int* __restrict__ left, const int* __restrict__ right, const uint32_t* __restrict__ size_int,
                            const uint32_t* __restrict__ node_first, const uint32_t* __restrict__ leaf_pos,
                            const Box* __restrict__ boxes, const uint32_t* __restrict__ vals, const Box* __restrict__ ibox,
                            const DpCost* __restrict__ dp, BvhNode4* __restrict__ nodes, uint32_t* __restrict__ stats)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwork) return;
    const Work4 me = work_in[w];
    int ch[4];
#if FOVPT_V_DPCOLLAPSE
    // the children the dynamic programme chose: (subtree, slots) pairs are split until every pair fills one slot
    int nch = 0;
    {
        int sn[4], ss[4], sp = 0;
        const int k = (int)(dp[me.node].dec & 3u);
        sn[sp] = right[me.node]; ss[sp++] = 4 - k;
        sn[sp] = left[me.node]; ss[sp++] = k;
        while (sp > 0) {
            const int y = sn[--sp];
            int sl = ss[sp];
            if (y < 0 || sl == 1) { ch[nch++] = y; continue; }
            const uint32_t d = dp[y].dec;
            if (sl == 3) {
                const uint32_t d3 = (d >> 4) & 3u;
                if (d3 == 1u) { sn[sp] = right[y]; ss[sp++] = 2; sn[sp] = left[y]; ss[sp++] = 1; continue; }
                if (d3 == 2u) { sn[sp] = right[y]; ss[sp++] = 1; sn[sp] = left[y]; ss[sp++] = 2; continue; }
                sl = 2;
            }
            if (d & 8u) { sn[sp] = right[y]; ss[sp++] = 1; sn[sp] = left[y]; ss[sp++] = 1; }
            else ch[nch++] = y;
        }
    }
#else
    // greedy: the two children are expanded, largest surface area first, until four slots are used
    int nch = 2;
    ch[0] = left[me.node]; ch[1] = right[me.node];
    for (;;) {
        if (nch == 4) break;
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < nch; k++) {
            const int x = ch[k];
            if (x < 0 || size_int[x] <= FOVPT_LEAF_MAX) continue;          // leaf or collapsed leaf: not expandable
            const float a = box_area(ibox[x]);
            if (a > best) { best = a; pick = k; }
        }
        if (pick < 0) break;
        const int x = ch[pick];
        ch[pick] = left[x];
        ch[nch++] = right[x];
    }
#endif
    BvhNode4 nd;
    for (int k = 0; k < 4; k++) {
        BvhChild& C = nd.c[k];
        C.lox = C.loy = C.loz = C.hix = C.hiy = C.hiz = INFINITY;   // empty slot: a point at +inf never passes the slab test
        C.code = leaf_code(0, 1);
        C.pad = (uint32_t)k;                                         // closest-hit tie rank: distinct within the node
    }
    for (int k = 0; k < nch; k++) {
        const int x = ch[k];
        Box b;
        int code;
        if (x < 0) { b = boxes[vals[~x]]; code = leaf_code((int)leaf_pos[~x], 1); }
        else {
            b = ibox[x];
#if FOVPT_V_DPCOLLAPSE
            if (dp[x].dec & 4u) code = leaf_code((int)node_first[x], (int)size_int[x]);
#else
            if (size_int[x] <= FOVPT_LEAF_MAX) code = leaf_code((int)node_first[x], (int)size_int[x]);
#endif
            else {
                const uint32_t slot = atomicAdd(&counters[0], 1u);          // next free wide node
                const uint32_t q = atomicAdd(&counters[1], 1u);             // next level's queue
                Work4 nw; nw.node = x; nw.out = slot; nw.depth = me.depth + 1;
                work_out[q] = nw;
                code = (int)slot;
            }
        }
        BvhChild& C = nd.c[k];
        C.lox = b.lo[0]; C.loy = b.lo[1]; C.loz = b.lo[2];
        C.hix = b.hi[0]; C.hiy = b.hi[1]; C.hiz = b.hi[2];
        C.code = code;
    }
    nodes[me.out] = nd;
    atomicMax(&stats[0], me.depth + 1u);
    atomicAdd(&stats[1], 1u);
}

// ---- child order for occlusion rays (after Ize & Hansen 2011, "RTSAH traversal order for occlusion rays") -----------------
// An occlusion ray ends at its FIRST front-facing hit and visits a wide node's children in storage order (sorting them by
// distance was measured slower, EXPERIMENTS.md), so storage order is free to be the order that ends rays soonest: bottom-up
// every subtree gets P = the chance that a ray passing its box is stopped inside and C = what finding out costs, in node steps,
//     leaf:  P = min(1, triangle area / box area)  (random lines through a box hit a one-sided triangle's front with A / SA),  C = leaf step
//     node:  children in descending P / C  (the classic rule for "test until one succeeds");
//            h_k = SA_k / SA_node,   C = 1 + sum_k prod_{m<k} (1 - h_m P_m) h_k C_k,   P = 1 - prod_k (1 - h_k P_k)
// One launch per level of the wide tree, deepest first (a level's nodes are contiguous).  No order can change a result
// (include/fovpt.h: the hit is defined by (t, primitive id), occlusion by existence).  Closest-hit rays visit nearest first and
// rank children they enter at the SAME distance -- rays that start inside several boxes, i.e. most secondary rays near the top of
// the tree -- by BvhChild::pad, which keeps the slot the collapse chose: ranking those by this pass's order costs the street's
// closest-hit launches 6 % (the atrium's nothing), smallest or largest box first is no better (EXPERIMENTS.md).
#ifndef FOVPT_COST_LEAF_STEP
#define FOVPT_COST_LEAF_STEP 1.4f          // a leaf step in node steps (measured: 1.36, profiles/r03_step_cycles.txt)
#endif
__global__ void k_order_children(uint32_t first, uint32_t count, BvhNode4* __restrict__ nodes, const TriRec* __restrict__ tris, float2* __restrict__ pc)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t i = first + t;
    const BvhNode4 nd = nodes[i];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float P[4], C[4], sa[4], key[4];
    for (int k = 0; k < 4; k++) {
        const BvhChild& c = nd.c[k];
        P[k] = 0.f; C[k] = 1.f; sa[k] = 0.f; key[k] = -1.f;
        if (!(c.lox < INFINITY)) continue;                                       // empty slot: stays behind the used ones
        lo[0] = fminf(lo[0], c.lox); lo[1] = fminf(lo[1], c.loy); lo[2] = fminf(lo[2], c.loz);
        hi[0] = fmaxf(hi[0], c.hix); hi[1] = fmaxf(hi[1], c.hiy); hi[2] = fmaxf(hi[2], c.hiz);
        const float dx = c.hix - c.lox, dy = c.hiy - c.loy, dz = c.hiz - c.loz;
        sa[k] = 2.0f * (dx * dy + dy * dz + dz * dx);
        if (c.code < 0) {
            const uint32_t lcode = (uint32_t)~c.code, n = (lcode & 7u) + 1u, t0 = (lcode >> 3) / 3u;
            float area = 0.f;
            for (uint32_t j = 0; j < n; j++) {
                const TriRec& T = tris[t0 + j];
                const float cx = T.e1y * T.e2z - T.e1z * T.e2y, cy = T.e1z * T.e2x - T.e1x * T.e2z, cz = T.e1x * T.e2y - T.e1y * T.e2x;
                area += 0.5f * sqrtf(cx * cx + cy * cy + cz * cz);
            }
            P[k] = sa[k] > 0.f ? fminf(1.0f, area / sa[k]) : 1.0f;
            C[k] = FOVPT_COST_LEAF_STEP;
        } else {
            const float2 v = pc[c.code];
            P[k] = v.x; C[k] = v.y;
        }
        key[k] = P[k] / C[k];
    }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float sa_node = 2.0f * (dx * dy + dy * dz + dz * dx);
    int ord[4] = {0, 1, 2, 3};
    for (int a = 1; a < 4; a++)                                                   // insertion sort, descending, stable
        for (int b = a; b > 0 && key[ord[b]] > key[ord[b - 1]]; b--) { const int tmp = ord[b]; ord[b] = ord[b - 1]; ord[b - 1] = tmp; }
    float pass = 1.0f, cost = 1.0f;
    BvhNode4 out;
    for (int k = 0; k < 4; k++) {
        const int j = ord[k];
        out.c[k] = nd.c[j];
        out.c[k].pad = (uint32_t)j;              // closest-hit rays keep ranking equal entry distances by the slot the collapse chose (see below)
        if (key[j] < 0.f) continue;
        const float h = sa_node > 0.f ? fminf(1.0f, sa[j] / sa_node) : 1.0f;
        cost += pass * h * C[j];
        pass *= 1.0f - h * P[j];
    }
    nodes[i] = out;
    pc[i] = make_float2(1.0f - pass, cost);
}

// tiny scenes (n <= FOVPT_LEAF_MAX): a root whose first child is the only leaf
__global__ void k_emit_tiny(int n, const Box* __restrict__ boxes, BvhNode4* __restrict__ nodes, uint32_t* __restrict__ stats, uint32_t* __restrict__ leaf_pos)
{
    Box u = boxes[0];
    for (int i = 1; i < n; i++)
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(u.lo[k], boxes[i].lo[k]); u.hi[k] = fmaxf(u.hi[k], boxes[i].hi[k]); }
    BvhNode4 nd;
    for (int k = 0; k < 4; k++) {
        BvhChild& C = nd.c[k];
        C.lox = C.loy = C.loz = C.hix = C.hiy = C.hiz = INFINITY;
        C.code = leaf_code(0, 1);
        C.pad = (uint32_t)k;
    }
    nd.c[0].lox = u.lo[0]; nd.c[0].loy = u.lo[1]; nd.c[0].loz = u.lo[2];
    nd.c[0].hix = u.hi[0]; nd.c[0].hiy = u.hi[1]; nd.c[0].hiz = u.hi[2];
    nd.c[0].code = leaf_code(0, n);
    nodes[0] = nd;
    stats[0] = 1; stats[1] = 1;
    for (int i = 0; i < n; i++) leaf_pos[i] = (uint32_t)i;
}

__global__ void k_emit_tris_generic(const float* __restrict__ flat, const uint32_t* __restrict__ mesh_of_prim, const uint32_t* __restrict__ vals,
                                    const uint32_t* __restrict__ ref_prim, const uint32_t* __restrict__ leaf_pos, uint32_t n, TriRec* __restrict__ tris)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = ref_prim ? ref_prim[vals[i]] : vals[i];   // a reference of a split triangle stands for the whole triangle
    const float* p = flat + (size_t)prim * 9;
    TriRec t;
    t.v0x = p[0]; t.v0y = p[1]; t.v0z = p[2];
    t.e1x = p[3] - p[0]; t.e1y = p[4] - p[1]; t.e1z = p[5] - p[2];
    t.e2x = p[6] - p[0]; t.e2y = p[7] - p[1]; t.e2z = p[8] - p[2];
    t.prim = prim; t.mesh = mesh_of_prim[prim]; t.pad = 0;
    tris[leaf_pos[i]] = t;
}

// ---- spatial splits before the sort (early split clipping) ---------------------------------------------------------------
// A triangle whose box is long against the scene's typical geometry -- a diagonal card, a long sliver -- is entered into the
// build as several REFERENCES: its box is cut into k slabs along its longest axis and each reference gets the bounds of the
// triangle clipped to its slab (Ernst & Greiner 2007; the budget idea of Karras & Aila 2013).  The hierarchy is built over
// references; every reference of a triangle becomes a leaf record of the WHOLE triangle with the triangle's global primitive
// id, so a ray may test a triangle more than once and the result -- minimum (t, primitive id), any front-facing candidate --
// is what it was (tests/test_gpu_parity.py).  k = clamp(floor(longest extent / s), 1, FOVPT_SPLIT_MAX); s is found by bisection
// so that the references number (1 + budget) x the triangles.
#define FOVPT_SPLIT_MAX 8
__device__ inline uint32_t split_count(const float* __restrict__ p, float s)
{
    float ext = 0.f;
    for (int a = 0; a < 3; a++) ext = fmaxf(ext, fmaxf(p[a], fmaxf(p[3 + a], p[6 + a])) - fminf(p[a], fminf(p[3 + a], p[6 + a])));
    const float k = floorf(ext / s);
    return k >= (float)FOVPT_SPLIT_MAX ? (uint32_t)FOVPT_SPLIT_MAX : k >= 1.f ? (uint32_t)k : 1u;
}
__global__ void k_split_count(const float* __restrict__ flat, uint32_t n, float s, uint32_t* __restrict__ count, unsigned long long* __restrict__ total)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = i < n ? split_count(flat + (size_t)i * 9, s) : 0u;
    if (count && i < n) count[i] = k;
    for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off);
    if ((threadIdx.x & 63) == 0 && k) atomicAdd(total, (unsigned long long)k);
}
__global__ void k_scene_extent(const float* __restrict__ flat, uint32_t n, uint32_t* __restrict__ bounds)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int a = 0; a < 3; a++) {
        float mn = INFINITY, mx = -INFINITY;
        if (i < n) { const float* p = flat + (size_t)i * 9; mn = fminf(p[a], fminf(p[3 + a], p[6 + a])); mx = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a])); }
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[a], f2ord(mn)); atomicMax(&bounds[3 + a], f2ord(mx)); }
    }
}
// references of triangle i: padded bounds of the triangle clipped to each slab; bounds[] collects the centroid bounds as k_tri_bounds does
__global__ void k_split_emit(const float* __restrict__ flat, uint32_t n, const uint32_t* __restrict__ count, const uint32_t* __restrict__ first,
                             Box* __restrict__ boxes, uint32_t* __restrict__ ref_prim, uint32_t* __restrict__ bounds)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        const float* p = flat + (size_t)i * 9;
        float lo[3], hi[3], ext = 0.f, mag = 0.f;
        int ax = 0;
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(p[a], fminf(p[3 + a], p[6 + a])); hi[a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
            if (hi[a] - lo[a] > ext) { ext = hi[a] - lo[a]; ax = a; }
            mag = fmaxf(mag, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
        }
        const float pad = 1e-4f * ext + 1e-5f * mag + 1e-20f;      // as k_tri_bounds: of the WHOLE triangle (also covers the clipping's rounding)
        const uint32_t k = count[i], f = first[i];
        for (uint32_t j = 0; j < k; j++) {
            Box b;
            if (k == 1u) { for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; } }
            else {
                const float c0 = j == 0u ? lo[ax] : lo[ax] + (hi[ax] - lo[ax]) * ((float)j / (float)k);
                const float c1 = j + 1u == k ? hi[ax] : lo[ax] + (hi[ax] - lo[ax]) * ((float)(j + 1u) / (float)k);
                for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
                for (int e = 0; e < 3; e++) {                          // vertices inside the slab, and where the edges cross its two planes
                    const float* u = p + 3 * e;
                    const float* v = p + 3 * ((e + 1) % 3);
                    if (u[ax] >= c0 && u[ax] <= c1) for (int a = 0; a < 3; a++) { b.lo[a] = fminf(b.lo[a], u[a]); b.hi[a] = fmaxf(b.hi[a], u[a]); }
                    for (int side = 0; side < 2; side++) {
                        const float c = side ? c1 : c0;
                        if ((u[ax] < c && v[ax] > c) || (u[ax] > c && v[ax] < c)) {
                            const float t = (c - u[ax]) / (v[ax] - u[ax]);
                            for (int a = 0; a < 3; a++) {
                                const float x = a == ax ? c : u[a] + (v[a] - u[a]) * t;
                                b.lo[a] = fminf(b.lo[a], x); b.hi[a] = fmaxf(b.hi[a], x);
                            }
                        }
                    }
                }
                if (!(b.lo[0] <= b.hi[0])) { for (int a = 0; a < 3; a++) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; } }     // (cannot happen: every slab meets the triangle)
                for (int a = 0; a < 3; a++) { b.lo[a] = fmaxf(b.lo[a], lo[a]); b.hi[a] = fminf(b.hi[a], hi[a]); }
            }
            for (int a = 0; a < 3; a++) {
                b.lo[a] -= pad; b.hi[a] += pad;
                const float c = 0.5f * (b.lo[a] + b.hi[a]);
                cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c);
            }
            boxes[f + j] = b;
            ref_prim[f + j] = i;
        }
    }
    for (int a = 0; a < 3; a++) {
        float mn = cmn[a], mx = cmx[a];
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off)); mx = fmaxf(mx, __shfl_xor(mx, off)); }
        if ((threadIdx.x & 63) == 0 && mn <= mx) { atomicMin(&bounds[a], f2ord(mn)); atomicMax(&bounds[3 + a], f2ord(mx)); }
    }
}

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, errlen, "%s failed: %s", #x, hipGetErrorString(e_)); goto fail; } } while (0)

}  // namespace

hipError_t fovpt_build_lbvh(hipStream_t st, const float* flat, const uint32_t* mesh_of_prim, uint32_t n_prims, int use_ploc, float split_budget,
                            int reinsert, BvhBuildResult* out, char* err, size_t errlen)
{
    uint32_t n = n_prims;                      // build primitives: triangles, or references of triangles when splits are on
    uint32_t *ref_prim = nullptr, *split_cnt = nullptr, *split_first = nullptr;
    unsigned long long* split_total = nullptr;
    void* split_temp = nullptr;
    Box *boxes = nullptr, *ibox = nullptr;
    uint32_t *bounds = nullptr, *vals = nullptr, *vals_s = nullptr, *arrive = nullptr, *stats = nullptr;
    uint64_t *keys = nullptr, *keys_s = nullptr;
    int *left = nullptr, *right = nullptr, *parent_int = nullptr, *parent_leaf = nullptr, *rfirst = nullptr, *rlast = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    // PLOC state
    int *c_node = nullptr, *t_node = nullptr, *nn = nullptr;
    Box *c_box = nullptr, *t_box = nullptr;
    uint32_t *valid = nullptr, *pos = nullptr, *node_counter = nullptr, *size_int = nullptr, *leaf_pos = nullptr, *node_first = nullptr, *node_depth = nullptr;
    unsigned char *side_int = nullptr, *side_leaf = nullptr;
    void* scan_temp = nullptr;
    size_t scan_bytes = 0;
    // collapse state
    Work4 *work_a = nullptr, *work_b = nullptr;
    DpCost* dp = nullptr;
    std::vector<uint32_t> round_end;           // PLOC: internal nodes created up to and including each round
    std::vector<uint32_t> level_first;         // wide tree: index of the first node of every level (+ the node count at the end)
    const char* order_env = getenv("FOVPT_BVH_ORDER");                            // 0: keep the collapse's child order (A/B)
    const bool order_children = !order_env || atoi(order_env) != 0;
    float2* order_pc = nullptr;
    // reinsertion rounds on the PLOC tree (k_reinsert_find): FOVPT_REINSERT=<rounds>, 0 = off
    const char* re_env = getenv("FOVPT_REINSERT");
    const int reinsert_rounds = reinsert >= 0 ? reinsert : re_env ? atoi(re_env) : FOVPT_REINSERT_DEFAULT;
    const bool verbose = getenv("FOVPT_BVH_VERBOSE") != nullptr;
    bool tree_changed = false;
    Move* re_moves = nullptr;
    unsigned long long* re_lock = nullptr;
    uint32_t *re_arrive = nullptr, *re_scalars = nullptr;      // re_scalars: [0] moves applied, [1] sum of internal areas (float), [2] candidates
    unsigned char* re_flags = nullptr;                         // [0, nall): the node is moved by a surviving move; [nall, 2 nall): the move yields
    uint32_t* dp_arrive = nullptr;
    uint32_t* counters = nullptr;
    BvhNode4* nodes = nullptr;
    TriRec* tris = nullptr;
    const int B = 256;
    uint32_t ni = 1, gn = 1, gi = 1;           // (assigned once the number of build primitives is known)
    bool have_boxes = false;
    uint32_t h_bounds[6];
    uint32_t h_stats[2] = {0, 0};
    uint32_t h_counters[2] = {1, 0};
    hipError_t rc = hipSuccess;
    err[0] = 0;
    // bottom-up sweeps over the binary tree until the root is finished (k_sweep_up); false + err on failure
    auto sweep_up = [&](int mode, const TreeView& T, float* cost) -> bool {
        const uint32_t nint = (uint32_t)T.n - 1u, g = (nint + 255u) / 256u;
        if (!re_arrive && hipMalloc(&re_arrive, 2ull * nint) != hipSuccess) { snprintf(err, errlen, "hipMalloc failed (sweep flags)"); return false; }
        if (hipMemsetAsync(re_arrive, 0, 2ull * nint, st) != hipSuccess) { snprintf(err, errlen, "hipMemsetAsync failed (sweep flags)"); return false; }
        unsigned short* done = (unsigned short*)re_arrive;
        for (uint32_t sweep = 0; sweep < 65000u;) {
            for (int k = 0; k < 16; k++, sweep++) {
                if (mode == 0) hipLaunchKernelGGL(k_sweep_up<0>, dim3(g), dim3(256), 0, st, T, sweep, done, size_int, side_int, side_leaf, cost, (DpCost*)nullptr);
                else hipLaunchKernelGGL(k_sweep_up<1>, dim3(g), dim3(256), 0, st, T, sweep, done, size_int, side_int, side_leaf, (float*)nullptr, dp);
            }
            unsigned short root_done = 0;
            if (hipMemcpyAsync(&root_done, done + (nint - 1u), 2, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                snprintf(err, errlen, "bottom-up sweep failed: %s", hipGetErrorString(hipGetLastError()));
                return false;
            }
            if (root_done) return true;
        }
        snprintf(err, errlen, "bottom-up sweeps did not reach the root");
        return false;
    };
    for (int a = 0; a < 3; a++) { h_bounds[a] = 0xffffffffu; h_bounds[3 + a] = 0u; }
    HC(hipMalloc(&bounds, 6 * 4));
    if (split_budget > 0.f && n_prims > FOVPT_LEAF_MAX) {
        // how long may a box be before it is cut: bisection on s for (1 + budget) x n_prims references
        const uint32_t gp = (n_prims + B - 1) / B;
        unsigned long long h_total = 0, want = (unsigned long long)((double)n_prims * (1.0 + (double)split_budget));
        HC(hipMalloc(&split_total, 8));
        HC(hipMemcpyAsync(bounds, h_bounds, sizeof(h_bounds), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scene_extent, dim3(gp), dim3(B), 0, st, flat, n_prims, bounds);
        uint32_t hb[6];
        HC(hipMemcpyAsync(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
        HC(hipStreamSynchronize(st));
        float hi_s = 0.f;
        for (int a = 0; a < 3; a++) hi_s = fmaxf(hi_s, ord2f(hb[3 + a]) - ord2f(hb[a]));
        float lo_s = hi_s * 1e-7f, s_cut = hi_s;
        for (int it = 0; it < 24 && hi_s > 0.f; it++) {
            const float mid = sqrtf(lo_s * hi_s);                   // the extents span orders of magnitude
            HC(hipMemsetAsync(split_total, 0, 8, st));
            hipLaunchKernelGGL(k_split_count, dim3(gp), dim3(B), 0, st, flat, n_prims, mid, (uint32_t*)nullptr, split_total);
            HC(hipMemcpyAsync(&h_total, split_total, 8, hipMemcpyDeviceToHost, st));
            HC(hipStreamSynchronize(st));
            if (h_total > want) lo_s = mid; else { hi_s = mid; s_cut = mid; }       // (the count falls as s grows)
        }
        HC(hipMalloc(&split_cnt, 4ull * n_prims)); HC(hipMalloc(&split_first, 4ull * n_prims));
        HC(hipMemsetAsync(split_total, 0, 8, st));
        hipLaunchKernelGGL(k_split_count, dim3(gp), dim3(B), 0, st, flat, n_prims, s_cut, split_cnt, split_total);
        HC(hipMemcpyAsync(&h_total, split_total, 8, hipMemcpyDeviceToHost, st));
        HC(hipStreamSynchronize(st));
        if (h_total > n_prims && h_total < (1ull << 27)) {
            size_t tb = 0;
            HC(rocprim::exclusive_scan(nullptr, tb, split_cnt, split_first, 0u, (size_t)n_prims, rocprim::plus<uint32_t>(), st));
            HC(hipMalloc(&split_temp, tb));
            HC(rocprim::exclusive_scan(split_temp, tb, split_cnt, split_first, 0u, (size_t)n_prims, rocprim::plus<uint32_t>(), st));
            n = (uint32_t)h_total;
            HC(hipMalloc(&boxes, sizeof(Box) * n));
            HC(hipMalloc(&ref_prim, 4ull * n));
            HC(hipMemcpyAsync(bounds, h_bounds, sizeof(h_bounds), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_split_emit, dim3(gp), dim3(B), 0, st, flat, n_prims, split_cnt, split_first, boxes, ref_prim, bounds);
            HC(hipGetLastError());
            have_boxes = true;
        }
    }
    ni = n > 1 ? n - 1 : 1;
    gn = (n + B - 1) / B; gi = (ni + B - 1) / B;

    if (!have_boxes) HC(hipMalloc(&boxes, sizeof(Box) * n));
    HC(hipMalloc(&ibox, sizeof(Box) * ni));
    HC(hipMalloc(&stats, 2 * 4));
    HC(hipMalloc(&counters, 2 * 4));
    HC(hipMalloc(&keys, 8ull * n)); HC(hipMalloc(&keys_s, 8ull * n));
    HC(hipMalloc(&vals, 4ull * n)); HC(hipMalloc(&vals_s, 4ull * n));
    HC(hipMalloc(&left, 4ull * ni)); HC(hipMalloc(&right, 4ull * ni));
    HC(hipMalloc(&parent_int, 4ull * ni)); HC(hipMalloc(&parent_leaf, 4ull * n));
    HC(hipMalloc(&size_int, 4ull * ni)); HC(hipMalloc(&leaf_pos, 4ull * n)); HC(hipMalloc(&node_first, 4ull * ni));
    HC(hipMalloc(&nodes, sizeof(BvhNode4) * ni));              // a wide node replaces >= 1 binary node
    HC(hipMalloc(&tris, sizeof(TriRec) * n));
    HC(hipMemsetAsync(stats, 0, 8, st));
    if (!have_boxes) {
        HC(hipMemcpyAsync(bounds, h_bounds, sizeof(h_bounds), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_tri_bounds, dim3(gn), dim3(B), 0, st, flat, n, boxes, bounds);
    }
    if (n <= FOVPT_LEAF_MAX) {
        hipLaunchKernelGGL(k_iota, dim3(1), dim3(64), 0, st, vals_s, n);
        hipLaunchKernelGGL(k_emit_tiny, dim3(1), dim3(1), 0, st, (int)n, boxes, nodes, stats, leaf_pos);
    } else {
        int root = 0;
        hipLaunchKernelGGL(k_morton, dim3(gn), dim3(B), 0, st, boxes, n, bounds, keys, vals);
        HC(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_s, vals, vals_s, (size_t)n, 0, 63, st));
        HC(hipMalloc(&temp, temp_bytes));
        HC(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_s, vals, vals_s, (size_t)n, 0, 63, st));
        if (!use_ploc) {
            HC(hipMalloc(&arrive, 4ull * ni));
            HC(hipMalloc(&rfirst, 4ull * ni)); HC(hipMalloc(&rlast, 4ull * ni));
            HC(hipMemsetAsync(arrive, 0, 4ull * ni, st));
            hipLaunchKernelGGL(k_hierarchy, dim3(gi), dim3(B), 0, st, keys_s, (int)n, left, right, parent_int, parent_leaf, rfirst, rlast);
            hipLaunchKernelGGL(k_refit, dim3(gn), dim3(B), 0, st, boxes, vals_s, (int)n, left, right, parent_int, parent_leaf, ibox, arrive);
            hipLaunchKernelGGL(k_lbvh_generic, dim3(gn), dim3(B), 0, st, (int)n, rfirst, rlast, size_int, node_first, leaf_pos);
            root = 0;
        } else {
            HC(hipMalloc(&c_node, 4ull * n)); HC(hipMalloc(&t_node, 4ull * n)); HC(hipMalloc(&nn, 4ull * n));
            HC(hipMalloc(&c_box, sizeof(Box) * n)); HC(hipMalloc(&t_box, sizeof(Box) * n));
            HC(hipMalloc(&valid, 4ull * n)); HC(hipMalloc(&pos, 4ull * n));
            HC(hipMalloc(&node_counter, 4)); HC(hipMalloc(&node_depth, 4ull * ni));
            HC(hipMalloc(&side_int, ni)); HC(hipMalloc(&side_leaf, n));
            HC(hipMemsetAsync(node_counter, 0, 4, st));
            HC(hipMemsetAsync(side_int, 0, ni, st));
            HC(rocprim::exclusive_scan(nullptr, scan_bytes, valid, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
            HC(hipMalloc(&scan_temp, scan_bytes));
            hipLaunchKernelGGL(k_ploc_init, dim3(gn), dim3(B), 0, st, (int)n, boxes, vals_s, c_node, c_box);
            uint32_t m = n;
            int force = 0, rounds = 0;
            round_end.clear();
            while (m > 1) {
                const uint32_t gm = (m + B - 1) / B;
                hipLaunchKernelGGL(k_ploc_nn, dim3(gm), dim3(B), 0, st, (int)m, force, c_box, nn);
                hipLaunchKernelGGL(k_ploc_merge, dim3(gm), dim3(B), 0, st, (int)m, c_node, c_box, nn, node_counter, left, right, parent_int,
                                   parent_leaf, side_int, side_leaf, ibox, size_int, valid, t_node, t_box);
                HC(rocprim::exclusive_scan(scan_temp, scan_bytes, valid, pos, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
                hipLaunchKernelGGL(k_ploc_compact, dim3(gm), dim3(B), 0, st, (int)m, valid, pos, t_node, t_box, c_node, c_box);
                uint32_t last_pos = 0, last_valid = 0, made = 0;
                HC(hipMemcpyAsync(&made, node_counter, 4, hipMemcpyDeviceToHost, st));
                HC(hipMemcpyAsync(&last_pos, pos + (m - 1), 4, hipMemcpyDeviceToHost, st));
                HC(hipMemcpyAsync(&last_valid, valid + (m - 1), 4, hipMemcpyDeviceToHost, st));
                HC(hipStreamSynchronize(st));
                round_end.push_back(made);
                const uint32_t m2 = last_pos + last_valid;
                // (Almost) no mutual pair this round -- coincident geometry: among equal boxes everybody picks the first
                // candidate of its window and one pair merges per round -- so the next round pairs neighbours.
                force = ((uint64_t)(m - m2) * 64u < m) ? 1 : 0;
                m = m2;
                if (++rounds > 4096) { snprintf(err, errlen, "PLOC did not converge"); goto fail; }
            }
            if (reinsert_rounds > 0 && n > 16) {
                // ---- reinsertion rounds on the PLOC tree (see k_reinsert_find)
                TreeView T;
                T.n = (int)n; T.left = left; T.right = right; T.parent_int = parent_int; T.parent_leaf = parent_leaf;
                T.ibox = ibox; T.boxes = boxes; T.vals = vals_s;
                const uint32_t nall = 2 * n - 1, gall2 = (nall + B - 1) / B;
                HC(hipMalloc(&re_moves, sizeof(Move) * (size_t)nall));
                HC(hipMalloc(&re_lock, 8ull * nall));
                HC(hipMalloc(&re_arrive, 2ull * ni));                     // (the sweep that finished each node, + 1)
                HC(hipMalloc(&re_scalars, 16));
                HC(hipMalloc(&re_flags, 2ull * nall));
                // the cost of the tree as PLOC left it (what the first round's gain is measured against)
                float cost_prev = 0.f;
                HC(hipMemsetAsync(re_scalars, 0, 16, st));
                if (!sweep_up(0, T, (float*)(re_scalars + 1))) goto fail;
                HC(hipMemcpyAsync(&cost_prev, re_scalars + 1, 4, hipMemcpyDeviceToHost, st));
                HC(hipStreamSynchronize(st));
                if (verbose) fprintf(stderr, "[fovpt bvh] PLOC tree: sum of internal areas %.6g\n", (double)cost_prev);
                for (int round = 0; round < reinsert_rounds; round++) {
                    uint32_t applied = 0;
                    float cost = 0.f;
                    hipLaunchKernelGGL(k_reinsert_find, dim3(gall2), dim3(B), 0, st, T, (int)n - 2, re_moves, 0u, 1u);
                    HC(hipMemsetAsync(re_lock, 0, 8ull * nall, st));
                    HC(hipMemsetAsync(re_flags, 0, 2ull * nall, st));
                    HC(hipMemsetAsync(re_scalars, 0, 16, st));
                    hipLaunchKernelGGL(k_reinsert_lock, dim3(gall2), dim3(B), 0, st, T, re_moves, re_lock, re_scalars);
                    hipLaunchKernelGGL(k_reinsert_check, dim3(gall2), dim3(B), 0, st, T, re_moves, re_lock, re_flags);
                    hipLaunchKernelGGL(k_reinsert_check2, dim3(gall2), dim3(B), 0, st, T, re_moves, re_lock, re_flags, re_flags + nall, re_scalars);
                    hipLaunchKernelGGL(k_reinsert_apply, dim3(gall2), dim3(B), 0, st, T, re_moves, re_flags + nall);
                    if (!sweep_up(0, T, (float*)(re_scalars + 1))) goto fail;
                    uint32_t candidates = 0;
                    HC(hipMemcpyAsync(&applied, re_scalars, 4, hipMemcpyDeviceToHost, st));
                    HC(hipMemcpyAsync(&cost, re_scalars + 1, 4, hipMemcpyDeviceToHost, st));
                    HC(hipMemcpyAsync(&candidates, re_scalars + 2, 4, hipMemcpyDeviceToHost, st));
                    HC(hipStreamSynchronize(st));
                    HC(hipGetLastError());
                    tree_changed = tree_changed || applied > 0;
                    if (verbose) fprintf(stderr, "[fovpt bvh] reinsertion round %d: %u of %u candidate moves, sum of internal areas %.6g\n", round, applied, candidates, (double)cost);
                    if (applied == 0 || !(cost_prev - cost > FOVPT_REINSERT_STOP * cost_prev)) break;      // (nothing much left to gain)
                    cost_prev = cost;
                }
            }
            const uint32_t gall = (2 * n - 1 + B - 1) / B;
            hipLaunchKernelGGL(k_dfs_offsets, dim3(gall), dim3(B), 0, st, (int)n, left, parent_int, parent_leaf, side_int, side_leaf, size_int,
                               leaf_pos, node_first, node_depth);
            root = (int)n - 2;                                 // the last merge created the root
        }
        // ---- 2 -> 4 collapse: costs bottom-up, then one launch per level of the wide tree
        HC(hipMalloc(&dp, sizeof(DpCost) * ni));
        if (use_ploc && !tree_changed) {
            uint32_t first = 0;
            for (uint32_t end : round_end) {
                if (end > first)
                    hipLaunchKernelGGL(k_dp_range, dim3((end - first + B - 1) / B), dim3(B), 0, st, (int)first, (int)(end - first), boxes, vals_s,
                                       left, right, size_int, ibox, dp);
                first = end;
            }
        } else if (use_ploc) {
            // a reinserted tree: the same programme in bottom-up sweeps (k_sweep_up)
            TreeView T;
            T.n = (int)n; T.left = left; T.right = right; T.parent_int = parent_int; T.parent_leaf = parent_leaf;
            T.ibox = ibox; T.boxes = boxes; T.vals = vals_s;
            if (!sweep_up(1, T, nullptr)) goto fail;
        } else {
            HC(hipMalloc(&dp_arrive, 4ull * ni));
            HC(hipMemsetAsync(dp_arrive, 0, 4ull * ni, st));
            hipLaunchKernelGGL(k_dp_collapse, dim3(gn), dim3(B), 0, st, (int)n, boxes, vals_s, left, right, parent_int, parent_leaf, size_int, ibox,
                               dp, dp_arrive);
        }
        HC(hipMalloc(&work_a, sizeof(Work4) * ni)); HC(hipMalloc(&work_b, sizeof(Work4) * ni));
        Work4 w0; w0.node = root; w0.out = 0; w0.depth = 0;
        HC(hipMemcpyAsync(work_a, &w0, sizeof(w0), hipMemcpyHostToDevice, st));
        HC(hipMemcpyAsync(counters, h_counters, 8, hipMemcpyHostToDevice, st));
        uint32_t nwork = 1;
        int levels = 0;
        level_first.push_back(0u);
        level_first.push_back(1u);             // the root is node 0; the nodes of level L + 1 are allocated by the launch of level L
        while (nwork > 0) {
            hipLaunchKernelGGL(k_collapse4, dim3((nwork + B - 1) / B), dim3(B), 0, st, (int)nwork, work_a, work_b, counters, left, right,
                               size_int, node_first, leaf_pos, boxes, vals_s, ibox, dp, nodes, stats);
            HC(hipMemcpyAsync(h_counters, counters, 8, hipMemcpyDeviceToHost, st));
            HC(hipStreamSynchronize(st));
            nwork = h_counters[1];
            if (nwork > 0) level_first.push_back(h_counters[0]);
            h_counters[1] = 0;
            HC(hipMemcpyAsync(counters + 1, &h_counters[1], 4, hipMemcpyHostToDevice, st));
            Work4* t = work_a; work_a = work_b; work_b = t;
            if (++levels > 4096) { snprintf(err, errlen, "BVH collapse did not terminate"); goto fail; }
        }
    }
    hipLaunchKernelGGL(k_emit_tris_generic, dim3(gn), dim3(B), 0, st, flat, mesh_of_prim, vals_s, ref_prim, leaf_pos, n, tris);
    HC(hipGetLastError());
    if (order_children && level_first.size() >= 2) {
        // children in the order that ends occlusion rays soonest: one launch per level, deepest first
        const uint32_t total = level_first.back();
        HC(hipMalloc(&order_pc, sizeof(float2) * (size_t)total));
        for (size_t L = level_first.size() - 1; L-- > 0;) {
            const uint32_t first = level_first[L], count = level_first[L + 1] - first;
            if (count) hipLaunchKernelGGL(k_order_children, dim3((count + B - 1) / B), dim3(B), 0, st, first, count, nodes, tris, order_pc);
        }
        HC(hipGetLastError());
    }
    HC(hipMemcpyAsync(h_stats, stats, 8, hipMemcpyDeviceToHost, st));
    HC(hipStreamSynchronize(st));

    {
        // The hierarchy that stays: ONE allocation, the nodes actually emitted (the build array is sized for the binary
        // tree, ~8 x as many) followed by the triangles, so the traversal reaches both through one base register and a
        // 32-bit offset, and the build's slack goes back to the allocator.
        const size_t node_bytes = sizeof(BvhNode4) * (size_t)h_stats[1], tri_bytes = sizeof(TriRec) * (size_t)n;
        const size_t tri_off = (node_bytes + 255) & ~(size_t)255;
        char* scene = nullptr;
        HC(hipMalloc(&scene, tri_off + tri_bytes + 64));          // + slack: a lane may fetch 16 bytes past its record
        if (hipMemcpyAsync(scene, nodes, node_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess
            || hipMemcpyAsync(scene + tri_off, tris, tri_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess
            || hipMemsetAsync(scene + tri_off + tri_bytes, 0, 64, st) != hipSuccess
            || hipStreamSynchronize(st) != hipSuccess) {
            (void)hipFree(scene);
            snprintf(err, errlen, "copying the hierarchy failed");
            goto fail;
        }
        out->nodes = (BvhNode4*)scene; out->tris = (TriRec*)(scene + tri_off);      // tris is INSIDE the nodes allocation
        out->num_nodes = h_stats[1];
        out->num_refs = n;
        out->max_depth = h_stats[0];
        out->reinserted = tree_changed ? 1u : 0u;
        out->node_bytes = node_bytes;
        out->tri_bytes = tri_bytes;
    }
fail:
    if (err[0]) rc = hipErrorUnknown;
    (void)hipFree(ref_prim); (void)hipFree(split_cnt); (void)hipFree(split_first); (void)hipFree(split_total); (void)hipFree(split_temp);
    (void)hipFree(boxes); (void)hipFree(ibox); (void)hipFree(bounds); (void)hipFree(stats); (void)hipFree(keys); (void)hipFree(keys_s);
    (void)hipFree(vals); (void)hipFree(vals_s); (void)hipFree(arrive); (void)hipFree(left); (void)hipFree(right); (void)hipFree(parent_int);
    (void)hipFree(parent_leaf); (void)hipFree(rfirst); (void)hipFree(rlast); (void)hipFree(temp); (void)hipFree(nodes); (void)hipFree(tris);
    (void)hipFree(c_node); (void)hipFree(t_node); (void)hipFree(nn); (void)hipFree(c_box); (void)hipFree(t_box); (void)hipFree(valid); (void)hipFree(pos);
    (void)hipFree(node_counter); (void)hipFree(size_int); (void)hipFree(leaf_pos); (void)hipFree(node_first); (void)hipFree(node_depth);
    (void)hipFree(dp); (void)hipFree(dp_arrive); (void)hipFree(order_pc);
    (void)hipFree(re_moves); (void)hipFree(re_lock); (void)hipFree(re_arrive); (void)hipFree(re_scalars); (void)hipFree(re_flags);
    (void)hipFree(side_int); (void)hipFree(side_leaf); (void)hipFree(scan_temp); (void)hipFree(work_a); (void)hipFree(work_b); (void)hipFree(counters);
    return rc;
}
