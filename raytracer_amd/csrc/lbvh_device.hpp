// The 4-wide hierarchies of accel_build.hpp built on the device, for scenes large enough that the host builder is what a
// scene upload waits for (the reference builds its octree on the host too, Octree.h:780-907; what is built here is the
// structure the device walks instead, with the same leaf records, so every query still returns the reference's result).
//
// A linear bounding-volume hierarchy: the primitives are sorted along a space-filling curve (63-bit Morton codes of their
// box centres, one radix sort), the binary radix tree over the sorted codes is formed in one pass with a thread per
// inner node (each finds its range and its split from the codes alone), boxes are fitted bottom-up (the second thread to
// arrive at a node joins its children's boxes), and the binary tree is folded into 4-wide nodes level by level from the
// root, so that nodes come out numbered breadth-first like the host builder's: a node takes its two children and, while
// it has room, replaces the inner child with the largest box by that child's two children; a subtree of at most
// `leaf_max` primitives becomes a leaf (its primitives are consecutive in the sorted order).  Boxes are single precision,
// rounded outward after the same relative slack the host builder gives them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/raytracer_amd.h"
#include "accel_build.hpp"

namespace lbvh {

struct Bounds {
    double lo[3], inv[3];        // scene box: lower corner and 1 / extent per axis (0 where the extent is 0)
};

struct Box6 {
    float lo[3], hi[3];
};

__device__ inline float f_down(double x)
{
    float f = __double2float_rd(x);
    return f;
}
__device__ inline float f_up(double x) { return __double2float_ru(x); }

__device__ inline unsigned long long spread21(unsigned long long v)
{
    // the low 21 bits of v, two zero bits between any two of them
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Item k's box (pushed outward like the host builder's store_box, then rounded outward to single precision) and the
// Morton code of its centre.  TRI: src[k] is a slot of `tri`; else of `sph`.
template <bool TRI>
__global__ __launch_bounds__(256) void k_item_boxes(const rt_tri_rec *tri, const rt_sph_rec *sph, const int32_t *src, int n, Bounds sb,
                                                    Box6 *boxes, unsigned long long *codes, uint32_t *order)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    double lo[3], hi[3];
    if (TRI) {
        const rt_tri_rec r = tri[src[k]];
        for (int a = 0; a < 3; ++a) {
            const double v0 = r.v0[a], v1 = r.v0[a] - r.ab[a], v2 = r.v0[a] - r.ac[a];
            lo[a] = fmin(v0, fmin(v1, v2));
            hi[a] = fmax(v0, fmax(v1, v2));
        }
    } else {
        const rt_sph_rec r = sph[src[k]];
        for (int a = 0; a < 3; ++a) { lo[a] = r.center[a] - r.radius; hi[a] = r.center[a] + r.radius; }
    }
    Box6 b;
    unsigned long long code = 0;
    for (int a = 0; a < 3; ++a) {
        const double scale = fmax(fmax(1.0, fabs(lo[a])), fmax(fabs(hi[a]), hi[a] - lo[a]));
        b.lo[a] = f_down(lo[a] - 1e-7 * scale);
        b.hi[a] = f_up(hi[a] + 1e-7 * scale);
        double c = (0.5 * (lo[a] + hi[a]) - sb.lo[a]) * sb.inv[a];
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        unsigned long long q = (unsigned long long)(c * 2097151.0);
        code |= spread21(q) << a;
    }
    boxes[k] = b;
    codes[k] = code;
    order[k] = (uint32_t)k;
}

// Length of the common prefix of the keys at sorted positions i and j (the position itself breaks ties between equal
// codes); -1 when j is outside the array.
__device__ inline int common_prefix(const unsigned long long *codes, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = codes[i], b = codes[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// The binary radix tree over n sorted codes: inner node i (0 .. n-2; node 0 is the root) covers the sorted positions
// first[i] .. last[i]; its children are inner nodes (>= 0) or leaves (~position, negative).
__global__ __launch_bounds__(256) void k_radix_tree(const unsigned long long *codes, int n, int32_t *left, int32_t *right, int32_t *first, int32_t *last,
                                                    int32_t *parent_inner, int32_t *parent_leaf)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(codes, n, i, i + 1) - common_prefix(codes, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(codes, n, i, i - d);
    int lmax = 2;
    while (common_prefix(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(codes, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(codes, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (common_prefix(codes, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    first[i] = lo;
    last[i] = hi;
    const int32_t lc = lo == gamma ? ~gamma : gamma, rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    if (lc >= 0) parent_inner[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_inner[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ inline Box6 join(const Box6 &a, const Box6 &b)
{
    Box6 r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return r;
}

// a box another thread of this launch has written (read past the compiler's and the first-level cache's copies)
__device__ inline Box6 fresh_box(const Box6 *p)
{
    Box6 r;
    const float *f = reinterpret_cast<const float *>(p);
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = __hip_atomic_load(f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.hi[k] = __hip_atomic_load(f + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return r;
}

// Boxes of the inner nodes, bottom-up: a thread per leaf climbs; at every node the first thread to arrive stops, the second
// (which then sees both children's boxes) joins them and goes on.  With the box goes the node's height: 0 for a subtree of at
// most `leaf_max` primitives (it becomes a leaf), else one more than its taller child's.
__global__ __launch_bounds__(256) void k_fit_boxes(const Box6 *item_boxes, const uint32_t *order, int n, const int32_t *left, const int32_t *right,
                                                   const int32_t *first, const int32_t *last, int leaf_max, const int32_t *parent_inner,
                                                   const int32_t *parent_leaf, Box6 *inner_boxes, int32_t *height, uint32_t *arrived)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    int node = parent_leaf[k];
    while (node >= 0) {
        // (boxes and heights are written and read with device-wide atomic accesses, past the caches: that this thread's own
        // are acknowledged before it announces itself is all the order needed; a device-wide fence would write the L2 back)
        __threadfence_block();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;
        __threadfence_block();
        const int32_t lc = left[node], rc = right[node];
        const Box6 a = lc >= 0 ? fresh_box(inner_boxes + lc) : item_boxes[order[~lc]];
        const Box6 b = rc >= 0 ? fresh_box(inner_boxes + rc) : item_boxes[order[~rc]];
        const Box6 u = join(a, b);
        const int32_t ha = lc >= 0 ? __hip_atomic_load(height + lc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int32_t hb = rc >= 0 ? __hip_atomic_load(height + rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int32_t h = last[node] - first[node] + 1 <= leaf_max ? 0 : 1 + (ha > hb ? ha : hb);
        __hip_atomic_store(height + node, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float *w = reinterpret_cast<float *>(inner_boxes + node);
        for (int c = 0; c < 3; ++c) {
            __hip_atomic_store(w + c, u.lo[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w + 3 + c, u.hi[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = parent_inner[node];
    }
}

struct FoldLevel {
    const int32_t *queue;        // binary inner nodes that become the 4-wide nodes of this level
    int32_t count;
    int32_t base;                // number of this level's first 4-wide node
    int32_t *next_queue;
    int32_t *next_count;
    int32_t next_cap;
    int32_t *overflow;
    int32_t level;               // of this level's nodes (the root's is 0)
    int32_t levels_wanted;       // the 4-wide tree should have no more levels than this (see k_fold_level)
};

struct Kid {
    int32_t code;                // >= 0: binary inner node; else leaf code -2 - (first * 4 + count - 1)
    int32_t height;              // of the binary subtree (0: a leaf)
    Box6 box;
};

__device__ inline float half_area(const Box6 &b)
{
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    return ex * ey + ey * ez + ez * ex;
}

// One level of the fold: queue entry q becomes 4-wide node base + q.
//
// Which child is opened decides the tree's depth, and the depth the stack a walk needs (three pending siblings a level),
// which has to fit the LDS beside a full compute unit's waves.  A binary subtree of height h folds into ceil(h / 2) levels
// when the taller child is opened first (each level then takes two off the height), so a child that could not end within
// the wanted number of levels even so is opened before the one with the largest box; curve-ordered trees over surfaces are
// uneven enough for this to matter from a few hundred thousand primitives on.
__global__ __launch_bounds__(256) void k_fold_level(FoldLevel L, const int32_t *left, const int32_t *right, const int32_t *first, const int32_t *last,
                                                    const Box6 *inner_boxes, const int32_t *height, const Box6 *item_boxes, const uint32_t *order,
                                                    int leaf_max, accel::Bvh4Node *nodes)
{
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= L.count) return;
    Kid kids[4];
    int nk = 0;
    auto add = [&](int32_t c) {
        Kid k;
        if (c < 0) {
            const int pos = ~c;
            k.code = -2 - (pos * 4);
            k.height = 0;
            k.box = item_boxes[order[pos]];
        } else {
            const int cnt = last[c] - first[c] + 1;
            k.code = cnt <= leaf_max ? -2 - (first[c] * 4 + (cnt - 1)) : c;
            k.height = cnt <= leaf_max ? 0 : height[c];
            k.box = inner_boxes[c];
        }
        kids[nk++] = k;
    };
    const int32_t me = L.queue[q];
    add(left[me]);
    add(right[me]);
    // open the inner child with the largest box while there is room for its two children
    const int levels_below = L.levels_wanted - 2 - L.level;      // levels a child of this node may have under it
    while (nk < 4) {
        int pick = -1, tall = -1;
        float best = -1.0f;
        for (int i = 0; i < nk; ++i) {
            if (kids[i].code < 0) continue;
            if (half_area(kids[i].box) > best) { best = half_area(kids[i].box); pick = i; }
            if (tall < 0 || kids[i].height > kids[tall].height) tall = i;
        }
        if (pick < 0) break;
        if ((kids[tall].height + 1) / 2 > levels_below + 1) pick = tall;
        const int32_t open = kids[pick].code;
        for (int i = pick; i + 1 < nk; ++i) kids[i] = kids[i + 1];
        --nk;
        add(left[open]);
        add(right[open]);
    }
    // the any-hit walk takes the last hit child first: the largest box goes there
    for (int i = 1; i < nk; ++i) {
        const Kid x = kids[i];
        int j = i;
        while (j > 0 && half_area(x.box) < half_area(kids[j - 1].box)) { kids[j] = kids[j - 1]; --j; }
        kids[j] = x;
    }
    accel::Bvh4Node nd;
    for (int c = 0; c < 4; ++c) {
        nd.child[c] = -1;
        nd._pad[c] = 0;
        for (int k = 0; k < 3; ++k) { nd.plane[k][0][c] = 3.402823466e38f; nd.plane[k][1][c] = -3.402823466e38f; }
    }
    for (int c = 0; c < nk; ++c) {
        for (int k = 0; k < 3; ++k) { nd.plane[k][0][c] = kids[c].box.lo[k]; nd.plane[k][1][c] = kids[c].box.hi[k]; }
        int32_t code = kids[c].code;
        if (code >= 0) {
            const int32_t at = atomicAdd(L.next_count, 1);
            if (at < L.next_cap) L.next_queue[at] = code; else *L.overflow = 1;
            code = L.base + L.count + at;
        }
        nd.child[c] = code;
    }
    nodes[L.base + q] = nd;
}

// Stack entries a depth-first walk below each node can have pending: going down from a node with k children leaves at
// most k - 1 of them behind.  Levels from the last to the first (children have larger numbers than their parents).
__global__ __launch_bounds__(256) void k_stack_need(const accel::Bvh4Node *nodes, int base, int count, int32_t *need)
{
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= count) return;
    const accel::Bvh4Node &nd = nodes[base + q];
    int k = 0, deepest = 0;
    for (int c = 0; c < 4; ++c) {
        if (nd.child[c] == -1) continue;
        ++k;
        if (nd.child[c] >= 0) deepest = need[nd.child[c]] > deepest ? need[nd.child[c]] : deepest;
    }
    need[base + q] = (k > 0 ? k - 1 : 0) + deepest;
}

// The leaf records in sorted order.
__global__ __launch_bounds__(256) void k_gather_tri(const rt_tri_rec *tri, const int32_t *src, const uint32_t *order, int n, rt_tri_rec *out)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * 6) return;
    const int rec = (int)(i / 6), piece = (int)(i - (long long)rec * 6);
    ((v4i *)out)[i] = ((const v4i *)(tri + src[order[rec]]))[piece];
}
__global__ __launch_bounds__(256) void k_gather_sph(const rt_sph_rec *sph, const uint32_t *sph_prim, const int32_t *sph_key, const int32_t *src,
                                                    const uint32_t *order, int n, rt_sph_rec *out, uint32_t *out_prim, int32_t *out_key)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int s = src[order[i]];
    out[i] = sph[s];
    out_prim[i] = sph_prim[s];
    out_key[i] = sph_key[s];
}

}  // namespace lbvh
