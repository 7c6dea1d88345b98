// The light cameras' point octree (Octree<Vector3d*>, Octree.h:1394-1497) built on the device.
//
// Same rules and the same arithmetic as the host builder (lightcam_build.cpp), level by level instead of by
// recursion.  The points stay in one array; a node is a contiguous range of it.  One level:
//   classify   every point of a node that subdivides gets the number of the FIRST child (octant order 0..7) whose
//              closed box holds it, or 8 when none does (Octree.h:1441-1446), as the low bits of a sort key whose high
//              bits are the position where its node's range begins; a point of a finished node keeps its own position
//              as its key.  Keys of different nodes cannot interleave (a node that subdivides has more than 10 points).
//   sort       one stable radix sort of (key, index) over all points: inside every node the points are now grouped by
//              child, in their previous order -- the stable partition of the host builder -- and nothing else has moved.
//   gather     points (and their owners) into the new order; where a run of equal keys begins is where a child begins.
//   children   one thread per subdividing node turns the run starts into child nodes (boxes from the parent's box by the
//              builder's arithmetic, Octree.h:1430-1465), the `leaf` flag (Octree.h:1452-1453) and the rest range.
//   owners     every point learns which node owns it on the next level, if that node subdivides (size/2 > 1e-8 and more
//              than 10 points, Octree.h:1367-1368,1427).
// Node numbers are handed out level by level (the root is node 0); the density query follows child indices only, so the
// numbering is free.  The point order is the host builder's exactly.  Weight sums are formed by one wave per node.
#pragma once
#include <hip/hip_runtime.h>

#include "lightcam_build.hpp"

namespace lcdev {

struct LevelCounters {
    int32_t n_nodes;        // nodes created so far
    int32_t n_next;         // nodes of the next level that subdivide
    int32_t overflow;       // set when the node array is full
    int32_t _pad;
};

// Impacts of a photon pass, packed: ray i's first counts[i] & 0xff slots go to out[offsets[i]...] (ray order kept: the order
// the reference inserts them in).  Also the number of castLightRay calls (counts >> 8), summed into *calls, and the
// largest |p| (sqrt((x*x + y*y) + z*z), the reference vector library's association), kept as the bit pattern of a
// non-negative double, which orders like the number itself.
__global__ __launch_bounds__(256) void k_pack_photons(const lightcam::Point *slots, const int32_t *counts, const uint32_t *offsets, int n_rays,
                                                      int slots_per_ray, lightcam::Point *out, unsigned long long *max_dist_bits,
                                                      unsigned long long *calls)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long my_calls = 0;
    double my_max = 0;
    if (i < n_rays) {
        const int n = counts[i] & 0xff;
        my_calls = (unsigned long long)(counts[i] >> 8);
        for (int k = 0; k < n; ++k) {
            const lightcam::Point p = slots[(size_t)i * slots_per_ray + k];
            out[offsets[i] + k] = p;
            const double dz = sqrt((p.x * p.x + p.y * p.y) + p.z * p.z);
            if (dz > my_max) my_max = dz;
        }
    }
    unsigned long long bits = (unsigned long long)__double_as_longlong(my_max);
    for (int off = 32; off > 0; off >>= 1) {
        my_calls += __shfl_xor(my_calls, off);
        const unsigned long long o = __shfl_xor(bits, off);
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) == 0) {
        if (my_calls) atomicAdd(calls, my_calls);
        if (bits) atomicMax(max_dist_bits, bits);
    }
}

__global__ __launch_bounds__(256) void k_impact_counts(const int32_t *counts, int n_rays, uint32_t *out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n_rays) out[i] = (uint32_t)(counts[i] & 0xff);
}

// The roots of the trees that are built together: tree t owns points [begins[t], begins[t+1]) of the shared array and is
// node t of the shared node array.  Cube centred on the origin with side 2 * (largest |impact|) + 1 (LightCam.h:312,325).
__global__ void k_roots(lightcam::Node *nodes, const unsigned long long *max_dist_bits /* stride 2 words */, const int32_t *begins, int n_trees,
                        uint8_t *subdivides, LevelCounters *ctr, int32_t *active_list)
{
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n_trees) return;
    const double max_dist = __longlong_as_double((long long)max_dist_bits[2 * t]);
    const double size = max_dist * 2 + 1;
    const double lo = 0 - size / 2;
    lightcam::Node nd;
    nd.size = size;
    nd.cx = lo + size / 2; nd.cy = lo + size / 2; nd.cz = lo + size / 2;
    nd.begin = begins[t]; nd.end = begins[t + 1];
    nd.leaf = 1;
    for (int k = 0; k < 8; ++k) nd.child[k] = -1;
    nd.rest_begin = nd.begin;
    nd.wsum = 0;
    nd.child_leaf = 0;
    nd._pad = 0;
    nodes[t] = nd;
    const bool sub = size / 2 > .00000001 && (nd.end - nd.begin) > 10;
    subdivides[t] = sub ? 1 : 0;
    if (sub) active_list[atomicAdd(&ctr->n_next, 1)] = t;
}

// Every point starts out owned by its tree's root, if that root subdivides.
__global__ __launch_bounds__(256) void k_fill_owner(int32_t *owner, int n, const int32_t *begins, int n_trees, const uint8_t *subdivides)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    int lo = 0, hi = n_trees - 1;          // the last tree whose range begins at or before i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (begins[mid] <= i) lo = mid; else hi = mid - 1;
    }
    owner[i] = subdivides[lo] ? lo : -1;
}

// Before a level: no run has begun yet in the nodes that subdivide on it; nobody is on the next level's list.
__global__ __launch_bounds__(256) void k_clear_runs(const int32_t *active, int n_active, int32_t *run_start, LevelCounters *ctr)
{
    const int a = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (a == 0) ctr->n_next = 0;
    if (a >= n_active) return;
    const int me = active[a];
    for (int k = 0; k < 9; ++k) run_start[(size_t)me * 9 + k] = -1;
}

// Sort key of every point for this level, and the identity permutation the sort carries along.
__global__ __launch_bounds__(256) void k_classify(const lightcam::Point *pts, const int32_t *owner, const lightcam::Node *nodes, int n,
                                                  unsigned long long *key, uint32_t *idx)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    idx[i] = (uint32_t)i;
    const int o = owner[i];
    if (o < 0) { key[i] = (unsigned long long)i << 4; return; }
    const lightcam::Node nd = nodes[o];
    const lightcam::Point p = pts[i];
    const double size = nd.size, h = size / 2;
    // the two intervals of every axis: [c - size/2, (c - size/2) + h] and [c, c + h]
    const double c[3] = {nd.cx, nd.cy, nd.cz}, q[3] = {p.x, p.y, p.z};
    int where = 0;
    for (int d = 0; d < 3; ++d) {
        const double lo0 = c[d] - size / 2, hi0 = lo0 + h, lo1 = c[d], hi1 = lo1 + h;
        if (lo0 <= q[d] && q[d] <= hi0) continue;
        if (lo1 <= q[d] && q[d] <= hi1) where |= 1 << d;
        else { where = 8; break; }
    }
    key[i] = ((unsigned long long)nd.begin << 4) | (unsigned long long)where;
}

// Points and owners into the sorted order; the first position of every (node, child) run.
__global__ __launch_bounds__(256) void k_gather(const lightcam::Point *pts_in, const int32_t *owner_in, const unsigned long long *key_sorted,
                                                const uint32_t *idx_sorted, int n, lightcam::Point *pts_out, int32_t *owner_out,
                                                int32_t *run_start /* per node 9 */)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n) return;
    const uint32_t src = idx_sorted[j];
    pts_out[j] = pts_in[src];
    const int o = owner_in[src];
    owner_out[j] = o;
    if (o >= 0) {
        const unsigned long long k = key_sorted[j];
        if (j == 0 || key_sorted[j - 1] != k) run_start[(size_t)o * 9 + (int)(k & 15ull)] = j;
    }
}

// Children of the nodes that subdivide on this level.
__global__ __launch_bounds__(64) void k_children(lightcam::Node *nodes, int node_capacity, const int32_t *active, int n_active,
                                                 const int32_t *run_start, LevelCounters *ctr, int32_t *next_active, uint8_t *subdivides)
{
    const int a = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (a >= n_active) return;
    const int me = active[a];
    lightcam::Node nd = nodes[me];
    int32_t start[10];
    for (int k = 0; k < 9; ++k) start[k] = run_start[(size_t)me * 9 + k];
    start[9] = nd.end;
    for (int k = 8; k >= 0; --k) if (start[k] < 0) start[k] = start[k + 1];
    int n_children = 0;
    for (int k = 0; k < 8; ++k) if (start[k + 1] > start[k]) ++n_children;
    if (start[9] == start[8]) nd.leaf = 0;          // every point found a child (Octree.h:1452-1453)
    nd.rest_begin = start[8];
    const int first = n_children ? atomicAdd(&ctr->n_nodes, n_children) : 0;
    if (first + n_children > node_capacity) { ctr->overflow = 1; nodes[me] = nd; return; }
    const double size = nd.size, h = size / 2;
    int slot = first;
    for (int k = 0; k < 8; ++k) {
        const int b = start[k], e = start[k + 1];
        if (e <= b) continue;
        const double lox = (k & 1) ? nd.cx : nd.cx - size / 2;
        const double loy = (k & 2) ? nd.cy : nd.cy - size / 2;
        const double loz = (k & 4) ? nd.cz : nd.cz - size / 2;
        const double csize = (lox + h) - lox;       // a child's side is hi.x - lo.x of its box (Octree.h:1465)
        lightcam::Node ch;
        ch.size = csize;
        ch.cx = lox + csize / 2; ch.cy = loy + csize / 2; ch.cz = loz + csize / 2;
        ch.begin = b; ch.end = e;
        ch.leaf = 1;
        for (int c = 0; c < 8; ++c) ch.child[c] = -1;
        ch.rest_begin = b;
        ch.wsum = 0;
        ch.child_leaf = 0;
        ch._pad = 0;
        nodes[slot] = ch;
        nd.child[k] = slot;
        const bool sub = csize / 2 > .00000001 && (e - b) > 10;
        subdivides[slot] = sub ? 1 : 0;
        if (sub) next_active[atomicAdd(&ctr->n_next, 1)] = slot;
        ++slot;
    }
    nodes[me] = nd;
}

// Which node owns a point on the next level: the child it went to, if that child subdivides.
__global__ __launch_bounds__(256) void k_next_owner(const unsigned long long *key_sorted, int32_t *owner, const lightcam::Node *nodes,
                                                    const uint8_t *subdivides, int n)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= n) return;
    const int o = owner[j];
    if (o < 0) return;
    const int where = (int)(key_sorted[j] & 15ull);
    int next = -1;
    if (where < 8) {
        const int ch = nodes[o].child[where];
        if (ch >= 0 && subdivides[ch]) next = ch;
    }
    owner[j] = next;
}

// After the last level, level by level from the deepest up: a node's weight sum is the sum of its children's (already
// formed) plus that of the points it keeps itself, and its children's leaf flags.  One wave per node; nodes [first, first +
// count) are one level.  (Summing every node's whole range instead made the few nodes at the top, which span all points,
// a serial tail of 4 ms per set-up.)
__global__ __launch_bounds__(256) void k_finish_level(lightcam::Node *nodes, int first, int count, const lightcam::Point *pts)
{
    const int lane = (int)(threadIdx.x & 63);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long k = first + wave; k < (long long)first + count; k += n_waves) {
        const int b = nodes[k].rest_begin, e = nodes[k].end;
        double s = 0;
        for (int i = b + lane; i < e; i += 64) s += pts[i].w;
        int bits = 0;
        if (lane < 8) {
            const int ch = nodes[k].child[lane];
            if (ch >= 0) {
                s += nodes[ch].wsum;
                if (nodes[ch].leaf) bits = 1 << lane;
            }
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        for (int off = 4; off > 0; off >>= 1) bits |= __shfl_xor(bits, off);
        if (lane == 0) { nodes[k].wsum = s; nodes[k].child_leaf = bits; }
    }
}

}  // namespace lcdev
