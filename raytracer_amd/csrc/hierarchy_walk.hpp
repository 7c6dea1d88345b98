// Queries over the derived 4-wide hierarchies (accel_build.hpp): nearest hit and any hit, with the
// traversal stack, the top of the trees and the material/light tables in LDS.
#pragma once
#include "primitives.hpp"

// With -DRT_COUNT the hierarchy walks count, per kernel class, what every lane did and how many
// loop trips its wave made for it; rt_debug_counters() reads the totals.  Normal builds carry none of it.
struct WalkCount {
#ifdef RT_COUNT
    unsigned long long rays, lane_nodes, wave_nodes, lane_prims, wave_prims, lane_leaves, wave_rounds;
#endif
};
#ifdef RT_COUNT
__device__ unsigned long long g_walk_counts[48];
#define WC_ADD(cnt, field) ((cnt).field += 1)
#define WC_WAVE(cnt, field)                                                              \
    do {                                                                                 \
        if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) (cnt).field += 1; \
    } while (0)
DEV void walk_count_flush(WalkCount &c, int base)
{
    unsigned long long v[7] = {c.rays, c.lane_nodes, c.wave_nodes, c.lane_prims, c.wave_prims, c.lane_leaves, c.wave_rounds};
    for (int k = 0; k < 7; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g_walk_counts[base + k], x);
    }
}
#else
#define WC_ADD(cnt, field) ((void)0)
#define WC_WAVE(cnt, field) ((void)0)
#endif

// With -DRT_STAMP the traversal kernels record, per wave and for its first chunk, the shader clock at fixed points
// (rt_debug_stamps reads them): where a pass of a small frame spends its time.  Normal builds carry none of it.
#ifdef RT_STAMP
#define RT_STAMP_WAVES 16384
#define RT_STAMP_SLOTS 12
__device__ unsigned long long g_stamps[RT_STAMP_WAVES * RT_STAMP_SLOTS];
__device__ unsigned int g_stamp_seen[RT_STAMP_WAVES * RT_STAMP_SLOTS];
__device__ int g_stamp_select;      // which launches record: kind * 8 + generation (kind 0 closest hit, 1 shadow), + 256 * n: the wave's
                                    // n-th passage of each point is the one recorded (0: the first chunk of the wave)
// (the stamps of the sampled chunk are kept in LDS and written out when the chunk is over: a stamp that went to memory at
// once would wait for every memory operation before it and measure that instead)
DEV unsigned long long *stamp_slots()
{
    __shared__ unsigned long long slots[16 * 32];       // per wave: 16 stamps, then 32 passage counters
    return slots + (threadIdx.x >> 6) * 32;
}
DEV void stamp_at(int k, int sel)
{
    const unsigned long long w = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sel == (g_stamp_select & 255) && (threadIdx.x & 63) == 0 && w < RT_STAMP_WAVES) {
        volatile unsigned long long *slot = stamp_slots();
        volatile unsigned int *seen = reinterpret_cast<volatile unsigned int *>(slot + 16);
        const unsigned int n = seen[k];
        seen[k] = n + 1u;
        if (n == (unsigned int)(g_stamp_select >> 8)) {
            slot[k] = __builtin_amdgcn_s_memtime();      // shader clock (s_memrealtime takes a microsecond to answer)
            if (k == 8)
                for (int j = 0; j < RT_STAMP_SLOTS; ++j) g_stamps[w * RT_STAMP_SLOTS + j] = slot[j];
        }
    }
}
DEV void stamp_reset()
{
    // (at kernel start: LDS keeps what the last block left)
    if ((threadIdx.x & 63) == 0) {
        volatile unsigned long long *slot = stamp_slots();
        for (int j = 0; j < 32; ++j) slot[j] = 0;
    }
}
#define STAMP(k) stamp_at(k, STAMP_SEL)
#define STAMP_RESET() stamp_reset()
#else
#define STAMP(k) ((void)0)
#define STAMP_RESET() ((void)0)
#endif

// Stack of pending nodes in LDS: entry k of a lane lives at stack[k * block size + thread-in-block] (the walks keep the
// offset k * block size itself and add a block size per push), so a
// wave's push or pop touches 64 consecutive banks.
#define BVH_STACK 64
#define BVH_NONE (-1)
DEV int leaf_begin(int code) { return (-2 - code) >> 2; }
DEV int leaf_count(int code) { return ((-2 - code) & 3) + 1; }

// Dynamic LDS of the traversal kernels: the per-thread stacks, then a copy of the first nodes of
// the two hierarchies (breadth-first order, so the top of each tree).  Every ray reads the top
// nodes, each lane its own 128 bytes per step; served from the vector cache those reads alone keep
// the cache's 64-byte/clock port busy for longer than the arithmetic takes.  A cached node
// occupies 36 words instead of 32: with a stride of 32 every node would start in the same bank.
extern __shared__ __align__(16) int s_lds[];
#define NODE_LDS_WORDS 36
DEV void fill_node_cache(const DevScene &S, bool with_nodes)
{
    // 16 bytes per thread and trip: a block of a small frame lives about as long as this copy takes
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef __attribute__((address_space(1))) const v4i glb_v4i;
    __attribute__((address_space(3))) int *lds = (__attribute__((address_space(3))) int *)s_lds;
    if (with_nodes) {
        const int pieces = (S.tri_cached + S.sph_cached) * 8;
        glb_v4i *tsrc = (glb_v4i *)S.tri_nodes;
        glb_v4i *ssrc = (glb_v4i *)S.sph_nodes;
        for (int i = (int)threadIdx.x; i < pieces; i += (int)blockDim.x) {
            const int node = i >> 3, q = i & 7;
            const v4i v = node < S.tri_cached ? tsrc[i] : ssrc[i - S.tri_cached * 8];
            *(lds_v4i *)(lds + (S.lds_cache_word + node * NODE_LDS_WORDS + q * 4)) = v;
        }
    }
    if (with_nodes) {
        // leaf triangle records: the 80 bytes a test reads out of each 96-byte record
        glb_v4i *src = (glb_v4i *)S.btri;
        for (int i = (int)threadIdx.x; i < S.btri_cached * 5; i += (int)blockDim.x) {
            const int rec = i / 5, q = i - rec * 5;
            *(lds_v4i *)(lds + (S.lds_tri_word + i * 4)) = src[rec * 6 + q];
        }
    }
    // material and light tables (64 bytes per entry): every hit reads one of each, always the same few
    const int mpieces = S.mat_cached * 4, lpieces = S.light_cached * 4;
    glb_v4i *msrc = (glb_v4i *)S.materials;
    glb_v4i *lsrc = (glb_v4i *)S.lights;
    for (int i = (int)threadIdx.x; i < mpieces + lpieces; i += (int)blockDim.x)
        *(lds_v4i *)(lds + (S.lds_table_word + i * 4)) = i < mpieces ? msrc[i] : lsrc[i - mpieces];
    if (threadIdx.x == 0) s_lds[S.lds_ticket_word] = 0;
    __syncthreads();
}

// With W waves per block, a block works on the chunk groups blockIdx.x, blockIdx.x + gridDim.x, ... of W
// consecutive chunks each (so a short list occupies as few blocks as it can, each with all its waves).  Its
// waves draw the chunks one at a time from a ticket in LDS instead of in a fixed stride, so that all waves of
// a block finish within a chunk of each other: a block keeps its LDS and its wave slots until its last wave
// is done, and with a fixed stride the other waves' slots idled a quarter of the time.
DEV long long next_chunk(const DevScene &S)
{
    int t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(&s_lds[S.lds_ticket_word], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    const int shift = S.stack_shift - 6;          // log2 of the waves per block
    return (((long long)blockIdx.x + (long long)gridDim.x * (t >> shift)) << shift) + (t & ((1 << shift) - 1));
}

// The same ticket for blocks of any number of waves: the block's t-th chunk is chunk blockIdx.x + t * gridDim.x.
DEV long long next_chunk_strided(const DevScene &S)
{
    int t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(&s_lds[S.lds_ticket_word], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    return (long long)blockIdx.x + (long long)gridDim.x * t;
}

// A 64-byte table entry (rt_material or rt_light) from LDS when it is cached there, else from memory.
template <typename T>
DEV T load_entry(const T *table, int index, int cached, int lds_word)
{
    static_assert(sizeof(T) == 64, "table entries are 64 bytes");
    typedef int v4i __attribute__((ext_vector_type(4)));
    union { v4i q[4]; T e; } u;
    if (index < cached) {
        const __attribute__((address_space(3))) v4i *p =
            (const __attribute__((address_space(3))) v4i *)((__attribute__((address_space(3))) const int *)s_lds + (lds_word + index * 16));
#pragma unroll
        for (int k = 0; k < 4; ++k) u.q[k] = p[k];
    } else {
        const __attribute__((address_space(1))) v4i *p = (const __attribute__((address_space(1))) v4i *)(table + index);
#pragma unroll
        for (int k = 0; k < 4; ++k) u.q[k] = p[k];
    }
    return u.e;
}
DEV rt_material load_material(const DevScene &S, int index, bool lds)
{
    return load_entry(S.materials, index, lds ? S.mat_cached : 0, S.lds_table_word);
}
DEV rt_light load_light(const DevScene &S, int index, bool lds)
{
    return load_entry(S.lights, index, lds ? S.light_cached : 0, S.lds_table_word + S.mat_cached * 16);
}

// Operands of the test of leaf triangle k: from LDS when the record is cached there, else from memory.
DEV TriOps load_leaf_tri(const DevScene &S, int k)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d a, b, c, e, f;
    if (k < S.btri_cached) {
        const __attribute__((address_space(3))) v2d *p =
            (const __attribute__((address_space(3))) v2d *)((__attribute__((address_space(3))) const int *)s_lds + (S.lds_tri_word + (int)__umul24((unsigned)k, 20u)));
        a = p[0]; b = p[1]; c = p[2]; e = p[3]; f = p[4];
    } else {
        const __attribute__((address_space(1))) v2d *p = (const __attribute__((address_space(1))) v2d *)&S.btri[k];
        a = p[0]; b = p[1]; c = p[2]; e = p[3]; f = p[4];
    }
    TriOps t;
    t.v0 = mk3(a.x, a.y, b.x);
    t.ab = mk3(b.y, c.x, c.y);
    t.ac = mk3(e.x, e.y, f.x);
    t.prim = __double2loint(f.y);
    t.key = __double2hiint(f.y);
    return t;
}

// Single-precision view of a ray for the slab tests.  A child box passes when the ray's
// parameter interval inside it, [tn, tf] clipped to [0, limit], is not empty.  Per axis
// t = plane * (1/d) - origin * (1/d), one fused multiply-add per plane; the near plane of an axis
// is the lower bound when the direction is positive, else the upper bound, so the node is read
// through a per-ray byte offset and no min/max per axis is needed.
//
// The test must never reject a box the exact ray touches.  The boxes are already rounded outward.
// Spatial errors (the origin converted to float, the product origin * (1/d), the sum with
// pad * |1/d| -- each at most 2^-24 (|o| + pad) once divided by |1/d|) are covered by moving the
// origin by `pad`, eight times that, away from each plane: lower bounds are taken against o + pad, upper bounds against
// o - pad, which is the box grown by pad.  Relative errors (1/d: conversion and division; the
// rounding of the fused operation) are a few 2^-24 of t and are covered by the factor 1 + 2^-17
// on tf in the comparison.
struct RayBox4 {
    float ix, iy, iz;        // 1 / direction
    float cnx, cny, cnz;     // (origin moved away from the near plane) * (1 / direction)
    float cfx, cfy, cfz;     // (origin moved away from the far plane) * (1 / direction)
    uint32_t sx, sy, sz;     // byte offset of the near planes of an axis in a node: 0 (lower) or 16 (upper) + 32 * axis
};
DEV RayBox4 make_raybox4(d3 o, d3 d, float bound)
{
    RayBox4 r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float omax = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    const float pad = 0x1p-21f * (omax + bound);
    float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    // an (almost) axis-parallel ray is nudged for the box test only: 0 * inf would be NaN.  Rare, so the
    // fix-up sits behind a test the whole wave makes at once.
    if (__ballot(fabsf(dx) < 1e-25f || fabsf(dy) < 1e-25f || fabsf(dz) < 1e-25f)) {
        if (fabsf(dx) < 1e-25f) dx = d.x < 0 ? -1e-25f : 1e-25f;
        if (fabsf(dy) < 1e-25f) dy = d.y < 0 ? -1e-25f : 1e-25f;
        if (fabsf(dz) < 1e-25f) dz = d.z < 0 ? -1e-25f : 1e-25f;
    }
    // one ulp each: inside the relative slack of the comparison
    r.ix = __builtin_amdgcn_rcpf(dx); r.iy = __builtin_amdgcn_rcpf(dy); r.iz = __builtin_amdgcn_rcpf(dz);
    r.sx = r.ix < 0.0f ? 16u : 0u; r.sy = r.iy < 0.0f ? 48u : 32u; r.sz = r.iz < 0.0f ? 80u : 64u;
    // Lower bounds pair with o + pad and upper bounds with o - pad; the near plane of an axis is the lower bound
    // for a positive direction and the upper one for a negative direction, so in both cases
    // near: (o +- pad) * (1/d) = o * (1/d) + pad * |1/d|,  far: o * (1/d) - pad * |1/d|.
    const float px = ox * r.ix, py = oy * r.iy, pz = oz * r.iz;
    const float ax = pad * fabsf(r.ix), ay = pad * fabsf(r.iy), az = pad * fabsf(r.iz);
    r.cnx = px + ax; r.cfx = px - ax;
    r.cny = py + ay; r.cfy = py - ay;
    r.cnz = pz + az; r.cfz = pz - az;
    if (__ballot(!(omax < 1e12f))) {
        if (!(omax < 1e12f)) {
            // an origin this far away would overflow the products: let every box pass instead
            r.ix = r.iy = r.iz = 0.0f;
            r.cnx = r.cny = r.cnz = 1.0f;
            r.cfx = r.cfy = r.cfz = -3.0e38f;
        }
    }
    return r;
}
DEV float limit_f32(double limit)
{
    float lf = (float)limit;
    return lf + fabsf(lf) * 0x1p-22f;   // never below the double it came from
}

// Conservative entry distances of the four children of node `cur` (+inf: missed within
// [0, limit] or no child), and their codes.  The first `cached` nodes of this hierarchy start at LDS word `lds_word`.
// h[i] says whether child i is hit; with HITS_ONLY k is left alone (a query that visits in any order needs no distances).
template <bool HITS_ONLY>
DEV void box4_test(const accel::Bvh4Node *nodes, int cached, int lds_word, int cur, const RayBox4 &r, float limitf,
                   float k[4], int c[4], bool h[4])
{
    // The two sources are read through pointers of explicit address spaces: with generic pointers the
    // compiler merges the branches into flat loads, which take the slow path for LDS addresses.
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f lds_v4f;
    typedef __attribute__((address_space(3))) const v4i lds_v4i;
    typedef __attribute__((address_space(1))) const v4f glb_v4f;
    typedef __attribute__((address_space(1))) const v4i glb_v4i;
    v4f nx, fx, ny, fy, nz, fz;
    v4i code;
    if (cur < cached) {
        // (a 24-bit multiply issues at full rate, the 32-bit one at a quarter)
        __attribute__((address_space(3))) const int *lw = (__attribute__((address_space(3))) const int *)s_lds + (lds_word + (int)__umul24((unsigned)cur, NODE_LDS_WORDS));
        nx = *(lds_v4f *)(lw + (r.sx >> 2));
        fx = *(lds_v4f *)(lw + ((r.sx ^ 16u) >> 2));
        ny = *(lds_v4f *)(lw + (r.sy >> 2));
        fy = *(lds_v4f *)(lw + ((r.sy ^ 16u) >> 2));
        nz = *(lds_v4f *)(lw + (r.sz >> 2));
        fz = *(lds_v4f *)(lw + ((r.sz ^ 16u) >> 2));
        code = *(lds_v4i *)(lw + 24);
    } else {
        __attribute__((address_space(1))) const char *base = (__attribute__((address_space(1))) const char *)nodes;
        const uint32_t nb = (uint32_t)cur * 128u;
        nx = *(glb_v4f *)(base + (nb + r.sx));
        fx = *(glb_v4f *)(base + (nb + (r.sx ^ 16u)));
        ny = *(glb_v4f *)(base + (nb + r.sy));
        fy = *(glb_v4f *)(base + (nb + (r.sy ^ 16u)));
        nz = *(glb_v4f *)(base + (nb + r.sz));
        fz = *(glb_v4f *)(base + (nb + (r.sz ^ 16u)));
        code = *(glb_v4i *)(base + (nb + 96u));
    }
    const float inf = __builtin_inff();
#define BOX4_CHILD(i, m)                                                                                          \
    do {                                                                                                          \
        const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(nx.m, r.ix, -r.cnx), __builtin_fmaf(ny.m, r.iy, -r.cny)), \
                                     __builtin_fmaf(nz.m, r.iz, -r.cnz)), 0.0f);                                   \
        const float tf = fminf(fminf(fminf(__builtin_fmaf(fx.m, r.ix, -r.cfx), __builtin_fmaf(fy.m, r.iy, -r.cfy)), \
                                     __builtin_fmaf(fz.m, r.iz, -r.cfz)), limitf);                                 \
        c[i] = code.m;                                                                                            \
        h[i] = code.m != BVH_NONE && tn <= tf * (1.0f + 0x1p-17f);                                                 \
        if (!HITS_ONLY) k[i] = h[i] ? tn : inf;                                                                   \
    } while (0)
    BOX4_CHILD(0, x);
    BOX4_CHILD(1, y);
    BOX4_CHILD(2, z);
    BOX4_CHILD(3, w);
#undef BOX4_CHILD
}

#define BVH_SORT2(ka, ca, kb, cb)                          \
    do {                                                   \
        const bool sw_ = kb < ka;                          \
        const float tk_ = sw_ ? kb : ka;                   \
        const int tc_ = sw_ ? cb : ca;                     \
        kb = sw_ ? ka : kb; cb = sw_ ? ca : cb;            \
        ka = tk_; ca = tc_;                                \
    } while (0)

// One step down from inner node `cur`: tests its (up to) four child boxes, continues with the
// nearest hit child and stacks the others, farthest first (or pops when nothing is hit).
template <bool ORDERED>
DEV int bvh4_step(const accel::Bvh4Node *nodes, int cached, int lds_word, int cur, const RayBox4 &rb, float limitf,
                  int *stack, int &sp, int step)
{
    float kk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int cc[4];
    bool hh[4] = {false, false, false, false};
    box4_test<!ORDERED>(nodes, cached, lds_word, cur, rb, limitf, kk, cc, hh);
    float k0 = kk[0], k1 = kk[1], k2 = kk[2], k3 = kk[3];
    int c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3];
    const float inf = __builtin_inff();
    if (ORDERED) {
        // no lane of the wave has two hit children: nothing to order, nothing to stack
        const int nh = (int)hh[0] + (int)hh[1] + (int)hh[2] + (int)hh[3];
        if (__ballot(nh >= 2) == 0) {
            if (nh) return hh[0] ? c0 : (hh[1] ? c1 : (hh[2] ? c2 : c3));
            if (sp == 0) return BVH_NONE;
            sp -= step;
            return stack[sp];
        }
        BVH_SORT2(k0, c0, k1, c1);
        BVH_SORT2(k2, c2, k3, c3);
        BVH_SORT2(k0, c0, k2, c2);
        BVH_SORT2(k1, c1, k3, c3);
        BVH_SORT2(k1, c1, k2, c2);
        if (k3 < inf) { stack[sp] = c3; sp += step; }
        if (k2 < inf) { stack[sp] = c2; sp += step; }
        if (k1 < inf) { stack[sp] = c1; sp += step; }
        if (k0 < inf) return c0;
    } else {
        // any order: the last hit child is kept in hand, the others go to the stack
        int next = BVH_NONE;
        if (hh[0]) next = c0;
        if (hh[1]) { if (next != BVH_NONE) { stack[sp] = next; sp += step; } next = c1; }
        if (hh[2]) { if (next != BVH_NONE) { stack[sp] = next; sp += step; } next = c2; }
        if (hh[3]) { if (next != BVH_NONE) { stack[sp] = next; sp += step; } next = c3; }
        if (next != BVH_NONE) return next;
    }
    if (sp == 0) return BVH_NONE;
    sp -= step;
    return stack[sp];
}

struct Best {
    double t;
    int key;      // tie-break: position in the reference's visiting order
    int slot;     // record index | SLOT_SPHERE
    int prim;     // primitive id of that record
};

// Nearest hit over one hierarchy.  TRI selects the leaf record type.  Two phases per round:
// every lane descends until it holds a leaf (or is done), then the wave tests leaves together.
template <bool TRI>
DEV void bvh_closest(const DevScene &S, d3 o, d3 d, double d2, const RayBox4 &rb, Best &best, int *stack, WalkCount &wc)
{
    const double minT = .01, maxT = 10000000000.0;
    const accel::Bvh4Node *nodes = TRI ? S.tri_nodes : S.sph_nodes;
    if ((TRI ? S.n_tri_nodes : S.n_sph_nodes) == 0) return;
    const int cached = TRI ? S.tri_cached : S.sph_cached;
    const int lds_word = S.lds_cache_word + (TRI ? 0 : S.tri_cached * NODE_LDS_WORDS);
    const int step = 1 << S.stack_shift;      // entries of one lane lie a block apart; sp counts in ints
    int cur = 0, sp = 0;
    while (true) {
        const float limitf = limit_f32(best.t * (1.0 + 1e-12) + 1e-12);
        WC_WAVE(wc, wave_rounds);
        while (cur >= 0) {
            WC_ADD(wc, lane_nodes);
            WC_WAVE(wc, wave_nodes);
            cur = bvh4_step<true>(nodes, cached, lds_word, cur, rb, limitf, stack, sp, step);
        }
        if (cur == BVH_NONE) break;
        WC_ADD(wc, lane_leaves);
        const int b = leaf_begin(cur), e = b + leaf_count(cur);
        for (int k = b; k < e; ++k) {
            double t;
            int key, prim = -1;
            WC_ADD(wc, lane_prims);
            WC_WAVE(wc, wave_prims);
            if (TRI) { const TriOps tr = load_leaf_tri(S, k); t = tri_test_fast(tr, o, d, minT, maxT, best.t); key = tr.key; prim = tr.prim; }
            else { t = sph_test(&S.bsph[k], o, d, d2, minT, maxT); key = S.bsph_key[k]; }
            if (t < best.t || (t == best.t && t < maxT && key < best.key)) {
                best.t = t; best.key = key; best.slot = TRI ? k : (k | SLOT_SPHERE);
                best.prim = TRI ? prim : (int)(S.bsph_prim[k] & ~RT_SPH_TRANSPARENT);
            }
        }
        if (sp == 0) break;
        sp -= step;
        cur = stack[sp];
    }
}

// Any hit in (minT, maxT) by a primitive other than `parent`.
template <bool TRI>
DEV bool bvh_any(const DevScene &S, d3 o, d3 d, double d2, const RayBox4 &rb, int parent, double maxT, int *stack,
                 WalkCount &wc)
{
    const double minT = .01;
    const accel::Bvh4Node *nodes = TRI ? S.tri_nodes : S.sph_nodes;
    if ((TRI ? S.n_tri_nodes : S.n_sph_nodes) == 0) return false;
    const int cached = TRI ? S.tri_cached : S.sph_cached;
    const int lds_word = S.lds_cache_word + (TRI ? 0 : S.tri_cached * NODE_LDS_WORDS);
    const int step = 1 << S.stack_shift;      // entries of one lane lie a block apart; sp counts in ints
    const float limitf = limit_f32(maxT * (1.0 + 1e-12) + 1e-12);
    int cur = 0, sp = 0;
    while (true) {
        WC_WAVE(wc, wave_rounds);
        while (cur >= 0) {
            WC_ADD(wc, lane_nodes);
            WC_WAVE(wc, wave_nodes);
            cur = bvh4_step<false>(nodes, cached, lds_word, cur, rb, limitf, stack, sp, step);
        }
        if (cur == BVH_NONE) break;
        WC_ADD(wc, lane_leaves);
        const int b = leaf_begin(cur), e = b + leaf_count(cur);
        for (int k = b; k < e; ++k) {
            WC_ADD(wc, lane_prims);
            WC_WAVE(wc, wave_prims);
            if (TRI) {
                const TriOps tr = load_leaf_tri(S, k);
                if (tr.prim == parent) continue;
                if (tri_any_fast(tr, o, d, minT, maxT)) return true;
            } else {
                if ((int)(S.bsph_prim[k] & ~RT_SPH_TRANSPARENT) == parent) continue;
                if (sph_test(&S.bsph[k], o, d, d2, minT, maxT) < maxT) return true;
            }
        }
        if (sp == 0) break;
        sp -= step;
        cur = stack[sp];
    }
    return false;
}

// The root node's own test (Octree.h:1006, 1210): when it fails the reference sees nothing at all.
DEV bool root_gate(const DevScene &S, d3 o, d3 d, double d2, double sentinel)
{
    // Origin at least a little inside the root sphere and |d| near 1: the far root is then
    // > 0.86*sqrt(4*d2*|C|) > 1, far above minT and far below the sentinel, so the test passes
    // whatever the last bits of the square root and the divisions are.
    {
        const rt_node *nd = &S.nodes[0];
        const d3 emp = sub3(o, ld3(nd->center));
        const double C = dot3(emp, emp) - nd->radius * nd->radius;
        if (C < -1.0 && C > -1e6 && d2 > 0.5 && d2 < 2.0) return true;
    }
    double entry;
    double ret = bound_test(&S.nodes[0], o, d, d2, .01, sentinel, &entry);
    return !(ret >= sentinel);
}

