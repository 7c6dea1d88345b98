// Queries answered from the grids of viewgrid.hpp: the nearest hit of a primary ray from the list of its 8 x 8-pixel
// tile, and "any hit" of a shadow ray from the list of the bin in which its direction lies on the cube map around its
// light.  A list holds every leaf record the ray can meet (and more); each entry is judged by the reference's own test,
// so the answers are those of the hierarchy walks (hierarchy_walk.hpp) and of the octree walk they stand for.
#pragma once
#include "hierarchy_walk.hpp"
#include "viewgrid.hpp"

#define GRID_SPHERE 0x80000000u

// Bin of the direction from light position L to point Pt on a cube map of res x res bins per face (faces +x, -x, +y,
// -y, +z, -z; on a face the two other coordinates divided by the major one, in axis order).  Single precision: the
// lists are complete for every direction within 2^-16 of a bin's border on either side.
DEV uint32_t light_bin(d3 Pt, d3 L, int res)
{
    const float dx = (float)(Pt.x - L.x), dy = (float)(Pt.y - L.y), dz = (float)(Pt.z - L.z);
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    int face;
    float w, a, b;
    if (ax >= ay && ax >= az) { face = dx < 0.0f ? 1 : 0; w = ax; a = dy; b = dz; }
    else if (ay >= az) { face = dy < 0.0f ? 3 : 2; w = ay; a = dx; b = dz; }
    else { face = dz < 0.0f ? 5 : 4; w = az; a = dx; b = dy; }
    if (!(w > 0.0f) || !(w < 3.0e38f)) return 0u;       // the point is the light itself (no hit is possible), or not a number
    const float iw = __builtin_amdgcn_rcpf(w), half = 0.5f * (float)res;
    int bx = (int)floorf(__builtin_fmaf(a * iw, half, half)), by = (int)floorf(__builtin_fmaf(b * iw, half, half));
    bx = bx < 0 ? 0 : (bx >= res ? res - 1 : bx);
    by = by < 0 ? 0 : (by >= res ? res - 1 : by);
    return (uint32_t)((face * res + by) * res + bx);
}

typedef uint32_t grid_v2u __attribute__((ext_vector_type(2)));
typedef uint32_t grid_v4u __attribute__((ext_vector_type(4)));

// The entries of one bin in order: the first two come with the bin's head, the others from the list (fetched one ahead).
struct GridList {
    const __attribute__((address_space(1))) grid_v2u *rest;     // the list (entry k at rest[k])
    uint32_t count;
    grid_v2u cur, after;
};
DEV GridList grid_list(const vg::Bin *bins, uint32_t bin, const vg::Entry *entries)
{
    const __attribute__((address_space(1))) grid_v4u *h = (const __attribute__((address_space(1))) grid_v4u *)(bins + bin);
    const grid_v4u a = h[0], b = h[1];
    GridList l;
    l.rest = (const __attribute__((address_space(1))) grid_v2u *)entries + a.x;
    l.count = a.y;
    l.cur.x = a.z; l.cur.y = a.w;
    l.after.x = b.x; l.after.y = b.y;
    return l;
}
// entry k is in l.cur; makes entry k + 1 current
DEV void grid_next(GridList &l, uint32_t k)
{
    l.cur = l.after;
    if (k + 2 < l.count) l.after = l.rest[k + 2];
}

// Any hit in (minT, maxT) by an entry of the bin's list other than `parent`.  A hit at parameter t lies maxT + .01 - t
// from the light (the ray is a unit vector towards it, maxT its length less .01), so at most maxT from it: the list is in
// ascending order of a lower bound of that distance and the loop ends at the first entry beyond.
DEV bool grid_any(const DevScene &S, const vg::Bin *bins, uint32_t bin, const vg::Entry *entries, d3 o, d3 d, double d2, int parent, double maxT,
                  WalkCount &wc)
{
    const double minT = .01;
    const float far = (float)maxT * (1.0f + 0x1p-16f) + 1e-6f;
    bool hit = false;
    GridList l = grid_list(bins, bin, entries);
    for (uint32_t k = 0; k < l.count && !hit; ++k) {
        const grid_v2u e = l.cur;
        if (__uint_as_float(e.y) > far) break;
        grid_next(l, k);
        WC_ADD(wc, lane_prims);
        WC_WAVE(wc, wave_prims);
        if (e.x & GRID_SPHERE) {
            const uint32_t s = e.x & ~GRID_SPHERE;
            if ((int)(S.bsph_prim[s] & ~RT_SPH_TRANSPARENT) == parent) continue;
            hit = sph_test(&S.bsph[s], o, d, d2, minT, maxT) < maxT;
        } else {
            const TriOps tr = load_leaf_tri(S, (int)e.x);
            if (tr.prim == parent) continue;
            hit = tri_any_fast(tr, o, d, minT, maxT);
        }
    }
    return hit;
}

// Nearest hit over the bin's list, ties by the records' keys (the reference's visiting order).  The parameter of a
// hit is its distance from the eye (unit direction): an entry whose lower bound lies beyond the nearest hit so far ends
// the loop (a tie is not beyond).
// (`l`: the bin's list with its head fetched, grid_list())
DEV void grid_closest_in(const DevScene &S, GridList l, d3 o, d3 d, double d2, Best &best, WalkCount &wc);
DEV void grid_closest(const DevScene &S, const vg::Bin *bins, uint32_t bin, const vg::Entry *entries, d3 o, d3 d, double d2, Best &best, WalkCount &wc)
{
    grid_closest_in(S, grid_list(bins, bin, entries), o, d, d2, best, wc);
}
DEV void grid_closest_in(const DevScene &S, GridList l, d3 o, d3 d, double d2, Best &best, WalkCount &wc)
{
    const double minT = .01, maxT = 10000000000.0;
    for (uint32_t k = 0; k < l.count; ++k) {
        const grid_v2u e = l.cur;
        if (best.t < maxT && __uint_as_float(e.y) > (float)best.t * (1.0f + 0x1p-16f) + 1e-6f) break;
        grid_next(l, k);
        WC_ADD(wc, lane_prims);
        WC_WAVE(wc, wave_prims);
        double t;
        int key, slot, prim;
        if (e.x & GRID_SPHERE) {
            const int s = (int)(e.x & ~GRID_SPHERE);
            t = sph_test(&S.bsph[s], o, d, d2, minT, maxT);
            key = S.bsph_key[s];
            slot = s | SLOT_SPHERE;
            prim = (int)(S.bsph_prim[s] & ~RT_SPH_TRANSPARENT);
        } else {
            const TriOps tr = load_leaf_tri(S, (int)e.x);
            t = tri_test_fast(tr, o, d, minT, maxT, best.t);
            key = tr.key;
            slot = (int)e.x;
            prim = tr.prim;
        }
        if (t < best.t || (t == best.t && t < maxT && key < best.key)) { best.t = t; best.key = key; best.slot = slot; best.prim = prim; }
    }
}
