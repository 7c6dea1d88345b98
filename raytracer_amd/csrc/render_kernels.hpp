// The kernels of the wavefront pipeline (one generation of the ray tree per launch):
//   k_closest<PRIMARY>  ray generation (Camera.h:112-230) fused with the closest-hit query, the hit
//                       record (Solid.h:110-125) and secondary-ray spawning (Camera.h:232-314)
//   k_closest<queued>   the same for generations 1..5, rays read from the queue
//   k_closest<.., GRID, FUSE>  the primary pass of a frame whose eye has a view grid (no walk compiled in); FUSE: it also forms
//                       the lights' terms at its hits (shade_primary_hit), so generation 0 has no hit records and no shadow pass
//   k_compact_direct    lists the groups of records a primary pass with direct placement filled
//   k_shadow_grid       the shadow pass of a scene whose lights have view grids and whose queries are order-free
//   k_pair, k_local, k_fused   a shadow pass and the next closest-hit pass in one launch; a whole frame in one launch of
//                       self-contained blocks; a lane per sample for scenes without secondary rays
//   k_shadow, k_shadow_hybrid   one item per (hit, light): shadow query (Octree.h:1198-1264), Phong terms
//                       (Camera.h:365-488), FP64 accumulation into the frame
//   k_lightcam          deferred light-camera density queries (Camera.h:406-451)
//   k_photons           a light camera's photon pass (LightCam.h:207-556)
//   k_collect_counts, k_finalize (/spp, clamp, *255, (int); Camera.h:217-227, Driver.cpp:722,758-760)
//   k_debug_intersect   one primitive test per thread (rt_debug_intersect)
#pragma once
#include "hierarchy_walk.hpp"
#include "grid_query.hpp"
#include "lightcam_query.hpp"
#include "octree_walk.hpp"
#include "queues.hpp"
#include "shading.hpp"

// The camera's 16 doubles go to LDS once per block and are read back per chunk.  As kernel
// arguments they would occupy 32 scalar registers for the whole kernel; the scalar file is
// already over-subscribed by the scene's pointers, and every value spilled from it costs a
// vector instruction each time it is brought back (a seventh of the primary kernel's vector work).
DEV void camera_to_lds(const FrameParams &P)
{
    double *cw = reinterpret_cast<double *>(&s_lds[P.S.lds_cam_word]);
    for (int k = 0; k < 3; ++k) { cw[k] = P.cam.u[k]; cw[3 + k] = P.cam.v[k]; cw[6 + k] = P.cam.w[k]; cw[9 + k] = P.cam.e[k]; }
    cw[12] = P.cam.d; cw[13] = P.cam.l; cw[14] = P.cam.t; cw[15] = P.cam.pix_width;
}

// Ray generation for position `lane` of 64-sample chunk `chunk` of a batch (Camera.h:112-230): stratified
// sub-samples with jitter, depth of field.  The camera's 16 doubles are read from LDS (lds_cam_word), where
// thread 0 of the block has put them.
// Primary ordering: pixels in 8x8 blocks (row-major inside a block), the N*N samples of a
// pixel in consecutive positions; a wave takes 64 consecutive positions.
// Positions are counted from the batch's first block, which keeps them in 32 bits, and every
// division is by a number fixed for the launch (a multiplication, see FastDiv).
// (in two halves: where the sample lies -- pixel, tile, sub-sample -- and, from that, its ray; a chunk whose tiles hold nothing
// needs only the first)
struct PrimarySample {
    bool active, first;
    uint32_t pixel, tile;
    int i, j, si, sj;
};
DEV PrimarySample primary_sample(const FrameParams &P, const QueueSet &Q, long long chunk, int lane)
{
    const int N = P.N;
    const uint32_t rel = (uint32_t)chunk * 64u + (uint32_t)lane;
    const uint32_t spp = (uint32_t)(N * N);
    const uint32_t pseq = fast_div(rel, P.div_spp);              // pixel, counted from the batch's first
    const int smp = (int)(rel - pseq * spp);
    const int within = (int)(pseq & 63u);
    const uint32_t col = (uint32_t)Q.block_col0 + (pseq >> 6);   // block column, not yet wrapped into the row
    const uint32_t wraps = fast_div(col, P.div_blocks_x);
    const int bx = (int)(col - wraps * (uint32_t)P.blocks_x);
    const long long by = Q.block_row0 + wraps;
    const int j = bx * 8 + (within & 7);
    const long long lrow_ll = by * 8 + (within >> 3);
    const int lrow = lrow_ll < P.local_rows ? (int)lrow_ll : P.local_rows;     // beyond the part: inactive below
    const int si = (int)fast_div((uint32_t)smp, P.div_n), sj = smp - si * N;
    PrimarySample ps;
    ps.active = j < P.cam.resx && lrow < P.local_rows;
    const int i = global_row(P, lrow);
    ps.pixel = (uint32_t)lrow * (uint32_t)P.cam.resx + (uint32_t)j;
    ps.tile = ((uint32_t)i >> 3) * (uint32_t)P.cg_tiles_x + ((uint32_t)j >> 3);      // the sample's 8 x 8-pixel tile of the whole frame
    ps.first = (si == 0 && sj == 0);
    ps.i = i; ps.j = j; ps.si = si; ps.sj = sj;
    return ps;
}
DEV void primary_ray_of(const FrameParams &P, const PrimarySample &ps, d3 *o_out, d3 *d_out)
{
    const int N = P.N;
    const int i = ps.i, j = ps.j, si = ps.si, sj = ps.sj;
    struct { double d, l, t, pix_width; int resx; } c;
    d3 u, v, w, e;
    {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const __attribute__((address_space(3))) v2d *cw =
            (const __attribute__((address_space(3))) v2d *)((__attribute__((address_space(3))) const int *)s_lds + P.S.lds_cam_word);
        const v2d c0 = cw[0], c1 = cw[1], c2 = cw[2], c3 = cw[3], c4 = cw[4], c5 = cw[5], c6 = cw[6], c7 = cw[7];
        u = mk3(c0.x, c0.y, c1.x); v = mk3(c1.y, c2.x, c2.y); w = mk3(c3.x, c3.y, c4.x); e = mk3(c4.y, c5.x, c5.y);
        c.d = c6.x; c.l = c6.y; c.t = c7.x; c.pix_width = c7.y;
        c.resx = P.cam.resx;
    }
    const uint32_t gpix = (uint32_t)i * (uint32_t)c.resx + (uint32_t)j;
    const uint32_t sample = (uint32_t)(si * N + sj);
    double p, q, r;
    if (N > 1) {
        const double spw = P.spw;
        double randX = normed_rand(gpix, sample, 0, P.seed);
        double randY = normed_rand(gpix, sample, 1, P.seed);
        randX -= .5;
        randY -= .5;
        double edgeX = (j - .5) * c.pix_width;
        double edgeY = (i - .5) * c.pix_width;
        double j_m = edgeX + sj * spw + .5 * spw;
        double i_m = edgeY + si * spw + .5 * spw;
        randX *= spw;
        randY *= spw;
        j_m += randX;
        i_m += randY;
        p = c.l + j_m;
        q = c.t - i_m;
    } else {
        p = c.l + j * c.pix_width;
        q = c.t - i * c.pix_width;
    }
    r = -c.d;
    d3 d = unit3(add3(add3(scale3(p, u), scale3(q, v)), scale3(r, w)));
    d3 o = e;
    if (P.aperture > 0) {
        d3 uOff = scale3(P.aperture, scale3(normed_rand(gpix, sample, 2, P.seed) - .5, u));
        d3 vOff = scale3(P.aperture, scale3(normed_rand(gpix, sample, 3, P.seed) - .5, v));
        o = add3(o, add3(uOff, vOff));
        p = c.l + j * c.pix_width;
        q = c.t - i * c.pix_width;
        d3 focus = add3(add3(add3(scale3(p, u), scale3(q, v)), scale3(r, w)), e);
        d = unit3(sub3(focus, o));
    }
    *o_out = o;
    *d_out = d;
}
DEV void primary_ray(const FrameParams &P, const QueueSet &Q, long long chunk, int lane, bool *active_out, d3 *o_out, d3 *d_out,
                     uint32_t *pixel_out, bool *first_out, uint32_t *tile_out)
{
    const PrimarySample ps = primary_sample(P, Q, chunk, lane);
    *active_out = ps.active; *pixel_out = ps.pixel; *first_out = ps.first; *tile_out = ps.tile;
    primary_ray_of(P, ps, o_out, d_out);
}

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif
// The lights' terms at a primary hit, formed where the hit was found (the grid-only primary kernel of a scene whose shadow
// queries are order-free and whose lights have grids, FUSE): what shadow_chunk does for a hit record, without the record --
// the hit is neither written nor read back.  Returns the sum of the terms of the lights that reach the hit (`*lit_any`:
// there is one): the caller adds it to the pixel in the one run-reduced sum of the chunk.
DEV d3 shade_primary_hit(const FrameParams &P, bool active, d3 Pt, d3 vv, d3 n2, d3 n3, bool flat, int prim, int mat, WalkCount &wc, bool *lit_any)
{
    const int nl = P.S.n_lights;
    const uint32_t per_light = 6u * (uint32_t)(P.S.lg_res * P.S.lg_res);
    d3 sum = mk3(0, 0, 0);
    bool any = false;
    for (int li = 0; li < nl; ++li) {
        const d3 lpos = ld3(load_light(P.S, li, true).pos);
        if (active) {
            const d3 toL = sub3(lpos, Pt);
            const d3 rd = unit3(toL);
            const double maxT = norm3(toL) - .01;
            const double d2 = dot3(rd, rd);
            bool occluded = false;
            if (root_gate(P.S, Pt, rd, d2, 10000000.0)) {
                WC_ADD(wc, rays);
                const uint32_t bin = (uint32_t)li * per_light + light_bin(Pt, lpos, P.S.lg_res);
                occluded = grid_any(P.S, P.S.lg_bins, bin, P.S.lg_entries, Pt, rd, d2, prim, maxT, wc);
            }
            if (!occluded) {
                const rt_light lgt = load_light(P.S, li, true);
                d3 normal = n2, nn = n3;
                if (flat) {
                    const void *src = (const void *)&P.S.prim_n23[(size_t)prim * 6];
                    const rec_v2d a = tab_load(src, 0), b = tab_load(src, 1), c = tab_load(src, 2);
                    normal = mk3(a.x, a.y, b.x); nn = mk3(b.y, c.x, c.y);
                }
                const d3 local = phong_terms(load_material(P.S, mat, true), normal, nn, vv, &lgt, rd, .999999);
                sum = any ? add3(sum, local) : local;
                any = true;
            }
        }
    }
    *lit_any = any;
    return sum;
}

// One 64-ray chunk of a closest-hit pass: the rays (generated, or read from the queue), the closest-hit query, the hit
// record and the pushes of the hit and of the secondary rays.
// FUSE (grid-only primary kernel): 1 the lights' terms of a hit are formed here, 2 the same for a scene whose hits are all
// flat triangles' (no normals held through the queries).
template <bool PRIMARY, bool BVH, bool GRID = false, int FUSE = 0>
DEV void closest_chunk(const FrameParams &P, const QueueSet &Q, const ShardMap &smap, long long chunk, int lane, int *stack, bool cull,
                       WalkCount &wc, uint32_t shard, uint32_t listed = ~0u)
{
#define STAMP_SEL (Q.generation)
    STAMP(10);
    bool active;
    d3 o, d;
    double A = 1.0, coef = 1.0;
    uint32_t pixel = 0, meta = 5u | 0x100u, tile = 0;
    bool first_sample = false;
    GridList tile_list{};
    bool traced = true;
    if (PRIMARY && GRID) {
        // Where the samples lie first, and their tiles' lists: a chunk none of whose samples has anything in its tile (sky)
        // hits nothing whatever its rays are, and they are not formed.
        const PrimarySample ps = primary_sample(P, Q, chunk, lane);
        active = ps.active; pixel = ps.pixel; first_sample = ps.first; tile = ps.tile;
        tile_list = grid_list(P.cg_bins, tile, P.cg_entries);
        traced = __ballot(active && tile_list.count != 0u) != 0ull;
        o = d = mk3(0, 0, 0);
        if (traced) primary_ray_of(P, ps, &o, &d);
    } else if (PRIMARY) {
        primary_ray(P, Q, chunk, lane, &active, &o, &d, &pixel, &first_sample, &tile);
    } else {
        int shard;
        const int local = shard_locate(smap, (uint32_t)chunk, &shard);
        active = local >= 0;
        size_t rslot = active ? (size_t)shard * Q.in_shard_cap + (uint32_t)local : 0;
        // (groups listed by k_compact_direct; `listed`: the chunk's entry, fetched by the caller while the chunk before was traced)
        if (Q.ray_list) rslot = (size_t)(listed != ~0u ? listed : Q.ray_list[rslot >> 6]) * 64u + (rslot & 63u);
        const char *rr = rec_at(Q.rays_in, rslot, 5);
        const rec_v2d r0 = rec_load(rr, 0), r1 = rec_load(rr, 1), r2 = rec_load(rr, 2), r3 = rec_load(rr, 3), r4 = rec_load(rr, 4);
        o = mk3(r0.x, r0.y, r1.x);
        d = mk3(r1.y, r2.x, r2.y);
        A = r3.x;
        coef = r3.y;
        pixel = (uint32_t)__double2loint(r4.x);
        meta = (uint32_t)__double2hiint(r4.x);
        if (pixel == RAY_NONE) active = false;          // a record that only fills its chunk up
    }

    double best_t = 0;
    int best = -1, best_prim = -1;
    STAMP(3);
    if (active && traced) {
        if (BVH) {
            const double d2 = dot3(d, d);
            Best bst;
            bst.t = 10000000000.0; bst.key = 0x7fffffff; bst.slot = -1; bst.prim = -1;
            if (root_gate(P.S, o, d, d2, 10000000000.0)) {
                WC_ADD(wc, rays);
                if (PRIMARY && GRID) {
                    grid_closest_in(P.S, tile_list, o, d, d2, bst, wc);
                } else if (GRID || (PRIMARY && P.cg_bins)) {
                    // every primary ray of a pinhole frame starts at the eye: the list of the sample's 8 x 8-pixel tile holds
                    // all it can meet (grid_query.hpp)
                    grid_closest(P.S, P.cg_bins, tile, P.cg_entries, o, d, d2, bst, wc);
                } else if (!GRID) {
                    const RayBox4 rb = make_raybox4(o, d, P.S.bvh_bound);
                    bvh_closest<true>(P.S, o, d, d2, rb, bst, stack, wc);
                    bvh_closest<false>(P.S, o, d, d2, rb, bst, stack, wc);
                }
            }
            best_t = bst.t;
            best = bst.slot;
            best_prim = bst.prim;
        } else {
            trace_closest(P.S, o, d, cull, .01, best_t, best);
        }
    } else if (active) {
        WC_ADD(wc, rays);       // (a sample whose tile is empty: counted as the ray it stands for)
    }
    const rt_tri_rec *tri_arr = BVH ? P.S.btri : P.S.tri;
    const rt_sph_rec *sph_arr = BVH ? P.S.bsph : P.S.sph;
    const uint32_t *sph_prim_arr = BVH ? P.S.bsph_prim : P.S.sph_prim;
    STAMP(4);

    const bool first = (meta & 0x100u) != 0;
    const int L = (int)(meta & 0xffu);
    const double wloc = first ? A : A * coef;
    bool hit = active && best >= 0;
    if (PRIMARY && GRID) {
        if (__ballot(hit) == 0ull) {
            // nothing was hit: the samples add the background and the chunk leaves no records
            if (active && first_sample && P.hit_id) P.hit_id[pixel] = -1;
            accum_add_equal(P.accum, active, pixel, wloc, ld3(P.S.bg));       // (a primary sample's weight is 1: the same amount in every lane)
            if (Q.direct_counts != nullptr) {
                if (lane == 0) Q.direct_counts[(uint32_t)chunk] = 0u;
            }
            return;
        }
    }

    // hit record (Solid.h:110-125)
    d3 n = mk3(0, 0, 0), Pt = mk3(0, 0, 0);
    int prim = -1, mat = 0;
    bool flat = false;
    rt_material m;
    m.Ks = 0; m.T = 0; m.ior = 1;
    if (hit) {
        hit_geometry(P.S, tri_arr, sph_arr, sph_prim_arr, P.flags, o, d, .01, best_t, best, best_prim, &prim, &n, &Pt, &flat);
        mat = P.S.prim_material[prim];
        m = load_material(P.S, mat, true);
    }
    if (PRIMARY && active && first_sample && P.hit_id) P.hit_id[pixel] = prim;
    // misses add the background (placed after the hit's table fetches have been issued, so that their
    // latency runs beside the shuffles and atomics of this sum)
    STAMP(5);
    if (!(PRIMARY && GRID && FUSE != 0)) accum_add_runs(P.accum, active, active && !hit, pixel, wloc, ld3(P.S.bg));      // (FUSE: below, with the lights' terms)
    STAMP(6);

    // Queue pushes.  The three reservations (hit record, refraction ray, reflection ray) are made
    // with two atomics -- both kinds of ray go to the same queue -- issued back to back by one lane,
    // so a wave waits for one round trip to the counters instead of three in a row.
    const bool want_hit = hit && P.S.n_lights > 0;
    constexpr bool fuse = PRIMARY && GRID && FUSE != 0;      // (the lights' terms are formed below: no hit record)
    const bool spawn = hit && L > 0 && !(coef < .01);       // Camera.h:244-307
    const double coef2 = coef * m.Ks;
    const bool want_refr = spawn && m.T > 0;
    const bool want_refl = spawn && (coef2 != 0.0 || (P.flags & RT_FLAG_KEEP_ZERO_WEIGHT));
    const unsigned long long mask_h = __ballot(want_hit && !fuse), mask_a = __ballot(want_refr), mask_b = __ballot(want_refl);
    const uint32_t n_h = (uint32_t)__popcll(mask_h), n_a = (uint32_t)__popcll(mask_a), n_b = (uint32_t)__popcll(mask_b);
    uint32_t base_h = 0, base_r = 0;
    // places taken: what is pushed, or (whole_chunks) that rounded up to whole chunks, the fillers counted apart for the statistics
    // (direct: the places follow from the chunk's number -- no counter is asked, no wave waits for one; whole groups of 64)
    const bool direct = PRIMARY && Q.direct_counts != nullptr;
    const bool whole = Q.whole_chunks != 0 || direct;
    const uint32_t take_h = whole ? (n_h + 63u) & ~63u : n_h, take_r = whole ? (n_a + n_b + 63u) & ~63u : n_a + n_b;
    if (direct) {
        const uint32_t c = (uint32_t)chunk;      // (a primary chunk's number counts from the batch's first)
        shard = 0u;
        base_h = c * 64u;
        base_r = c * 64u * Q.direct_fan;
        const uint32_t n_f = fuse ? (uint32_t)__popcll(__ballot(want_hit)) : 0u;      // (hits whose lights are dealt with below)
        if (lane == 0) Q.direct_counts[c] = n_h | ((n_a + n_b) << 8) | (n_f << 16);
    } else
    if (lane == 0) {
        if (n_h) base_h = atomicAdd(&Q.counts[CNT_HITS(Q.generation) + shard], take_h);
        if (n_a + n_b) base_r = atomicAdd(&Q.counts[CNT_RAYS(Q.generation + 1) + shard], take_r);
        if (take_h != n_h) atomicAdd(&Q.counts[CNT_PAD_HITS(Q.generation) + shard], take_h - n_h);
        if (take_r != n_a + n_b) atomicAdd(&Q.counts[CNT_PAD_RAYS(Q.generation + 1) + shard], take_r - (n_a + n_b));
    }
    // While the counters answer (a trip to memory and back), the records' contents are formed: the unit vector to the
    // ray's origin for the hit record, the directions of the secondary rays.
    d3 vv = mk3(0, 0, 0), n2 = mk3(0, 0, 0), n3 = mk3(0, 0, 0), dir_a = mk3(0, 0, 0), dir_b = mk3(0, 0, 0);
    if (want_hit) {
        vv = unit3(sub3(o, Pt));
        if (!flat) { n2 = unit3(n); n3 = unit3(n2); }
    }
    if (want_refr) dir_a = refract_dir(d, n, m.ior, false);
    if (want_refl) dir_b = unit3(add3(d, scale3(2.0 * -dot3(d, n), n)));      // hr.reflectionDir (Solid.h:121-122), only where a ray follows it
    if (!direct) {
        base_h = (uint32_t)__builtin_amdgcn_readlane((int)base_h, 0);
        base_r = (uint32_t)__builtin_amdgcn_readlane((int)base_r, 0);
    }
    STAMP(7);
    const uint32_t slot_h = base_h + rank_below(mask_h);
    const uint32_t slot_a = base_r + rank_below(mask_a);
    const uint32_t slot_b = base_r + n_a + rank_below(mask_b);
    if ((want_hit && !fuse && slot_h >= Q.hit_shard_cap) || (want_refr && slot_a >= Q.out_shard_cap) ||
        (want_refl && slot_b >= Q.out_shard_cap))
        Q.counts[CNT_OVERFLOW] = 1u;

    if (whole) {
        // the fillers: of a record only the word that marks it is written (at most 63 of either kind, one per lane)
        const uint32_t fill_h = base_h + n_h + (uint32_t)lane, fill_r = base_r + n_a + n_b + (uint32_t)lane;
        if (n_h + (uint32_t)lane < take_h && fill_h < Q.hit_shard_cap) {
            const rec_v2d q[1] = {rec_pack(0.0, __hiloint2double(-1, (int)HIT_NONE))};
            rec_store(rec_at(Q.hits, (size_t)shard * Q.hit_shard_cap + fill_h, Q.hit_stride >> 4), 3, q);
        }
        if (n_a + n_b + (uint32_t)lane < take_r && fill_r < Q.out_shard_cap) {
            const rec_v2d q[1] = {rec_pack_words(RAY_NONE, 0u, 0u, 0u)};
            rec_store(rec_at(Q.rays_out, (size_t)shard * Q.out_shard_cap + fill_r, 5), 4, q);
        }
    }
    if (want_hit && !fuse && slot_h < Q.hit_shard_cap) {
        char *dst = rec_at(Q.hits, (size_t)shard * Q.hit_shard_cap + slot_h, Q.hit_stride >> 4);
        // HitRec: P, v, weight, pixel, prim -- and for a hit whose normal is not its primitive's tabulated one
        // (a sphere, interpolated vertex normals) the material and the normal normalised twice and thrice
        const rec_v2d q[4] = {rec_pack(Pt.x, Pt.y), rec_pack(Pt.z, vv.x), rec_pack(vv.y, vv.z),
                              rec_pack(wloc, __hiloint2double(prim, (int)(pixel | (flat ? HIT_FLAT : 0u))))};
        rec_store(dst, 0, q);
        if (!flat) {
            const rec_v2d r[4] = {rec_pack_words(0u, 0u, (uint32_t)mat, 0u), rec_pack(n2.x, n2.y), rec_pack(n2.z, n3.x), rec_pack(n3.y, n3.z)};
            rec_store(dst, 4, r);
        }
    }
    if (want_refr && slot_a < Q.out_shard_cap) {
        // RayRec: o, d, A, coef, pixel, meta, pad
        const rec_v2d q[5] = {rec_pack(Pt.x, Pt.y), rec_pack(Pt.z, dir_a.x), rec_pack(dir_a.y, dir_a.z), rec_pack(A * m.T, m.T * coef),
                              rec_pack_words(pixel, (uint32_t)(L - 1) | 0x100u, 0u, 0u)};
        rec_store(rec_at(Q.rays_out, (size_t)shard * Q.out_shard_cap + slot_a, 5), 0, q);
    }
    if (want_refl && slot_b < Q.out_shard_cap) {
        const rec_v2d q[5] = {rec_pack(Pt.x, Pt.y), rec_pack(Pt.z, dir_b.x), rec_pack(dir_b.y, dir_b.z), rec_pack(A, coef2),
                              rec_pack_words(pixel, (uint32_t)(L - 1), 0u, 0u)};
        rec_store(rec_at(Q.rays_out, (size_t)shard * Q.out_shard_cap + slot_b, 5), 0, q);
    }
    if (Q.commit && take_r) {
        // (k_local) the rays are written: count them into the 64-record pieces of the list they lie in, so that a piece
        // that is full can be traced before the rest of this generation is done
        __threadfence_block();
        if (lane == 0) {
            uint32_t s0 = base_r, e0 = base_r + take_r;
            if (e0 > Q.out_shard_cap) e0 = Q.out_shard_cap;
            while (s0 < e0) {
                const uint32_t ce = (s0 | 63u) + 1u < e0 ? (s0 | 63u) + 1u : e0;
                atomicAdd(&Q.commit[s0 >> 6], ce - s0);
                s0 = ce;
            }
        }
    }
    STAMP(8);
    if (PRIMARY && GRID) {
        // (a scene whose hits are all flat triangles' takes the normals from the primitives' table: none held)
        // (a sample adds either the background or the sum of its lights' terms: one run-reduced sum for the chunk)
        if (FUSE != 0) {
            bool lit_any = false;
            d3 terms;
            if (FUSE == 2) terms = shade_primary_hit(P, want_hit, Pt, vv, mk3(0, 0, 0), mk3(0, 0, 0), true, prim, mat, wc, &lit_any);
            else terms = shade_primary_hit(P, want_hit, Pt, vv, n2, n3, flat, prim, mat, wc, &lit_any);
            const d3 bg = ld3(P.S.bg);
            accum_add_runs(P.accum, active, active && (!hit || lit_any), pixel, wloc, hit ? terms : bg);
        }
    }

#undef STAMP_SEL
}

// GRID: the primary pass of a frame whose eye has a grid (viewgrid.hpp), as a kernel of its own: no hierarchy walk is
// compiled in, so it needs no traversal stacks in LDS and fewer registers (blocks of RT_GRID_BLOCK threads).
template <bool PRIMARY, bool BVH, bool GRID = false, int FUSE = 0>
__global__ __launch_bounds__(GRID ? RT_GRID_BLOCK : RT_BLOCK, GRID ? RT_GRID_WAVES : RT_MIN_WAVES) void k_closest(FrameParams P, QueueSet Q)
{
#define STAMP_SEL (Q.generation)
    // traversal stack, sized at launch to the hierarchies' depth (entry k of a thread at [k * block size + thread])
    int *stack = &s_lds[BVH ? threadIdx.x : 0];
    const int lane = (int)(threadIdx.x & 63);
    STAMP_RESET();
    ShardMap smap;
    smap.items = smap.first = smap.total = 0;
    if (!PRIMARY) smap = shard_map(&Q.counts[CNT_RAYS(Q.generation)], Q.in_shard_cap, 1u);
    const long long n_chunks = PRIMARY ? (Q.chunk_end - Q.chunk_begin) : (long long)smap.total;
    // a block none of whose waves gets a chunk leaves before it copies anything into LDS (late
    // generations of small frames fill only a few of the launched blocks)
    if ((long long)blockIdx.x * (blockDim.x >> 6) >= n_chunks) return;
    if (PRIMARY && threadIdx.x == 0) camera_to_lds(P);
    // (units of work drawn while the pass runs, below: the block's first unit is its own index)
    volatile int *ring = &s_lds[P.S.lds_ticket_word + 4];
    if (PRIMARY && threadIdx.x == 0) {
        const uint32_t nu = Q.unit_chunks ? (uint32_t)((Q.chunk_end - Q.chunk_begin + Q.unit_chunks - 1) / Q.unit_chunks) : 0u;
        uint32_t id = blockIdx.x;
        if ((Q.unit_mult & 0x7fffffffu) && id < nu) id = (uint32_t)(((unsigned long long)id * (Q.unit_mult & 0x7fffffffu)) % nu);
        ring[0] = (int)((1u << 24) | id);
        // The other three places start with a tag no unit waits for.  LDS keeps what the last block on this compute unit
        // left there -- the ring of the previous frame's pass, whose k-th word carries the very tag this block's k-th unit
        // waits for -- and a wave that gets to a unit before its word is written (waves race ahead where chunks are
        // background) would take that block's unit for its own: some chunks traced twice, others never.
#ifndef RT_TEST_STALE_RING      // (a test build leaves them as they are, to show that the ring_poison test sees the fault)
        ring[1] = ring[2] = ring[3] = 0;
#endif
    }
    STAMP(1);
    fill_node_cache(P.S, BVH);      // (GRID: no nodes are cached, S.tri_cached = S.sph_cached = 0; the leaf records are)      // ends with the block's barrier
    STAMP(2);
    const bool cull = P.cull_closest != 0;
    WalkCount wc{};

    // A large primary pass (Q.unit_chunks != 0) shares its chunks out while it runs.  Blocks that keep to a fixed share of
    // the image end as far apart as their shares differ in work (the lower third of the 8K teapot frame: block lifetimes
    // of 5.3 to 7.0 ms, and the chip idles towards the end of every batch).  Here a block works on units of `unit_chunks`
    // consecutive chunks: its first unit is its own index, further ones come from a counter in memory, and inside a unit
    // its waves draw chunks from the ticket in LDS as always.  The wave that draws the middle chunk of a unit fetches the
    // next unit and leaves it in a ring of four LDS words (tag in the top byte, so a stale word is never taken for the
    // awaited one); the others wait for that word only if they get there first.  Records go to the shard given by the
    // unit, not by the block, so that a shard's load does not depend on which blocks were quick.
    const uint32_t G = PRIMARY ? Q.unit_chunks : 0u, gshift = G ? (uint32_t)(31 - __clz((int)G)) : 0u;
    const uint32_t n_units = G ? (uint32_t)((n_chunks + G - 1) >> gshift) : 0u;
    long long ahead_chunk = -1;
    uint32_t ahead_listed = ~0u;
    for (;;) {
        long long chunk;
        uint32_t shard = blockIdx.x & (N_SHARDS - 1), listed = ~0u;
        if (PRIMARY && G) {
            int t = 0;
            if (lane == 0) t = atomicAdd(&s_lds[P.S.lds_ticket_word], 1);
            t = __builtin_amdgcn_readfirstlane(t);
            const uint32_t j = (uint32_t)t >> gshift, r = (uint32_t)t & (G - 1);
            int word = 0;
            if (lane == 0) {
                if (r == (G >> 1)) {
                    uint32_t id = gridDim.x + atomicAdd(&Q.counts[CNT_UNITS], 1u);
                    if (id >= n_units) id = 0xffffffu;
                    else if (Q.unit_mult & 0x7fffffffu) id = (uint32_t)(((unsigned long long)id * (Q.unit_mult & 0x7fffffffu)) % n_units);
                    // (diagnostic, bit 31 of unit_mult: the word is handed over late, so that the block's other waves get
                    // to the next unit first and have to wait for it -- with k_poison_ring's words in LDS meanwhile)
                    if (Q.unit_mult >> 31)
                        for (int k = 0; k < 12; ++k) __builtin_amdgcn_s_sleep(127);
                    ring[(j + 1) & 3] = (int)((((j + 2) & 0xffu) << 24) | id);
                }
                // The word of unit j carries tag j + 1.  The wait is bounded: a tag ahead of the awaited one (the place was
                // written again before this wave read it) or a word that never arrives ends the block's share with the
                // frame marked incomplete (the host reports it and renders the frame again) instead of a spin without end.
                const uint32_t want = (j + 1) & 0xffu;
                uint32_t spins = 0;
                for (;;) {
                    word = ring[j & 3];
                    const uint32_t ahead = (((uint32_t)word >> 24) - want) & 0xffu;
                    if (ahead == 0) break;
                    if ((ahead < 128u && (ahead & 3u) == 0u) || ++spins > (1u << 22)) {
                        Q.counts[CNT_OVERFLOW] = 1u;
                        word = (int)0xffffffu;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const uint32_t id = (uint32_t)__builtin_amdgcn_readfirstlane(word) & 0xffffffu;
            if (id >= n_units) break;                  // the counter has run past the batch: nothing left for this block
            chunk = ((long long)id << gshift) + r;
            if (chunk >= n_chunks) continue;
            shard = id & (N_SHARDS - 1);
        } else if (!PRIMARY && Q.ray_list) {
            // (a list of groups: the rays lie in shard 0, chunk k of the pass is entry k; the wave's next ticket and its entry
            // are fetched while this chunk is traced, so the entry is not one more fetch in every chunk's chain)
            if (ahead_chunk < 0) {
                ahead_chunk = next_chunk(P.S);
                ahead_listed = ahead_chunk < n_chunks ? Q.ray_list[ahead_chunk] : ~0u;
            }
            chunk = ahead_chunk;
            listed = ahead_listed;
            if (chunk >= n_chunks) break;
            ahead_chunk = next_chunk(P.S);
            ahead_listed = ahead_chunk < n_chunks ? Q.ray_list[ahead_chunk] : ~0u;
        } else {
            chunk = (GRID && (RT_GRID_BLOCK & (RT_GRID_BLOCK - 1))) ? next_chunk_strided(P.S) : next_chunk(P.S);
            if (chunk >= n_chunks) break;
        }
        closest_chunk<PRIMARY, BVH, GRID, FUSE>(P, Q, smap, chunk, lane, stack, cull, wc, shard, listed);
    }
    STAMP(9);
#ifdef RT_COUNT
    walk_count_flush(wc, PRIMARY ? 0 : 8);
#endif
#undef STAMP_SEL
}


// After a primary pass with direct placement: the groups of 64 records that hold anything, listed in the order of the chunks
// (inside a block of 256 chunks; the blocks' runs in the order they arrive), and the counters the next passes and the
// statistics read, all in shard 0: records listed (groups * 64) and how many of them are fillers.
__global__ __launch_bounds__(256) void k_compact_direct(const uint32_t *direct_counts, uint32_t n_chunks, uint32_t fan, int generation, uint32_t *counts,
                                                        uint32_t *hit_list, uint32_t *ray_list)
{
    __shared__ uint32_t scan_h[256], scan_r[256], sum_h, sum_r, sum_f, base_h, base_r;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    const uint32_t word = c < n_chunks ? direct_counts[c] : 0u;
    const uint32_t nh = word & 0xffu, nr = (word >> 8) & 0xffu, nf = word >> 16;       // (nf: hits shaded where they were found)
    const uint32_t gh = nh ? 1u : 0u, gr = (nr + 63u) >> 6;
    scan_h[threadIdx.x] = gh; scan_r[threadIdx.x] = gr;
    if (threadIdx.x == 0) sum_h = sum_r = sum_f = 0u;
    __syncthreads();
    if (nh) atomicAdd(&sum_h, nh);
    if (nr) atomicAdd(&sum_r, nr);
    if (nf) atomicAdd(&sum_f, nf);
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t a = threadIdx.x >= off ? scan_h[threadIdx.x - off] : 0u, b = threadIdx.x >= off ? scan_r[threadIdx.x - off] : 0u;
        __syncthreads();
        scan_h[threadIdx.x] += a; scan_r[threadIdx.x] += b;
        __syncthreads();
    }
    if (threadIdx.x == 255) {
        const uint32_t th = scan_h[255], tr = scan_r[255];
        base_h = th ? atomicAdd(&counts[CNT_HITS(generation)], th * 64u) >> 6 : 0u;
        base_r = tr ? atomicAdd(&counts[CNT_RAYS(generation + 1)], tr * 64u) >> 6 : 0u;
        if (th * 64u != sum_h) atomicAdd(&counts[CNT_PAD_HITS(generation)], th * 64u - sum_h);
        if (tr * 64u != sum_r) atomicAdd(&counts[CNT_PAD_RAYS(generation + 1)], tr * 64u - sum_r);
        if (sum_f) atomicAdd(&counts[CNT_FUSED_HITS], sum_f);
    }
    __syncthreads();
    if (gh) hit_list[base_h + scan_h[threadIdx.x] - 1u] = c;
    for (uint32_t j = 0; j < gr; ++j) ray_list[base_r + scan_r[threadIdx.x] - gr + j] = c * fan + j;
}

// Diagnostic (option "ring_poison", run before a primary pass that hands its chunks out in units): leaves in every compute
// unit's LDS, in the four words where that pass keeps its ring of unit numbers, the tags its blocks' units 0..3 wait for --
// with the number of a unit that is not theirs.  LDS keeps what the last block on a compute unit wrote, so this is what the
// ring of an earlier pass can look like to the next one; a pass that took such a word for the awaited one would trace unit
// 0 again and again and never its own.  One block per compute unit (same block size and LDS size as the pass); a block
// stays until all have started, so that no compute unit takes two of them and another none (bounded wait).
__global__ __launch_bounds__(RT_BLOCK) void k_poison_ring(int ring_word, uint32_t *arrived, uint32_t blocks)
{
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; ++k) ((volatile int *)s_lds)[ring_word + k] = (int)(((uint32_t)(k + 1) << 24) | 0u);
        atomicAdd(arrived, 1u);
        for (uint32_t spins = 0; spins < (1u << 18); ++spins) {
            if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= blocks) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}

#ifndef RT_MIN_WAVES_SHADOW
#define RT_MIN_WAVES_SHADOW 4
#endif
// The shadow/shading pass.  MODE selects how the occlusion query is answered: SHADOW_EXACT walks the
// octree in the reference's order, SHADOW_BVH takes the hierarchies (scenes without transparent spheres),
// SHADOW_HYBRID decides per ray.
#define SHADOW_EXACT 0
#define SHADOW_BVH 1
#define SHADOW_HYBRID 2
#define MAX_HYBRID_SPHERES 64
// One 64-item chunk of a shadow/shading pass; the items are (hit, light) pairs, 64 consecutive hits with one light (shard_map_by_light).
// SKIP_DARK: lights that add nothing at a hit are not queried (below).  Taken by the kernels of small frames, where a wave
// lasts as long as its slowest lane's walk and whole waves lie on the side of an object a light does not reach
// (512 x 512 frames 1.5 to 5 % faster); over a large frame 8 % of the queries go, the other lanes of their waves walk on
// all the same, and the test costs what that saves (8K frame: shadow passes 30.9 -> 31.4 ms), so k_shadow asks every light.
// The two normals a hit is shaded with (normalised twice and three times) and its material: from the record, or, for a
// triangle shaded with its own normal, from the primitive's table entries.
DEV void hit_normals(const DevScene &S, const char *hr, bool flat, int prim, rec_v2d &n0, rec_v2d &n1, rec_v2d &n2)
{
    if (flat) {
        const void *src = (const void *)&S.prim_n23[(size_t)prim * 6];
        n0 = tab_load(src, 0); n1 = tab_load(src, 1); n2 = tab_load(src, 2);
    } else {
        n0 = rec_load(hr, 5); n1 = rec_load(hr, 6); n2 = rec_load(hr, 7);
    }
}
DEV int hit_material(const DevScene &S, const char *hr, bool flat, int prim)
{
    if (flat) return S.prim_material[prim];
    return __double2loint(rec_load(hr, 4).y);       // (HitRec::mat: the third word of piece 4)
}

// What the chunks of one group of 64 hits -- one chunk per light, the same hits in the same lanes -- add to their pixels,
// summed per lane over the lights, so that the group needs one run-reduced sum instead of one per light.
struct GroupTerms {
    d3 sum;
    bool any, active;
    uint32_t pixel;
    double weight;
};
template <int MODE, bool SKIP_DARK, bool GRID = false>
DEV void shadow_chunk(const FrameParams &P, const QueueSet &Q, const ShardMap &smap, uint32_t nl, long long chunk, int *stack, bool cull,
                      WalkCount &wc, uint32_t listed = ~0u, GroupTerms *group = nullptr)
{
#define STAMP_SEL (8 + Q.generation)
    STAMP(10);
    int shard, li;
    const int item = shard_locate_by_light(smap, (uint32_t)chunk, P.div_lights, &shard, &li);      // one light per chunk
    const uint32_t h = item >= 0 ? (uint32_t)item : 0u;
    size_t hslot = (size_t)shard * Q.hit_shard_cap + h;
    // (groups listed by k_compact_direct; `listed`: the group's entry, fetched by the caller while the group before was worked on)
    if (Q.hit_list) hslot = (size_t)(listed != ~0u ? listed : Q.hit_list[hslot >> 6]) * 64u + (hslot & 63u);
    const char *hr = rec_at(Q.hits, hslot, Q.hit_stride >> 4);
    // head of the record: P, weight, pixel, prim, material -- four 16-byte pieces fetched together
    const rec_v2d h0 = rec_load(hr, 0), h1 = rec_load(hr, 1), h3 = rec_load(hr, 3);
    const bool active = item >= 0 && ((uint32_t)__double2loint(h3.y) & ~HIT_FLAT) != HIT_NONE;      // (not a record that only fills its chunk up)
    const d3 Pt = mk3(h0.x, h0.y, h1.x);
    const double weight = h3.x;
    const int hit_prim = __double2hiint(h3.y);
    const bool hit_flat = ((uint32_t)__double2loint(h3.y) & HIT_FLAT) != 0;
    const d3 lpos = ld3(load_light(P.S, li, true).pos);
    const uint32_t pixel = (uint32_t)__double2loint(h3.y) & ~HIT_FLAT;
    bool lit = false;
    int defer_cam = -1;
    d3 local = mk3(0, 0, 0);
#ifdef RT_STAMP
    if (__ballot(active) == 0xdeadbeefull) return;      // (the record has arrived)
#endif
    STAMP(3);
    if (active) {
        const d3 toL = sub3(lpos, Pt);
        const d3 rd = unit3(toL);
        const double maxT = norm3(toL) - .01;
        bool only = false;
        int closest = -1;
        bool occluded;
        bool order_free = MODE == SHADOW_BVH;
        if (MODE == SHADOW_HYBRID) {
            // checkOccluded3 depends on the visiting order only through transparent spheres the ray hits
            // (Octree.h:1222-1243); a ray that hits none of them is blocked iff it hits anything
            const double d2 = dot3(rd, rd);
            order_free = true;
            for (int k = 0; k < P.S.n_tsph; ++k)
                if (P.S.tsph_prim[k] != hit_prim && sph_test(&P.S.tsph[k], Pt, rd, d2, .01, maxT) < maxT) order_free = false;
        }
        if (order_free) {
            // no transparent sphere in the way: "occluded" is "any hit", and never "only transparent"
            const double d2 = dot3(rd, rd);
            // A light that adds exactly nothing at this hit needs no answer (the reference asks, and multiplies the
            // answer's consequence by zero): such an item counts as shadowed.
            bool dark = false;
            if (SKIP_DARK && P.S.shade_bounded && fabs(weight) < 1e300) {
                const rec_v2d g2 = rec_load(hr, 2);
                rec_v2d n0, n1, n2;
                hit_normals(P.S, hr, hit_flat, hit_prim, n0, n1, n2);
                const rt_material m = load_material(P.S, hit_material(P.S, hr, hit_flat, hit_prim), true);
                dark = light_adds_nothing(m.Kd, m.Ks, m.Shine, mk3(n0.x, n0.y, n1.x), mk3(n1.y, n2.x, n2.y), mk3(h1.y, g2.x, g2.y), rd);
            }
            occluded = dark;
            if (!dark && root_gate(P.S, Pt, rd, d2, 10000000.0)) {
                WC_ADD(wc, rays);
                if (GRID || P.S.lg_bins) {
                    // every shadow ray towards this light ends at it: the list of the direction's bin on the light's cube map
                    // holds all that can lie in between (grid_query.hpp)
                    const uint32_t bin = (uint32_t)li * 6u * (uint32_t)(P.S.lg_res * P.S.lg_res) + light_bin(Pt, lpos, P.S.lg_res);
                    occluded = grid_any(P.S, P.S.lg_bins, bin, P.S.lg_entries, Pt, rd, d2, hit_prim, maxT, wc);
                } else if (!GRID) {
                    const RayBox4 rb = make_raybox4(Pt, rd, P.S.bvh_bound);
                    occluded = bvh_any<true>(P.S, Pt, rd, d2, rb, hit_prim, maxT, stack, wc) ||
                               bvh_any<false>(P.S, Pt, rd, d2, rb, hit_prim, maxT, stack, wc);
                }
            }
        } else if (MODE != SHADOW_BVH) {
            occluded = trace_shadow(P.S, Pt, rd, hit_prim, maxT, cull, &only, &closest);
        } else {
            occluded = false;
        }
        if (occluded && only) {
            // shadowed by transparent spheres only: the light camera of (that sphere, this light)
            // decides how much light arrives (Camera.h:406-451); evaluated in k_lightcam
            if (P.lc_mask) P.lc_mask[pixel] = 1;
            if (P.S.lc_table && closest >= 0) {
                const int rank = P.S.prim_lc_rank[closest];
                if (rank >= 0) {
                    defer_cam = rank * P.S.n_lights + li;
                    P.S.lc_needed[defer_cam] = 1u;       // cameras are set up when a frame first asks for them (Camera.h:417-440)
                }
            }
        }
        lit = !occluded;
        STAMP(4);
        if (lit) {
            const rt_light lgt = load_light(P.S, li, true);     // read again here rather than kept in registers through the walk
            // the view vector from the record; the two normals from the record too, or from the primitive's table
            // entry when they are the flat triangle's own -- four 16-byte pieces fetched together
            const rec_v2d g2 = rec_load(hr, 2);
            rec_v2d n0, n1, n2;
            hit_normals(P.S, hr, hit_flat, hit_prim, n0, n1, n2);
            const int hit_mat = hit_material(P.S, hr, hit_flat, hit_prim);
            local = phong_terms(load_material(P.S, hit_mat, true), mk3(n0.x, n0.y, n1.x), mk3(n1.y, n2.x, n2.y),
                                mk3(h1.y, g2.x, g2.y), &lgt, rd, .999999);
        }
    }
    if (MODE != SHADOW_BVH && P.S.lc_table) {
        bool ok;
        const bool want = defer_cam >= 0;
        const uint32_t slot = wave_reserve(want, Q.defer_counts ? Q.defer_counts : &Q.counts[CNT_DEFER(Q.generation)], Q.defer_shard_cap,
                                           &Q.counts[CNT_OVERFLOW + 1], &ok);
        if (ok) {
            // the deferred queue keeps its records whole (k_lightcam reads them field by field)
            HitRec *dr = &Q.deferred[slot];
            const rec_v2d g2 = rec_load(hr, 2);
            const rec_v2d a[4] = {h0, h1, g2, rec_pack(weight, __hiloint2double(hit_prim, (int)(pixel | (hit_flat ? HIT_FLAT : 0u))))};
            tab_store(dr, a);
            rec_v2d n0 = rec_pack(0, 0), n1 = n0, n2 = n0;
            if (!hit_flat) hit_normals(P.S, hr, false, hit_prim, n0, n1, n2);
            const rec_v2d b[4] = {rec_pack_words((uint32_t)li, (uint32_t)defer_cam, (uint32_t)hit_material(P.S, hr, hit_flat, hit_prim), 0u), n0, n1, n2};
            tab_store(&dr->light, b);
        } else if (want) {
            Q.counts[CNT_OVERFLOW] = 1u;      // the queue holds every (hit, light) pair of its generation: not expected; reported
        }
    }
#ifdef RT_STAMP
    if (__ballot(lit && local.x > 1e300) == 0xdeadbeefull) return;      // (the terms are formed)
#endif
    STAMP(5);
    if (group) {
        if (lit) { group->sum = group->any ? add3(group->sum, local) : local; group->any = true; }
        group->active = active; group->pixel = pixel; group->weight = weight;
    } else {
        accum_add_runs(P.accum, active, lit, pixel, weight, local);
    }
    STAMP(8);

#undef STAMP_SEL
}

template <int MODE, bool GRID = false>
DEV void shadow_pass(const FrameParams &P, const QueueSet &Q)
{
    constexpr bool BVH = MODE != SHADOW_EXACT;      // the hierarchies (and their LDS stacks and node cache) are in use
#define STAMP_SEL (8 + Q.generation)
#ifdef RT_COUNT
    const long long t_begin = wall_clock64();
#endif
    // traversal stack, sized at launch to the hierarchies' depth (entry k of a thread at [k * block size + thread])
    int *stack = &s_lds[BVH ? threadIdx.x : 0];
    const uint32_t nl = (uint32_t)P.S.n_lights;
    STAMP_RESET();
    // work items are (hit, light) pairs: per 64 hits of a shard one chunk for each light
    const ShardMap smap = shard_map_by_light(&Q.counts[CNT_HITS(Q.generation)], Q.hit_shard_cap, nl);
    // no chunk for this block (GRID: a block's t-th share is the group of 64 hits number blockIdx + t * gridDim, all its lights)
    if (GRID ? (nl == 0 || (long long)blockIdx.x >= (long long)(smap.total / nl)) : (long long)blockIdx.x * (blockDim.x >> 6) >= (long long)smap.total) return;
    STAMP(1);
    fill_node_cache(P.S, BVH);      // (GRID: no nodes are cached, S.tri_cached = S.sph_cached = 0; the leaf records are)
    STAMP(2);
    const bool cull = P.cull_shadow != 0;
    WalkCount wc{};
    if (GRID) {
        // A wave takes the chunks of all lights of one group of 64 hits, one after the other: the group's records come from
        // memory once (the later lights' fetches meet them in the compute unit's cache), where chunks dealt out one by one
        // sent the lights of a group to blocks on different dies, each of which fetched the records for itself.
        const long long groups = nl ? (long long)(smap.total / nl) : 0;
        // (with a list of groups -- the hits lie in shard 0 then, group k of the pass is entry k -- the wave's next ticket and
        // its list entry are fetched while this group is worked on: the entry would be one more fetch in every chunk's chain)
        long long group = next_chunk_strided(P.S);
        uint32_t listed = Q.hit_list && group < groups ? Q.hit_list[group] : ~0u;
        while (group < groups) {
            const long long after = next_chunk_strided(P.S);
            const uint32_t listed_after = Q.hit_list && after < groups ? Q.hit_list[after] : ~0u;
            GroupTerms terms;
            terms.sum = mk3(0, 0, 0); terms.any = terms.active = false; terms.pixel = 0u; terms.weight = 0.0;
            for (uint32_t l = 0; l < nl; ++l) shadow_chunk<MODE, false, GRID>(P, Q, smap, nl, group * nl + l, stack, cull, wc, listed, &terms);
            accum_add_runs(P.accum, terms.active, terms.any, terms.pixel, terms.weight, terms.sum);
            group = after;
            listed = listed_after;
        }
    } else {
        for (long long chunk = next_chunk(P.S); chunk < (long long)smap.total; chunk = next_chunk(P.S))
            shadow_chunk<MODE, false, GRID>(P, Q, smap, nl, chunk, stack, cull, wc);
    }
    STAMP(9);
#ifdef RT_COUNT
    walk_count_flush(wc, 16);
    // lifetime of the wave in 100 MHz ticks: summed over a launch and set against its duration it tells how full
    // the resident wave slots were (tools/exp_tail.py)
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_walk_counts[40], (unsigned long long)(wall_clock64() - t_begin));
#endif
#undef STAMP_SEL
}

template <bool BVH>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_SHADOW) void k_shadow(FrameParams P, QueueSet Q)
{
    shadow_pass<BVH ? SHADOW_BVH : SHADOW_EXACT>(P, Q);
}
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_SHADOW) void k_shadow_hybrid(FrameParams P, QueueSet Q)
{
    shadow_pass<SHADOW_HYBRID>(P, Q);
}
// The shadow pass of a scene whose lights have grids and whose queries are order-free, as a kernel of its own (see k_closest, GRID).
__global__ __launch_bounds__(RT_GRID_BLOCK, RT_GRID_WAVES) void k_shadow_grid(FrameParams P, QueueSet Q)
{
    shadow_pass<SHADOW_BVH, true>(P, Q);
}



// The shadow pass of generation g and the closest-hit pass of generation g+1 as ONE launch.  Both consume what the
// closest-hit pass of generation g produced (its hits, its secondary rays) and neither reads what the other writes, so
// their chunks form one list: the shadow chunks first, then the closest-hit chunks, dealt to the blocks like the chunks
// of any pass.  A batch then takes 7 launches instead of 12 (C0, S0+C1, ..., S4+C5, S5), and every launch that is left
// out is a ramp (LDS fill, first chunks) and a tail (the last blocks' last chunks) the chip does not idle through.
// (One list, not two worked through one after the other: in a small frame a wave should get one chunk of either kind,
// not one of each -- teapot-3 at 512 x 512: 0.389 ms with one list, 0.456 with two, 0.435 with a launch per pass.)
template <bool BVHC, int MODE>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_SHADOW) void k_pair(FrameParams P, QueueSet Qs, QueueSet Qc)
{
    constexpr bool WITH_NODES = BVHC || MODE != SHADOW_EXACT;
    int *stack = &s_lds[WITH_NODES ? threadIdx.x : 0];
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t nl = (uint32_t)P.S.n_lights;
    const ShardMap sm_s = shard_map_by_light(&Qs.counts[CNT_HITS(Qs.generation)], Qs.hit_shard_cap, nl);
    const ShardMap sm_c = shard_map(&Qc.counts[CNT_RAYS(Qc.generation)], Qc.in_shard_cap, 1u);
    const long long n_s = (long long)sm_s.total, total = n_s + (long long)sm_c.total;
    if ((long long)blockIdx.x * (blockDim.x >> 6) >= total) return;
    fill_node_cache(P.S, WITH_NODES);
    const bool cull_s = P.cull_shadow != 0, cull_c = P.cull_closest != 0;
    WalkCount wc{};
    for (long long chunk = next_chunk(P.S); chunk < total; chunk = next_chunk(P.S)) {
        if (chunk < n_s) shadow_chunk<MODE, true>(P, Qs, sm_s, nl, chunk, stack, cull_s, wc);
        else closest_chunk<false, BVHC>(P, Qc, sm_c, chunk - n_s, lane, stack, cull_c, wc, blockIdx.x & (N_SHARDS - 1));
    }
#ifdef RT_COUNT
    walk_count_flush(wc, 8);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// A small frame as ONE launch of blocks that each render their own samples through all generations.
//
// Pass by pass over the whole frame, a frame of a few hundred thousand samples is a chain of launches each as long as
// its slowest wave (a wave whose rays graze the geometry walks 2.5 times as long as the median wave), and the other
// four thousand waves wait for it seven times over.  Here a block takes a fixed run of primary chunks and keeps what
// they spawn to itself: its hit and ray queues are its own slices of the queue buffers, their counters live in LDS
// (the same counter layout, shard 0 only, so the pass functions run unchanged), and the passes -- primary, then the
// shadow pass of generation g together with the closest-hit pass of generation g+1, as in k_pair -- are separated by
// the block's barrier only.  A slow wave then holds up the fifteen other waves of its block for one pass, not the chip,
// and blocks drift apart freely.  Rays are still regrouped into full waves, per block.  At the end the block adds its
// counters to the frame's (shard 0), for the statistics and the overflow report.
// For scenes whose shadow queries take the hierarchies, all of them or (with a few transparent spheres) ray by ray; the
// light-camera evaluations such a query postpones go to the frame's sharded queue in memory, all generations in one
// list, and are answered by the launch after this one.
struct LocalPlan {
    uint32_t chunks_per_block;   // primary chunks of a block
    uint32_t rotate;             // the blocks' shares move on by this many places from one round of primary chunks to the next
    uint32_t ray_cap, hit_cap;   // records of one block's ray list / hit list of one generation
    int32_t lds_counts_word;     // first LDS word of the block's CNT_WORDS counters, followed by 32 words of bookkeeping and a word
                                 // per 64 records of the ray lists of generations 1..5
    RayRec *rays;                // one slice per generation (1..6) and block
    HitRec *hits;                // one slice per generation and block
    uint32_t *frame_counts;      // the frame's counters
    HitRec *deferred;            // the frame's queue of postponed light-camera evaluations (all generations in one), or null
    uint32_t defer_shard_cap;
};
template <int MODE>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_SHADOW) void k_local(FrameParams P, QueueSet Q0, LocalPlan L)
{
    int *stack = &s_lds[threadIdx.x];
    const int lane = (int)(threadIdx.x & 63);
    const long long n_chunks = Q0.chunk_end - Q0.chunk_begin;
    // a block's primary chunks lie a grid apart, all over the frame: every block then gets its share of the parts of
    // the image where the trees are deep
    // (and not one column of the image: the k-th chunk of block b is chunk k * grid + (b + k * rotate) mod grid, so with
    // a grid that is a multiple of the blocks per image row a block still visits a different column every time)
    if ((long long)blockIdx.x >= n_chunks) return;
    const uint32_t full_rounds = (uint32_t)(n_chunks / gridDim.x), last_round = (uint32_t)(n_chunks - (long long)full_rounds * gridDim.x);
    const uint32_t my_chunks = full_rounds + ((blockIdx.x + full_rounds * L.rotate) % gridDim.x < last_round ? 1u : 0u);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(&s_lds[L.lds_counts_word]);
    for (int i = (int)threadIdx.x; i < CNT_WORDS + 32 + 5 * (int)(L.ray_cap >> 6); i += (int)blockDim.x) cnt[i] = 0u;
    if (threadIdx.x == 0) camera_to_lds(P);
    fill_node_cache(P.S, true);      // ends with the block's barrier
    const bool cull_s = P.cull_shadow != 0, cull_c = P.cull_closest != 0;
    const uint32_t nl = (uint32_t)P.S.n_lights;
    const size_t hit_bytes = (size_t)L.hit_cap * Q0.hit_stride;
    WalkCount wc{};
    auto queue_set = [&](int g) {
        QueueSet Q = Q0;
        // (a slice per generation, not two in turn: the first chunks of generation g+1 write the rays of generation g+2
        // while the last chunks of generation g still read theirs)
        Q.rays_in = L.rays + ((size_t)g * gridDim.x + blockIdx.x) * L.ray_cap;
        Q.rays_out = L.rays + ((size_t)(g + 1) * gridDim.x + blockIdx.x) * L.ray_cap;
        Q.hits = reinterpret_cast<HitRec *>(reinterpret_cast<char *>(L.hits) + ((size_t)g * gridDim.x + blockIdx.x) * hit_bytes);
        Q.counts = cnt;
        Q.in_shard_cap = L.ray_cap;
        Q.out_shard_cap = g < 5 ? L.ray_cap : 0u;
        Q.hit_shard_cap = L.hit_cap;
        Q.generation = g;
        Q.unit_chunks = 0;
        Q.whole_chunks = (uint32_t)g < Q0.whole_chunks ? 1u : 0u;        // (Q0.whole_chunks: so many generations push whole chunks)
        Q.deferred = L.deferred;
        Q.defer_shard_cap = L.defer_shard_cap;
        Q.defer_counts = L.deferred ? &L.frame_counts[CNT_DEFER(0)] : nullptr;
        return Q;
    };
    auto local_map = [&](uint32_t records, uint32_t cap, uint32_t per_record) {
        ShardMap m;                       // one shard, number 0
        const uint32_t n = records < cap ? records : cap;
        m.items = lane == 0 ? n * per_record : 0u;
        m.first = 0u;
        m.total = (n * per_record + 63u) >> 6;
        return m;
    };
    // No barrier between the passes.  The closest-hit passes C_0 .. C_5 are the only chain of dependences (C_g+1 traces
    // what C_g spawned); the shadow pass of a generation can run any time after that generation's closest-hit pass.  Each
    // wave keeps its own view of how far the block is (generations below `gd` are complete) and looks for work in this
    // order: a closest-hit chunk of generation gd (its ray list is final); a closest-hit chunk of a later generation whose
    // 64 rays have all been written already (the writers count them per 64-record piece, closest_chunk) -- so the
    // stragglers of one generation run beside the first chunks of the next; a shadow chunk of any complete generation;
    // else it waits until generation gd's last chunks are done (counted in LDS after a fence) and moves on.  A chunk is
    // only ever taken when everything it reads is written, so a wave that holds work never waits for another.
    const uint32_t pieces = L.ray_cap >> 6;       // 64-record pieces of a ray list
    volatile uint32_t *sync = reinterpret_cast<volatile uint32_t *>(&s_lds[L.lds_counts_word + CNT_WORDS]);   // taken_c[6], done_c[6], taken_s[6]
    uint32_t *sync_rw = const_cast<uint32_t *>(sync);
    volatile uint32_t *written = sync + 32;       // [generation - 1][piece], generations 1..5
    auto take = [&](int word, uint32_t total) {       // a ticket below `total`, or ~0u
        uint32_t t = ~0u;
        if (sync[word] < total) {
            if (lane == 0) t = atomicAdd(&sync_rw[word], 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= total) t = ~0u;
        }
        return t;
    };
    ShardMap none;
    none.items = none.first = none.total = 0u;
    int gd = 0;
    uint32_t total_c = my_chunks;
    uint32_t spins = 0;
    for (;;) {
        int cg = -1;                // a closest-hit chunk to trace: generation, chunk, records its list is known to hold
        uint32_t ct = 0, cn = 0;
        bool retry = false;
        if (gd < 6) {
            const uint32_t t = take(gd, total_c);
            if (t != ~0u) {
                cg = gd; ct = t;
                cn = gd ? cnt[CNT_RAYS(gd)] : 0u;
            }
            for (int gg = gd + 1; gg < 6 && cg < 0 && !retry; ++gg) {
                const uint32_t t2 = sync[gg];
                if (t2 < pieces && written[(gg - 1) * pieces + t2] == 64u) {
                    int won = 0;
                    if (lane == 0) won = atomicCAS(&sync_rw[gg], t2, t2 + 1u) == t2 ? 1 : 0;
                    won = __builtin_amdgcn_readfirstlane(won);
                    if (won) { cg = gg; ct = t2; cn = (t2 + 1u) << 6; }
                    else retry = true;            // another wave took it: look again
                }
            }
        }
        if (cg >= 0) {
            __threadfence_block();                // what the chunk reads was written before it was announced
            QueueSet Qc = queue_set(cg);
            Qc.commit = cg < 5 ? const_cast<uint32_t *>(written) + cg * pieces : nullptr;
            if (cg == 0) closest_chunk<true, true>(P, Qc, none, (long long)((blockIdx.x + ct * L.rotate) % gridDim.x) + (long long)ct * gridDim.x, lane, stack, cull_c, wc, 0u);
            else closest_chunk<false, true>(P, Qc, local_map(cn, L.ray_cap, 1u), (long long)ct, lane, stack, cull_c, wc, 0u);
            __threadfence_block();                // the records are written before the chunk counts as done
            if (lane == 0) atomicAdd(&sync_rw[6 + cg], 1u);
            spins = 0;
            continue;
        }
        if (retry) continue;
        bool shaded = false;
        for (int gg = 0; gg < gd && !shaded; ++gg) {
            ShardMap sm_s = local_map(nl ? cnt[CNT_HITS(gg)] : 0u, L.hit_cap, 1u);     // by light: a chunk per light for every 64 hits
            sm_s.total *= nl;
            const uint32_t t = take(12 + gg, sm_s.total);
            if (t != ~0u) {
                const QueueSet Qs = queue_set(gg);
                shadow_chunk<MODE, true>(P, Qs, sm_s, nl, (long long)t, stack, cull_s, wc);
                shaded = true;
            }
        }
        if (shaded) { spins = 0; continue; }
        if (gd == 6) break;                       // every generation is traced and no shadow chunk is left to take
        if (sync[6 + gd] == total_c) {
            __threadfence_block();                // generation gd is complete: what it wrote may be read
            ++gd;
            total_c = 0u;
            if (gd < 6) {
                const uint32_t n = cnt[CNT_RAYS(gd)];
                total_c = ((n < L.ray_cap ? n : L.ray_cap) + 63u) >> 6;
            }
            continue;
        }
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) {               // cannot happen (a chunk taken is a chunk finished); reported, not waited out
            cnt[CNT_OVERFLOW] = 1u;
            break;
        }
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < CNT_WORDS; i += (int)blockDim.x)
        if (cnt[i]) atomicAdd(&L.frame_counts[i], cnt[i]);
#ifdef RT_COUNT
    walk_count_flush(wc, 8);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The whole ray tree of a sample in one launch, for frames too small to fill the chip pass by pass.
//
// The wavefront pipeline above spends a launch per generation and kind of ray; a frame of a few hundred thousand
// samples fills the chip once at most, so each of its twelve passes lasts as long as one wave's walk plus what every
// pass pays again (LDS fill, queue counters, record round trips through memory, ramp and tail).  Here a lane keeps
// its sample from the primary ray to the last secondary ray: closest hit, the shadow queries of that hit towards
// every light, then the next ray of the tree (Camera.h:232-314 unrolled depth first: the refraction ray continues in
// registers, the reflection ray of the same hit waits in a small per-lane stack in private memory).  Nothing goes
// through queues except the deferred light-camera queries, the sample's colour is summed in registers and added to
// the accumulation frame once.  Lanes whose tree has ended idle until the wave's longest tree is done: acceptable
// exactly when there are not enough samples to fill the idle lanes anyway, so large frames keep the wavefront path.
struct PendingRay {          // a ray of the tree that waits for its turn
    double o[3], d[3], A, coef;
    uint32_t meta, _pad;
};
#define FUSED_PENDING 6      // a hit spawns at most two rays, one of which continues at once: at most one entry per level
#ifndef RT_MIN_WAVES_FUSED
#define RT_MIN_WAVES_FUSED 1
#endif
template <bool BVH, int MODE>
__global__ __launch_bounds__(256, RT_MIN_WAVES_FUSED) void k_fused(FrameParams P, QueueSet Q)
{
#define STAMP_SEL 16
    int *stack = &s_lds[BVH ? threadIdx.x : 0];
    const int lane = (int)(threadIdx.x & 63);
    const long long n_chunks = Q.chunk_end - Q.chunk_begin;
    if ((long long)blockIdx.x * (blockDim.x >> 6) >= n_chunks) return;
    STAMP(0);
    if (threadIdx.x == 0) camera_to_lds(P);
    fill_node_cache(P.S, BVH);
    STAMP(2);
    const bool cull = P.cull_closest != 0, cull_sh = P.cull_shadow != 0;
    const int nl = P.S.n_lights;
    WalkCount wc{};
    // rays and hits per generation, counted per wave (generation 0's rays are the samples: the host knows them)
    unsigned int w_rays[6] = {0, 0, 0, 0, 0, 0}, w_hits[6] = {0, 0, 0, 0, 0, 0};

    for (long long chunk = next_chunk(P.S); chunk < n_chunks; chunk = next_chunk(P.S)) {
        bool active, first_sample;
        d3 o, d;
        uint32_t pixel;
        uint32_t tile_unused;
        primary_ray(P, Q, chunk, lane, &active, &o, &d, &pixel, &first_sample, &tile_unused);
        double A = 1.0, coef = 1.0;
        uint32_t meta = 5u | 0x100u;
        bool have = active, primary = true;
        d3 sum = mk3(0, 0, 0);
        PendingRay pend[FUSED_PENDING];
        int sp = 0;
#ifdef RT_STAMP
        int stamp_iter = 0;
#endif
        while (__ballot(have) != 0) {
            const int L = (int)(meta & 0xffu);
#ifdef RT_STAMP
            if (stamp_iter == 0) STAMP(3);
            if (stamp_iter == 1) STAMP(4);
            if (stamp_iter == 2) STAMP(9);
            ++stamp_iter;
#define STAMP_IT(k) do { if (stamp_iter == 2) STAMP(k); } while (0)
#else
#define STAMP_IT(k) ((void)0)
#endif
            // ---- closest hit (Camera.h:318-363)
            double best_t = 0;
            int best = -1, best_prim = -1;
            if (have) {
                if (BVH) {
                    const double d2 = dot3(d, d);
                    Best bst;
                    bst.t = 10000000000.0; bst.key = 0x7fffffff; bst.slot = -1; bst.prim = -1;
                    if (root_gate(P.S, o, d, d2, 10000000000.0)) {
                        const RayBox4 rb = make_raybox4(o, d, P.S.bvh_bound);
                        bvh_closest<true>(P.S, o, d, d2, rb, bst, stack, wc);
                        bvh_closest<false>(P.S, o, d, d2, rb, bst, stack, wc);
                    }
                    best_t = bst.t; best = bst.slot; best_prim = bst.prim;
                } else {
                    trace_closest(P.S, o, d, cull, .01, best_t, best);
                }
            }
            STAMP_IT(5);
            const rt_tri_rec *tri_arr = BVH ? P.S.btri : P.S.tri;
            const rt_sph_rec *sph_arr = BVH ? P.S.bsph : P.S.sph;
            const uint32_t *sph_prim_arr = BVH ? P.S.bsph_prim : P.S.sph_prim;
            const bool first = (meta & 0x100u) != 0;
            const double wloc = first ? A : A * coef;
            const bool hit = have && best >= 0;
            d3 n = mk3(0, 0, 0), Pt = mk3(0, 0, 0);
            int prim = -1, mat = 0;
            bool flat = false;
            rt_material m;
            m.Ks = 0; m.T = 0; m.ior = 1;
            if (hit) {
                hit_geometry(P.S, tri_arr, sph_arr, sph_prim_arr, P.flags, o, d, .01, best_t, best, best_prim, &prim, &n, &Pt, &flat);
                mat = P.S.prim_material[prim];
                m = load_material(P.S, mat, true);
            }
            if (primary && active && first_sample && P.hit_id) P.hit_id[pixel] = prim;
            if (have && !hit) {      // a miss adds the background
                const d3 bg = ld3(P.S.bg);
                sum = add3(sum, mk3(wloc * bg.x, wloc * bg.y, wloc * bg.z));
            }
            const bool want_hit = hit && nl > 0;
            // per-generation counts of the wave (uniform values: ballots and scalar adds)
            {
                const int gen = 5 - L;
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    if (g > 0) w_rays[g] += (unsigned int)__popcll(__ballot(have && gen == g));
                    w_hits[g] += (unsigned int)__popcll(__ballot(want_hit && gen == g));
                }
            }

            STAMP_IT(6);
            // ---- the lights of this hit (Camera.h:385-485)
            if (__ballot(want_hit) != 0) {
                d3 vv = mk3(0, 0, 0), n2 = mk3(0, 0, 0), n3 = mk3(0, 0, 0);
                if (want_hit) {
                    vv = unit3(sub3(o, Pt));
                    if (!flat) { n2 = unit3(n); n3 = unit3(n2); }
                }
                for (int li = 0; li < nl; ++li) {
                    const d3 lpos = ld3(load_light(P.S, li, true).pos);
                    bool lit = false;
                    int defer_cam = -1;
                    d3 local = mk3(0, 0, 0);
                    if (want_hit) {
                        const d3 toL = sub3(lpos, Pt);
                        const d3 rd = unit3(toL);
                        const double maxT = norm3(toL) - .01;
                        bool only = false;
                        int closest = -1;
                        bool occluded;
                        bool order_free = MODE == SHADOW_BVH;
                        if (MODE == SHADOW_HYBRID) {
                            const double d2 = dot3(rd, rd);
                            order_free = true;
                            for (int k = 0; k < P.S.n_tsph; ++k)
                                if (P.S.tsph_prim[k] != prim && sph_test(&P.S.tsph[k], Pt, rd, d2, .01, maxT) < maxT) order_free = false;
                        }
                        if (order_free) {
                            const double d2 = dot3(rd, rd);
                            occluded = false;
                            if (root_gate(P.S, Pt, rd, d2, 10000000.0)) {
                                const RayBox4 rb = make_raybox4(Pt, rd, P.S.bvh_bound);
                                occluded = bvh_any<true>(P.S, Pt, rd, d2, rb, prim, maxT, stack, wc) ||
                                           bvh_any<false>(P.S, Pt, rd, d2, rb, prim, maxT, stack, wc);
                            }
                        } else if (MODE != SHADOW_BVH) {
                            occluded = trace_shadow(P.S, Pt, rd, prim, maxT, cull_sh, &only, &closest);
                        } else {
                            occluded = false;
                        }
                        if (occluded && only) {
                            if (P.lc_mask) P.lc_mask[pixel] = 1;
                            if (P.S.lc_table && closest >= 0) {
                                const int rank = P.S.prim_lc_rank[closest];
                                if (rank >= 0) {
                                    defer_cam = rank * nl + li;
                                    P.S.lc_needed[defer_cam] = 1u;
                                }
                            }
                        }
                        lit = !occluded;
                        if (lit) {
                            const rt_light lgt = load_light(P.S, li, true);
                            d3 na = n2, nb = n3;
                            if (flat) {
                                const void *nsrc = (const void *)&P.S.prim_n23[(size_t)prim * 6];
                                const rec_v2d q0 = tab_load(nsrc, 0), q1 = tab_load(nsrc, 1), q2 = tab_load(nsrc, 2);
                                na = mk3(q0.x, q0.y, q1.x); nb = mk3(q1.y, q2.x, q2.y);
                            }
                            local = phong_terms(load_material(P.S, mat, true), na, nb, vv, &lgt, rd, .999999);
                        }
                    }
                    if (MODE != SHADOW_BVH && P.S.lc_table) {
                        // the light camera decides (Camera.h:406-451): postponed to k_lightcam, which runs such queries in full waves
                        bool ok;
                        const bool want = defer_cam >= 0;
                        const uint32_t slot = wave_reserve(want, &Q.counts[CNT_DEFER(0)], Q.defer_shard_cap, &Q.counts[CNT_OVERFLOW + 1], &ok);
                        if (ok) {
                            HitRec *dr = &Q.deferred[slot];
                            const rec_v2d q[4] = {rec_pack(Pt.x, Pt.y), rec_pack(Pt.z, vv.x), rec_pack(vv.y, vv.z),
                                                  rec_pack(wloc, __hiloint2double(prim, (int)(pixel | (flat ? HIT_FLAT : 0u))))};
                            tab_store(dr, q);
                            const rec_v2d r[4] = {rec_pack_words((uint32_t)li, (uint32_t)defer_cam, (uint32_t)mat, 0u), rec_pack(n2.x, n2.y),
                                                  rec_pack(n2.z, n3.x), rec_pack(n3.y, n3.z)};
                            tab_store(&dr->light, r);
                        } else if (want) {
                            Q.counts[CNT_OVERFLOW] = 1u;      // queue full: the frame is incomplete and is reported as such
                        }
                    }
                    if (lit) sum = add3(sum, mk3(wloc * local.x, wloc * local.y, wloc * local.z));
                }
            }

            STAMP_IT(7);
            // ---- the next ray of this sample's tree (Camera.h:244-307)
            const bool spawn = hit && L > 0 && !(coef < .01);
            const double coef2 = coef * m.Ks;
            const bool want_refr = spawn && m.T > 0;
            const bool want_refl = spawn && (coef2 != 0.0 || (P.flags & RT_FLAG_KEEP_ZERO_WEIGHT));
            bool cont = false;
            d3 nd = d;
            double nA = A, ncoef = coef;
            uint32_t nmeta = meta;
            if (want_refl) {
                nd = unit3(add3(d, scale3(2.0 * -dot3(d, n), n)));      // hr.reflectionDir (Solid.h:121-122)
                ncoef = coef2;
                nmeta = (uint32_t)(L - 1);
                cont = true;
            }
            if (want_refr) {
                if (cont) {
                    if (sp < FUSED_PENDING) {
                        PendingRay &pr = pend[sp];
                        pr.o[0] = Pt.x; pr.o[1] = Pt.y; pr.o[2] = Pt.z;
                        pr.d[0] = nd.x; pr.d[1] = nd.y; pr.d[2] = nd.z;
                        pr.A = nA; pr.coef = ncoef; pr.meta = nmeta;
                        ++sp;
                    } else {
                        Q.counts[CNT_OVERFLOW] = 1u;      // cannot happen (one entry per level); reported if it does
                    }
                }
                nd = refract_dir(d, n, m.ior, false);
                nA = A * m.T;
                ncoef = m.T * coef;
                nmeta = (uint32_t)(L - 1) | 0x100u;
                cont = true;
            }
            d3 no = Pt;
            if (have && !cont && sp > 0) {
                --sp;
                const PendingRay &pr = pend[sp];
                no = ld3(pr.o); nd = ld3(pr.d);
                nA = pr.A; ncoef = pr.coef; nmeta = pr.meta;
                cont = true;
            }
            have = have && cont;
            o = no; d = nd; A = nA; coef = ncoef; meta = nmeta;
            primary = false;
        }
        STAMP(10);
        accum_add_runs(P.accum, active, true, pixel, 1.0, sum);
    }
    STAMP(11);
    // totals: [0] closest rays, [1] shadow rays, [2+g] rays and [8+g] hits of generation g, [15] overflow
    if (lane == 0) {
        unsigned long long r = 0, h = 0;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (w_rays[g]) atomicAdd(&Q.totals[2 + g], (unsigned long long)w_rays[g]);
            if (w_hits[g]) atomicAdd(&Q.totals[8 + g], (unsigned long long)w_hits[g]);
            r += w_rays[g]; h += w_hits[g];
        }
        if (r) atomicAdd(&Q.totals[0], r);
        if (h) atomicAdd(&Q.totals[1], h * (unsigned long long)nl);
    }
#ifdef RT_COUNT
    walk_count_flush(wc, 0);
#endif
#undef STAMP_SEL
}

// Deferred light-camera evaluations of a batch, all six generations in one launch: density query,
// then the same shading as a lit point with the query result as the factor (Camera.h:444-476).
// A query is a long chain of dependent fetches and a launch lasts as long as its longest one, so
// the generations are not given a launch each: their 64-query chunks form one list, dealt round
// to the waves.
struct LightcamQueues {
    const HitRec *deferred[6];
    uint32_t shard_cap[6];
    const uint32_t *counts;
    uint32_t *overflow;          // the frame's "incomplete" word
    uint32_t *next_chunk;        // chunks handed out so far (zero at launch)
    int32_t pool_cap;            // places of a wave's pool of pending cells that are used (LC_POOL)
};
#ifndef LC_MIN_WAVES
#define LC_MIN_WAVES 4
#endif
__global__ __launch_bounds__(LC_BLOCK, LC_MIN_WAVES) void k_lightcam(FrameParams P, LightcamQueues L)
{
    ShardMap smap[6];
    long long first[7];
    first[0] = 0;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        smap[g] = shard_map(&L.counts[CNT_DEFER(g)], L.shard_cap[g], 1u);
        first[g + 1] = first[g] + (long long)smap[g].total;
    }
    // chunks differ in length by a factor of five: waves draw them from a counter, in order, instead of keeping to a fixed share
    for (;;) {
        uint32_t drawn = 0;
        if ((threadIdx.x & 63) == 0) drawn = atomicAdd(L.next_chunk, 1u);
        const long long work = (long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
        if (work >= first[6]) break;
        int g = 0;
#pragma unroll
        for (int k = 1; k < 6; ++k) if (work >= first[k]) g = k;
        // the generation is the same for the whole wave; pick its map without indexing registers dynamically
        ShardMap m = smap[0];
        const HitRec *base = L.deferred[0];
        uint32_t cap = L.shard_cap[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) if (g == k) { m = smap[k]; base = L.deferred[k]; cap = L.shard_cap[k]; }
        int shard;
        const int item = shard_locate(m, (uint32_t)(work - first[g]), &shard);
        const bool active = item >= 0;
        const HitRec *hr = &base[(size_t)shard * cap + (active ? (uint32_t)item : 0u)];
        bool lit = false;
        d3 local = mk3(0, 0, 0);
        const d3 Pt = ld3(hr->P);
        const double kfac = lc_calc_intensity_wave(P.S.lc_table, active ? hr->cam : 0, Pt, active, L.overflow, L.pool_cap);
        if (active) {
            const rt_light *lt = &P.S.lights[hr->light];
            if (kfac > .01) {
                lit = true;
                const double *nsrc = (hr->pixel & HIT_FLAT) ? &P.S.prim_n23[(size_t)hr->prim * 6] : hr->n;
                local = phong_terms(load_material(P.S, hr->mat, false), ld3(nsrc), ld3(nsrc + 3), ld3(hr->v), lt,
                                    unit3(sub3(ld3(lt->pos), Pt)), kfac);
            }
        }
        accum_add_runs(P.accum, active, lit, hr->pixel & ~HIT_FLAT, hr->weight, local);
    }
}

struct PhotonParams {
    DevScene S;
    double u[3], v[3], w[3], e[3];
    double d, l, t, pix_width;
    int32_t N;
    uint32_t flags;
    int32_t cull;
    lightcam::Point *slots;     // N*N rows of 21
    int32_t *counts;            // N*N
};
#define LC_MAX_IMPACTS 21

__global__ __launch_bounds__(256) void k_photons(PhotonParams Q)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)Q.N * Q.N;
    if (idx >= total) return;
    const int si = (int)(idx / Q.N), sj = (int)(idx - (long long)si * Q.N);
    const double spw = Q.pix_width / Q.N;
    double p, q;
    if (Q.N > 1) {
        const double edgeX = (0 - .5) * Q.pix_width;
        const double edgeY = (0 - .5) * Q.pix_width;
        const double j_m = edgeX + sj * spw + .5 * spw;
        const double i_m = edgeY + si * spw + .5 * spw;
        p = Q.l + j_m;
        q = Q.t - i_m;
    } else {
        p = Q.l + 0 * Q.pix_width;
        q = Q.t - 0 * Q.pix_width;
    }
    const double r = -Q.d;
    d3 d = unit3(add3(add3(scale3(p, ld3(Q.u)), scale3(q, ld3(Q.v))), scale3(r, ld3(Q.w))));
    d3 o = ld3(Q.e);
    double coef = 1.0, CoR = 1.0;
    int budget = 20, n_out = 0, calls = 0;
    bool top = true;
    lightcam::Point *row = Q.slots + idx * LC_MAX_IMPACTS;
    while (n_out < LC_MAX_IMPACTS) {
        ++calls;                                                   // one castLightRay call
        if (!(coef > .05)) break;                                  // LightCam.h:502
        double t;
        int best;
        trace_closest(Q.S, o, d, Q.cull != 0, .000001, t, best);    // selfIntersectBias 1e-6, LightCam.h:44
        if (best < 0) break;
        int prim;
        d3 n, Pt;
        hit_geometry(Q.S, Q.S.tri, Q.S.sph, Q.S.sph_prim, Q.flags, o, d, .000001, t, best, -1, &prim, &n, &Pt);
        const rt_material m = Q.S.materials[Q.S.prim_material[prim]];
        lightcam::Point pt;
        pt.x = Pt.x; pt.y = Pt.y; pt.z = Pt.z;
        pt.w = coef * (1.0 - m.T);                                 // LightCam.h:536
        row[n_out++] = pt;
        if (top && !(best & SLOT_SPHERE)) break;                    // LightCam.h:360
        if (budget <= 0) break;
        CoR *= m.Ks;                                               // LightCam.h:380
        if (coef < .01) break;
        if (CoR < .01 && m.T == 0) break;
        if (!(m.T > 0)) break;                                     // only refraction continues the chain
        d = refract_dir(d, n, m.ior, true);
        o = Pt;
        coef = coef * m.T;
        budget -= 1;
        top = false;
    }
    Q.counts[idx] = n_out | (calls << 8);
}

// totals[0]=closest rays, [1]=shadow rays, [2..7]=rays per generation (generation 0 is filled in by the
// host, which knows the number of primary samples), [8..13]=hits per generation, [15]=overflow
// Packs the impacts of a photon pass: ray i's first `counts[i] & 0xff` slots go to out[offsets[i]...], ray order kept.
__global__ __launch_bounds__(256) void k_pack_photons(const lightcam::Point *slots, const int32_t *counts, const uint32_t *offsets,
                                                      int n_rays, int slots_per_ray, lightcam::Point *out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_rays) return;
    const int n = counts[i] & 0xff;
    for (int k = 0; k < n; ++k) out[offsets[i] + k] = slots[(size_t)i * slots_per_ray + k];
}

// One wave, lane = shard: a batch's queue counters added to the frame's totals.
DEV void collect_counts(const uint32_t *counts, unsigned long long *totals, int n_lights)
{
    const int lane = (int)threadIdx.x;
    for (int g = 0; g < 6; ++g) {
        unsigned long long r = (g == 0) ? 0ull : counts[CNT_RAYS(g) + lane];
        unsigned long long h = counts[CNT_HITS(g) + lane];
        h -= counts[CNT_PAD_HITS(g) + lane];                 // records that only fill chunks up are no hits and no rays
        if (g > 0) r -= counts[CNT_PAD_RAYS(g) + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off);
            h += __shfl_down(h, off);
        }
        if (lane == 0) {
            if (g == 0) h += counts[CNT_FUSED_HITS];        // (primary hits shaded where they were found: no records, the same rays)
            totals[2 + g] += r;
            totals[8 + g] += h;
            totals[0] += r;
            totals[1] += h * (unsigned long long)n_lights;
        }
    }
    if (lane == 0) totals[15] += counts[CNT_OVERFLOW];
}
__global__ void k_collect_counts(const uint32_t *counts, unsigned long long *totals, int n_lights)
{
    if (blockIdx.x != 0 || threadIdx.x >= N_SHARDS) return;
    collect_counts(counts, totals, n_lights);
}

// (the accumulation frame is left at zero: the next frame of the context starts from it)
// The frame's last kernel also hands the totals to the host: its first wave rolls up the counters of a frame that was one
// launch (`counts`, else null: the batches' own roll-ups have run) and copies the totals to `host_totals` (pinned memory).
__global__ __launch_bounds__(256) void k_finalize(double *accum, long long n_values, int spp, double *out_f64, uint8_t *out_u8,
                                                  const uint32_t *counts, unsigned long long *totals, unsigned long long *host_totals,
                                                  int n_lights)
{
    if (blockIdx.x == 0 && threadIdx.x < N_SHARDS) {
        if (counts) collect_counts(counts, totals, n_lights);
        if (threadIdx.x == 0)
            for (int k = 0; k < 16; ++k) host_totals[k] = totals[k];
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) {
        double c = accum[i] / (double)spp;
        accum[i] = 0.0;
        c = (c < 1.0) ? c : 1.0;        // std::min(1.0, c)
        c = (0.0 < c) ? c : 0.0;        // std::max(0.0, c)
        double v = c * 255;
        if (out_f64) out_f64[i] = v;
        if (out_u8) out_u8[i] = (uint8_t)(int)v;
    }
}

// One primitive test per thread with the device functions the kernels use (rt_debug_intersect).  Kinds: 0 triangle,
// 1 its division-sparing form, 2 sphere, 3 node bound, 4 any-hit triangle form, 5 nearest-hit triangle form with cut-off maxT,
// 6 the power of the specular term (geom: base, exponent, -, -), 7 a vector's normalisation (geom: x, y, z, -; the unit
// vector comes back where a sphere's normal does).
__global__ void k_debug_intersect(int kind, int n, const double *geom, const double *rays, double minT, double maxT, double *out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const d3 o = ld3(rays + (size_t)i * 6), d = ld3(rays + (size_t)i * 6 + 3);
    double t = maxT;
    d3 nrm = mk3(0, 0, 0);
    if (kind == 0 || kind == 1 || kind == 4 || kind == 5) {
        const double *g = geom + (size_t)i * 9;
        rt_tri_rec tr;
        for (int k = 0; k < 3; ++k) { tr.v0[k] = g[k]; tr.ab[k] = g[k] - g[3 + k]; tr.ac[k] = g[k] - g[6 + k]; }   // Triangle.h:40-46
        tr.prim = 0; tr._pad0 = 0; tr._pad1[0] = tr._pad1[1] = 0;
        if (kind == 0) { double al, be; t = tri_test(&tr, o, d, minT, maxT, &al, &be); }
        else {
            TriOps ops;
            ops.v0 = ld3(tr.v0); ops.ab = ld3(tr.ab); ops.ac = ld3(tr.ac); ops.prim = 0; ops.key = 0;
            if (kind == 1) t = tri_test_fast(ops, o, d, minT, maxT, maxT);
            else if (kind == 4) t = tri_any_fast(ops, o, d, minT, maxT) ? 0.0 : maxT;       // any-hit form: 0 for a hit
            else t = tri_test_fast(ops, o, d, minT, 10000000000.0, maxT);                   // nearest-hit form, maxT is its cut-off
        }
    } else if (kind == 6) {
        t = spec_pow(geom[(size_t)i * 4], geom[(size_t)i * 4 + 1]);
    } else if (kind == 7) {
        nrm = unit3(ld3(geom + (size_t)i * 4));
    } else if (kind == 2) {
        const double *g = geom + (size_t)i * 4;
        rt_sph_rec sp;
        sp.center[0] = g[0]; sp.center[1] = g[1]; sp.center[2] = g[2]; sp.radius = g[3];
        t = sph_test(&sp, o, d, dot3(d, d), minT, maxT);
        if (t < maxT) nrm = unit3(div3(sub3(add3(o, scale3(t, d)), ld3(sp.center)), sp.radius));
    } else {
        const double *g = geom + (size_t)i * 4;
        rt_node nd;
        nd.center[0] = g[0]; nd.center[1] = g[1]; nd.center[2] = g[2]; nd.radius = g[3];
        double entry;
        t = bound_test(&nd, o, d, dot3(d, d), minT, maxT, &entry);
    }
    out[(size_t)i * 4] = t; out[(size_t)i * 4 + 1] = nrm.x; out[(size_t)i * 4 + 2] = nrm.y; out[(size_t)i * 4 + 3] = nrm.z;
}

