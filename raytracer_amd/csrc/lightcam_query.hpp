// Density query of a light camera (LightCam.h:121-203, 571-620) over its flattened point octree.
#pragma once
#include "device_math.hpp"
#include "device_records.hpp"

// LightCam::calcIntensity + getIntensity (LightCam.h:121-203, 571-620): density of light-ray
// impacts around `pos`.  The reference descends the point octree while a node is at least
// 2*maxDist wide and near enough, and at the first smaller node (or a `leaf`) adds up the
// weights of ALL of that node's points that lie within maxDist.  Which nodes are reached is
// reproduced exactly.  Inside such a node the sum is taken hierarchically: a sub-cell wholly
// inside the query sphere contributes its precomputed weight sum, one wholly outside nothing,
// and only cells the sphere's surface cuts are opened down to their points, each point decided by
// the reference's own comparison.  The caustic foci put thousands of impacts into single cells;
// this visits a few hundred cells and points instead.  (The order of the additions differs from
// the reference's, as it already must for any parallel sum: a rounding-level difference.)
//
// A wave answers its 64 queries together.  A query is a chain of dependent fetches, one per cell it
// visits, and queries differ by a factor of ten in how many cells that is: with a query per lane a wave
// lasts as long as its longest query while most of its lanes idle.  Here the cells still to be visited by
// any of the 64 queries lie in a pool in LDS, each with the lane that owns its query; every trip the 64
// lanes take the 64 newest cells of one kind, whoever's they are, fetch them together, and put the children
// to be visited back.  The two kinds -- cells whose children are to be judged, cells whose points are to be
// summed -- fill the pool from its two ends, and a trip handles one kind, so that its lanes run the same code.
// The queries' positions and radii, and the sums (LDS atomic adds), are in LDS as well.
// Point ranges longer than LC_COOP_MIN are scanned by all 64 lanes.
// `overflow` is set to 1 if the pool ever fills up, so that a sum with a cell missing is never returned silently.
#define LC_TERMINAL 0x40000000   // pool code: sum this cell's points within range
#define LC_EXPAND 0x20000000     // pool code: a large, near cell whose children are to be judged
#ifndef LC_POOL
#define LC_POOL 1536             // cells pending for a wave's 64 queries
#endif
#ifndef LC_OPEN_MIN
#define LC_OPEN_MIN 2        // cells with more points than this are classified against the query sphere
#endif
#ifndef LC_COOP_MIN
#define LC_COOP_MIN 24       // point ranges longer than this are scanned by the whole wave
#endif
#ifndef LC_BLOCK
#define LC_BLOCK 64          // threads per block of the kernel that calls lc_query_wave
#endif
DEV lightcam::Node load_lc_node(const lightcam::Node *n)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) v4i *p = (const __attribute__((address_space(1))) v4i *)n;
    union { v4i q[6]; lightcam::Node nd; } u;
#pragma unroll
    for (int k = 0; k < 6; ++k) u.q[k] = p[k];
    return u.nd;
}

struct LcWaveLds {               // one wave's share of LDS
    int code[LC_POOL];           // pending cells: node index | LC_TERMINAL / LC_EXPAND
    unsigned char owner[LC_POOL];   // the lane whose query the cell belongs to
    double px[64], py[64], pz[64], maxd[64], total[64];
    const lightcam::Node *nodes[64];
    const lightcam::Point *points[64];
};

DEV void lc_total_add(LcWaveLds &W, int own, double v)
{
    __hip_atomic_fetch_add(&W.total[own], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Every lane of the wave calls it together; `active` lanes hold a query.
// `pool_cap` (64 < pool_cap <= LC_POOL) is how much of the pool is used: smaller values only exist to exercise a full pool.
DEV double lc_query_wave(const DevLightCam *table, int cam_index, d3 pos, bool active, uint32_t *overflow, int pool_cap)
{
    __shared__ LcWaveLds s_lc[LC_BLOCK / 64];
    LcWaveLds &W = s_lc[threadIdx.x >> 6];
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long below = (1ull << lane) - 1ull;
    const DevLightCam cam = table[active ? cam_index : 0];
    double my_maxDist;
    {
        const d3 lp = ld3(cam.light_pos), spos = ld3(cam.sphere_pos);
        const double dist = norm3(sub3(pos, lp));
        const double sphereDistToLight = norm3(sub3(lp, spos));
        my_maxDist = cam.spw * (dist / sphereDistToLight);
        my_maxDist *= 1.5;                              // smoothingFactor
    }
    W.px[lane] = pos.x; W.py[lane] = pos.y; W.pz[lane] = pos.z;
    W.maxd[lane] = my_maxDist;
    W.total[lane] = 0;
    W.nodes[lane] = cam.nodes;
    W.points[lane] = cam.points;
    // cells to expand: code[0 .. nE), newest last; cells to sum: code[LC_POOL - nT .. LC_POOL), newest first
    int nE, nT = 0;     // (the same in every lane)
    {
        const bool has = active && cam.n_nodes > 0;
        const unsigned long long m = __ballot(has);
        if (has) { const int s = __popcll(m & below); W.code[s] = cam.root; W.owner[s] = (unsigned char)lane; }
        nE = __popcll(m);
    }
#ifdef RT_COUNT
    unsigned long long c_desc = 0, c_term = 0, c_sum = 0, c_pts = 0, c_coop = 0, c_cooppts = 0;
    unsigned long long t_iters = 0, t_coop = 0, t_phases = 0, t_fill = 0, t_pool = 0, t_exp = 0, n_exp = 0, t_sum = 0, n_sum = 0;
    const unsigned long long t_begin = wall_clock64();
#define LCC(x) (x)
#else
#define LCC(x) ((void)0)
#endif
    // puts the cells push[0..7] of the lanes into the pool, children 7..0 of all lanes in turn (so that the newest,
    // taken first, are the first children); a group that does not fit is left out and the frame reported incomplete
#define LC_PUSH_ALL(ALSO_EXPAND)                                                                      \
    _Pragma("unroll") for (int c = 7; c >= 0; --c) {                                                  \
        const bool pt = push[c] >= 0 && (!(ALSO_EXPAND) || (push[c] & LC_TERMINAL));                  \
        const bool pe = (ALSO_EXPAND) && push[c] >= 0 && !(push[c] & LC_TERMINAL);                    \
        const unsigned long long mt = __ballot(pt), me = (ALSO_EXPAND) ? __ballot(pe) : 0ull;         \
        if (mt) {                                                                                     \
            const int cnt = __popcll(mt);                                                             \
            if (nE + nT + cnt <= pool_cap) {                                                           \
                if (pt) { const int s = pool_cap - 1 - (nT + __popcll(mt & below)); W.code[s] = push[c]; W.owner[s] = (unsigned char)own; } \
                nT += cnt;                                                                            \
            } else *overflow = 1u;                                                                    \
        }                                                                                             \
        if (me) {                                                                                     \
            const int cnt = __popcll(me);                                                             \
            if (nE + nT + cnt <= pool_cap) {                                                           \
                if (pe) { const int s = nE + __popcll(me & below); W.code[s] = push[c]; W.owner[s] = (unsigned char)own; } \
                nE += cnt;                                                                            \
            } else *overflow = 1u;                                                                    \
        }                                                                                             \
    }
    while (nE + nT > 0) {
        LCC(++t_iters);
        LCC(t_pool = (unsigned long long)(nE + nT) > t_pool ? (unsigned long long)(nE + nT) : t_pool);
        // A full trip of cells to sum if there is one; otherwise cells to expand, which bring more of both kinds.  A cell
        // taken frees one place and may need eight: a trip takes no more cells than the room in the pool allows (with 64
        // places kept back from the expanding trips, so that the summing ones, which drain the pool, can always run), and
        // a nearly full pool slows a wave down instead of losing cells.
        const int room = pool_cap - nE - nT;
        const int cap_e = room - 64 >= 7 * 64 ? 64 : (room - 64 < 7 ? 0 : (room - 64) / 7);
        const int cap_t = room >= 7 * 64 ? 64 : (room < 7 ? 1 : room / 7);
        const bool sum_trip = nT > 0 && (nT >= 64 || nE == 0 || cap_e == 0);
        const int avail = sum_trip ? nT : nE;
        const int cap = sum_trip ? cap_t : (cap_e > 0 ? cap_e : 1);
        const int take = avail < cap ? avail : cap;
        const bool have = lane < take;
        LCC(t_fill += (unsigned long long)take);
#ifdef RT_COUNT
        const unsigned long long t_trip = wall_clock64();
#endif
        const int at = !have ? 0 : (sum_trip ? pool_cap - nT + lane : nE - 1 - lane);
        const int code = W.code[at];
        const int own = (int)W.owner[at];
        if (sum_trip) nT -= take; else nE -= take;
        const d3 qp = mk3(W.px[own], W.py[own], W.pz[own]);
        const double maxDist = W.maxd[own];
        int push[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) push[c] = -1;
        if (!sum_trip) {
            if (have) {
                // the whole 96-byte cell record in six 16-byte loads issued together: fields fetched one by one
                // where they are used would each cost a trip to memory of their own
                const lightcam::Node nd_ = load_lc_node(&W.nodes[own][code & ~(LC_TERMINAL | LC_EXPAND)]);
                const lightcam::Node *r = &nd_;
                bool expand = (code & LC_EXPAND) != 0;
                const double epsilon = maxDist * 0.1;
                if (!expand) {       // only the root arrives undecided
                    if (r->size < maxDist * 2 || r->leaf) push[0] = code | LC_TERMINAL;
                    else if (norm3(sub3(qp, mk3(r->cx, r->cy, r->cz))) < (r->size + epsilon)) expand = true;
                }
                LCC(++c_desc);
                if (expand) {
                    // The children's cubes follow from this node's cube by the builder's own arithmetic
                    // (Octree.h:1430-1465), so each child is judged here, before it is fetched: a small
                    // or `leaf` child is summed, a large one is descended into only if it is near.
                    // A child's low corner is the node's centre or that minus half its size, per axis; its size
                    // follows from the x corner: two values per axis, worked out once for the eight children.
                    const double size = r->size, hh = size / 2;
                    const double lx[2] = {r->cx - size / 2, r->cx}, ly[2] = {r->cy - size / 2, r->cy}, lz[2] = {r->cz - size / 2, r->cz};
                    double lim[2], lim_lo[2], lim_hi[2], dx2[2], dy2[2][2], dz2[2][2];
                    bool small[2];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const double csize = (lx[a] + hh) - lx[a];
                        small[a] = csize < maxDist * 2;
                        lim[a] = csize + epsilon;
                        const double lim2 = lim[a] * lim[a];
                        lim_lo[a] = lim2 * (1.0 - 0x1p-48);
                        lim_hi[a] = lim2 * (1.0 + 0x1p-48);
                        const double dx = qp.x - (lx[a] + csize / 2);
                        dx2[a] = dx * dx;
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) {
                            const double dy = qp.y - (ly[b2] + csize / 2), dz = qp.z - (lz[b2] + csize / 2);
                            dy2[a][b2] = dy * dy;
                            dz2[a][b2] = dz * dz;
                        }
                    }
                    // |pos - centre| < csize + epsilon is decided on the squares; where they are a few ulps apart (kept out
                    // of the eight children's code: a square root is as long as the rest of a child) the reference's own comparison
                    int unsure = 0;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int ch = r->child[c];
                        if (ch < 0) continue;
                        const int a = c & 1;
                        if (small[a] || ((r->child_leaf >> c) & 1)) push[c] = ch | LC_TERMINAL;
                        else {
                            const double s2 = (dx2[a] + dy2[a][(c >> 1) & 1]) + dz2[a][(c >> 2) & 1];
                            if (s2 < lim_lo[a]) push[c] = ch | LC_EXPAND;
                            else if (!(s2 > lim_hi[a])) unsure |= 1 << c;
                        }
                    }
                    if (__ballot(unsure != 0)) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const int a = c & 1;
                            const double s2 = (dx2[a] + dy2[a][(c >> 1) & 1]) + dz2[a][(c >> 2) & 1];
                            if (((unsure >> c) & 1) && sqrt(s2) < lim[a]) push[c] = r->child[c] | LC_EXPAND;
                        }
                    }
                }
            }
            LC_PUSH_ALL(true)
            LCC(t_exp += wall_clock64() - t_trip); LCC(++n_exp);
            continue;
        }
        // sqrt is monotonic and correctly rounded: |v| < maxDist is decided on the squared length
        // unless it lies within a few ulps of maxDist^2, where the reference's own comparison runs
        const double m2 = maxDist * maxDist, m2_lo = m2 * (1.0 - 0x1p-48), m2_hi = m2 * (1.0 + 0x1p-48);
        int pend_b = 0, pend_e = 0;
        bool pending = false;
        if (have) {
            const lightcam::Node nd_ = load_lc_node(&W.nodes[own][code & ~(LC_TERMINAL | LC_EXPAND)]);
            const lightcam::Node *r = &nd_;
            LCC(++c_term);
            // every point of this cell that lies within maxDist counts
            int sb = r->begin, se = r->end;
            bool scan = true;
            if (se - sb > LC_OPEN_MIN) {
                const double m2_in = m2 * (1.0 - 1e-9), m2_out = m2 * (1.0 + 1e-9);   // cell classification, with a margin
                const double h = r->size * 0.5;
                const double ax = fabs(qp.x - r->cx), ay = fabs(qp.y - r->cy), az = fabs(qp.z - r->cz);
                const double nx = fmax(ax - h, 0.0), ny = fmax(ay - h, 0.0), nz = fmax(az - h, 0.0);
                const double fx = ax + h, fy = ay + h, fz = az + h;
                const double dmin2 = (nx * nx + ny * ny) + nz * nz, dmax2 = (fx * fx + fy * fy) + fz * fz;
                if (dmin2 > m2_out) scan = false;
                else if (dmax2 < m2_in) { lc_total_add(W, own, r->wsum); LCC(++c_sum); scan = false; }
                else if (r->rest_begin > sb) {   // has children: open it, its own leftover points are scanned
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (r->child[c] >= 0) push[c] = r->child[c] | LC_TERMINAL;
                    sb = r->rest_begin;
                }
            }
            if (scan && se > sb) {
                if (se - sb > LC_COOP_MIN) {
                    pend_b = sb; pend_e = se; pending = true;
                    LCC(++c_coop); LCC(c_cooppts += (unsigned long long)(se - sb));
                } else {
                    LCC(c_pts += (unsigned long long)(se - sb));
                    // four points per trip, fetched together: one lane alone would otherwise wait out a
                    // full memory latency for every single point
                    const lightcam::Point *pts = W.points[own];
                    double sum = 0;
                    for (int k = sb; k < se; k += 4) {
                        lightcam::Point q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) q[u] = pts[k + u < se ? k + u : se - 1];
                        double s2[4];
                        bool inside[4], unsure = false;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const d3 dv = sub3(mk3(q[u].x, q[u].y, q[u].z), qp);
                            s2[u] = dot3(dv, dv);
                            inside[u] = s2[u] < m2_lo;
                            unsure = unsure || (!inside[u] && !(s2[u] > m2_hi));
                        }
                        if (__ballot(unsure)) {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (!inside[u] && !(s2[u] > m2_hi)) inside[u] = sqrt(s2[u]) < maxDist;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (inside[u] && k + u < se) sum += q[u].w;
                    }
                    if (sum != 0) lc_total_add(W, own, sum);
                }
            }
        }
        LC_PUSH_ALL(false)
        LCC(t_sum += wall_clock64() - t_trip); LCC(++n_sum);
        unsigned long long mask = __ballot(pending);
        if (mask) {
#ifdef RT_COUNT
            const unsigned long long t_c0 = wall_clock64();
            ++t_phases;
#endif
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int b = __shfl(pend_b, l), e = __shfl(pend_e, l);
                const int o = __shfl(own, l);
                const lightcam::Point *pts = W.points[o];
                const d3 cp = mk3(W.px[o], W.py[o], W.pz[o]);
                const double qmax = W.maxd[o], q2 = qmax * qmax, qlo = q2 * (1.0 - 0x1p-48), qhi = q2 * (1.0 + 0x1p-48);
                double part = 0;
                for (int k = b + lane; k < e; k += 64) {
                    const lightcam::Point *q = &pts[k];
                    const d3 dv = sub3(mk3(q->x, q->y, q->z), cp);
                    const double s2 = dot3(dv, dv);
                    bool inside = s2 < qlo;
                    const bool unsure = !inside && !(s2 > qhi);
                    if (__ballot(unsure)) { if (unsure) inside = sqrt(s2) < qmax; }
                    if (inside) part += q->w;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                if (lane == 0 && part != 0) lc_total_add(W, o, part);
            }
#ifdef RT_COUNT
            t_coop += wall_clock64() - t_c0;
#endif
        }
    }
#undef LC_PUSH_ALL
#ifdef RT_COUNT
    {
        // per wave: wall-clock ticks in the query, in its cooperative phases, its trips
        if (lane == 0) {
            atomicAdd(&g_walk_counts[32], 1ull);
            atomicAdd(&g_walk_counts[33], wall_clock64() - t_begin);
            atomicAdd(&g_walk_counts[34], t_coop);
            atomicAdd(&g_walk_counts[35], t_iters);
            atomicAdd(&g_walk_counts[36], t_phases);
            atomicMax(&g_walk_counts[37], wall_clock64() - t_begin);
            atomicMax(&g_walk_counts[38], t_iters);
            atomicAdd(&g_walk_counts[30], t_fill);
            atomicMax(&g_walk_counts[31], t_pool);
            atomicAdd(&g_walk_counts[41], t_exp); atomicAdd(&g_walk_counts[42], n_exp);      // ticks and number of the trips of either kind
            atomicAdd(&g_walk_counts[43], t_sum); atomicAdd(&g_walk_counts[44], n_sum);
        }
    }
    {
        unsigned long long v[6] = {active ? 1ull : 0ull, c_desc, c_term, c_sum, c_pts + c_cooppts, c_coop};
        for (int k = 0; k < 6; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0 && x) atomicAdd(&g_walk_counts[24 + k], x);
        }
    }
#endif
#undef LCC
    const double total = W.total[lane];
    const double expected = (my_maxDist * my_maxDist) / (cam.spw * cam.spw);
    return total / expected;
}
DEV double lc_calc_intensity_wave(const DevLightCam *table, int cam_index, d3 pos, bool active, uint32_t *overflow, int pool_cap)
{
    return lc_query_wave(table, cam_index, pos, active, overflow, pool_cap);
}
