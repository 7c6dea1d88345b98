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
// COOP: every lane of the wave calls it together (`active` lanes hold a query); point ranges
// longer than 24 are then scanned by all 64 lanes.  Without COOP a lane works alone.
// `overflow` is set to 1 if a lane's stack of pending cells ever fills up (7 siblings per level, 224
// entries: a tree more than 32 levels deep), so that a sum with a cell missing is never returned silently.
#define LC_TERMINAL 0x40000000   // stack code: sum this cell's points within range
#define LC_EXPAND 0x20000000     // stack code: a large, near cell whose children are to be judged
#ifndef LC_LDS_STACK
#define LC_LDS_STACK 32
#endif
#ifndef LC_OPEN_MIN
#define LC_OPEN_MIN 2        // cells with more points than this are classified against the query sphere
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

template <bool COOP>
DEV double lc_query(const DevLightCam *table, int cam_index, d3 pos, bool active, uint32_t *overflow)
{
    const int lane = (int)(threadIdx.x & 63);
    const DevLightCam cam = table[active ? cam_index : 0];
    const d3 lp = ld3(cam.light_pos), spos = ld3(cam.sphere_pos);
    const double dist = norm3(sub3(pos, lp));
    const double sphereDistToLight = norm3(sub3(lp, spos));
    double maxDist = cam.spw * (dist / sphereDistToLight);
    maxDist *= 1.5;                                  // smoothingFactor
    // sqrt is monotonic and correctly rounded: |v| < maxDist is decided on the squared length
    // unless it lies within a few ulps of maxDist^2, where the reference's own comparison runs
    const double m2 = maxDist * maxDist, m2_lo = m2 * (1.0 - 0x1p-48), m2_hi = m2 * (1.0 + 0x1p-48);
    const double m2_in = m2 * (1.0 - 1e-9), m2_out = m2 * (1.0 + 1e-9);   // cell classification, with a margin
    double total = 0;
#ifdef RT_COUNT
    unsigned long long c_desc = 0, c_term = 0, c_sum = 0, c_pts = 0, c_coop = 0, c_cooppts = 0;
    unsigned long long t_iters = 0, t_coop = 0, t_phases = 0;
    const unsigned long long t_begin = wall_clock64();
#define LCC(x) (x)
#else
#define LCC(x) ((void)0)
#endif
    // depth-first, children 0..7; at most 7 pending siblings per level.  The first LC_LDS_STACK
    // entries of a lane's stack live in LDS (entry k of a thread at [k*256 + thread]), deeper ones,
    // which are rare, in private memory: a pop is then not one more trip to memory in a chain of
    // dependent fetches.
    __shared__ int lds_stack[COOP ? LC_LDS_STACK * 256 : 1];
    int deep[192];
    int sp_ = 0;
#define LC_PUSH(v)                                                      \
    do {                                                                \
        if (COOP && sp_ < LC_LDS_STACK) lds_stack[sp_ * 256 + (int)threadIdx.x] = (v); \
        else deep[COOP ? sp_ - LC_LDS_STACK : sp_] = (v);               \
        ++sp_;                                                          \
    } while (0)
#define LC_CAP (COOP ? 192 + LC_LDS_STACK : 192)
    if (active && cam.n_nodes > 0) LC_PUSH(cam.root);
    int pend_b = 0, pend_e = 0;
    while (true) {
        bool pending = false;
        while (sp_ > 0 && !pending) {
            LCC(++t_iters);
            --sp_;
            const int code = (COOP && sp_ < LC_LDS_STACK) ? lds_stack[sp_ * 256 + (int)threadIdx.x]
                                                          : deep[COOP ? sp_ - LC_LDS_STACK : sp_];
            // the whole 96-byte cell record in six 16-byte loads issued together: fields fetched one by one
            // where they are used would each cost a trip to memory of their own
            const lightcam::Node nd_ = load_lc_node(&cam.nodes[code & ~(LC_TERMINAL | LC_EXPAND)]);
            const lightcam::Node *r = &nd_;
            bool terminal = (code & LC_TERMINAL) != 0;
            LCC(terminal ? ++c_term : ++c_desc);
            if (!terminal) {
                const double epsilon = maxDist * 0.1;
                bool expand = (code & LC_EXPAND) != 0;
                if (!expand) {       // only the root arrives undecided
                    if (r->size < maxDist * 2 || r->leaf) terminal = true;
                    else if (norm3(sub3(pos, mk3(r->cx, r->cy, r->cz))) < (r->size + epsilon)) expand = true;
                    else continue;
                }
                if (expand) {
                    // The children's cubes follow from this node's cube by the builder's own arithmetic
                    // (Octree.h:1430-1465), so each child is judged here, before it is fetched: a small
                    // or `leaf` child is summed, a large one is descended into only if it is near.
                    const double size = r->size, hh = size / 2;
                    for (int c = 7; c >= 0; --c) {
                        const int ch = r->child[c];
                        if (ch < 0) continue;
                        if (sp_ >= LC_CAP) { *overflow = 1u; continue; }   // a cell left out: the frame is reported incomplete
                        const double lox = (c & 1) ? r->cx : r->cx - size / 2;
                        const double loy = (c & 2) ? r->cy : r->cy - size / 2;
                        const double loz = (c & 4) ? r->cz : r->cz - size / 2;
                        const double csize = (lox + hh) - lox;
                        if (csize < maxDist * 2 || ((r->child_leaf >> c) & 1)) LC_PUSH(ch | LC_TERMINAL);
                        else {
                            // |pos - centre| < csize + epsilon, decided on the squares unless they are a few ulps apart
                            const d3 dc = sub3(pos, mk3(lox + csize / 2, loy + csize / 2, loz + csize / 2));
                            const double s2 = dot3(dc, dc), lim = csize + epsilon, lim2 = lim * lim;
                            bool near = s2 < lim2 * (1.0 - 0x1p-48);
                            if (!near && !(s2 > lim2 * (1.0 + 0x1p-48))) near = sqrt(s2) < lim;
                            if (near) LC_PUSH(ch | LC_EXPAND);
                        }
                    }
                    continue;
                }
            }
            // every point of this cell that lies within maxDist counts
            int sb = r->begin, se = r->end;
            if (se - sb > LC_OPEN_MIN) {
                const double h = r->size * 0.5;
                const double ax = fabs(pos.x - r->cx), ay = fabs(pos.y - r->cy), az = fabs(pos.z - r->cz);
                const double nx = fmax(ax - h, 0.0), ny = fmax(ay - h, 0.0), nz = fmax(az - h, 0.0);
                const double fx = ax + h, fy = ay + h, fz = az + h;
                const double dmin2 = (nx * nx + ny * ny) + nz * nz, dmax2 = (fx * fx + fy * fy) + fz * fz;
                if (dmin2 > m2_out) continue;
                if (dmax2 < m2_in) { total += r->wsum; LCC(++c_sum); continue; }
                if (r->rest_begin > sb && sp_ + 8 <= LC_CAP) {   // has children: open it, its own leftover points are scanned
                    for (int c = 7; c >= 0; --c)
                        if (r->child[c] >= 0) LC_PUSH(r->child[c] | LC_TERMINAL);
                    sb = r->rest_begin;
                }
            }
            if (COOP && se - sb > 24) {
                pend_b = sb; pend_e = se; pending = true;
                LCC(++c_coop); LCC(c_cooppts += (unsigned long long)(se - sb));
            } else {
                LCC(c_pts += (unsigned long long)(se - sb));
                // four points per trip, fetched together: one lane alone would otherwise wait out a
                // full memory latency for every single point
                for (int k = sb; k < se; k += 4) {
                    lightcam::Point q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = cam.points[k + u < se ? k + u : se - 1];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const d3 dv = sub3(mk3(q[u].x, q[u].y, q[u].z), pos);
                        const double s2 = dot3(dv, dv);
                        bool inside = s2 < m2_lo;
                        if (!inside && !(s2 > m2_hi)) inside = sqrt(s2) < maxDist;
                        if (inside && k + u < se) total += q[u].w;
                    }
                }
            }
        }
        if (!COOP) break;
        unsigned long long mask = __ballot(pending);
        if (mask == 0) break;     // every lane has emptied its stack
#ifdef RT_COUNT
        const unsigned long long t_c0 = wall_clock64();
        ++t_phases;
#endif
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int b = __shfl(pend_b, l), e = __shfl(pend_e, l);
            const int ci = __shfl(cam_index, l);
            const lightcam::Point *pts = table[ci].points;
            const d3 qp = mk3(__shfl(pos.x, l), __shfl(pos.y, l), __shfl(pos.z, l));
            const double qmax = __shfl(maxDist, l), qlo = __shfl(m2_lo, l), qhi = __shfl(m2_hi, l);
            double part = 0;
            for (int k = b + lane; k < e; k += 64) {
                const lightcam::Point *q = &pts[k];
                const d3 dv = sub3(mk3(q->x, q->y, q->z), qp);
                const double s2 = dot3(dv, dv);
                bool inside = s2 < qlo;
                if (!inside && !(s2 > qhi)) inside = sqrt(s2) < qmax;
                if (inside) part += q->w;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == l) total += part;
        }
#ifdef RT_COUNT
        t_coop += wall_clock64() - t_c0;
#endif
    }
#ifdef RT_COUNT
    if (COOP) {
        // per wave: wall-clock ticks in the query, in its cooperative phases, the trips of its slowest lane
        unsigned long long it = t_iters;
        for (int off = 32; off > 0; off >>= 1) { unsigned long long o = __shfl_xor(it, off); it = o > it ? o : it; }
        if (lane == 0) {
            atomicAdd(&g_walk_counts[32], 1ull);
            atomicAdd(&g_walk_counts[33], wall_clock64() - t_begin);
            atomicAdd(&g_walk_counts[34], t_coop);
            atomicAdd(&g_walk_counts[35], it);
            atomicAdd(&g_walk_counts[36], t_phases);
            atomicMax(&g_walk_counts[37], wall_clock64() - t_begin);
            atomicMax(&g_walk_counts[38], it);
        }
    }
    {
        unsigned long long v[6] = {active ? 1ull : 0ull, c_desc, c_term, c_sum, c_pts + c_cooppts, c_coop};
        for (int k = 0; k < 6; ++k) {
            unsigned long long x = v[k], mx = v[k];
            for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off); unsigned long long o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
            if (lane == 0 && x) atomicAdd(&g_walk_counts[24 + k], x);
            if (lane == 0 && k == 2) atomicMax(&g_walk_counts[31], mx);
            if (lane == 0 && k == 4) atomicMax(&g_walk_counts[30], mx);
        }
    }
#endif
#undef LCC
#undef LC_PUSH
#undef LC_CAP
    const double expected = (maxDist * maxDist) / (cam.spw * cam.spw);
    return total / expected;
}
DEV double lc_calc_intensity_wave(const DevLightCam *table, int cam_index, d3 pos, bool active, uint32_t *overflow)
{
    return lc_query<true>(table, cam_index, pos, active, overflow);
}


