// Queries over the flattened octree in the reference's own visiting order (depth-first pre-order with
// skip links, so no stack): closest hit (Octree.h:996-1062) and the order-dependent shadow query
// checkOccluded3 (Octree.h:1198-1264).  Used whenever the equivalent hierarchy may not be (transparent
// spheres, spheres outside the root cube, RT_FLAG_NO_CULL) and by the light camera's photon pass.
#pragma once
#include "primitives.hpp"

// Closest hit over the flattened octree (Octree.h:996-1062).  `best` is a triangle slot,
// a sphere slot | SLOT_SPHERE, or -1.  With `cull`, a node whose bounding sphere is entered
// beyond the current best hit is pruned; the nearest hit is unchanged by that.
DEV void trace_closest(const DevScene &S, d3 o, d3 d, bool cull, double minT, double &best_t, int &best)
{
    const double maxT = 10000000000.0;
    const double A = dot3(d, d);
    best_t = maxT;
    best = -1;
    int idx = 0;
    const int end = S.n_nodes;
    int tcur = 0, tend = 0, scur = 0, send = 0;
    while (true) {
        // find the next node with resident primitives
        while (tcur >= tend && scur >= send && idx < end) {
            const rt_node *nd = &S.nodes[idx];
            double entry;
            double ret = bound_test(nd, o, d, A, minT, maxT, &entry);
            bool pass = !(ret >= maxT);
            if (pass && cull && entry > best_t * (1.0 + 1e-12) + 1e-12) pass = false;
            if (pass) {
                tcur = nd->tri_begin; tend = tcur + nd->tri_count;
                scur = nd->sph_begin; send = scur + nd->sph_count;
                idx = idx + 1;
            } else {
                idx = nd->skip;
            }
        }
        if (tcur >= tend && scur >= send) break;
        while (tcur < tend) {
            double t = tri_test_fast(load_tri(&S.tri[tcur]), o, d, minT, maxT, best_t);
            if (t < best_t) { best_t = t; best = tcur; }
            ++tcur;
        }
        while (scur < send) {
            double t = sph_test(&S.sph[scur], o, d, A, minT, maxT);
            if (t < best_t) { best_t = t; best = scur | SLOT_SPHERE; }
            ++scur;
        }
    }
}

// checkOccluded3 (Octree.h:1198-1264) over the flattened octree.
// Returns occluded; *only_transparent and *closest as the reference leaves them.
// The reference returns from a subtree as soon as that subtree's own node saw any
// occluder, skipping every later sibling; in pre-order that is "shrink the end of the
// walk to this node's skip link".
DEV bool trace_shadow(const DevScene &S, d3 o, d3 d, int parent, double maxT, bool cull, bool *only_transparent,
                      int *closest)
{
    const double minT = .01, sentinel = 10000000.0;
    const double A = dot3(d, d);
    bool seen = false;
    *only_transparent = true;
    *closest = -1;
    int idx = 0;
    int end = S.n_nodes;
    while (idx < end) {
        const rt_node *nd = &S.nodes[idx];
        double entry;
        double ret = bound_test(nd, o, d, A, minT, sentinel, &entry);
        bool pass = !(ret >= sentinel);
        if (pass && cull && entry > maxT * (1.0 + 1e-12) + 1e-12) pass = false;
        if (!pass) { idx = nd->skip; continue; }
        bool anything = false;
        double furthest = minT;
        const int tb = nd->tri_begin, te = tb + nd->tri_count;
        for (int s = tb; s < te; ++s) {
            const TriOps tr = load_tri(&S.tri[s]);
            if (tr.prim == parent) continue;
            double t = tri_test_fast(tr, o, d, minT, maxT, maxT);
            if (t < maxT) {   // a triangle always ends the query (Octree.h:1241-1251)
                if (t > furthest) *closest = tr.prim;
                *only_transparent = false;
                return true;
            }
        }
        const int sb = nd->sph_begin, se = sb + nd->sph_count;
        for (int s = sb; s < se; ++s) {
            uint32_t tag = S.sph_prim[s];
            int prim = (int)(tag & ~RT_SPH_TRANSPARENT);
            if (prim == parent) continue;
            double t = sph_test(&S.sph[s], o, d, A, minT, maxT);
            if (t < maxT) {
                anything = true;
                if (t > furthest) { furthest = t; *closest = prim; }
                if (!(tag & RT_SPH_TRANSPARENT)) { *only_transparent = false; return true; }
            }
        }
        if (anything) { seen = true; end = nd->skip; }
        idx = idx + 1;
    }
    return seen;
}

