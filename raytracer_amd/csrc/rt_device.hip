// MI355X (gfx950) render path: wavefront pipeline of HIP kernels behind the C ABI
// declared in include/raytracer_amd.h.
//
// Pipeline per batch of primary samples (one pass = one ray generation):
//   k_closest<PRIMARY>  ray generation (Camera.h:112-230) fused with closest-hit traversal
//                       (Octree.h:996-1062, Triangle.h:91-158, Sphere.h:35-77, 155-204),
//                       hit record (Solid.h:110-125) and secondary-ray spawning
//                       (Camera.h:232-314); pushes hit records and next-generation rays
//                       with one wave-aggregated atomic per queue.
//   k_shadow            one work item per (hit, light): shadow traversal
//                       (Octree.h:1198-1264), Phong terms (Solid.h:128-172,
//                       Camera.h:365-488), f64 atomic accumulation into the frame.
//   k_closest<QUEUE>    the same traversal for generations 1..5, rays read from the queue.
//   k_finalize          /spp, clamp, *255, (int) truncation (Camera.h:217-227, Driver.cpp:722,758-760).
//
// The octree arrives flattened in depth-first pre-order with skip links, so the
// traversal carries no stack: "descend" is index+1 and "prune" is node.skip; the
// visit order is exactly the reference's recursion order (children 0..7).
//
// Arithmetic is FP64 in the reference's evaluation order; this file is compiled with
// -ffp-contract=off so that every +,-,*,/ and sqrt is the same IEEE operation the CPU
// performs.  Only pow() (specular term) and the order in which a pixel's contributions
// are summed can differ from a CPU evaluation.
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "accel_build.hpp"
#include "lightcam_build.hpp"

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string g_error;
int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}
}  // namespace

extern "C" void rt_set_error_string(const char *msg) { g_error = msg ? msg : ""; }
extern "C" const char *rt_last_error(void) { return g_error.c_str(); }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// ------------------------------------------------------------------ device structs
struct alignas(16) RayRec {      // 80 bytes
    double o[3];
    double d[3];
    double A;                    // product of T along the chain of refraction calls
    double coef;                 // coefOfReflection carried by the call (Camera.h:241,305)
    uint32_t pixel;              // local pixel index
    uint32_t meta;               // bits 0..7: remaining loop iterations, bit 8: first ray of a call
    uint32_t _pad[2];
};

struct alignas(16) HitRec {      // 128 bytes
    double P[3];                 // impactPos
    double n[3];                 // hr.normal normalised a second time, as computeLocalColor does (Camera.h:382)
    double nn[3];                // ... and a third time, as computeSpecularComp does (Solid.h:153)
    double v[3];                 // unit vector from the hit point to the ray origin (Solid.h:156)
    double weight;               // factor applied to the local colour
    uint32_t pixel;
    int32_t prim;                // the solid that was hit (never occludes itself)
    int32_t light, cam;          // only in the deferred light-camera queue: light index, camera index
    int32_t mat;                 // material index of `prim`
    int32_t _pad;
};
static_assert(sizeof(HitRec) == 128, "HitRec must be 128 bytes");

struct DevScene {
    const rt_node *nodes;
    const rt_tri_rec *tri;
    const rt_sph_rec *sph;
    const uint32_t *sph_prim;
    const uint32_t *prim_flags;
    const int32_t *prim_material;
    const double *prim_normals;
    const rt_material *materials;
    const rt_light *lights;
    int32_t n_nodes, n_lights;
    double bg[3];
    // derived hierarchies (accel_build.hpp); leaf records share the layout of the octree slots
    const accel::Bvh4Node *tri_nodes;
    const rt_tri_rec *btri;
    const accel::Bvh4Node *sph_nodes;
    const rt_sph_rec *bsph;
    const uint32_t *bsph_prim;
    const int32_t *bsph_key;
    int32_t n_tri_nodes, n_sph_nodes;
    int32_t tri_cached, sph_cached;       // the first so many nodes of each hierarchy are copied into LDS by every block
    int32_t lds_cache_word;               // first LDS word of the node cache (after the traversal stacks)
    int32_t mat_cached, light_cached;     // the first so many materials / lights are copied into LDS as well
    int32_t lds_table_word;               // first LDS word of those tables (materials, then lights; 16 words each)
    float bvh_bound;                      // largest |coordinate| of any hierarchy box
    // light cameras (LightCam.h): one per (transparent sphere, light) pair, index rank*n_lights+light
    const int32_t *prim_lc_rank;          // per primitive: rank among transparent spheres, or -1
    const struct DevLightCam *lc_table;   // null until the cameras are built
};

struct DevLightCam {
    const lightcam::Node *nodes;
    const lightcam::Point *points;
    int32_t n_nodes, n_points;
    double spw;                           // sub-pixel width of the light camera (LightCam.h:223)
    double light_pos[3], sphere_pos[3];
};

struct FrameParams {
    DevScene S;
    rt_camera cam;
    int32_t N;                   // samples per axis
    double aperture;
    uint32_t flags, seed;
    int32_t band_rows, n_parts, part, local_rows;
    int32_t blocks_x;            // 8x8-pixel blocks per row of blocks (primary sample ordering)
    int32_t cull_closest, cull_shadow;
    double *accum;               // local_rows*resx*3
    int32_t *hit_id;             // local_rows*resx or null
    uint8_t *lc_mask;            // local_rows*resx or null
};

// Queues are cut into N_SHARDS independent shards (a block pushes into shard blockIdx % N_SHARDS),
// so that no single counter word serialises the pushes of the whole chip; a consumer walks the
// shards' 64-ray chunks as one list.
#define N_SHARDS 64
#define CNT_RAYS(g) ((g) * N_SHARDS)             // g = 0..6
#define CNT_HITS(g) ((7 + (g)) * N_SHARDS)       // g = 0..5
#define CNT_DEFER(g) ((13 + (g)) * N_SHARDS)     // g = 0..5: deferred light-camera evaluations
#define CNT_OVERFLOW (19 * N_SHARDS)
#define CNT_WORDS (19 * N_SHARDS + 16)

struct QueueSet {
    RayRec *rays_in;
    RayRec *rays_out;
    HitRec *hits;
    HitRec *deferred;            // light-camera evaluations postponed by the shadow pass (same shard size as hits)
    uint32_t *counts;
    uint32_t in_shard_cap, out_shard_cap, hit_shard_cap;   // records per shard
    int32_t generation;
    int64_t chunk_begin, chunk_end;  // PRIMARY only: 64-sample chunks of the primary ordering
};

// ------------------------------------------------------------------ device math
#define DEV __device__ __forceinline__

struct d3 { double x, y, z; };
DEV d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV d3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }
DEV d3 sub3(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV d3 add3(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV d3 neg3(d3 a) { return mk3(-a.x, -a.y, -a.z); }
DEV d3 scale3(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
DEV d3 div3(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
DEV double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV d3 unit3(d3 a)
{
    double z = dot3(a, a);
    if (z > 0) { double s = sqrt(z); return div3(a, s); }
    return a;
}
DEV double norm3(d3 a) { return sqrt(dot3(a, a)); }

// det of the matrix with columns a, b, c in the reference's association (Triangle.h:388-393)
DEV double det3(d3 a, d3 b, d3 c)
{
    return a.x * (b.y * c.z - c.y * b.z) + b.x * (c.y * a.z - a.y * c.z) + c.x * (a.y * b.z - b.y * a.z);
}

// Bounding-sphere test of a node (Sphere.h:155-204).  Returns the value the reference
// returns (t2, t1 or `sentinel`).  The reference takes sqrt(B*B-4AC) without halving it, so
// its roots are wider than the sphere's; *entry receives the true distance at which the
// ray enters the sphere, (-b - root/2)/A, which is what pruning may compare against.
DEV double bound_test(const rt_node *nd, d3 o, d3 d, double A, double minT, double sentinel, double *entry)
{
    d3 emp = sub3(o, ld3(nd->center));
    double r = nd->radius;
    double C = dot3(emp, emp) - r * r;
    double B = 2.0 * dot3(d, emp);
    double disc = B * B - 4 * A * C;
    *entry = sentinel;
    if (disc < 0) return sentinel;
    double b = B / 2;
    double root = sqrt(disc);
    double t1 = (-b + root) / A;
    double t2 = (-b - root) / A;
    *entry = (-b - 0.5 * root) / A;
    if (t2 > minT) return t2;
    else if (t1 > minT) return t1;
    return sentinel;
}

// Triangle::intersect (Triangle.h:91-158) without the hit-record update; returns t or maxT.
DEV double tri_test(const rt_tri_rec *tr, d3 o, d3 d, double minT, double maxT, double *alpha_out, double *beta_out)
{
    d3 v0 = ld3(tr->v0), ab = ld3(tr->ab), ac = ld3(tr->ac);
    d3 ae = sub3(v0, o);
    double detA = det3(ab, ac, d);
    double alpha = det3(ae, ac, d) / detA;
    if (alpha < 0 || alpha > 1) return maxT;
    double beta = det3(ab, ae, d) / detA;
    if (beta < 0 || beta > 1 - alpha) return maxT;
    double t = det3(ab, ac, ae) / detA;
    if (t < minT || t > maxT) return maxT;
    *alpha_out = alpha;
    *beta_out = beta;
    return t;
}

// Operands of one triangle test, fetched as five 16-byte loads (the first 80 bytes of the record:
// v0, ab, ac, primitive id, tie key).  The memory pipeline spends the same time on a one-word load
// of a wave as on a four-word one, so the number of load instructions is what counts.
struct TriOps {
    d3 v0, ab, ac;
    int prim, key;
};
DEV TriOps load_tri(const rt_tri_rec *tr)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const __attribute__((address_space(1))) v2d *p = (const __attribute__((address_space(1))) v2d *)tr;
    const v2d a = p[0], b = p[1], c = p[2], e = p[3], f = p[4];
    TriOps t;
    t.v0 = mk3(a.x, a.y, b.x);
    t.ab = mk3(b.y, c.x, c.y);
    t.ac = mk3(e.x, e.y, f.x);
    t.prim = __double2loint(f.y);
    t.key = __double2hiint(f.y);
    return t;
}

// The same test with its rejections decided before any division.  The four determinants are
// built from shared 2x2 minors in the reference's association, so detA, X, Y, Z are bit-identical
// to det3(); a miss is declared early only when the quotient the reference would form is beyond
// its threshold by more than the rounding of one division (factor 1 +- 2^-40), otherwise the exact
// sequence above runs.  `tcut` <= maxT: a hit farther than tcut may be reported as maxT (it
// cannot replace the current nearest hit, and for shadow queries tcut == maxT).
DEV double tri_test_fast(const TriOps &tr, d3 o, d3 d, double minT, double maxT, double tcut)
{
    const d3 v0 = tr.v0, ab = tr.ab, ac = tr.ac;
    const double m3 = ab.y * ac.z - ac.y * ab.z;
    const d3 ae = sub3(v0, o);
    const double m1 = ac.y * d.z - d.y * ac.z;
    const double m2 = d.y * ab.z - ab.y * d.z;
    const double detA = ab.x * m1 + ac.x * m2 + d.x * m3;
    const double m4 = d.y * ae.z - ae.y * d.z;
    const double m5 = ae.y * ac.z - ac.y * ae.z;
    const double m6 = ab.y * ae.z - ae.y * ab.z;
    const double X = ae.x * m1 + ac.x * m4 + d.x * m5;          // det3(ae, ac, d)
    const double Y = ab.x * (-m4) + ae.x * m2 + d.x * m6;       // det3(ab, ae, d)
    const double Z = ab.x * (-m5) + ac.x * (-m6) + ae.x * m3;   // det3(ab, ac, ae)
    const double ad = fabs(detA);
    if (ad > 1e-150 && ad < 1e150) {
        const double sg = detA < 0 ? -1.0 : 1.0;
        const double sX = sg * X, sY = sg * Y, sZ = sg * Z;
        const double up = ad * (1.0 + 0x1p-40);
        bool miss = (sX < -1e-150) || (sX > up) || (sY < -1e-150) || (sX + sY > ad * (1.0 + 0x1p-30)) ||
                    (sZ < (minT * ad) * (1.0 - 0x1p-40)) || (sZ > (tcut * ad) * (1.0 + 0x1p-40));
        if (miss && fabs(X) < 1e150 && fabs(Y) < 1e150 && fabs(Z) < 1e150) return maxT;
    }
    double alpha = X / detA;
    if (alpha < 0 || alpha > 1) return maxT;
    double beta = Y / detA;
    if (beta < 0 || beta > 1 - alpha) return maxT;
    double t = Z / detA;
    if (t < minT || t > maxT) return maxT;
    return t;
}

// Sphere::intersect (Sphere.h:35-77) without the hit-record update; returns t or maxT.
DEV double sph_test(const rt_sph_rec *sp, d3 o, d3 d, double d2, double minT, double maxT)
{
    d3 emc = sub3(o, ld3(sp->center));
    double r = sp->radius;
    double ddotemc = dot3(d, emc);
    double disc = (ddotemc * ddotemc) - d2 * (dot3(emc, emc) - r * r);
    if (disc < 0) return maxT;
    double sq = sqrt(disc);
    double root1 = (-ddotemc + sq) / d2;
    double root2 = (-ddotemc - sq) / d2;
    double t = root1;
    if (root1 < 0 || (root2 > minT && root2 < root1)) t = root2;
    if (t < minT || t > maxT) return maxT;
    return t;
}

#define SLOT_SPHERE 0x40000000

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

// ------------------------------------------------------------------ traversal statistics (diagnostic builds only)
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

// ------------------------------------------------------------------ hierarchy traversal
// Stack of pending nodes in LDS: entry k of a lane lives at stack[k * 256 + lane-in-block], so a
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
    if (with_nodes) {
        const int words = (S.tri_cached + S.sph_cached) * 32;
        const int *tsrc = reinterpret_cast<const int *>(S.tri_nodes);
        const int *ssrc = reinterpret_cast<const int *>(S.sph_nodes);
        for (int i = (int)threadIdx.x; i < words; i += (int)blockDim.x) {
            const int node = i >> 5, w = i & 31;
            const int v = node < S.tri_cached ? tsrc[i] : ssrc[i - S.tri_cached * 32];
            s_lds[S.lds_cache_word + node * NODE_LDS_WORDS + w] = v;
        }
    }
    // material and light tables (64 bytes per entry): every hit reads one of each, always the same few
    const int mwords = S.mat_cached * 16, lwords = S.light_cached * 16;
    const int *msrc = reinterpret_cast<const int *>(S.materials);
    const int *lsrc = reinterpret_cast<const int *>(S.lights);
    for (int i = (int)threadIdx.x; i < mwords + lwords; i += (int)blockDim.x)
        s_lds[S.lds_table_word + i] = i < mwords ? msrc[i] : lsrc[i - mwords];
    __syncthreads();
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

// Single-precision view of a ray for the slab tests.  A child box passes when the ray's
// parameter interval inside it, [tn, tf] clipped to [0, limit], is not empty.  Per axis
// t = plane * (1/d) - origin * (1/d), one fused multiply-add per plane; the near plane of an axis
// is the lower bound when the direction is positive, else the upper bound, so the node is read
// through a per-ray byte offset and no min/max per axis is needed.
//
// The test must never reject a box the exact ray touches.  The boxes are already rounded outward.
// Spatial errors (the origin converted to float, the product origin * (1/d) -- each at most
// 2^-24 (|o| + pad) once divided by |1/d|) are covered by moving the origin by `pad`, eight times
// that, away from each plane: lower bounds are taken against o + pad, upper bounds against
// o - pad, which is the box grown by pad.  Relative errors (1/d: conversion and division; the
// rounding of the fused operation) are a few 2^-24 of t and are covered by the factor 1 + 2^-17
// on tf in the comparison.
struct RayBox4 {
    float ix, iy, iz;        // 1 / direction
    float cnx, cny, cnz;     // (origin moved away from the near plane) * (1 / direction)
    float cfx, cfy, cfz;     // (origin moved away from the far plane) * (1 / direction)
    uint32_t sx, sy, sz;     // byte offset of the near planes of an axis in a node: 0 (lower) or 16 (upper) + 32 * axis
};
DEV float inv_dir(double d)
{
    // an (almost) axis-parallel ray is nudged for the box test only: 0 * inf would be NaN
    const float df = (float)d;
    const float dd = fabsf(df) < 1e-25f ? (d < 0 ? -1e-25f : 1e-25f) : df;
    return __builtin_amdgcn_rcpf(dd);   // one ulp: inside the relative slack of the comparison
}
DEV RayBox4 make_raybox4(d3 o, d3 d, float bound)
{
    RayBox4 r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float omax = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    const float pad = 0x1p-21f * (omax + bound);
    r.ix = inv_dir(d.x); r.iy = inv_dir(d.y); r.iz = inv_dir(d.z);
    const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
    r.sx = nx ? 16u : 0u; r.sy = ny ? 48u : 32u; r.sz = nz ? 80u : 64u;
    // lower bounds pair with o + pad, upper bounds with o - pad
    r.cnx = (nx ? ox - pad : ox + pad) * r.ix; r.cfx = (nx ? ox + pad : ox - pad) * r.ix;
    r.cny = (ny ? oy - pad : oy + pad) * r.iy; r.cfy = (ny ? oy + pad : oy - pad) * r.iy;
    r.cnz = (nz ? oz - pad : oz + pad) * r.iz; r.cfz = (nz ? oz + pad : oz - pad) * r.iz;
    if (!(omax < 1e12f)) {
        // an origin this far away would overflow the products: let every box pass instead
        r.ix = r.iy = r.iz = 0.0f;
        r.cnx = r.cny = r.cnz = 1.0f;
        r.cfx = r.cfy = r.cfz = -3.0e38f;
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
DEV void box4_test(const accel::Bvh4Node *nodes, int cached, int lds_word, int cur, const RayBox4 &r, float limitf,
                   float k[4], int c[4])
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
        __attribute__((address_space(3))) const int *lw = (__attribute__((address_space(3))) const int *)s_lds + (lds_word + cur * NODE_LDS_WORDS);
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
        k[i] = (code.m != BVH_NONE && tn <= tf * (1.0f + 0x1p-17f)) ? tn : inf;                                    \
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
DEV int bvh4_step(const accel::Bvh4Node *nodes, int cached, int lds_word, int cur, const RayBox4 &rb, float limitf,
                  bool ordered, int *stack, int &sp)
{
    float kk[4];
    int cc[4];
    box4_test(nodes, cached, lds_word, cur, rb, limitf, kk, cc);
    float k0 = kk[0], k1 = kk[1], k2 = kk[2], k3 = kk[3];
    int c0 = cc[0], c1 = cc[1], c2 = cc[2], c3 = cc[3];
    const float inf = __builtin_inff();
    if (ordered) {
        BVH_SORT2(k0, c0, k1, c1);
        BVH_SORT2(k2, c2, k3, c3);
        BVH_SORT2(k0, c0, k2, c2);
        BVH_SORT2(k1, c1, k3, c3);
        BVH_SORT2(k1, c1, k2, c2);
        if (k3 < inf) { stack[sp * 256] = c3; ++sp; }
        if (k2 < inf) { stack[sp * 256] = c2; ++sp; }
        if (k1 < inf) { stack[sp * 256] = c1; ++sp; }
        if (k0 < inf) return c0;
    } else {
        // any order: the last hit child is kept in hand, the others go to the stack
        int next = BVH_NONE;
        if (k0 < inf) next = c0;
        if (k1 < inf) { if (next != BVH_NONE) { stack[sp * 256] = next; ++sp; } next = c1; }
        if (k2 < inf) { if (next != BVH_NONE) { stack[sp * 256] = next; ++sp; } next = c2; }
        if (k3 < inf) { if (next != BVH_NONE) { stack[sp * 256] = next; ++sp; } next = c3; }
        if (next != BVH_NONE) return next;
    }
    if (sp == 0) return BVH_NONE;
    --sp;
    return stack[sp * 256];
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
    int cur = 0, sp = 0;
    while (true) {
        const float limitf = limit_f32(best.t * (1.0 + 1e-12) + 1e-12);
        WC_WAVE(wc, wave_rounds);
        while (cur >= 0) {
            WC_ADD(wc, lane_nodes);
            WC_WAVE(wc, wave_nodes);
            cur = bvh4_step(nodes, cached, lds_word, cur, rb, limitf, true, stack, sp);
        }
        if (cur == BVH_NONE) break;
        WC_ADD(wc, lane_leaves);
        const int b = leaf_begin(cur), e = b + leaf_count(cur);
        for (int k = b; k < e; ++k) {
            double t;
            int key, prim = -1;
            WC_ADD(wc, lane_prims);
            WC_WAVE(wc, wave_prims);
            if (TRI) { const TriOps tr = load_tri(&S.btri[k]); t = tri_test_fast(tr, o, d, minT, maxT, best.t); key = tr.key; prim = tr.prim; }
            else { t = sph_test(&S.bsph[k], o, d, d2, minT, maxT); key = S.bsph_key[k]; }
            if (t < best.t || (t == best.t && t < maxT && key < best.key)) {
                best.t = t; best.key = key; best.slot = TRI ? k : (k | SLOT_SPHERE);
                best.prim = TRI ? prim : (int)(S.bsph_prim[k] & ~RT_SPH_TRANSPARENT);
            }
        }
        if (sp == 0) break;
        --sp;
        cur = stack[sp * 256];
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
    const float limitf = limit_f32(maxT * (1.0 + 1e-12) + 1e-12);
    int cur = 0, sp = 0;
    while (true) {
        WC_WAVE(wc, wave_rounds);
        while (cur >= 0) {
            WC_ADD(wc, lane_nodes);
            WC_WAVE(wc, wave_nodes);
            cur = bvh4_step(nodes, cached, lds_word, cur, rb, limitf, false, stack, sp);
        }
        if (cur == BVH_NONE) break;
        WC_ADD(wc, lane_leaves);
        const int b = leaf_begin(cur), e = b + leaf_count(cur);
        for (int k = b; k < e; ++k) {
            double t;
            WC_ADD(wc, lane_prims);
            WC_WAVE(wc, wave_prims);
            if (TRI) {
                const TriOps tr = load_tri(&S.btri[k]);
                if (tr.prim == parent) continue;
                t = tri_test_fast(tr, o, d, minT, maxT, maxT);
            } else {
                if ((int)(S.bsph_prim[k] & ~RT_SPH_TRANSPARENT) == parent) continue;
                t = sph_test(&S.bsph[k], o, d, d2, minT, maxT);
            }
            if (t < maxT) return true;
        }
        if (sp == 0) break;
        --sp;
        cur = stack[sp * 256];
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

// Geometry of a found hit (Solid.h:110-125): primitive id, unit normal, impact point.
// `best` is a slot of the given record arrays (| SLOT_SPHERE for spheres).
DEV void hit_geometry(const DevScene &S, const rt_tri_rec *tri_arr, const rt_sph_rec *sph_arr, const uint32_t *sph_prim_arr,
                      uint32_t flags, d3 o, d3 d, double minT, double t, int best, int known_prim, int *prim_out, d3 *n_out,
                      d3 *P_out)
{
    d3 nraw;
    int prim = known_prim;       // >= 0: the walk already knows it, no need to go through the record
    if (best & SLOT_SPHERE) {
        const int s = best & ~SLOT_SPHERE;
        if (known_prim < 0) prim = (int)(sph_prim_arr[s] & ~RT_SPH_TRANSPARENT);
        const rt_sph_rec *sp = &sph_arr[s];
        nraw = div3(sub3(add3(o, scale3(t, d)), ld3(sp->center)), sp->radius);
    } else {
        const rt_tri_rec *tr = &tri_arr[best];
        if (known_prim < 0) prim = tr->prim;
        const double *pn = &S.prim_normals[(size_t)prim * 12];
        nraw = ld3(pn);
        if ((flags & RT_FLAG_PHONG_NORMALS) && (S.prim_flags[prim] & RT_PRIM_HAS_VNORMALS)) {
            double al, be;
            (void)tri_test(tr, o, d, minT, 10000000000.0, &al, &be);
            d3 n0 = ld3(pn + 3), n1 = ld3(pn + 6), n2 = ld3(pn + 9);
            nraw = add3(add3(scale3(al, n1), scale3(be, n2)), scale3(1 - (al + be), n0));
        }
    }
    *prim_out = prim;
    *n_out = unit3(nraw);
    *P_out = mk3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
}

// Refraction / total-internal-reflection direction at a transparent hit (Camera.h:260-302,
// LightCam.h:397-472).  `renormalize` is the light camera's variant (LightCam.h:440).
DEV d3 refract_dir(d3 d, d3 n, double ior, bool renormalize)
{
    d3 nn = n;
    d3 V = neg3(d);
    double dt = dot3(nn, V);
    if (dt >= 0) ior = 1.0 / ior;
    else { nn = neg3(nn); dt = -dt; }
    double disc = 1 - (1 - (dt * dt)) * (ior * ior);
    if (disc > 0) {
        double sq = sqrt(disc);
        d3 rd = mk3((nn.x * dt - V.x) * ior - nn.x * sq, (nn.y * dt - V.y) * ior - nn.y * sq,
                    (nn.z * dt - V.z) * ior - nn.z * sq);
        return renormalize ? unit3(rd) : rd;
    }
    return unit3(add3(d, scale3(2.0 * -dot3(d, nn), nn)));
}

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
DEV double lc_query(const DevLightCam *table, int cam_index, d3 pos, bool active)
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
    if (active && cam.n_nodes > 0) LC_PUSH(0);
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
                        if (ch < 0 || sp_ >= LC_CAP) continue;
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
DEV double lc_calc_intensity(const DevLightCam *table, int cam_index, d3 pos) { return lc_query<false>(table, cam_index, pos, true); }
DEV double lc_calc_intensity_wave(const DevLightCam *table, int cam_index, d3 pos, bool active)
{
    return lc_query<true>(table, cam_index, pos, active);
}


// Counter-based stand-in for the reference's shared rand() % 1000 / 1000 (Camera.h:99-103).
DEV uint32_t hash_rng(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    uint32_t h = seed * 0x9E3779B1u + 0x85EBCA6Bu;
    h ^= pixel + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= sample + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= dim + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
DEV double normed_rand(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    return (double)(hash_rng(pixel, sample, dim, seed) % 1000u) / 1000.0;
}

DEV int global_row(const FrameParams &P, int local_row)
{
    int band = local_row / P.band_rows;
    int within = local_row - band * P.band_rows;
    return (band * P.n_parts + P.part) * P.band_rows + within;
}

// Wave-aggregated queue push into this block's shard: one atomic per wave, lanes take
// consecutive slots.  Returns the record index in the whole queue.
DEV uint32_t wave_reserve(bool want, uint32_t *shard_counters, uint32_t shard_cap, uint32_t *overflow, bool *ok)
{
    const uint32_t shard = blockIdx.x & (N_SHARDS - 1);
    unsigned long long mask = __ballot(want);
    uint32_t n = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (n) {
        int leader = __ffsll((long long)mask) - 1;
        int lane = (int)(threadIdx.x & 63);
        if (lane == leader) base = atomicAdd(&shard_counters[shard], n);
        base = (uint32_t)__shfl((int)base, leader);
    }
    uint32_t rank = (uint32_t)__popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
    uint32_t slot = base + rank;
    *ok = want && slot < shard_cap;
    if (want && slot >= shard_cap) *overflow = 1u;
    return shard * shard_cap + slot;
}

// View of a sharded queue as one list of 64-item chunks.  Lane l keeps shard l's item count and
// the index of its first chunk; chunk c belongs to the last shard whose first chunk is <= c.
struct ShardMap {
    uint32_t items;      // this lane's shard: number of items
    uint32_t first;      // this lane's shard: first chunk index
    uint32_t total;      // chunks in all shards
};
DEV ShardMap shard_map(const uint32_t *shard_counters, uint32_t shard_cap, uint32_t items_per_record)
{
    const int lane = (int)(threadIdx.x & 63);
    ShardMap m;
    uint32_t n = shard_counters[lane];
    if (n > shard_cap) n = shard_cap;
    m.items = n * items_per_record;
    uint32_t chunks = (m.items + 63u) >> 6;
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    m.first = incl - chunks;
    m.total = (uint32_t)__shfl((int)incl, 63);
    return m;
}
// For chunk c (same in all lanes): shard, and this lane's item index inside the shard (or -1).
DEV int shard_locate(const ShardMap &m, uint32_t c, int *shard_out, int pos_in_chunk = -1)
{
    const unsigned long long le = __ballot(m.first <= c && m.items > 0);
    const int shard = 63 - __clzll((long long)le);
    const uint32_t first = (uint32_t)__shfl((int)m.first, shard);
    const uint32_t items = (uint32_t)__shfl((int)m.items, shard);
    const uint32_t idx = (c - first) * 64u + (uint32_t)(pos_in_chunk >= 0 ? pos_in_chunk : (int)(threadIdx.x & 63));
    *shard_out = shard;
    return idx < items ? (int)idx : -1;
}

// Adds w*c to a pixel.  Lanes that hold the same pixel in a contiguous run of lanes are summed
// first and the run's first lane issues the (memory-side) FP64 atomics.  `in_queue` lanes take
// part in runs even when they contribute nothing (`valid` false), so a shadowed light does not
// split the run of its hit.
DEV void accum_add_runs(double *accum, bool in_queue, bool valid, uint32_t pixel, double w, d3 c)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t key = in_queue ? pixel : 0xFFFFFFFFu;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool head = (lane == 0) || prev != key;
    // run id = number of run heads at or before this lane: equal ids <=> same contiguous run
    const unsigned long long heads = __ballot(head);
    const int run = __popcll(heads & ((2ull << lane) - 1ull));
    double vx = valid ? w * c.x : 0.0, vy = valid ? w * c.y : 0.0, vz = valid ? w * c.z : 0.0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int orun = __shfl_down(run, off);
        const bool same = lane + off < 64 && orun == run;
        if (__ballot(same) == 0) break;      // runs are contiguous: no partner at this distance, none farther
        const double ox = __shfl_down(vx, off), oy = __shfl_down(vy, off), oz = __shfl_down(vz, off);
        if (same) { vx += ox; vy += oy; vz += oz; }
    }
    if (in_queue && head && (vx != 0.0 || vy != 0.0 || vz != 0.0)) {
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 0], vx);
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 1], vy);
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 2], vz);
    }
}

// ------------------------------------------------------------------ kernels
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif
template <bool PRIMARY, bool BVH>
__global__ __launch_bounds__(256, RT_MIN_WAVES) void k_closest(FrameParams P, QueueSet Q)
{
    // traversal stack, sized at launch to the hierarchies' depth (entry k of a thread at [k*256 + thread])
    int *stack = &s_lds[BVH ? threadIdx.x : 0];
    fill_node_cache(P.S, BVH);
    const int lane = (int)(threadIdx.x & 63);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    ShardMap smap;
    smap.items = smap.first = smap.total = 0;
    if (!PRIMARY) smap = shard_map(&Q.counts[CNT_RAYS(Q.generation)], Q.in_shard_cap, 1u);
    const long long n_chunks = PRIMARY ? (Q.chunk_end - Q.chunk_begin) : (long long)smap.total;
    const bool cull = P.cull_closest != 0;
    WalkCount wc{};

    for (long long chunk = wave; chunk < n_chunks; chunk += n_waves) {
        bool active;
        d3 o, d;
        double A = 1.0, coef = 1.0;
        uint32_t pixel = 0, meta = 5u | 0x100u;
        bool first_sample = false;
        if (PRIMARY) {
            // Primary ordering: pixels in 8x8 blocks (row-major inside a block), the N*N samples of a
            // pixel in consecutive positions; a wave takes 64 consecutive positions.
            const int N = P.N;
            const long long pos = (Q.chunk_begin + chunk) * 64 + lane;
            const int spp = N * N;
            const long long pseq = pos / spp;
            const int smp = (int)(pos - pseq * spp);
            const long long blk = pseq >> 6;
            const int within = (int)(pseq & 63);
            const int by = (int)(blk / P.blocks_x), bx = (int)(blk - (long long)by * P.blocks_x);
            const int j = bx * 8 + (within & 7), lrow = by * 8 + (within >> 3);
            const int si = smp / N, sj = smp - si * N;
            active = j < P.cam.resx && lrow < P.local_rows;
            const int i = global_row(P, lrow);
            pixel = (uint32_t)lrow * (uint32_t)P.cam.resx + (uint32_t)j;
            first_sample = (si == 0 && sj == 0);
            const rt_camera &c = P.cam;
            const uint32_t gpix = (uint32_t)i * (uint32_t)c.resx + (uint32_t)j;
            const uint32_t sample = (uint32_t)(si * N + sj);
            double p, q, r;
            if (N > 1) {
                double spw = c.pix_width / N;
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
            d3 u = ld3(c.u), v = ld3(c.v), w = ld3(c.w), e = ld3(c.e);
            d = unit3(add3(add3(scale3(p, u), scale3(q, v)), scale3(r, w)));
            o = e;
            if (P.aperture > 0) {
                d3 uOff = scale3(P.aperture, scale3(normed_rand(gpix, sample, 2, P.seed) - .5, u));
                d3 vOff = scale3(P.aperture, scale3(normed_rand(gpix, sample, 3, P.seed) - .5, v));
                o = add3(o, add3(uOff, vOff));
                p = c.l + j * c.pix_width;
                q = c.t - i * c.pix_width;
                d3 focus = add3(add3(add3(scale3(p, u), scale3(q, v)), scale3(r, w)), e);
                d = unit3(sub3(focus, o));
            }
        } else {
            int shard;
            const int local = shard_locate(smap, (uint32_t)chunk, &shard);
            active = local >= 0;
            const RayRec *rr = &Q.rays_in[active ? (size_t)shard * Q.in_shard_cap + (uint32_t)local : 0];
            o = ld3(rr->o);
            d = ld3(rr->d);
            A = rr->A;
            coef = rr->coef;
            pixel = rr->pixel;
            meta = rr->meta;
        }

        double best_t = 0;
        int best = -1, best_prim = -1;
        if (active) {
            if (BVH) {
                const double d2 = dot3(d, d);
                Best bst;
                bst.t = 10000000000.0; bst.key = 0x7fffffff; bst.slot = -1; bst.prim = -1;
                if (root_gate(P.S, o, d, d2, 10000000000.0)) {
                    const RayBox4 rb = make_raybox4(o, d, P.S.bvh_bound);
                    WC_ADD(wc, rays);
                    bvh_closest<true>(P.S, o, d, d2, rb, bst, stack, wc);
                    bvh_closest<false>(P.S, o, d, d2, rb, bst, stack, wc);
                }
                best_t = bst.t;
                best = bst.slot;
                best_prim = bst.prim;
            } else {
                trace_closest(P.S, o, d, cull, .01, best_t, best);
            }
        }
        const rt_tri_rec *tri_arr = BVH ? P.S.btri : P.S.tri;
        const rt_sph_rec *sph_arr = BVH ? P.S.bsph : P.S.sph;
        const uint32_t *sph_prim_arr = BVH ? P.S.bsph_prim : P.S.sph_prim;

        const bool first = (meta & 0x100u) != 0;
        const int L = (int)(meta & 0xffu);
        const double wloc = first ? A : A * coef;
        bool hit = active && best >= 0;
        accum_add_runs(P.accum, active, active && !hit, pixel, wloc, ld3(P.S.bg));

        // hit record (Solid.h:110-125)
        d3 n = mk3(0, 0, 0), Pt = mk3(0, 0, 0), refl = mk3(0, 0, 0);
        int prim = -1, mat = 0;
        rt_material m;
        m.Ks = 0; m.T = 0; m.ior = 1;
        if (hit) {
            hit_geometry(P.S, tri_arr, sph_arr, sph_prim_arr, P.flags, o, d, .01, best_t, best, best_prim, &prim, &n, &Pt);
            refl = unit3(add3(d, scale3(2.0 * -dot3(d, n), n)));
            mat = P.S.prim_material[prim];
            m = load_material(P.S, mat, true);
        }
        if (PRIMARY && active && first_sample && P.hit_id) P.hit_id[pixel] = prim;

        // Queue pushes.  The three reservations (hit record, refraction ray, reflection ray) are made
        // with two atomics -- both kinds of ray go to the same queue -- issued back to back by one lane,
        // so a wave waits for one round trip to the counters instead of three in a row.
        const bool want_hit = hit && P.S.n_lights > 0;
        const bool spawn = hit && L > 0 && !(coef < .01);       // Camera.h:244-307
        const double coef2 = coef * m.Ks;
        const bool want_refr = spawn && m.T > 0;
        const bool want_refl = spawn && (coef2 != 0.0 || (P.flags & RT_FLAG_KEEP_ZERO_WEIGHT));
        const unsigned long long mask_h = __ballot(want_hit), mask_a = __ballot(want_refr), mask_b = __ballot(want_refl);
        const uint32_t n_h = (uint32_t)__popcll(mask_h), n_a = (uint32_t)__popcll(mask_a), n_b = (uint32_t)__popcll(mask_b);
        const uint32_t shard = blockIdx.x & (N_SHARDS - 1);
        uint32_t base_h = 0, base_r = 0;
        if (lane == 0) {
            if (n_h) base_h = atomicAdd(&Q.counts[CNT_HITS(Q.generation) + shard], n_h);
            if (n_a + n_b) base_r = atomicAdd(&Q.counts[CNT_RAYS(Q.generation + 1) + shard], n_a + n_b);
        }
        base_h = (uint32_t)__shfl((int)base_h, 0);
        base_r = (uint32_t)__shfl((int)base_r, 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t slot_h = base_h + (uint32_t)__popcll(mask_h & below);
        const uint32_t slot_a = base_r + (uint32_t)__popcll(mask_a & below);
        const uint32_t slot_b = base_r + n_a + (uint32_t)__popcll(mask_b & below);
        if ((want_hit && slot_h >= Q.hit_shard_cap) || (want_refr && slot_a >= Q.out_shard_cap) ||
            (want_refl && slot_b >= Q.out_shard_cap))
            Q.counts[CNT_OVERFLOW] = 1u;

        if (want_hit && slot_h < Q.hit_shard_cap) {
            HitRec *h = &Q.hits[(size_t)shard * Q.hit_shard_cap + slot_h];
            h->P[0] = Pt.x; h->P[1] = Pt.y; h->P[2] = Pt.z;
            const d3 n2 = unit3(n);
            const d3 vv = unit3(sub3(o, Pt));
            const d3 n3 = unit3(n2);
            h->n[0] = n2.x; h->n[1] = n2.y; h->n[2] = n2.z;
            h->nn[0] = n3.x; h->nn[1] = n3.y; h->nn[2] = n3.z;
            h->v[0] = vv.x; h->v[1] = vv.y; h->v[2] = vv.z;
            h->weight = wloc;
            h->pixel = pixel;
            h->prim = prim;
            h->mat = mat;
        }
        if (want_refr && slot_a < Q.out_shard_cap) {
            const d3 rd = refract_dir(d, n, m.ior, false);
            RayRec *r = &Q.rays_out[(size_t)shard * Q.out_shard_cap + slot_a];
            r->o[0] = Pt.x; r->o[1] = Pt.y; r->o[2] = Pt.z;
            r->d[0] = rd.x; r->d[1] = rd.y; r->d[2] = rd.z;
            r->A = A * m.T;
            r->coef = m.T * coef;
            r->pixel = pixel;
            r->meta = (uint32_t)(L - 1) | 0x100u;
        }
        if (want_refl && slot_b < Q.out_shard_cap) {
            RayRec *r = &Q.rays_out[(size_t)shard * Q.out_shard_cap + slot_b];
            r->o[0] = Pt.x; r->o[1] = Pt.y; r->o[2] = Pt.z;
            r->d[0] = refl.x; r->d[1] = refl.y; r->d[2] = refl.z;
            r->A = A;
            r->coef = coef2;
            r->pixel = pixel;
            r->meta = (uint32_t)(L - 1);
        }
    }
#ifdef RT_COUNT
    walk_count_flush(wc, PRIMARY ? 0 : 8);
#endif
}


// Diffuse + specular terms of one light at one hit (Solid.h:128-172), scaled by `kfac`
// (Camera.h:465-476).  `rd` is the unit vector from the hit point to the light.
DEV d3 phong_terms(const rt_material &m, const HitRec *hr, const rt_light *lt, d3 rd, double kfac)
{
    const d3 normal = ld3(hr->n);
    d3 local = mk3(0, 0, 0);
    if (m.Ks > 0) {
        d3 nn = ld3(hr->nn);
        d3 v = ld3(hr->v);
        d3 rdir = unit3(add3(neg3(rd), scale3(2.0 * dot3(rd, nn), nn)));
        double a = dot3(rdir, v);
        double pw = pow((a < 0.0) ? 0.0 : a, m.Shine);
        double ik = lt->intensity * m.Ks;
        d3 comp = mk3((ik * lt->color[0]) * pw, (ik * lt->color[1]) * pw, (ik * lt->color[2]) * pw);
        local = add3(local, scale3(kfac, comp));
    }
    {
        double a = dot3(rd, normal);
        double dm = (a < 0.0) ? 0.0 : a;
        double ik = lt->intensity * m.Kd;
        d3 comp = mk3((ik * m.r) * dm, (ik * m.g) * dm, (ik * m.b) * dm);
        local = add3(local, scale3(kfac, comp));
    }
    return local;
}

#ifndef RT_MIN_WAVES_SHADOW
#define RT_MIN_WAVES_SHADOW 4
#endif
template <bool BVH>
__global__ __launch_bounds__(256, RT_MIN_WAVES_SHADOW) void k_shadow(FrameParams P, QueueSet Q)
{
    // traversal stack, sized at launch to the hierarchies' depth (entry k of a thread at [k*256 + thread])
    int *stack = &s_lds[BVH ? threadIdx.x : 0];
    fill_node_cache(P.S, BVH);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const uint32_t nl = (uint32_t)P.S.n_lights;
    // work items are (hit, light) pairs, the lights of one hit in consecutive items
    const ShardMap smap = shard_map(&Q.counts[CNT_HITS(Q.generation)], Q.hit_shard_cap, nl);
    const bool cull = P.cull_shadow != 0;
    WalkCount wc{};
    for (long long chunk = wave; chunk < (long long)smap.total; chunk += n_waves) {
        int shard;
        const int item = shard_locate(smap, (uint32_t)chunk, &shard);
        const bool active = item >= 0;
        const uint32_t h = active ? (uint32_t)item / nl : 0u;
        const int li = active ? (int)((uint32_t)item - h * nl) : 0;
        const HitRec *hr = &Q.hits[(size_t)shard * Q.hit_shard_cap + h];
        const d3 Pt = ld3(hr->P);
        const d3 lpos = ld3(load_light(P.S, li, true).pos);
        const uint32_t pixel = hr->pixel;
        bool lit = false;
        int defer_cam = -1;
        d3 local = mk3(0, 0, 0);
        if (active) {
            const d3 toL = sub3(lpos, Pt);
            const d3 rd = unit3(toL);
            const double maxT = norm3(toL) - .01;
            bool only = false;
            int closest = -1;
            bool occluded;
            if (BVH) {
                // no transparent sphere in the scene: "occluded" is "any hit", and never "only transparent"
                const double d2 = dot3(rd, rd);
                occluded = false;
#ifdef RT_EXP_SHADOW_SKIP          /* experiment: cost of the kernel without the traversal */
                if (false) {
#else
                if (root_gate(P.S, Pt, rd, d2, 10000000.0)) {
#endif
                    const RayBox4 rb = make_raybox4(Pt, rd, P.S.bvh_bound);
                    WC_ADD(wc, rays);
                    occluded = bvh_any<true>(P.S, Pt, rd, d2, rb, hr->prim, maxT, stack, wc) ||
                               bvh_any<false>(P.S, Pt, rd, d2, rb, hr->prim, maxT, stack, wc);
                }
            } else {
                occluded = trace_shadow(P.S, Pt, rd, hr->prim, maxT, cull, &only, &closest);
            }
            if (occluded && only) {
                // shadowed by transparent spheres only: the light camera of (that sphere, this light)
                // decides how much light arrives (Camera.h:406-451); evaluated in k_lightcam
                if (P.lc_mask) P.lc_mask[pixel] = 1;
                if (P.S.lc_table && closest >= 0) {
                    const int rank = P.S.prim_lc_rank[closest];
                    if (rank >= 0) defer_cam = rank * P.S.n_lights + li;
                }
            }
            lit = !occluded;
            if (lit) {
                const rt_light lgt = load_light(P.S, li, true);     // read again here rather than kept in registers through the walk
                local = phong_terms(load_material(P.S, hr->mat, true), hr, &lgt, rd, .999999);
            }
        }
        if (!BVH && P.S.lc_table) {
            bool ok;
            const bool want = defer_cam >= 0;
            const uint32_t slot = wave_reserve(want, &Q.counts[CNT_DEFER(Q.generation)], Q.hit_shard_cap, &Q.counts[CNT_OVERFLOW + 1], &ok);
            if (ok) {
                HitRec *dr = &Q.deferred[slot];
                *dr = *hr;
                dr->light = li;
                dr->cam = defer_cam;
            } else if (want) {   // queue full: evaluate here
                const double kfac = lc_calc_intensity(P.S.lc_table, defer_cam, Pt);
                if (kfac > .01) {
                    lit = true;
                    const rt_light lgt = load_light(P.S, li, true);
                    local = phong_terms(load_material(P.S, hr->mat, true), hr, &lgt, unit3(sub3(lpos, Pt)), kfac);
                }
            }
        }
        accum_add_runs(P.accum, active, lit, pixel, hr->weight, local);
    }
#ifdef RT_COUNT
    walk_count_flush(wc, 16);
#endif
}

// Deferred light-camera evaluations of one generation: density query, then the same shading as a
// lit point with the query result as the factor (Camera.h:444-476).
// LC_PER_WAVE < 64 spreads a generation's queries over more waves (measured: no gain, the time of a
// pass is the chain of dependent fetches of its single longest query).
#ifndef LC_PER_WAVE
#define LC_PER_WAVE 64
#endif
__global__ __launch_bounds__(256) void k_lightcam(FrameParams P, QueueSet Q)
{
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    const int lane = (int)(threadIdx.x & 63);
    const int parts = 64 / LC_PER_WAVE;
    const ShardMap smap = shard_map(&Q.counts[CNT_DEFER(Q.generation)], Q.hit_shard_cap, 1u);
    for (long long vc = wave; vc < (long long)smap.total * parts; vc += n_waves) {
        const long long chunk = vc / parts;
        const int part = (int)(vc - chunk * parts);
        int shard;
        const int item = shard_locate(smap, (uint32_t)chunk, &shard, part * LC_PER_WAVE + (lane % LC_PER_WAVE));
        const bool active = item >= 0 && lane < LC_PER_WAVE;
        const HitRec *hr = &Q.deferred[(size_t)shard * Q.hit_shard_cap + (active ? (uint32_t)item : 0u)];
        bool lit = false;
        d3 local = mk3(0, 0, 0);
        const d3 Pt = ld3(hr->P);
        const double kfac = lc_calc_intensity_wave(P.S.lc_table, active ? hr->cam : 0, Pt, active);
        if (active) {
            const rt_light *lt = &P.S.lights[hr->light];
            if (kfac > .01) {
                lit = true;
                local = phong_terms(load_material(P.S, hr->mat, false), hr, lt, unit3(sub3(ld3(lt->pos), Pt)), kfac);
            }
        }
        accum_add_runs(P.accum, active, lit, hr->pixel, hr->weight, local);
    }
}

// Light rays of one light camera (LightCam::lightCast, recursiveCastLightRays, castLightRay:
// LightCam.h:207-306, 340-556).  One lane per sub-pixel (si, sj) of the N x N grid; each lane follows
// its refraction / total-internal-reflection chain and writes every impact into its own row of
// `slots` (21 entries: the first hit plus a depth budget of 20), so the host can read them back in
// the reference's order.
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
__global__ void k_collect_counts(const uint32_t *counts, unsigned long long *totals, int n_lights)
{
    const int lane = (int)threadIdx.x;   // one wave, lane = shard
    if (blockIdx.x != 0 || lane >= N_SHARDS) return;
    for (int g = 0; g < 6; ++g) {
        unsigned long long r = (g == 0) ? 0ull : counts[CNT_RAYS(g) + lane];
        unsigned long long h = counts[CNT_HITS(g) + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off);
            h += __shfl_down(h, off);
        }
        if (lane == 0) {
            totals[2 + g] += r;
            totals[8 + g] += h;
            totals[0] += r;
            totals[1] += h * (unsigned long long)n_lights;
        }
    }
    if (lane == 0) totals[15] += counts[CNT_OVERFLOW];
}

__global__ __launch_bounds__(256) void k_finalize(const double *accum, long long n_values, int spp, double *out_f64,
                                                  uint8_t *out_u8)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) {
        double c = accum[i] / (double)spp;
        c = (c < 1.0) ? c : 1.0;        // std::min(1.0, c)
        c = (0.0 < c) ? c : 0.0;        // std::max(0.0, c)
        double v = c * 255;
        if (out_f64) out_f64[i] = v;
        if (out_u8) out_u8[i] = (uint8_t)(int)v;
    }
}

// ------------------------------------------------------------------ host side
struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // scene
    DevScene S{};
    std::vector<void *> scene_allocs;
    bool has_scene = false;
    bool any_transparent = false;          // any material with T > 0
    bool any_transparent_sphere = false;
    bool spheres_inside_root = true;       // every sphere lies inside the root cube
    bool accel_usable = false;             // hierarchies built and shallow enough for the LDS stack
    int accel_depth = 0;                   // deepest hierarchy, in nodes: the stack entries a ray can need
    // host copies needed to set up the light cameras
    std::vector<rt_light> h_lights;
    std::vector<std::array<double, 4>> h_lc_spheres;   // centre + radius of every transparent sphere, by rank
    std::vector<void *> lc_allocs;
    bool lc_built = false;
    uint32_t lc_built_flags = 0;
    unsigned long long lc_rays = 0;
    int32_t lightcam_samples = 150;
    // frame resources (grow-only)
    double *accum = nullptr; size_t accum_values = 0;
    RayRec *rays[2] = {nullptr, nullptr}; size_t ray_capacity[2] = {0, 0};
    HitRec *hits[4] = {nullptr, nullptr, nullptr, nullptr}; size_t hit_capacity[4] = {0, 0, 0, 0};
    HitRec *deferred = nullptr; size_t deferred_capacity = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_c[6] = {}, ev_s[6] = {}, ev_batch[2] = {};
    uint32_t *counts = nullptr;
    unsigned long long *totals = nullptr;
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr}; size_t stage_bytes[4] = {0, 0, 0, 0};
    int64_t batch_override = 0;
    // pending async frame
    bool pending = false;
    uint32_t pending_flags = 0;
    unsigned long long pending_primaries = 0;
    int pending_batches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_closest, ev_shadow;
    size_t ev_closest_used = 0, ev_shadow_used = 0;
    int cu_count = 256;
    int32_t n_materials = 0;
    size_t lds_per_cu = 64 * 1024;   // LDS of one compute unit (maxSharedMemoryPerMultiProcessor)
};

static void free_lightcams(rt_ctx *ctx)
{
    for (void *p : ctx->lc_allocs) (void)hipFree(p);
    ctx->lc_allocs.clear();
    ctx->lc_built = false;
    ctx->S.lc_table = nullptr;
    ctx->lc_rays = 0;
}

static int free_scene(rt_ctx *ctx)
{
    free_lightcams(ctx);
    for (void *p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->has_scene = false;
    return RT_OK;
}

template <typename T>
static int upload(rt_ctx *ctx, const T *src, size_t n, const T **dst)
{
    *dst = nullptr;
    size_t bytes = sizeof(T) * (n ? n : 1);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    ctx->scene_allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, sizeof(T) * n, hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return RT_OK;
}

extern "C" int rt_device_count(int *count)
{
    if (!count) return fail(RT_ERR_INVALID, "rt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

extern "C" int rt_create(int device_id, rt_ctx **out)
{
    if (!out) return fail(RT_ERR_INVALID, "rt_create: null argument");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(RT_ERR_DEVICE, "rt_create: no HIP device visible; this library has no CPU path");
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID, "rt_create: device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (int k = 0; k < 6; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_c[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_s[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_batch[k], hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&ctx->ev_begin));
    HIP_TRY(hipEventCreate(&ctx->ev_end));
    HIP_TRY(hipMalloc((void **)&ctx->counts, 2 * CNT_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->totals, 16 * sizeof(unsigned long long)));
    if (getenv("RT_DEBUG")) {
        int n = 0;
        for (size_t lds : {(size_t)0, (size_t)12 * 1024, (size_t)32 * 1024}) {
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_closest<true, true>, 256, lds);
            fprintf(stderr, "[rt] k_closest<primary,bvh> lds=%zu: %d blocks/CU\n", lds, n);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_shadow<true>, 256, lds);
            fprintf(stderr, "[rt] k_shadow<bvh> lds=%zu: %d blocks/CU\n", lds, n);
        }
        fprintf(stderr, "[rt] CUs=%d LDS/CU=%zu sharedMemPerBlock=%zu\n", ctx->cu_count, ctx->lds_per_cu, (size_t)prop.sharedMemPerBlock);
    }
    *out = ctx;
    return RT_OK;
}

extern "C" int rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    free_scene(ctx);
    (void)hipFree(ctx->accum);
    (void)hipFree(ctx->rays[0]);
    (void)hipFree(ctx->rays[1]);
    for (HitRec *h : ctx->hits) (void)hipFree(h);
    (void)hipFree(ctx->deferred);
    for (int k = 0; k < 6; ++k) { (void)hipEventDestroy(ctx->ev_c[k]); (void)hipEventDestroy(ctx->ev_s[k]); }
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ctx->ev_batch[k]);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipFree(ctx->counts);
    (void)hipFree(ctx->totals);
    for (void *p : ctx->stage) (void)hipFree(p);
    for (auto &pr : ctx->ev_closest) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : ctx->ev_shadow) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    (void)hipEventDestroy(ctx->ev_begin);
    (void)hipEventDestroy(ctx->ev_end);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

// One primitive test per thread with the device functions the kernels use (rt_debug_intersect).
__global__ void k_debug_intersect(int kind, int n, const double *geom, const double *rays, double minT, double maxT, double *out)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const d3 o = ld3(rays + (size_t)i * 6), d = ld3(rays + (size_t)i * 6 + 3);
    double t = maxT;
    d3 nrm = mk3(0, 0, 0);
    if (kind == 0 || kind == 1) {
        const double *g = geom + (size_t)i * 9;
        rt_tri_rec tr;
        for (int k = 0; k < 3; ++k) { tr.v0[k] = g[k]; tr.ab[k] = g[k] - g[3 + k]; tr.ac[k] = g[k] - g[6 + k]; }   // Triangle.h:40-46
        tr.prim = 0; tr._pad0 = 0; tr._pad1[0] = tr._pad1[1] = 0;
        if (kind == 0) { double al, be; t = tri_test(&tr, o, d, minT, maxT, &al, &be); }
        else {
            TriOps ops;
            ops.v0 = ld3(tr.v0); ops.ab = ld3(tr.ab); ops.ac = ld3(tr.ac); ops.prim = 0; ops.key = 0;
            t = tri_test_fast(ops, o, d, minT, maxT, maxT);
        }
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

extern "C" int rt_debug_intersect(rt_ctx *ctx, int32_t kind, int32_t n, const double *geom, const double *rays, double minT,
                                  double maxT, double *out)
{
    if (!ctx || !geom || !rays || !out || n < 0 || kind < 0 || kind > 3) return fail(RT_ERR_INVALID, "rt_debug_intersect: bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t gdoubles = (size_t)n * (kind <= 1 ? 9 : 4);
    double *d_geom = nullptr, *d_rays = nullptr, *d_out = nullptr;
    int rc = RT_OK;
    if (hipMalloc((void **)&d_geom, gdoubles * 8) != hipSuccess || hipMalloc((void **)&d_rays, (size_t)n * 48) != hipSuccess ||
        hipMalloc((void **)&d_out, (size_t)n * 32) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: hipMalloc failed");
    if (rc == RT_OK && (hipMemcpy(d_geom, geom, gdoubles * 8, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(d_rays, rays, (size_t)n * 48, hipMemcpyHostToDevice) != hipSuccess))
        rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: upload failed");
    if (rc == RT_OK) {
        hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)kind, (int)n,
                           d_geom, d_rays, minT, maxT, d_out);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(out, d_out, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: kernel or download failed");
    }
    (void)hipFree(d_geom); (void)hipFree(d_rays); (void)hipFree(d_out);
    return rc;
}

#ifdef RT_COUNT
// Diagnostic builds: totals since the last call: 3 groups (primary, queued closest, shadow) x 8 words,
// then 8 words of the light-camera queries.
extern "C" int rt_debug_counters(unsigned long long *out48)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_walk_counts), 48 * sizeof(unsigned long long)));
    unsigned long long zero[48] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_counts), zero, sizeof zero));
    return RT_OK;
}
#endif

extern "C" int rt_set_batch(rt_ctx *ctx, int64_t primaries)
{
    if (!ctx || primaries < 0) return fail(RT_ERR_INVALID, "rt_set_batch: bad argument");
    ctx->batch_override = primaries;
    return RT_OK;
}

// Validates every index the kernels will follow, so that a malformed scene is an error
// code here and never an out-of-bounds access on the GPU.
static int validate_scene(const rt_scene_desc *s)
{
    if (!s->nodes || s->n_nodes <= 0) return fail(RT_ERR_INVALID, "scene: no nodes");
    if (s->n_tri_slots < 0 || s->n_sph_slots < 0 || s->n_prims < 0 || s->n_materials < 0 || s->n_lights < 0)
        return fail(RT_ERR_INVALID, "scene: negative count");
    if ((s->n_tri_slots && !s->tri) || (s->n_sph_slots && (!s->sph || !s->sph_prim)))
        return fail(RT_ERR_INVALID, "scene: missing primitive records");
    if (s->n_prims && (!s->prim_flags || !s->prim_material || !s->prim_normals))
        return fail(RT_ERR_INVALID, "scene: missing per-primitive arrays");
    if ((s->n_materials && !s->materials) || (s->n_lights && !s->lights)) return fail(RT_ERR_INVALID, "scene: missing materials or lights");
    for (int32_t i = 0; i < s->n_nodes; ++i) {
        const rt_node &n = s->nodes[i];
        if (n.skip <= i || n.skip > s->n_nodes) return fail(RT_ERR_INVALID, "scene: node skip link out of order");
        if (n.tri_count < 0 || n.tri_begin < 0 || (int64_t)n.tri_begin + n.tri_count > s->n_tri_slots)
            return fail(RT_ERR_INVALID, "scene: node triangle range out of bounds");
        if (n.sph_count < 0 || n.sph_begin < 0 || (int64_t)n.sph_begin + n.sph_count > s->n_sph_slots)
            return fail(RT_ERR_INVALID, "scene: node sphere range out of bounds");
    }
    for (int32_t i = 0; i < s->n_tri_slots; ++i)
        if (s->tri[i].prim < 0 || s->tri[i].prim >= s->n_prims || (s->prim_flags[s->tri[i].prim] & RT_PRIM_SPHERE))
            return fail(RT_ERR_INVALID, "scene: triangle slot names a bad primitive");
    for (int32_t i = 0; i < s->n_sph_slots; ++i) {
        uint32_t p = s->sph_prim[i] & ~RT_SPH_TRANSPARENT;
        if (p >= (uint32_t)s->n_prims || !(s->prim_flags[p] & RT_PRIM_SPHERE))
            return fail(RT_ERR_INVALID, "scene: sphere slot names a bad primitive");
    }
    for (int32_t i = 0; i < s->n_prims; ++i)
        if (s->prim_material[i] < 0 || s->prim_material[i] >= s->n_materials)
            return fail(RT_ERR_INVALID, "scene: primitive material out of range");
    return RT_OK;
}

extern "C" int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *s)
{
    if (!ctx || !s) return fail(RT_ERR_INVALID, "rt_upload_scene: null argument");
    int rc = validate_scene(s);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    DevScene D{};
    if ((rc = upload(ctx, s->nodes, (size_t)s->n_nodes, &D.nodes)) != RT_OK) return rc;
    accel::Accel acc;
    std::vector<rt_tri_rec> octree_tri;
    std::vector<int32_t> octree_sph_key;
    accel::build(*s, acc, octree_tri, octree_sph_key);
    if ((rc = upload(ctx, octree_tri.data(), octree_tri.size(), &D.tri)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.tri_bvh4.nodes.data(), acc.tri_bvh4.nodes.size(), &D.tri_nodes)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.sph_bvh4.nodes.data(), acc.sph_bvh4.nodes.size(), &D.sph_nodes)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.tri.data(), acc.tri.size(), &D.btri)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.sph.data(), acc.sph.size(), &D.bsph)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.sph_prim.data(), acc.sph_prim.size(), &D.bsph_prim)) != RT_OK) return rc;
    if ((rc = upload(ctx, acc.sph_key.data(), acc.sph_key.size(), &D.bsph_key)) != RT_OK) return rc;
    D.n_tri_nodes = (int32_t)acc.tri_bvh4.nodes.size();
    D.n_sph_nodes = (int32_t)acc.sph_bvh4.nodes.size();
    {
        const double bb = acc.tri_bvh4.max_abs_coord > acc.sph_bvh4.max_abs_coord ? acc.tri_bvh4.max_abs_coord
                                                                                   : acc.sph_bvh4.max_abs_coord;
        D.bvh_bound = (float)(bb * 1.000001 + 1e-30);
    }
    ctx->accel_usable = acc.usable;
    ctx->accel_depth = acc.tri_bvh4.max_depth > acc.sph_bvh4.max_depth ? acc.tri_bvh4.max_depth : acc.sph_bvh4.max_depth;
    if ((rc = upload(ctx, s->sph, (size_t)s->n_sph_slots, &D.sph)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->sph_prim, (size_t)s->n_sph_slots, &D.sph_prim)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->prim_flags, (size_t)s->n_prims, &D.prim_flags)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->prim_material, (size_t)s->n_prims, &D.prim_material)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->prim_normals, (size_t)s->n_prims * 12, &D.prim_normals)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->materials, (size_t)s->n_materials, &D.materials)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->lights, (size_t)s->n_lights, &D.lights)) != RT_OK) return rc;
    {
        std::vector<int32_t> rank(static_cast<size_t>(s->n_prims), -1);
        ctx->h_lc_spheres.clear();
        for (int32_t i = 0; i < s->n_sph_slots; ++i) {
            if (!(s->sph_prim[i] & RT_SPH_TRANSPARENT)) continue;
            const int32_t p = static_cast<int32_t>(s->sph_prim[i] & ~RT_SPH_TRANSPARENT);
            if (rank[p] >= 0) continue;
            rank[p] = static_cast<int32_t>(ctx->h_lc_spheres.size());
            ctx->h_lc_spheres.push_back({s->sph[i].center[0], s->sph[i].center[1], s->sph[i].center[2], s->sph[i].radius});
        }
        if ((rc = upload(ctx, rank.data(), rank.size(), &D.prim_lc_rank)) != RT_OK) return rc;
        ctx->h_lights.assign(s->lights, s->lights + s->n_lights);
    }
    D.lc_table = nullptr;
    D.n_nodes = s->n_nodes;
    D.n_lights = s->n_lights;
    D.bg[0] = s->background[0]; D.bg[1] = s->background[1]; D.bg[2] = s->background[2];
    ctx->S = D;
    ctx->lightcam_samples = s->lightcam_samples;
    ctx->n_materials = s->n_materials;
    ctx->any_transparent = false;
    for (int32_t i = 0; i < s->n_materials; ++i) if (s->materials[i].T > 0) ctx->any_transparent = true;
    ctx->any_transparent_sphere = false;
    for (int32_t i = 0; i < s->n_sph_slots; ++i) if (s->sph_prim[i] & RT_SPH_TRANSPARENT) ctx->any_transparent_sphere = true;
    // Pruning by entry distance is only result-preserving when every sphere surface point
    // lies inside the root cube (then some leaf that contains the hit point also holds the sphere).
    ctx->spheres_inside_root = true;
    const rt_node &root = s->nodes[0];
    const double half = root.radius / (std::sqrt(3.0) / 2.0) / 2.0;
    for (int32_t i = 0; i < s->n_sph_slots; ++i)
        for (int k = 0; k < 3; ++k) {
            double c = s->sph[i].center[k] - root.center[k];
            if (!(std::fabs(c) + s->sph[i].radius < half * (1.0 - 1e-9))) ctx->spheres_inside_root = false;
        }
    ctx->has_scene = true;
    return RT_OK;
}

static int local_rows_of(const rt_camera *cam, const rt_render_opts *o)
{
    int band = o->band_rows > 0 ? o->band_rows : cam->resy;
    int parts = o->n_parts > 0 ? o->n_parts : 1;
    int rows = 0;
    int n_bands = (cam->resy + band - 1) / band;
    for (int b = o->part; b < n_bands; b += parts) {
        int r0 = b * band, r1 = r0 + band;
        if (r1 > cam->resy) r1 = cam->resy;
        rows += r1 - r0;
    }
    return rows;
}

extern "C" int rt_part_rows(const rt_camera *cam, const rt_render_opts *opts, int32_t *rows)
{
    if (!cam || !opts || !rows) return fail(RT_ERR_INVALID, "rt_part_rows: null argument");
    if (cam->resx <= 0 || cam->resy <= 0) return fail(RT_ERR_INVALID, "rt_part_rows: empty frame");
    int parts = opts->n_parts > 0 ? opts->n_parts : 1;
    if (opts->part < 0 || opts->part >= parts) return fail(RT_ERR_INVALID, "rt_part_rows: part out of range");
    *rows = local_rows_of(cam, opts);
    return RT_OK;
}

template <typename T>
static int ensure(T **ptr, size_t *have, size_t want)
{
    if (*have >= want && *ptr) return RT_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc((void **)ptr, sizeof(T) * want));
    *have = want;
    return RT_OK;
}

static std::pair<hipEvent_t, hipEvent_t> *next_event_pair(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used)
{
    if (used == v.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        v.emplace_back(a, b);
    }
    return &v[used++];
}

// Builds the light camera of every (transparent sphere, light) pair: photon pass on the device,
// point octree on the host (lightcam_build.cpp), flattened tree back to the device.  The reference
// creates a camera the first time a shading point needs it (Camera.h:417-440); a camera depends
// only on its pair, so building all of them ahead changes no pixel.
#define LC_MAX_PAIRS 64
static int build_lightcams(rt_ctx *ctx, uint32_t flags, bool cull)
{
    free_lightcams(ctx);
    const size_t n_pairs = ctx->h_lc_spheres.size() * ctx->h_lights.size();
    if (n_pairs == 0 || n_pairs > LC_MAX_PAIRS) return RT_OK;   // beyond the cap the term stays off (masked)
    const int N = ctx->lightcam_samples;
    if (N < 1 || N > 4096) return fail(RT_ERR_INVALID, "light camera sample count (Z) must be in 1..4096");
    const size_t n_rays = (size_t)N * N;
    lightcam::Point *d_slots = nullptr;
    int32_t *d_counts = nullptr;
    HIP_TRY(hipMalloc((void **)&d_slots, n_rays * LC_MAX_IMPACTS * sizeof(lightcam::Point)));
    if (hipMalloc((void **)&d_counts, n_rays * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(d_slots);
        return fail(RT_ERR_DEVICE, "hipMalloc failed for the light-camera counts");
    }
    std::vector<lightcam::Point> h_slots(n_rays * LC_MAX_IMPACTS);
    std::vector<int32_t> h_counts(n_rays);
    std::vector<DevLightCam> table(n_pairs);
    int rc = RT_OK;
    for (size_t r = 0; r < ctx->h_lc_spheres.size() && rc == RT_OK; ++r)
        for (size_t li = 0; li < ctx->h_lights.size() && rc == RT_OK; ++li) {
            const auto &sp = ctx->h_lc_spheres[r];
            const rt_light &lt = ctx->h_lights[li];
            lightcam::Frame fr;
            const double spos[3] = {sp[0], sp[1], sp[2]};
            lightcam::setup_frame(lt.pos, spos, sp[3], fr);
            PhotonParams Q{};
            Q.S = ctx->S;
            for (int k = 0; k < 3; ++k) { Q.u[k] = fr.u[k]; Q.v[k] = fr.v[k]; Q.w[k] = fr.w[k]; Q.e[k] = fr.e[k]; }
            Q.d = fr.d; Q.l = fr.l; Q.t = fr.t; Q.pix_width = fr.pix_width;
            Q.N = N;
            Q.flags = flags;
            Q.cull = cull ? 1 : 0;
            Q.slots = d_slots;
            Q.counts = d_counts;
            hipLaunchKernelGGL(k_photons, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, ctx->stream, Q);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
                hipMemcpy(h_counts.data(), d_counts, n_rays * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(h_slots.data(), d_slots, h_slots.size() * sizeof(lightcam::Point), hipMemcpyDeviceToHost) != hipSuccess) {
                rc = fail(RT_ERR_DEVICE, "light-camera photon pass failed");
                break;
            }
            std::vector<lightcam::Point> pts;
            double max_dist = 0;
            for (size_t i = 0; i < n_rays; ++i) {
                const int n = h_counts[i] & 0xff;
                ctx->lc_rays += (unsigned long long)(h_counts[i] >> 8);
                for (int k = 0; k < n; ++k) {
                    const lightcam::Point &p = h_slots[i * LC_MAX_IMPACTS + k];
                    pts.push_back(p);
                    const double dz = lightcam::length3(p.x, p.y, p.z);
                    if (dz > max_dist) max_dist = dz;     // std::max(distToOrigin, maxDistFromZero)
                }
            }
            lightcam::Tree tree;
            lightcam::build(pts, max_dist, tree);
            DevLightCam cam{};
            void *dn = nullptr, *dp = nullptr;
            if (hipMalloc(&dn, sizeof(lightcam::Node) * (tree.nodes.size() ? tree.nodes.size() : 1)) != hipSuccess ||
                hipMalloc(&dp, sizeof(lightcam::Point) * (tree.points.size() ? tree.points.size() : 1)) != hipSuccess) {
                if (dn) (void)hipFree(dn);
                rc = fail(RT_ERR_DEVICE, "hipMalloc failed for a light-camera tree");
                break;
            }
            ctx->lc_allocs.push_back(dn);
            ctx->lc_allocs.push_back(dp);
            if (hipMemcpy(dn, tree.nodes.data(), sizeof(lightcam::Node) * tree.nodes.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dp, tree.points.data(), sizeof(lightcam::Point) * tree.points.size(), hipMemcpyHostToDevice) != hipSuccess) {
                rc = fail(RT_ERR_DEVICE, "upload of a light-camera tree failed");
                break;
            }
            cam.nodes = static_cast<const lightcam::Node *>(dn);
            cam.points = static_cast<const lightcam::Point *>(dp);
            cam.n_nodes = (int32_t)tree.nodes.size();
            cam.n_points = (int32_t)tree.points.size();
            cam.spw = fr.pix_width / N;
            for (int k = 0; k < 3; ++k) { cam.light_pos[k] = lt.pos[k]; cam.sphere_pos[k] = spos[k]; }
            table[r * ctx->h_lights.size() + li] = cam;
        }
    (void)hipFree(d_slots);
    (void)hipFree(d_counts);
    if (rc != RT_OK) { free_lightcams(ctx); return rc; }
    void *dt = nullptr;
    HIP_TRY(hipMalloc(&dt, sizeof(DevLightCam) * n_pairs));
    ctx->lc_allocs.push_back(dt);
    HIP_TRY(hipMemcpy(dt, table.data(), sizeof(DevLightCam) * n_pairs, hipMemcpyHostToDevice));
    ctx->S.lc_table = static_cast<const DevLightCam *>(dt);
    ctx->lc_built = true;
    ctx->lc_built_flags = flags;
    return RT_OK;
}

static int enqueue_frame(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *o, const rt_frame_out *out,
                         bool outputs_on_device)
{
    if (!ctx || !cam || !o || !out) return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (!ctx->has_scene) return fail(RT_ERR_INVALID, "rt_render: no scene uploaded");
    if (cam->resx <= 0 || cam->resy <= 0) return fail(RT_ERR_INVALID, "rt_render: empty frame");
    if (o->num_samples < 1 || o->num_samples > 64) return fail(RT_ERR_INVALID, "rt_render: num_samples must be in 1..64");
    const int parts = o->n_parts > 0 ? o->n_parts : 1;
    const int band = o->band_rows > 0 ? o->band_rows : cam->resy;
    if (o->part < 0 || o->part >= parts) return fail(RT_ERR_INVALID, "rt_render: part out of range");
    if (ctx->pending) return fail(RT_ERR_INVALID, "rt_render: a frame is already in flight; call rt_sync first");
    HIP_TRY(hipSetDevice(ctx->device));

    const int local_rows = local_rows_of(cam, o);
    const int N = o->num_samples;
    const size_t n_pixels = (size_t)local_rows * cam->resx;
    if (n_pixels >= (size_t)1 << 31) return fail(RT_ERR_INVALID, "rt_render: more than 2^31 pixels in one part");

    const bool may_cull = !(o->flags & RT_FLAG_NO_CULL) && ctx->spheres_inside_root;
    {
        // light cameras: built once per scene (and per normal-interpolation mode, which the light rays see)
        const uint32_t lc_flags = o->flags & RT_FLAG_PHONG_NORMALS;
        const bool want_lc = !(o->flags & RT_FLAG_NO_LIGHTCAM) && ctx->any_transparent_sphere && !ctx->h_lights.empty();
        if (want_lc && (!ctx->lc_built || ctx->lc_built_flags != lc_flags)) {
            int rc = build_lightcams(ctx, lc_flags, may_cull);
            if (rc != RT_OK) return rc;
        }
    }
    FrameParams P{};
    P.S = ctx->S;
    if (o->flags & RT_FLAG_NO_LIGHTCAM) P.S.lc_table = nullptr;
    P.cam = *cam;
    P.N = N;
    P.aperture = o->aperture;
    P.flags = o->flags;
    P.seed = o->seed;
    P.band_rows = band;
    P.n_parts = parts;
    P.part = o->part;
    P.local_rows = local_rows;
    P.blocks_x = (cam->resx + 7) / 8;
    P.cull_closest = may_cull ? 1 : 0;
    P.cull_shadow = (may_cull && !ctx->any_transparent_sphere) ? 1 : 0;
    // The hierarchies answer "nearest / any hit over all primitives"; that equals the reference's
    // walk when every sphere is inside the root cube, and for shadow queries only when no
    // transparent sphere exists (checkOccluded3 is then order-free).
    const bool bvh_closest_ok = may_cull && ctx->accel_usable;
    const bool bvh_shadow_ok = bvh_closest_ok && !ctx->any_transparent_sphere;
    // at most three pending siblings per 4-wide level (BVH_STACK is the cap the builder checks against)
    int stack_entries = 3 * ctx->accel_depth + 1;
    if (stack_entries > BVH_STACK) stack_entries = BVH_STACK;
    if (stack_entries < 4) stack_entries = 4;
    // LDS per block: the stacks, the material and light tables, then as many top-of-tree nodes as fit
    // beside them when four blocks share a compute unit (the register budget allows four waves per SIMD)
    const size_t lds_stack = (size_t)stack_entries * 256 * sizeof(int);
    P.S.mat_cached = ctx->n_materials < 64 ? ctx->n_materials : 64;
    P.S.light_cached = P.S.n_lights < 16 ? P.S.n_lights : 16;
    const size_t lds_tables = (size_t)(P.S.mat_cached + P.S.light_cached) * 64;
    {
        const size_t per_block = ctx->lds_per_cu / 4;
        const size_t fixed = lds_stack + lds_tables + 512;
        long long room = per_block > fixed ? (long long)((per_block - fixed) / (NODE_LDS_WORDS * 4)) : 0;
        if (const char *e = std::getenv("RT_NODE_CACHE")) room = std::atoll(e) < room ? std::atoll(e) : room;   // experiments
        // the triangle hierarchy first, the sphere hierarchy takes what is left; split evenly when both are large
        long long nt = P.S.n_tri_nodes, ns = P.S.n_sph_nodes;
        if (nt + ns > room) {
            const long long half = room / 2;
            if (nt <= half) ns = room - nt;
            else if (ns <= half) nt = room - ns;
            else { nt = half; ns = room - half; }
        }
        P.S.tri_cached = (int32_t)nt;
        P.S.sph_cached = (int32_t)ns;
        P.S.lds_cache_word = (int32_t)(lds_stack / sizeof(int));
        P.S.lds_table_word = P.S.lds_cache_word + (P.S.tri_cached + P.S.sph_cached) * NODE_LDS_WORDS;
    }
    // every traversal kernel gets the same layout (the octree-walk variants use the tables only)
    const size_t lds_bvh = lds_stack + (size_t)(P.S.tri_cached + P.S.sph_cached) * NODE_LDS_WORDS * sizeof(int) + lds_tables;
    const size_t lds_closest = lds_bvh;
    const size_t lds_shadow = lds_bvh;

    hipStream_t st = ctx->stream;
    HIP_TRY(hipEventRecord(ctx->ev_begin, st));
    HIP_TRY(hipMemsetAsync(ctx->totals, 0, 16 * sizeof(unsigned long long), st));
    ctx->ev_closest_used = ctx->ev_shadow_used = 0;
    ctx->pending_flags = o->flags;
    ctx->pending_batches = 0;
    ctx->pending_primaries = (unsigned long long)n_pixels * (unsigned long long)(N * N);

    if (local_rows > 0) {
        int rc;
        if ((rc = ensure(&ctx->accum, &ctx->accum_values, n_pixels * 3)) != RT_OK) return rc;
        HIP_TRY(hipMemsetAsync(ctx->accum, 0, n_pixels * 3 * sizeof(double), st));
        P.accum = ctx->accum;

        // outputs: device pointers are used directly, host pointers go through staging buffers
        void *dst[4] = {out->rgb_f64, out->rgb_u8, out->hit_id, out->lightcam_mask};
        const size_t bytes[4] = {n_pixels * 3 * sizeof(double), n_pixels * 3, n_pixels * sizeof(int32_t), n_pixels};
        void *dev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 4; ++k) {
            if (!dst[k]) continue;
            if (outputs_on_device) dev[k] = dst[k];
            else {
                if (ctx->stage_bytes[k] < bytes[k]) {
                    if (ctx->stage[k]) HIP_TRY(hipFree(ctx->stage[k]));
                    ctx->stage[k] = nullptr; ctx->stage_bytes[k] = 0;
                    HIP_TRY(hipMalloc(&ctx->stage[k], bytes[k]));
                    ctx->stage_bytes[k] = bytes[k];
                }
                dev[k] = ctx->stage[k];
            }
        }
        P.hit_id = static_cast<int32_t *>(dev[2]);
        P.lc_mask = static_cast<uint8_t *>(dev[3]);
        if (dev[3]) HIP_TRY(hipMemsetAsync(dev[3], 0, bytes[3], st));

        // primary ordering: 8x8-pixel blocks, N*N samples per pixel -> (N*N) 64-sample chunks per block
        const long long blocks_y = ((long long)local_rows + 7) / 8;
        const long long n_chunks = blocks_y * P.blocks_x * (long long)(N * N);
        long long batch_chunks = ctx->batch_override > 0 ? (ctx->batch_override + 63) / 64
                                                          : (ctx->any_transparent ? (1ll << 22) / 64 : (1ll << 26) / 64);
        if (batch_chunks > n_chunks) batch_chunks = n_chunks;
        if (batch_chunks < 1) batch_chunks = 1;

        // Shard capacities, generation by generation: what one wave can be handed, times the waves
        // that feed one shard, times the fan-out of a hit (reflection, plus refraction if any
        // material is transparent).  These are upper bounds, so a push can never overflow.
        const int max_blocks = ctx->cu_count * 8;
        const long long fan_out = ctx->any_transparent ? 2 : 1;
        long long blocks0 = (batch_chunks + 3) / 4;
        if (blocks0 > max_blocks) blocks0 = max_blocks;
        const long long waves_q = (long long)max_blocks * 4, wps_q = ((max_blocks + N_SHARDS - 1) / N_SHARDS) * 4ll;
        long long hit_cap[6], ray_cap[7];
        ray_cap[0] = 0;
        {
            const long long w0 = blocks0 * 4, wps0 = ((blocks0 + N_SHARDS - 1) / N_SHARDS) * 4;
            hit_cap[0] = ((batch_chunks + w0 - 1) / w0) * 64 * wps0;
            ray_cap[1] = hit_cap[0] * fan_out;
        }
        for (int g = 1; g < 6; ++g) {
            const long long c_in = N_SHARDS * ((ray_cap[g] + 63) / 64);
            hit_cap[g] = ((c_in + waves_q - 1) / waves_q) * 64 * wps_q;
            ray_cap[g + 1] = hit_cap[g] * fan_out;
        }
        long long need_rays[2] = {0, 0}, need_hits = 0;
        for (int g = 1; g < 6; ++g) if (ray_cap[g] * N_SHARDS > need_rays[g & 1]) need_rays[g & 1] = ray_cap[g] * N_SHARDS;
        for (int g = 0; g < 6; ++g) if (hit_cap[g] * N_SHARDS > need_hits) need_hits = hit_cap[g] * N_SHARDS;
        if (need_rays[0] >= (1ll << 32) || need_rays[1] >= (1ll << 32) || need_hits >= (1ll << 32))
            return fail(RT_ERR_INVALID, "rt_render: batch too large for 32-bit queue indices");
        if ((rc = ensure(&ctx->rays[0], &ctx->ray_capacity[0], (size_t)(need_rays[0] ? need_rays[0] : 64))) != RT_OK) return rc;
        if ((rc = ensure(&ctx->rays[1], &ctx->ray_capacity[1], (size_t)(need_rays[1] ? need_rays[1] : 64))) != RT_OK) return rc;
        // Overlap pays when a pass cannot fill the chip by itself (small frames: each pass is one
        // ray's latency long); big batches already saturate it with one persistent grid, and there
        // the extra ordering constraints only cost, so they run on one stream.
        const bool overlap = batch_chunks * 64 <= (1ll << 22);
        // with overlap, four hit buffers: [batch parity][generation parity], so that the shadow pass of
        // one generation (second stream) can run beside the closest-hit pass of the next one; on one
        // stream a single buffer serves every pass
        for (int k = 0; k < (overlap ? 4 : 1); ++k)
            if ((rc = ensure(&ctx->hits[k], &ctx->hit_capacity[k], (size_t)need_hits)) != RT_OK) return rc;
        const bool use_lightcam = P.S.lc_table != nullptr && !bvh_shadow_ok && P.S.n_lights > 0;
        if (use_lightcam && (rc = ensure(&ctx->deferred, &ctx->deferred_capacity, (size_t)need_hits)) != RT_OK) return rc;

        // Two streams.  A: counters memset, closest-hit passes C0..C5 of every batch, in order.
        // B: shadow/shading passes S0..S5 and the batch's counter roll-up.  S_g waits for C_g; C_g+2
        // waits for S_g (it re-uses S_g's hit buffer); a batch's memset waits for the roll-up of
        // the batch two before it (same counter block).  Pixel accumulation is atomic, so the
        // order in which the passes finish does not matter.
        hipStream_t sb = overlap ? ctx->stream2 : st;
        const bool timing = (o->flags & RT_FLAG_TIMING) != 0;
        long long batch_no = 0;
        for (long long t0 = 0; t0 < n_chunks; t0 += batch_chunks, ++batch_no) {
            const long long t1 = (t0 + batch_chunks < n_chunks) ? t0 + batch_chunks : n_chunks;
            const int par = (int)(batch_no & 1);
            uint32_t *counts = ctx->counts + (size_t)par * CNT_WORDS;
            if (batch_no >= 2) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[par], 0));
            HIP_TRY(hipMemsetAsync(counts, 0, CNT_WORDS * sizeof(uint32_t), st));
            for (int g = 0; g < 6; ++g) {
                QueueSet Q{};
                Q.rays_in = ctx->rays[g & 1];
                Q.rays_out = ctx->rays[(g + 1) & 1];
                Q.hits = ctx->hits[overlap ? par * 2 + (g & 1) : 0];
                Q.deferred = ctx->deferred;
                Q.counts = counts;
                Q.in_shard_cap = (uint32_t)ray_cap[g];
                Q.out_shard_cap = (uint32_t)(g < 5 ? ray_cap[g + 1] : 0);
                Q.hit_shard_cap = (uint32_t)hit_cap[g];
                Q.generation = g;
                Q.chunk_begin = t0;
                Q.chunk_end = t1;
                if (g >= 2 && P.S.n_lights > 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_s[g - 2], 0));
                std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
                if (timing && !(ev = next_event_pair(ctx->ev_closest, ctx->ev_closest_used)))
                    return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                if (g == 0) {
                    long long blocks = (t1 - t0 + 3) / 4;
                    if (blocks > blocks0) blocks = blocks0;
                    if (bvh_closest_ok) hipLaunchKernelGGL((k_closest<true, true>), dim3((unsigned)blocks), dim3(256), lds_closest, st, P, Q);
                    else hipLaunchKernelGGL((k_closest<true, false>), dim3((unsigned)blocks), dim3(256), lds_closest, st, P, Q);
                } else {
                    if (bvh_closest_ok) hipLaunchKernelGGL((k_closest<false, true>), dim3((unsigned)max_blocks), dim3(256), lds_closest, st, P, Q);
                    else hipLaunchKernelGGL((k_closest<false, false>), dim3((unsigned)max_blocks), dim3(256), lds_closest, st, P, Q);
                }
                if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                HIP_TRY(hipEventRecord(ctx->ev_c[g], st));
                if (P.S.n_lights > 0) {
                    HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_c[g], 0));
                    ev = nullptr;
                    if (timing && !(ev = next_event_pair(ctx->ev_shadow, ctx->ev_shadow_used)))
                        return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev) HIP_TRY(hipEventRecord(ev->first, sb));
                    if (bvh_shadow_ok) hipLaunchKernelGGL((k_shadow<true>), dim3((unsigned)max_blocks), dim3(256), lds_shadow, sb, P, Q);
                    else hipLaunchKernelGGL((k_shadow<false>), dim3((unsigned)max_blocks), dim3(256), lds_shadow, sb, P, Q);
                    if (use_lightcam) hipLaunchKernelGGL(k_lightcam, dim3((unsigned)max_blocks), dim3(256), 0, sb, P, Q);
                    if (ev) HIP_TRY(hipEventRecord(ev->second, sb));
                    HIP_TRY(hipEventRecord(ctx->ev_s[g], sb));
                }
            }
            HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_c[5], 0));
            hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, sb, counts, ctx->totals, P.S.n_lights);
            HIP_TRY(hipEventRecord(ctx->ev_batch[par], sb));
            ctx->pending_batches++;
        }
        // the frame is complete when the second stream has drained
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[0], 0));
        if (batch_no >= 2) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[1], 0));
        HIP_TRY(hipGetLastError());
        {
            const long long n_values = (long long)n_pixels * 3;
            long long blocks = (n_values + 255) / 256;
            if (blocks > max_blocks) blocks = max_blocks;
            hipLaunchKernelGGL(k_finalize, dim3((unsigned)blocks), dim3(256), 0, st, ctx->accum, n_values, N * N,
                               static_cast<double *>(dev[0]), static_cast<uint8_t *>(dev[1]));
            HIP_TRY(hipGetLastError());
        }
        if (!outputs_on_device)
            for (int k = 0; k < 4; ++k)
                if (dst[k]) HIP_TRY(hipMemcpyAsync(dst[k], dev[k], bytes[k], hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(ctx->ev_end, st));
    ctx->pending = true;
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx *ctx, rt_stats *stats)
{
    if (!ctx) return fail(RT_ERR_INVALID, "rt_sync: null argument");
    if (!ctx->pending) return fail(RT_ERR_INVALID, "rt_sync: no frame in flight");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->pending = false;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long tot[16];
    HIP_TRY(hipMemcpy(tot, ctx->totals, sizeof tot, hipMemcpyDeviceToHost));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        tot[0] += ctx->pending_primaries;   // generation 0 is counted on the host
        tot[2] = ctx->pending_primaries;
        stats->closest_rays = tot[0];
        stats->lightcam_rays = ctx->lc_rays;
        stats->shadow_rays = tot[1];
        for (int g = 0; g < 6; ++g) { stats->rays_per_generation[g] = tot[2 + g]; stats->hits_per_generation[g] = tot[8 + g]; }
        stats->queue_overflow = tot[15] ? 1 : 0;
        stats->batches = ctx->pending_batches;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        stats->gpu_ms = ms;
        if (ctx->pending_flags & RT_FLAG_TIMING) {
            double c = 0, s = 0;
            for (size_t i = 0; i < ctx->ev_closest_used; ++i) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, ctx->ev_closest[i].first, ctx->ev_closest[i].second));
                c += m;
            }
            for (size_t i = 0; i < ctx->ev_shadow_used; ++i) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, ctx->ev_shadow[i].first, ctx->ev_shadow[i].second));
                s += m;
            }
            stats->closest_ms = c;
            stats->shadow_ms = s;
            stats->other_ms = ms - c - s;
            stats->closest_launches = ctx->ev_closest_used;
            stats->shadow_launches = ctx->ev_shadow_used;
        }
    }
    if (tot[15]) return fail(RT_ERR_CAPACITY, "rt_render: a ray queue overflowed; the frame is incomplete");
    return RT_OK;
}

extern "C" int rt_render_async(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out)
{
    if (out && !out->on_device) return fail(RT_ERR_INVALID, "rt_render_async: outputs must be device memory");
    return enqueue_frame(ctx, cam, opts, out, true);
}

extern "C" int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out,
                         rt_stats *stats)
{
    int rc = enqueue_frame(ctx, cam, opts, out, out && out->on_device);
    if (rc != RT_OK) {
        if (ctx) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->stream2); ctx->pending = false; }
        return rc;
    }
    return rt_sync(ctx, stats);
}
