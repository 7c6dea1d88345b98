// Grids of what can be seen from one point: the primitives of the scene binned by the direction in which they lie
// from a point that many rays share -- the eye (Camera.h:182-191: every primary ray of a pinhole frame starts at
// `from`) or a point light (Camera.h:389-404: every shadow ray ends at one).
//
// A ray through such a point can only meet primitives whose central projection from that point covers the ray's
// direction.  So the directions are cut into bins (the camera's image plane in tiles of 8 x 8 pixels; a cube map of
// res x res bins per face around a light) and every bin gets the list of leaf records whose projection touches it.
// A query then reads ONE list and runs the reference's own FP64 test (primitives.hpp) on each entry: no hierarchy
// walk at all.  The lists only have to be complete, never exact: every decision about a primitive is still the
// reference's, so results are the walk's bit for bit.
//
// Completeness.  A primitive is entered by conservative rasterisation of its projection: a triangle is clipped against
// the plane `depth >= z_near` of the face (Sutherland-Hodgman, at most four corners), projected in FP64, grown by
// `margin` in projected units, and entered into every bin its bounding rectangle covers that no edge of the polygon
// separates from it; a sphere by the bounding rectangle of its silhouette (the tangent lines through the point in the
// two coordinate planes), or the whole face when it reaches the plane depth = z_near.  z_near is below the depth of
// any point a query can report (a hit lies at least minT from the eye, and at least .01 from the light because a
// shadow query ends .01 before it, Camera.h:404); `margin` is 2^-16 of the grid's extent, sixty times the rounding of
// the single-precision bin look-up of a shadow ray and far beyond the FP64 rounding of the projection itself.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/raytracer_amd.h"

namespace vg {

#define VG_SPHERE 0x80000000u      // entry: index of a leaf sphere record (else of a leaf triangle record)
#define VG_LARGE_BINS 128          // an item whose rectangle holds more bins is entered by a whole block

// An entry of a bin's list: the leaf record, and a lower bound (single precision, rounded down) of the distance from the
// grid's point to any part of the primitive that lies in the bin's directions.  A list is sorted by that distance, so a
// query stops at the first entry that lies beyond what it can still report (the nearest hit found so far; the far end of
// a shadow ray): most shadow rays of lit points test nothing at all.  For a triangle the bound is the smallest distance
// at which the bin's four corner directions (grown by the margin) meet its plane, when they all meet it in front; for a
// sphere the distance to its surface; 0 where no such statement can be made.
struct Entry {
    uint32_t id;
    float depth;
};
// What a query reads first of a bin: where its list lies, and the list's first two entries themselves -- most lists are no
// longer than that, and a query then needs one trip to memory instead of two in a row.
struct Bin {
    uint32_t begin, count;
    Entry first[2];
    uint32_t _pad[2];
};
static_assert(sizeof(Bin) == 32, "a bin head is 32 bytes");

struct Face {
    double u[3], v[3], w[3];       // a point V (relative to the grid's point) with V.w > 0 projects to x = V.u / V.w, y = V.v / V.w
    double x0, y0, inv_cell;       // bin column of x: floor((x - x0) * inv_cell), rows likewise
    int32_t nx, ny;
    uint32_t first_bin;            // of this face in the counter / offset arrays
    int32_t _pad;
};

struct Build {
    double origin[3];
    double z_near;
    double margin;                 // growth of every projected shape, in projected units
    Face face[6];
    int32_t n_faces;
    int32_t n_tri, n_sph;          // items: the first n_tri are triangles (leaf record tri_index[k]), then n_sph spheres (leaf record k)
    const int32_t *tri_index;
    const rt_tri_rec *tri;
    const rt_sph_rec *sph;
    uint32_t *bins;                // count pass: entries per bin; fill pass: the bin's next free place in `entries`
    Entry *entries;
    uint32_t *large;               // (item * 8 + face) of the items left to the block-per-item kernel
    uint32_t *n_large;
    uint32_t large_cap;
    uint32_t *overflow;
    unsigned long long *large_total;   // entries of the large items, summed while they are counted            // raised when `large` is full (the grid is then not used)
};

// The projection of one item on one face: a convex polygon of at most four corners (n = 0: nothing of it in front),
// or `all` (the whole face), and the rectangle of bins its grown bounding box covers.
struct Shape {
    int n;
    bool all, edges;               // edges: the polygon's edge functions may be used to leave bins out
    double x[4], y[4];
    double sgn;                    // orientation of the polygon (+1 counter-clockwise)
    int bx0, bx1, by0, by1;
    // depth: a triangle's plane in the face's coordinates, depth(x, y) = pc / (x pa + y pb + pw) along w (pc = 0: no statement);
    // a sphere's constant bound in `flat`
    double pa, pb, pw, pc, flat;
};

__device__ inline double dot3a(const double *a, double x, double y, double z) { return (a[0] * x + a[1] * y) + a[2] * z; }

__device__ inline bool bin_range(const Face &f, double lo, double hi, double margin, double x0, int n, int *b0, int *b1)
{
    // columns (or rows) touched by [lo - margin, hi + margin]; false when the interval misses the grid
    const double a = (lo - margin - x0) * f.inv_cell, b = (hi + margin - x0) * f.inv_cell;
    if (!(b >= 0.0) || !(a <= (double)n)) return false;
    *b0 = a <= 0.0 ? 0 : (a >= (double)(n - 1) ? n - 1 : (int)a);
    *b1 = b >= (double)(n - 1) ? n - 1 : (int)b;
    if (*b1 < *b0) *b1 = *b0;
    return true;
}

__device__ inline Shape whole_face(const Face &f)
{
    Shape s;
    s.n = 4; s.all = true; s.edges = false; s.sgn = 1.0;
    s.bx0 = 0; s.bx1 = f.nx - 1; s.by0 = 0; s.by1 = f.ny - 1;
    for (int k = 0; k < 4; ++k) s.x[k] = s.y[k] = 0.0;
    s.pa = s.pb = s.pw = s.pc = s.flat = 0.0;
    return s;
}

__device__ inline Shape project_triangle(const Build &B, const Face &f, const rt_tri_rec &t)
{
    Shape s;
    s.n = 0; s.all = false; s.edges = false; s.sgn = 1.0;
    s.bx0 = s.bx1 = s.by0 = s.by1 = 0;
    s.pa = s.pb = s.pw = s.pc = s.flat = 0.0;
    {
        // the plane N.(V - v0) = 0 with V = depth (x a + y b + w), a = u / |u|^2, b = v / |v|^2 (|w| = 1): depth = N.v0 / (x N.a + y N.b + N.w)
        const double nx = t.ab[1] * t.ac[2] - t.ac[1] * t.ab[2], ny = t.ab[2] * t.ac[0] - t.ac[2] * t.ab[0], nz = t.ab[0] * t.ac[1] - t.ac[0] * t.ab[1];
        const double uu = dot3a(f.u, f.u[0], f.u[1], f.u[2]), vv = dot3a(f.v, f.v[0], f.v[1], f.v[2]);
        s.pa = dot3a(f.u, nx, ny, nz) / uu;
        s.pb = dot3a(f.v, nx, ny, nz) / vv;
        s.pw = dot3a(f.w, nx, ny, nz);
        s.pc = (nx * (t.v0[0] - B.origin[0]) + ny * (t.v0[1] - B.origin[1])) + nz * (t.v0[2] - B.origin[2]);
        if (!(fabs(s.pc) < 1e300) || !(fabs(s.pa) < 1e300) || !(fabs(s.pb) < 1e300) || !(fabs(s.pw) < 1e300)) s.pc = 0.0;
    }
    double X[3], Y[3], Z[3];
    for (int k = 0; k < 3; ++k) {
        const double px = t.v0[0] - (k == 1 ? t.ab[0] : (k == 2 ? t.ac[0] : 0.0)) - B.origin[0];
        const double py = t.v0[1] - (k == 1 ? t.ab[1] : (k == 2 ? t.ac[1] : 0.0)) - B.origin[1];
        const double pz = t.v0[2] - (k == 1 ? t.ab[2] : (k == 2 ? t.ac[2] : 0.0)) - B.origin[2];
        X[k] = dot3a(f.u, px, py, pz); Y[k] = dot3a(f.v, px, py, pz); Z[k] = dot3a(f.w, px, py, pz);
    }
    // clip against Z >= z_near
    const double zn = B.z_near;
    for (int k = 0; k < 3; ++k) {
        const int m = k == 2 ? 0 : k + 1;
        const bool in_k = Z[k] >= zn, in_m = Z[m] >= zn;
        if (in_k) { s.x[s.n] = X[k] / Z[k]; s.y[s.n] = Y[k] / Z[k]; ++s.n; }
        if (in_k != in_m) {
            const double a = (zn - Z[k]) / (Z[m] - Z[k]);
            const double cx = X[k] + a * (X[m] - X[k]), cy = Y[k] + a * (Y[m] - Y[k]);
            s.x[s.n] = cx / zn; s.y[s.n] = cy / zn; ++s.n;
        }
    }
    if (s.n == 0) return s;
    double xl = s.x[0], xh = s.x[0], yl = s.y[0], yh = s.y[0];
    for (int k = 1; k < s.n; ++k) {
        xl = s.x[k] < xl ? s.x[k] : xl; xh = s.x[k] > xh ? s.x[k] : xh;
        yl = s.y[k] < yl ? s.y[k] : yl; yh = s.y[k] > yh ? s.y[k] : yh;
    }
    if (!(xl == xl) || !(xh == xh) || !(yl == yl) || !(yh == yh)) return whole_face(f);     // (not a number: no statement about it)
    if (!bin_range(f, xl, xh, B.margin, f.x0, f.nx, &s.bx0, &s.bx1) || !bin_range(f, yl, yh, B.margin, f.y0, f.ny, &s.by0, &s.by1)) {
        s.n = 0;
        return s;
    }
    // orientation; an (almost) edge-on polygon keeps to its rectangle
    double area2 = 0.0;
    for (int k = 0; k < s.n; ++k) {
        const int m = k + 1 == s.n ? 0 : k + 1;
        area2 += s.x[k] * s.y[m] - s.x[m] * s.y[k];
    }
    const double ext = (xh - xl) + (yh - yl);
    s.sgn = area2 < 0.0 ? -1.0 : 1.0;
    s.edges = s.n >= 3 && fabs(area2) > 1e-9 * ext * ext && fabs(area2) < 1e290;
    return s;
}

__device__ inline Shape project_sphere(const Build &B, const Face &f, const rt_sph_rec &sp)
{
    Shape s;
    s.n = 0; s.all = false; s.edges = false; s.sgn = 1.0;
    s.bx0 = s.bx1 = s.by0 = s.by1 = 0;
    s.flat = 0.0;
    const double px = sp.center[0] - B.origin[0], py = sp.center[1] - B.origin[1], pz = sp.center[2] - B.origin[2];
    const double cx = dot3a(f.u, px, py, pz), cy = dot3a(f.v, px, py, pz), cz = dot3a(f.w, px, py, pz);
    const double r = fabs(sp.radius) * (1.0 + 1e-9) + 1e-12 * (fabs(px) + fabs(py) + fabs(pz) + 1.0);
    s.pa = s.pb = s.pw = s.pc = 0.0;
    {
        const double dist = sqrt((px * px + py * py) + pz * pz) - r;
        s.flat = dist > 0.0 ? dist : 0.0;
    }
    // (the axes are orthogonal but need not be unit vectors -- the camera's u and v carry the image plane's distance --
    // so lengths along an axis are world lengths times the axis' own)
    const double wl = sqrt(dot3a(f.w, f.w[0], f.w[1], f.w[2]));
    if (!(cz + r * wl >= B.z_near)) {
        if (cz == cz && r == r) return s;      // wholly behind the plane depth = z_near
        return whole_face(f);
    }
    const double flat_depth = s.flat == s.flat ? s.flat : 0.0;
    // The point lies inside the sphere (or as good as): every direction meets it.
    if (!(s.flat > 0.0)) { Shape a = whole_face(f); a.flat = 0.0; return a; }
    // Tangent lines through the point in the planes (u, w) and (v, w), in world lengths.  With the centre at (a, z), at the
    // distance D from the point, and L^2 = D^2 - r^2 > 0, the directions (x, depth) of the two tangents are
    // (a L + z r, z L - a r) and (a L - z r, z L + a r); the sphere's directions lie between them.  A tangent that points
    // backward (depth <= 0: the sphere reaches through the plane depth = 0 on that side) leaves that side open; a sphere that
    // does not reach the plane has both in front, and the slopes are (a z +- r L) / (z^2 - r^2).  L^2 <= 0 (the point is inside
    // the sphere's shadow on that plane) leaves both sides open.
    const double ul = sqrt(dot3a(f.u, f.u[0], f.u[1], f.u[2])), vl = sqrt(dot3a(f.v, f.v[0], f.v[1], f.v[2]));
    const double z = cz / wl;
    const double open_end = 1e300;
    double lo[2], hi[2];
    for (int k = 0; k < 2; ++k) {
        const double a = (k == 0 ? cx / ul : cy / vl), scale = (k == 0 ? ul : vl) / wl;
        const double l2 = (a * a + z * z) - r * r;
        lo[k] = -open_end; hi[k] = open_end;
        if (!(l2 > 1e-12 * (a * a + z * z))) continue;
        const double L = sqrt(l2);
        const double z_hi = z * L - a * r, z_lo = z * L + a * r, tiny = 1e-9 * (fabs(z) * L + fabs(a) * r);
        if (z_hi > tiny) {
            hi[k] = (a * L + z * r) / z_hi * scale;
            hi[k] += 1e-9 * fabs(hi[k]);
        }
        if (z_lo > tiny) {
            lo[k] = (a * L - z * r) / z_lo * scale;
            lo[k] -= 1e-9 * fabs(lo[k]);
        }
    }
    if (!(lo[0] == lo[0]) || !(hi[0] == hi[0]) || !(lo[1] == lo[1]) || !(hi[1] == hi[1])) { Shape a = whole_face(f); a.flat = flat_depth; return a; }
    s.flat = flat_depth;
    if (!bin_range(f, lo[0], hi[0], B.margin, f.x0, f.nx, &s.bx0, &s.bx1) || !bin_range(f, lo[1], hi[1], B.margin, f.y0, f.ny, &s.by0, &s.by1))
        return s;
    s.n = 4;
    return s;
}

__device__ inline Shape project_item(const Build &B, int item, int face)
{
    const Face &f = B.face[face];
    if (item < B.n_tri) return project_triangle(B, f, B.tri[B.tri_index[item]]);
    return project_sphere(B, f, B.sph[item - B.n_tri]);
}

// Does bin (bx, by) of the face possibly meet the (grown) polygon?
__device__ inline bool bin_touches(const Build &B, const Face &f, const Shape &s, int bx, int by)
{
#ifdef VG_TEST_DROP      // (a test build leaves entries out, to show that the tests see an incomplete list)
    if ((bx * 3 + by) % VG_TEST_DROP == 0) return false;
#endif
    if (!s.edges) return true;
    const double cell = 1.0 / f.inv_cell;
    const double X0 = f.x0 + bx * cell, X1 = X0 + cell, Y0 = f.y0 + by * cell, Y1 = Y0 + cell;
    for (int k = 0; k < s.n; ++k) {
        const int m = k + 1 == s.n ? 0 : k + 1;
        const double dx = (s.x[m] - s.x[k]) * s.sgn, dy = (s.y[m] - s.y[k]) * s.sgn;
        if (dx == 0.0 && dy == 0.0) continue;
        // the corner of the bin farthest on the inner side of the edge
        const double px = dy < 0.0 ? X1 : X0, py = dx > 0.0 ? Y1 : Y0;
        const double e = dx * (py - s.y[k]) - dy * (px - s.x[k]);
        const double slack = (fabs(dx) + fabs(dy)) * (B.margin + 1e-12 * (fabs(px) + fabs(py) + fabs(s.x[k]) + fabs(s.y[k])));
        if (e < -slack) return false;
    }
    return true;
}

// Lower bound of the distance from the grid's point to the part of the shape that lies in the directions of bin (bx, by).
__device__ inline float bin_depth(const Build &B, const Face &f, const Shape &s, int bx, int by)
{
    double bound = s.flat;
    if (s.pc != 0.0) {
        const double cell = 1.0 / f.inv_cell;
        const double X0 = f.x0 + bx * cell - B.margin, X1 = f.x0 + (bx + 1) * cell + B.margin;
        const double Y0 = f.y0 + by * cell - B.margin, Y1 = f.y0 + (by + 1) * cell + B.margin;
        // the denominators at the four corners, as multiples of the sign of pc: all positive <=> every direction of the bin
        // meets the plane in front
        const double sg = s.pc < 0.0 ? -1.0 : 1.0;
        const double d00 = sg * ((X0 * s.pa + Y0 * s.pb) + s.pw), d10 = sg * ((X1 * s.pa + Y0 * s.pb) + s.pw);
        const double d01 = sg * ((X0 * s.pa + Y1 * s.pb) + s.pw), d11 = sg * ((X1 * s.pa + Y1 * s.pb) + s.pw);
        double dmax = d00 > d10 ? d00 : d10;
        dmax = d01 > dmax ? d01 : dmax;
        dmax = d11 > dmax ? d11 : dmax;
        const double dmin = fmin(fmin(d00, d10), fmin(d01, d11));
        const double scale = (fabs(s.pa) * (fabs(X0) + fabs(X1)) + fabs(s.pb) * (fabs(Y0) + fabs(Y1))) + fabs(s.pw);
        if (dmin > 1e-9 * scale && dmax < 1e300) {
            const double zmin = fabs(s.pc) / dmax;
            // |x a + y b + w| over the bin is at least this (a, b orthogonal to w; |a| = 1/|u|, |b| = 1/|v|)
            const double xm = (X0 <= 0.0 && X1 >= 0.0) ? 0.0 : fmin(fabs(X0), fabs(X1));
            const double ym = (Y0 <= 0.0 && Y1 >= 0.0) ? 0.0 : fmin(fabs(Y0), fabs(Y1));
            const double uu = dot3a(f.u, f.u[0], f.u[1], f.u[2]), vv = dot3a(f.v, f.v[0], f.v[1], f.v[2]);
            bound = zmin * sqrt((xm * xm / uu + ym * ym / vv) + 1.0);
        } else {
            bound = 0.0;
        }
    }
    if (!(bound > 0.0) || !(bound < 1e30)) return 0.0f;
    // a little below, and rounded towards zero
    float r = (float)(bound * (1.0 - 0x1p-20));
    if ((double)r > bound * (1.0 - 0x1p-21)) r = __uint_as_float(__float_as_uint(r) - 1u);
    return r > 0.0f ? r : 0.0f;
}

__device__ inline uint32_t entry_of(const Build &B, int item)
{
    return item < B.n_tri ? (uint32_t)B.tri_index[item] : ((uint32_t)(item - B.n_tri) | VG_SPHERE);
}

// One thread per (item, face).  FILL false: bins[] counts the entries; true: bins[] holds each bin's next free place.
template <bool FILL>
__global__ __launch_bounds__(256) void k_grid_items(Build B)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)(B.n_tri + B.n_sph) * B.n_faces;
    if (idx >= total) return;
    const int item = (int)(idx / B.n_faces), face = (int)(idx - (long long)item * B.n_faces);
    const Shape s = project_item(B, item, face);
    if (s.n == 0) return;
    const Face &f = B.face[face];
    const long long nb = (long long)(s.bx1 - s.bx0 + 1) * (s.by1 - s.by0 + 1);
    if (nb > VG_LARGE_BINS) {
        if (!FILL) {
            const uint32_t at = atomicAdd(B.n_large, 1u);
            if (at < B.large_cap) B.large[at] = (uint32_t)item * 8u + (uint32_t)face;
            else *B.overflow = 1u;
        }
        return;
    }
    const uint32_t e = entry_of(B, item);
    for (int by = s.by0; by <= s.by1; ++by)
        for (int bx = s.bx0; bx <= s.bx1; ++bx) {
            if (!bin_touches(B, f, s, bx, by)) continue;
            const uint32_t bin = f.first_bin + (uint32_t)by * (uint32_t)f.nx + (uint32_t)bx;
            const uint32_t at = atomicAdd(&B.bins[bin], 1u);
            if (FILL) { Entry en; en.id = e; en.depth = bin_depth(B, f, s, bx, by); B.entries[at] = en; }
        }
}

// The items with many bins: VG_LARGE_SLICES blocks each (a floor under a light covers a whole face of its cube map -- 65 536
// bins -- and a scene has a handful of such items: one block per item left the rest of the chip idle for 0.4 ms a pass),
// a block's threads striding over its share of the rectangle.
#define VG_LARGE_SLICES 32
template <bool FILL>
__global__ __launch_bounds__(256) void k_grid_large(Build B)
{
    uint32_t n = *B.n_large;
    if (n > B.large_cap) n = B.large_cap;
    const unsigned long long units = (unsigned long long)n * VG_LARGE_SLICES;
    for (unsigned long long u = blockIdx.x; u < units; u += gridDim.x) {
        const uint32_t code = B.large[u / VG_LARGE_SLICES];
        const uint32_t slice = (uint32_t)(u % VG_LARGE_SLICES);
        const int item = (int)(code >> 3), face = (int)(code & 7u);
        const Shape s = project_item(B, item, face);
        if (s.n == 0) continue;
        const Face &f = B.face[face];
        const uint32_t e = entry_of(B, item);
        const int w = s.bx1 - s.bx0 + 1;
        const long long nb = (long long)w * (s.by1 - s.by0 + 1);
        unsigned long long mine = 0;
        for (long long k = (long long)slice * blockDim.x + threadIdx.x; k < nb; k += (long long)blockDim.x * VG_LARGE_SLICES) {
            const int by = s.by0 + (int)(k / w), bx = s.bx0 + (int)(k - (long long)(by - s.by0) * w);
            if (!bin_touches(B, f, s, bx, by)) continue;
            const uint32_t bin = f.first_bin + (uint32_t)by * (uint32_t)f.nx + (uint32_t)bx;
            const uint32_t at = atomicAdd(&B.bins[bin], 1u);
            if (FILL) { Entry en; en.id = e; en.depth = bin_depth(B, f, s, bx, by); B.entries[at] = en; }
            ++mine;
        }
        if (!FILL && mine) atomicAdd(B.large_total, mine);
    }
}

// Every bin's list in ascending order of depth (ties by record, so that a grid is the same every time it is made).
// One thread per bin: the lists are short (insertion sort; a heap sort beyond 64 entries).
__device__ inline bool entry_less(const Entry &a, const Entry &b) { return a.depth < b.depth || (a.depth == b.depth && a.id < b.id); }
__global__ __launch_bounds__(256) void k_grid_sort(const uint32_t *begin, uint32_t n_bins, Entry *entries)
{
    const uint32_t bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= n_bins) return;
    Entry *a = entries + begin[bin];
    const uint32_t n = begin[bin + 1] - begin[bin];
    if (n < 2) return;
    if (n <= 64) {
        for (uint32_t i = 1; i < n; ++i) {
            const Entry x = a[i];
            uint32_t j = i;
            while (j > 0 && entry_less(x, a[j - 1])) { a[j] = a[j - 1]; --j; }
            a[j] = x;
        }
        return;
    }
    auto sift = [&](uint32_t root, uint32_t end) {
        for (;;) {
            uint32_t child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && entry_less(a[child], a[child + 1])) ++child;
            if (!entry_less(a[root], a[child])) break;
            const Entry t = a[root]; a[root] = a[child]; a[child] = t;
            root = child;
        }
    };
    for (uint32_t i = n / 2; i-- > 0;) sift(i, n);
    for (uint32_t end = n - 1; end > 0; --end) {
        const Entry t = a[0]; a[0] = a[end]; a[end] = t;
        sift(0, end);
    }
}

// Bins that hold anything (for the decision whether the lists are short enough to be worth reading).
__global__ __launch_bounds__(256) void k_grid_occupied(const uint32_t *begin, uint32_t n_bins, uint32_t *occupied)
{
    const uint32_t bin = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = __syncthreads_count(bin < n_bins && begin[bin + 1] > begin[bin]);
    if (threadIdx.x == 0 && n) atomicAdd(occupied, (uint32_t)n);
}

// The bins' heads, from the offsets and the sorted lists.
__global__ __launch_bounds__(256) void k_grid_heads(const uint32_t *begin, uint32_t n_bins, const Entry *entries, Bin *bins)
{
    const uint32_t bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= n_bins) return;
    Bin b;
    b.begin = begin[bin];
    b.count = begin[bin + 1] - b.begin;
    const Entry none = {0u, 0.0f};
    b.first[0] = b.count > 0 ? entries[b.begin] : none;
    b.first[1] = b.count > 1 ? entries[b.begin + 1] : none;
    b._pad[0] = b._pad[1] = 0u;
    bins[bin] = b;
}

}  // namespace vg
