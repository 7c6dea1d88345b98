// The light cameras' point octrees (Octree<Vector3d*>, Octree.h:1394-1497) built on the device with two sorts instead of
// one per level.
//
// Which child of a node takes a point is a matter of the point's coordinates and the node's cube alone (the FIRST child,
// in octant order, whose closed box holds it, or none: Octree.h:1441-1446), and a child's cube follows from its parent's
// (Octree.h:1430-1465): a point's way down the tree, as far as it could ever go, is known without looking at any other
// point.  What the other points decide is how far it does go: a node subdivides when it holds more than 10 points and its
// half side exceeds 1e-8 (Octree.h:1367-1368,1427).  So:
//   paths      every point writes its whole way as a key: its tree, then a digit per level (1..8: child octant + 1, 9:
//              no child holds it, the point stays in that node; 0: the way has ended);
//   sort       by key: the points of any node, at any depth, are now one run -- those that share the digits down to it;
//   depths     a point sees from its ten neighbours on either side which of the nodes along its way hold more than ten
//              points (a run is longer than ten exactly when eleven consecutive places around the point share the digits),
//              hence where its way really ends; the digits beyond are cleared;
//   sort       the points in their ORIGINAL order by the cleared keys, stably: inside a node that did not subdivide the
//              points keep the order they came in, which is what the reference's level-by-level stable partition leaves;
//   nodes      every place where a run of shared digits begins is a node: numbered by a prefix sum, range by binary
//              search, cube by replaying the digits from the root's cube with the builder's arithmetic, children linked
//              by each child looking up its parent.
// The result is the tree of lightcam_device.hpp node for node, except for the nodes' numbers (the query follows child
// indices only); weight sums are formed by the same kernel, level by level from the deepest.  A way that does not fit the
// key's digits makes the build give up, and the level-by-level builder takes over.
#pragma once
#include <hip/hip_runtime.h>

#include "lightcam_build.hpp"

namespace lcpath {

#define LCP_MAX_DIGITS 30      // digits of 4 bits in two 64-bit words, less the bits of the tree number

struct Key {
    unsigned long long hi, lo;
};

struct Params {
    int n;                  // points
    int n_trees;
    int tree_bits;          // the tree number takes the top bits of Key::hi
    int digits;             // digits that follow it: (128 - tree_bits) / 4, at most LCP_MAX_DIGITS
    const int32_t *begins;  // tree t owns points [begins[t], begins[t + 1]) of the original order
};

__device__ inline int digit_of(const Key &k, int tree_bits, int l)
{
    const int bit = tree_bits + 4 * l;            // from the top of the 128-bit key
    return bit < 64 ? (int)((k.hi >> (60 - bit)) & 15ull) : (int)((k.lo >> (124 - bit)) & 15ull);
}
__device__ inline void set_digit(Key &k, int tree_bits, int l, int d)
{
    const int bit = tree_bits + 4 * l;
    if (bit < 64) k.hi |= (unsigned long long)d << (60 - bit);
    else k.lo |= (unsigned long long)d << (124 - bit);
}
// (tree_bits is a multiple of 4, so no digit straddles the two words)

// Digits two keys share from the top: -1 when their trees differ.
__device__ inline int shared_digits(const Key &a, const Key &b, int tree_bits)
{
    int same_bits;
    if (a.hi != b.hi) same_bits = __clzll((long long)(a.hi ^ b.hi));
    else if (a.lo != b.lo) same_bits = 64 + __clzll((long long)(a.lo ^ b.lo));
    else same_bits = 128;
    if (same_bits < tree_bits) return -1;
    return (same_bits - tree_bits) >> 2;
}

struct Cube {
    double cx, cy, cz, size;
};
// Octant of `nd` that takes point q: the first child whose closed box holds it, 8 when none does (Octree.h:1441-1446).
__device__ inline int octant_of(const Cube &nd, const double q[3])
{
    const double size = nd.size, h = size / 2;
    const double c[3] = {nd.cx, nd.cy, nd.cz};
    int where = 0;
    for (int d = 0; d < 3; ++d) {
        // the two intervals of every axis: [c - size/2, (c - size/2) + h] and [c, c + h]
        const double lo0 = c[d] - size / 2, hi0 = lo0 + h, lo1 = c[d], hi1 = lo1 + h;
        if (lo0 <= q[d] && q[d] <= hi0) continue;
        if (lo1 <= q[d] && q[d] <= hi1) where |= 1 << d;
        else return 8;
    }
    return where;
}
// Child k's cube (Octree.h:1430-1465: a child's side is hi.x - lo.x of its box).
__device__ inline Cube child_cube(const Cube &nd, int k)
{
    const double size = nd.size, h = size / 2;
    const double lox = (k & 1) ? nd.cx : nd.cx - size / 2;
    const double loy = (k & 2) ? nd.cy : nd.cy - size / 2;
    const double loz = (k & 4) ? nd.cz : nd.cz - size / 2;
    const double csize = (lox + h) - lox;
    Cube ch;
    ch.size = csize;
    ch.cx = lox + csize / 2; ch.cy = loy + csize / 2; ch.cz = loz + csize / 2;
    return ch;
}
__device__ inline Cube root_cube(const lightcam::Node &r)
{
    Cube c;
    c.cx = r.cx; c.cy = r.cy; c.cz = r.cz; c.size = r.size;
    return c;
}
__device__ inline int tree_of(const int32_t *begins, int n_trees, int i)
{
    int lo = 0, hi = n_trees - 1;          // the last tree whose range begins at or before i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (begins[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The whole way of every point (roots[t] is tree t's root as k_roots made it).  geo[i]: the first depth at which a node on
// the way has a half side of 1e-8 or less and cannot subdivide whatever it holds (255: none within the key's digits).
__global__ __launch_bounds__(256) void k_paths(Params P, const lightcam::Point *pts, const lightcam::Node *roots, unsigned long long *key_hi,
                                               unsigned long long *key_lo, uint32_t *idx, uint8_t *geo)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P.n) return;
    const int t = tree_of(P.begins, P.n_trees, i);
    const lightcam::Point p = pts[i];
    const double q[3] = {p.x, p.y, p.z};
    Cube c = root_cube(roots[t]);
    Key k;
    k.hi = P.tree_bits ? (unsigned long long)t << (64 - P.tree_bits) : 0ull;
    k.lo = 0ull;
    int g = 255;
    for (int l = 0; l < P.digits; ++l) {
        if (!(c.size / 2 > .00000001)) { g = l; break; }
        const int where = octant_of(c, q);
        set_digit(k, P.tree_bits, l, where + 1);
        if (where == 8) break;
        c = child_cube(c, where);
    }
    key_hi[i] = k.hi; key_lo[i] = k.lo;
    idx[i] = (uint32_t)i;
    geo[i] = (uint8_t)g;
}

// values[j] = source[index[j]] (the second key word follows the first sort's order, and so on)
__global__ __launch_bounds__(256) void k_gather_words(const unsigned long long *source, const uint32_t *index, int n, unsigned long long *out)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j < n) out[j] = source[index[j]];
}

// Where every point's way really ends.  In the order of the whole keys, position j; `orig[j]` is the point's place in the
// original order, where its cleared key goes (the second sort starts from that order).  *too_deep is raised when a node at
// the last depth the key can express would still subdivide.
__global__ __launch_bounds__(256) void k_depths(Params P, const unsigned long long *hi, const unsigned long long *lo, const uint32_t *orig,
                                                const uint8_t *geo, unsigned long long *cleared_hi, unsigned long long *cleared_lo, int32_t *too_deep)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= P.n) return;
    Key me;
    me.hi = hi[j]; me.lo = lo[j];
    // digits shared with the k-th neighbour on either side; runs are contiguous, so the numbers do not grow with k
    int left[10], right[10];
    for (int k = 0; k < 10; ++k) {
        left[k] = right[k] = -2;
        if (j - 1 - k >= 0 && (k == 0 || left[k - 1] >= 0)) {
            Key o; o.hi = hi[j - 1 - k]; o.lo = lo[j - 1 - k];
            left[k] = shared_digits(o, me, P.tree_bits);
        }
        if (j + 1 + k < P.n && (k == 0 || right[k - 1] >= 0)) {
            Key o; o.hi = hi[j + 1 + k]; o.lo = lo[j + 1 + k];
            right[k] = shared_digits(o, me, P.tree_bits);
        }
    }
    const int g = geo[orig[j]];
    int depth = 0;        // of the node the point ends in
    bool rest = false;    // it ends there as a point no child of a subdivided node holds
    for (;;) {
        // the node at `depth` holds the points that share `depth` digits with this one
        int count = 1;
        for (int k = 0; k < 10; ++k) count += (left[k] >= depth ? 1 : 0) + (right[k] >= depth ? 1 : 0);
        const bool subdivides = count > 10 && depth < g;
        if (!subdivides) break;
        if (depth >= P.digits) { *too_deep = 1; break; }          // no digit left to say where the point goes from here
        const int d = digit_of(me, P.tree_bits, depth);
        if (d == 9) { rest = true; break; }
        if (d == 0) { *too_deep = 1; break; }                     // (cannot be: a node that may subdivide gave the point a digit)
        ++depth;
    }
    const int keep = depth + (rest ? 1 : 0);      // digits that stay
    Key c;
    const int bits = P.tree_bits + 4 * keep;
    c.hi = bits >= 64 ? me.hi : (bits == 0 ? 0ull : me.hi & ~((1ull << (64 - bits)) - 1ull));
    c.lo = bits <= 64 ? 0ull : (bits >= 128 ? me.lo : me.lo & ~((1ull << (128 - bits)) - 1ull));
    cleared_hi[orig[j]] = c.hi;
    cleared_lo[orig[j]] = c.lo;
}

__device__ inline int kept_digits(const Key &k, int tree_bits, int digits)
{
    int n = 0;
    while (n < digits && digit_of(k, tree_bits, n) != 0) ++n;
    return n;
}

// In the final order: the points themselves, and how many nodes below the roots begin at every place (for the prefix sum
// that numbers them).  A point with k kept digits lies in the nodes at depths 0..k, less one when its last digit is 9.
__global__ __launch_bounds__(256) void k_place(Params P, const unsigned long long *hi, const unsigned long long *lo, const uint32_t *orig,
                                               const lightcam::Point *pts_in, lightcam::Point *pts_out, uint32_t *begins_here)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= P.n) return;
    pts_out[j] = pts_in[orig[j]];
    Key me; me.hi = hi[j]; me.lo = lo[j];
    const int kept = kept_digits(me, P.tree_bits, P.digits);
    const int depth = kept - ((kept > 0 && digit_of(me, P.tree_bits, kept - 1) == 9) ? 1 : 0);
    int shared = -1;
    if (j > 0) { Key o; o.hi = hi[j - 1]; o.lo = lo[j - 1]; shared = shared_digits(o, me, P.tree_bits); }
    // nodes at depths shared + 1 .. depth begin here; depth 0 is the root, which has its number already
    const int first = shared + 1 > 1 ? shared + 1 : 1;
    begins_here[j] = depth >= first ? (uint32_t)(depth - first + 1) : 0u;
}

// First position at or after `from` (up to `end`) whose key shares fewer than `d` digits with `me`: the end of the run of
// the node at depth d that holds `me`.
__device__ inline int run_end(const unsigned long long *hi, const unsigned long long *lo, int tree_bits, const Key &me, int d, int from, int end)
{
    int a = from, b = end;          // the answer lies in (a, b]; position a shares d digits
    while (b - a > 1) {
        const int mid = a + ((b - a) >> 1);
        Key o; o.hi = hi[mid]; o.lo = lo[mid];
        if (shared_digits(o, me, tree_bits) >= d) a = mid; else b = mid;
    }
    return b;
}
// First position of that run, searching down from `from`.
__device__ inline int run_begin(const unsigned long long *hi, const unsigned long long *lo, int tree_bits, const Key &me, int d, int from, int floor_)
{
    int a = floor_ - 1, b = from;   // the answer lies in (a, b]; position b shares d digits
    while (b - a > 1) {
        const int mid = a + ((b - a) >> 1);
        Key o; o.hi = hi[mid]; o.lo = lo[mid];
        if (shared_digits(o, me, tree_bits) >= d) b = mid; else a = mid;
    }
    return b;
}

// Number of the node at depth d whose run begins at position p (d >= 1), given the prefix sums.
__device__ inline int node_number(const Params &P, const unsigned long long *hi, const unsigned long long *lo, const uint32_t *number_base, int p, int d)
{
    Key me; me.hi = hi[p]; me.lo = lo[p];
    int shared = -1;
    if (p > 0) { Key o; o.hi = hi[p - 1]; o.lo = lo[p - 1]; shared = shared_digits(o, me, P.tree_bits); }
    const int first = shared + 1 > 1 ? shared + 1 : 1;
    return P.n_trees + (int)number_base[p] + (d - first);
}

// The nodes that begin at every place: cube, range, what stays in them, and their own entry in their parent.  Roots (depth
// 0, node t of tree t) get their range's split here too.  level_of[node] = depth.
__global__ __launch_bounds__(256) void k_nodes(Params P, const unsigned long long *hi, const unsigned long long *lo, const uint32_t *number_base,
                                               const lightcam::Node *roots, lightcam::Node *nodes, int node_capacity, uint8_t *level_of, int32_t *overflow)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= P.n) return;
    Key me; me.hi = hi[j]; me.lo = lo[j];
    const int kept = kept_digits(me, P.tree_bits, P.digits);
    const int depth = kept - ((kept > 0 && digit_of(me, P.tree_bits, kept - 1) == 9) ? 1 : 0);
    int shared = -1;
    if (j > 0) { Key o; o.hi = hi[j - 1]; o.lo = lo[j - 1]; shared = shared_digits(o, me, P.tree_bits); }
    if (depth < shared + 1) return;                       // no node begins here
    const int t = P.tree_bits ? (int)(me.hi >> (64 - P.tree_bits)) : 0;
    const int tree_end = P.begins[t + 1];
    const int first = shared + 1 > 1 ? shared + 1 : 1;
    // (the roots' cubes are read before any root is written: k_nodes is launched with a copy of them in `roots`)
    Cube c = root_cube(roots[t]);
    int end_above = tree_end;                             // end of the run one level up (bounds the search)
    for (int d = 0; d <= depth; ++d) {
        if (d > 0) c = child_cube(c, digit_of(me, P.tree_bits, d - 1) - 1);
        if (d < shared + 1) continue;                     // that node began at an earlier place
        const int id = d == 0 ? t : P.n_trees + (int)number_base[j] + (d - first);
        if (id >= node_capacity) { *overflow = 1; return; }
        const int end = d == 0 ? tree_end : run_end(hi, lo, P.tree_bits, me, d, j, end_above);
        end_above = end;
        lightcam::Node nd;
        nd.cx = c.cx; nd.cy = c.cy; nd.cz = c.cz; nd.size = c.size;
        nd.begin = j; nd.end = end;
        nd.leaf = 1;
        for (int k = 0; k < 8; ++k) nd.child[k] = -1;
        nd.rest_begin = j;
        nd.wsum = 0;
        nd.child_leaf = 0;
        nd._pad = 0;
        const bool subdivides = c.size / 2 > .00000001 && (end - j) > 10;
        if (subdivides && d < P.digits) {
            // the points no child holds lie at the run's end (digit 9 sorts last): the first of them, by bisection
            int a = j - 1, b = end;          // positions <= a go to children, positions >= b (if any) stay
            while (b - a > 1) {
                const int mid = a + ((b - a) >> 1);
                Key o; o.hi = hi[mid]; o.lo = lo[mid];
                if (digit_of(o, P.tree_bits, d) == 9) b = mid; else a = mid;
            }
            nd.rest_begin = b;
            if (b == end) nd.leaf = 0;       // every point found a child (Octree.h:1452-1453)
        }
        nodes[id] = nd;
        level_of[id] = (uint8_t)d;
    }
}

// Every node below a root enters itself in its parent (run after k_nodes: the parents' child entries start at -1).
__global__ __launch_bounds__(256) void k_link(Params P, const unsigned long long *hi, const unsigned long long *lo, const uint32_t *number_base,
                                              lightcam::Node *nodes, int node_capacity)
{
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= P.n) return;
    Key me; me.hi = hi[j]; me.lo = lo[j];
    const int kept = kept_digits(me, P.tree_bits, P.digits);
    const int depth = kept - ((kept > 0 && digit_of(me, P.tree_bits, kept - 1) == 9) ? 1 : 0);
    int shared = -1;
    if (j > 0) { Key o; o.hi = hi[j - 1]; o.lo = lo[j - 1]; shared = shared_digits(o, me, P.tree_bits); }
    const int first = shared + 1 > 1 ? shared + 1 : 1;
    if (depth < first) return;
    const int t = P.tree_bits ? (int)(me.hi >> (64 - P.tree_bits)) : 0;
    for (int d = first; d <= depth; ++d) {
        const int id = P.n_trees + (int)number_base[j] + (d - first);
        if (id >= node_capacity) return;
        int parent;
        if (d - 1 == 0) parent = t;
        else if (d - 1 >= first) parent = id - 1;            // the node one level up begins here as well
        else {
            const int p = run_begin(hi, lo, P.tree_bits, me, d - 1, j, P.begins[t]);
            parent = node_number(P, hi, lo, number_base, p, d - 1);
        }
        if (parent < node_capacity) nodes[parent].child[digit_of(me, P.tree_bits, d - 1) - 1] = id;
    }
}

// Nodes per depth (a block counts in LDS first: a few dozen counters in memory would take every node's atomic in turn).
#define LCP_LEVELS 64
__global__ __launch_bounds__(256) void k_level_hist(const uint8_t *level_of, int n_nodes, uint32_t *level_count)
{
    __shared__ uint32_t mine[LCP_LEVELS];
    if (threadIdx.x < LCP_LEVELS) mine[threadIdx.x] = 0u;
    __syncthreads();
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < n_nodes) atomicAdd(&mine[level_of[k] & (LCP_LEVELS - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < LCP_LEVELS && mine[threadIdx.x]) atomicAdd(&level_count[threadIdx.x], mine[threadIdx.x]);
}
// The nodes' numbers grouped by depth: list[first_of_level[d] + k], in any order inside a level.
__global__ __launch_bounds__(256) void k_level_lists(const uint8_t *level_of, int n_nodes, const uint32_t *first_of_level, uint32_t *cursor, int32_t *list)
{
    __shared__ uint32_t mine[LCP_LEVELS], base[LCP_LEVELS];
    if (threadIdx.x < LCP_LEVELS) mine[threadIdx.x] = 0u;
    __syncthreads();
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int d = 0;
    uint32_t rank = 0;
    if (k < n_nodes) { d = level_of[k] & (LCP_LEVELS - 1); rank = atomicAdd(&mine[d], 1u); }
    __syncthreads();
    if (threadIdx.x < LCP_LEVELS && mine[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], mine[threadIdx.x]);
    __syncthreads();
    if (k < n_nodes) list[first_of_level[d] + base[d] + rank] = k;
}

__global__ __launch_bounds__(256) void k_iota(uint32_t *idx, int n)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) idx[i] = (uint32_t)i;
}

// Weight sums and children's leaf flags of the nodes list[0 .. count) (one level; the levels below are done): the arithmetic
// of lcdev::k_finish_level.
__global__ __launch_bounds__(256) void k_finish_list(lightcam::Node *nodes, const int32_t *list, int count, const lightcam::Point *pts)
{
    const int lane = (int)(threadIdx.x & 63);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long i = wave; i < (long long)count; i += n_waves) {
        const int k = list[i];
        const int b = nodes[k].rest_begin, e = nodes[k].end;
        double s = 0;
        for (int p = b + lane; p < e; p += 64) s += pts[p].w;
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

}  // namespace lcpath
