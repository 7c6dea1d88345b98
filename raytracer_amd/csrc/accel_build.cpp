// Bounding-volume hierarchies derived from the flattened octree (see accel_build.hpp).
//
// Why the answer cannot change.  The reference's closest-hit query (Octree.h:996-1062)
// tests a primitive only if every node on the path from the root to a node that holds it
// passes the bounding-sphere test (Sphere.h:155-204).  A triangle is resident in a node only
// if it lies inside that node's cube (Triangle.h:359-374), and a sphere is handed down to
// every child cube it overlaps (Sphere.h:211-232) until subdivision stops; so for any hit
// point inside the root cube there is a node holding the primitive whose cube, and every
// ancestor's cube, contains the hit point -- and a ray that reaches a point inside a cube
// passes that cube's bounding-sphere test.  Hence, once the root's own test has passed, the
// reference's result is "nearest intersection over all primitives", ties going to the
// primitive met first in its depth-first order.  The hierarchies below answer exactly that,
// with the tie rule carried as an explicit key.  The premise "every hit point lies inside
// the root cube" is checked for spheres at upload (triangles outside the cube stay resident
// in the root, whose test is evaluated per ray on the device); when it fails the device keeps
// to the octree walk.
#include "accel_build.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace accel {
namespace {

struct Item {
    double lo[3], hi[3], c[3];
    int32_t src;   // index into the source record array
};

struct Box {
    double lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
    }
    void grow(const double *l, const double *h)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
    }
    double area() const
    {
        double e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (e[0] < 0 || e[1] < 0 || e[2] < 0) return 0;
        return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    }
};

constexpr int kLeafMaxDefault = 2;   // 1..4 (the leaf code keeps count-1 in two bits); 2 measured best: box tests are cheap, FP64 records are not
constexpr int kSplitBudgetPercent = 100;   // growth allowed to split_references()
constexpr int kBins = 16;
constexpr int kSahDepth = 14;   // below this depth splits are medians, which bounds the tree height

class Builder {
public:
    Builder(std::vector<Item> &items, Bvh &out, int leaf_max) : it_(items), out_(out), kLeafMax(leaf_max) {}

    void run()
    {
        out_.nodes.clear();
        out_.max_depth = 0;
        const int n = static_cast<int>(it_.size());
        if (n == 0) return;
        if (n <= kLeafMax) {   // a root whose first child is the only leaf
            BvhNode root = empty_node();
            set_leaf(root, 0, 0, n);
            out_.nodes.push_back(root);
            out_.max_depth = 1;
            return;
        }
        split(0, n, 1);
    }

private:
    static BvhNode empty_node()
    {
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        for (int s = 0; s < 2; ++s) {
            nd.child[s] = -1;
            for (int k = 0; k < 3; ++k) { nd.lo[s][k] = 1.0; nd.hi[s][k] = -1.0; }
        }
        return nd;
    }

    Box bounds(int b, int e) const
    {
        Box bx;
        bx.reset();
        for (int i = b; i < e; ++i) bx.grow(it_[i].lo, it_[i].hi);
        return bx;
    }

    // Boxes are pushed outward so that the slab test on the device is conservative with
    // respect to the FP64 primitive tests it guards.
    static void store_box(BvhNode &nd, int side, const Box &bx)
    {
        for (int k = 0; k < 3; ++k) {
            double scale = std::max({1.0, std::fabs(bx.lo[k]), std::fabs(bx.hi[k]), bx.hi[k] - bx.lo[k]});
            nd.lo[side][k] = bx.lo[k] - 1e-7 * scale;
            nd.hi[side][k] = bx.hi[k] + 1e-7 * scale;
        }
    }

    void set_leaf(BvhNode &nd, int side, int b, int e)
    {
        // largest box first: an any-hit walk stops at the first occluder it finds
        std::stable_sort(it_.begin() + b, it_.begin() + e, [](const Item &x, const Item &y) {
            auto area = [](const Item &i) {
                const double ex = i.hi[0] - i.lo[0], ey = i.hi[1] - i.lo[1], ez = i.hi[2] - i.lo[2];
                return ex * ey + ey * ez + ez * ex;
            };
            return area(x) > area(y);
        });
        store_box(nd, side, bounds(b, e));
        // two pieces of one primitive in the same leaf: the record is tested once (the second copy stays in
        // the array behind the leaf's range, unreferenced)
        int count = e - b;
        for (int i = b + 1; i < b + count;) {
            bool dup = false;
            for (int j = b; j < i; ++j) dup = dup || it_[j].src == it_[i].src;
            if (dup) { std::rotate(it_.begin() + i, it_.begin() + i + 1, it_.begin() + b + count); --count; }
            else ++i;
        }
        nd.child[side] = -2 - (b * 4 + (count - 1));   // leaf code: first record and count-1 in two bits
        nd.count[side] = count;
    }

    int choose_split(int b, int e, int depth)
    {
        const int n = e - b;
        Box cb;
        cb.reset();
        for (int i = b; i < e; ++i) cb.grow(it_[i].c, it_[i].c);
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        auto median = [&](int ax) {
            const int mid = b + n / 2;
            std::nth_element(it_.begin() + b, it_.begin() + mid, it_.begin() + e,
                             [ax](const Item &x, const Item &y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.src < y.src); });
            return mid;
        };
        if (depth >= kSahDepth || !(cb.hi[axis] - cb.lo[axis] > 0)) return median(axis);

        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            const double ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0)) continue;
            Box bin_box[kBins];
            int bin_n[kBins] = {0};
            for (auto &bb : bin_box) bb.reset();
            for (int i = b; i < e; ++i) {
                int k = static_cast<int>(kBins * ((it_[i].c[ax] - cb.lo[ax]) / ext));
                k = std::min(std::max(k, 0), kBins - 1);
                bin_box[k].grow(it_[i].lo, it_[i].hi);
                bin_n[k]++;
            }
            double right_area[kBins];
            int right_n[kBins];
            Box acc;
            acc.reset();
            int cnt = 0;
            for (int k = kBins - 1; k > 0; --k) {
                if (bin_n[k]) acc.grow(bin_box[k].lo, bin_box[k].hi);
                cnt += bin_n[k];
                right_area[k] = acc.area();
                right_n[k] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                if (bin_n[k]) acc.grow(bin_box[k].lo, bin_box[k].hi);
                cnt += bin_n[k];
                if (cnt == 0 || right_n[k + 1] == 0) continue;
                double cost = acc.area() * cnt + right_area[k + 1] * right_n[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
            }
        }
        if (best_axis < 0) return median(axis);
        const double ext = cb.hi[best_axis] - cb.lo[best_axis];
        const double lo = cb.lo[best_axis];
        auto mid_it = std::partition(it_.begin() + b, it_.begin() + e, [&](const Item &x) {
            int k = static_cast<int>(kBins * ((x.c[best_axis] - lo) / ext));
            k = std::min(std::max(k, 0), kBins - 1);
            return k <= best_bin;
        });
        int mid = static_cast<int>(mid_it - it_.begin());
        if (mid == b || mid == e) return median(axis);
        return mid;
    }

    // Builds the node for [b,e) (more than kLeafMax items) and returns its index.
    int split(int b, int e, int depth)
    {
        out_.max_depth = std::max(out_.max_depth, depth);
        const int me = static_cast<int>(out_.nodes.size());
        out_.nodes.push_back(empty_node());
        const int mid = choose_split(b, e, depth);
        const int rb[2] = {b, mid}, re[2] = {mid, e};
        for (int s = 0; s < 2; ++s) {
            if (re[s] - rb[s] <= kLeafMax) {
                BvhNode nd = out_.nodes[me];
                set_leaf(nd, s, rb[s], re[s]);
                out_.nodes[me] = nd;
                out_.max_depth = std::max(out_.max_depth, depth + 1);
            } else {
                int c = split(rb[s], re[s], depth + 1);
                BvhNode nd = out_.nodes[me];
                store_box(nd, s, bounds(rb[s], re[s]));
                nd.child[s] = c;
                nd.count[s] = 0;
                out_.nodes[me] = nd;
            }
        }
        return me;
    }

    std::vector<Item> &it_;
    Bvh &out_;
    const int kLeafMax;
};

// ---- reference splitting ----
// A long triangle that lies obliquely in space has a box far larger than itself, and every ray
// through that box pays for an exact test.  Before the tree is built such triangles are
// represented by several items: the triangle is cut by axis-aligned planes and each piece gets
// the (tight) box of its own part.  All pieces point to the same record, so a query still sees
// the same primitive with the same result; a primitive found through two pieces is the same
// (t, key) candidate twice.  The pieces' boxes cover the triangle, which is all the walk needs.
// A triangle cut by axis-aligned planes is a convex polygon of at most 9 corners.
struct Poly {
    double p[12][3];
    int n = 0;
    void add(const double *x) { if (n < 12) { p[n][0] = x[0]; p[n][1] = x[1]; p[n][2] = x[2]; ++n; } }
};
struct Piece {
    Poly poly;        // the part of the triangle inside the item's box
    Item item;
    double gain = 0;  // box area the best cut of this piece saves (0: do not cut)
};

Box poly_bounds(const Poly &poly)
{
    Box b;
    b.reset();
    for (int i = 0; i < poly.n; ++i) b.grow(poly.p[i], poly.p[i]);
    return b;
}

// The part of a convex polygon with coordinate `axis` <= at (keep_low) or >= at.
void clip_poly(const Poly &poly, int axis, double at, bool keep_low, Poly &out)
{
    out.n = 0;
    for (int i = 0; i < poly.n; ++i) {
        const double *a = poly.p[i], *b = poly.p[(i + 1) % poly.n];
        const bool ia = keep_low ? a[axis] <= at : a[axis] >= at;
        const bool ib = keep_low ? b[axis] <= at : b[axis] >= at;
        if (ia) out.add(a);
        if (ia != ib) {
            const double t = (at - a[axis]) / (b[axis] - a[axis]);
            double x[3];
            for (int k = 0; k < 3; ++k) x[k] = a[k] + t * (b[k] - a[k]);
            x[axis] = at;
            out.add(x);
        }
    }
}

void set_item_box(Item &it, const Box &b)
{
    for (int k = 0; k < 3; ++k) { it.lo[k] = b.lo[k]; it.hi[k] = b.hi[k]; it.c[k] = 0.5 * (b.lo[k] + b.hi[k]); }
}

// Halves of `p` on either side of the middle of its box's longest axis; false when there is nothing to cut.
bool cut_piece(const Piece &p, Piece &lo, Piece &hi)
{
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (p.item.hi[k] - p.item.lo[k] > p.item.hi[axis] - p.item.lo[axis]) axis = k;
    const double at = 0.5 * (p.item.lo[axis] + p.item.hi[axis]);
    if (!(at > p.item.lo[axis]) || !(at < p.item.hi[axis])) return false;
    clip_poly(p.poly, axis, at, true, lo.poly);
    clip_poly(p.poly, axis, at, false, hi.poly);
    if (lo.poly.n < 3 || hi.poly.n < 3) return false;
    lo.item = hi.item = p.item;
    set_item_box(lo.item, poly_bounds(lo.poly));
    set_item_box(hi.item, poly_bounds(hi.poly));
    return true;
}

double item_area(const Item &it)
{
    Box b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = it.lo[k]; b.hi[k] = it.hi[k]; }
    return b.area();
}

// Cuts, most profitable first, until the items have grown by `budget_percent` or no cut saves at
// least `kMinSaving` of a box; the result replaces `items` only if it pays for the scene as a whole.  (A cut that saves less -- an axis-aligned right triangle saves a
// quarter -- makes the tree larger for little.)
void split_references(std::vector<Item> &items, const rt_tri_rec *records, int budget_percent, bool verbose)
{
    constexpr double kMinSaving = 0.3, kMinTotalSaving = 0.25;
    const size_t budget = items.size() * static_cast<size_t>(budget_percent) / 100;
    if (budget == 0) return;
    auto prepare = [&](Piece &p) {
        Piece a, b;
        p.gain = 0;
        if (!cut_piece(p, a, b)) return;
        const double whole = item_area(p.item), parts = item_area(a.item) + item_area(b.item);
        if (whole > 0 && parts <= (1 - kMinSaving) * whole) p.gain = whole - parts;
    };
    // pieces that may still be cut wait in `pool`, ordered through a heap of (gain, index)
    std::vector<Piece> pool;
    std::vector<std::pair<double, size_t>> heap;
    std::vector<Item> done;
    done.reserve(items.size() + budget);
    auto whole_piece = [&](const Item &it) {
        const rt_tri_rec &r = records[it.src];
        Piece p;
        p.item = it;
        for (int k = 0; k < 3; ++k) { p.poly.p[0][k] = r.v0[k]; p.poly.p[1][k] = r.v0[k] - r.ab[k]; p.poly.p[2][k] = r.v0[k] - r.ac[k]; }
        p.poly.n = 3;
        prepare(p);
        return p;
    };
    // the first cut of every triangle already tells a compact mesh from a scene of slivers
    double first_gain = 0, whole_area = 0;
    for (const Item &it : items) {
        whole_area += item_area(it);
        first_gain += whole_piece(it).gain;
    }
    if (first_gain < 0.5 * kMinTotalSaving * whole_area) return;
    for (const Item &it : items) {
        const Piece p = whole_piece(it);
        if (p.gain > 0) { heap.emplace_back(p.gain, pool.size()); pool.push_back(p); }
        else done.push_back(it);
    }
    std::make_heap(heap.begin(), heap.end());
    size_t extra = 0;
    while (!heap.empty() && extra < budget) {
        std::pop_heap(heap.begin(), heap.end());
        const size_t at = heap.back().second;
        heap.pop_back();
        const Piece p = pool[at];
        Piece half[2];
        if (!cut_piece(p, half[0], half[1])) { done.push_back(p.item); continue; }
        ++extra;
        for (int h = 0; h < 2; ++h) {
            prepare(half[h]);
            if (half[h].gain <= 0) { done.push_back(half[h].item); continue; }
            size_t slot = pool.size();
            if (h == 0) { slot = at; pool[at] = half[h]; }      // the parent's place is free
            else pool.push_back(half[h]);
            heap.emplace_back(half[h].gain, slot);
            std::push_heap(heap.begin(), heap.end());
        }
    }
    for (const auto &e : heap) done.push_back(pool[e.second].item);
    // Summed box area is what the number of exact tests of a random ray is proportional to.  The cut is
    // kept only where it removes a good part of it (gears.nff, whose faces are fans of slivers: 64 %);
    // on compact meshes (the teapots: 9-10 %) the larger tree costs more than the tighter boxes save.
    double before = 0, after = 0;
    for (const Item &it : items) before += item_area(it);
    for (const Item &it : done) after += item_area(it);
    const bool keep = after <= (1 - kMinTotalSaving) * before;
    if (verbose)
        std::fprintf(stderr, "[rt] reference splitting: %zu -> %zu items, box area %.4g -> %.4g (%.1f %% saved): %s\n", items.size(),
                     done.size(), before, after, before > 0 ? 100 * (before - after) / before : 0.0, keep ? "kept" : "dropped");
    if (keep) items.swap(done);
}

}  // namespace


// ---- 4-wide collapse ----
namespace {

struct Child4 {
    int32_t code;            // code in the binary tree (inner index, leaf code)
    double lo[3], hi[3];
};

float round_down(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}
double box_area(const Child4 &c)
{
    double e[3] = {c.hi[0] - c.lo[0], c.hi[1] - c.lo[1], c.hi[2] - c.lo[2]};
    return 2 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
}

int32_t collapse_node(const Bvh &b2, int32_t idx2, Bvh4 &out, int depth)
{
    out.max_depth = std::max(out.max_depth, depth);
    const int32_t me = static_cast<int32_t>(out.nodes.size());
    out.nodes.emplace_back();
    std::vector<Child4> kids;
    auto add_children_of = [&](int32_t n2) {
        const BvhNode &nd = b2.nodes[n2];
        for (int s = 0; s < 2; ++s) {
            if (nd.child[s] == -1) continue;
            Child4 c;
            c.code = nd.child[s];
            for (int k = 0; k < 3; ++k) { c.lo[k] = nd.lo[s][k]; c.hi[k] = nd.hi[s][k]; }
            kids.push_back(c);
        }
    };
    add_children_of(idx2);
    // open the inner child with the largest box while there is room for its two children
    while (kids.size() < 4) {
        int pick = -1;
        double best = -1;
        for (size_t i = 0; i < kids.size(); ++i)
            if (kids[i].code >= 0 && box_area(kids[i]) > best) { best = box_area(kids[i]); pick = static_cast<int>(i); }
        if (pick < 0) break;
        const int32_t open = kids[pick].code;
        kids.erase(kids.begin() + pick);
        add_children_of(open);
    }
    // The any-hit walk takes the last hit child first: put the largest box there, it is the likeliest
    // to hold an occluder (the nearest-hit walk orders children by entry distance and does not care).
    std::stable_sort(kids.begin(), kids.end(), [](const Child4 &a, const Child4 &b) { return box_area(a) < box_area(b); });
    Bvh4Node nd;
    std::memset(&nd, 0, sizeof nd);
    for (int c = 0; c < 4; ++c) {
        nd.child[c] = -1;
        for (int k = 0; k < 3; ++k) {
            nd.plane[k][0][c] = std::numeric_limits<float>::max();
            nd.plane[k][1][c] = -std::numeric_limits<float>::max();
        }
    }
    for (size_t c = 0; c < kids.size(); ++c) {
        for (int k = 0; k < 3; ++k) {
            nd.plane[k][0][c] = round_down(kids[c].lo[k]);
            nd.plane[k][1][c] = round_up(kids[c].hi[k]);
            out.max_abs_coord = std::max({out.max_abs_coord, std::fabs(kids[c].lo[k]), std::fabs(kids[c].hi[k])});
        }
        nd.child[c] = kids[c].code;   // leaf codes pass through; inner codes are rewritten below
    }
    out.nodes[me] = nd;
    for (size_t c = 0; c < kids.size(); ++c)
        if (kids[c].code >= 0) {
            const int32_t sub = collapse_node(b2, kids[c].code, out, depth + 1);
            out.nodes[me].child[c] = sub;
        }
    return me;
}

// Renumbers the nodes breadth-first, so that the first K nodes are the K nodes nearest the root:
// the device keeps a prefix of the array in on-chip memory, and the top of the tree is what
// every ray visits.
void breadth_first(Bvh4 &t)
{
    const size_t n = t.nodes.size();
    std::vector<int32_t> order;           // order[new] = old
    order.reserve(n);
    std::vector<int32_t> new_of(n, -1);
    order.push_back(0);
    for (size_t head = 0; head < order.size(); ++head) {
        const Bvh4Node &nd = t.nodes[order[head]];
        for (int c = 0; c < 4; ++c)
            if (nd.child[c] >= 0) order.push_back(nd.child[c]);
    }
    for (size_t i = 0; i < order.size(); ++i) new_of[order[i]] = static_cast<int32_t>(i);
    std::vector<Bvh4Node> out(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        Bvh4Node nd = t.nodes[order[i]];
        for (int c = 0; c < 4; ++c)
            if (nd.child[c] >= 0) nd.child[c] = new_of[nd.child[c]];
        out[i] = nd;
    }
    t.nodes.swap(out);
}

void collapse(const Bvh &b2, Bvh4 &out)
{
    out = Bvh4();
    if (b2.nodes.empty()) return;
    collapse_node(b2, 0, out, 1);
    breadth_first(out);
    // Stack bound: going down from a node with k children leaves at most k-1 of them pending.  Children
    // follow their parent in breadth-first order, so one backward sweep sees every child before its parent.
    std::vector<int32_t> need(out.nodes.size(), 0);
    for (size_t i = out.nodes.size(); i-- > 0;) {
        const Bvh4Node &nd = out.nodes[i];
        int k = 0, deepest = 0;
        for (int c = 0; c < 4; ++c) {
            if (nd.child[c] == -1) continue;
            ++k;
            if (nd.child[c] >= 0) deepest = std::max(deepest, need[nd.child[c]]);
        }
        need[i] = (k > 0 ? k - 1 : 0) + deepest;
    }
    out.max_stack = need[0];
}

}  // namespace

void build(const rt_scene_desc &d, Accel &out, std::vector<rt_tri_rec> &octree_tri, std::vector<int32_t> &octree_sph_key,
           const Options &opt)
{
    out = Accel();
    const int leaf_max = (opt.leaf_max >= 1 && opt.leaf_max <= 4) ? opt.leaf_max : kLeafMaxDefault;
    octree_tri.assign(d.tri, d.tri + d.n_tri_slots);
    octree_sph_key.assign(static_cast<size_t>(d.n_sph_slots), 0);

    // Tie-break key of a primitive: its first position in the reference's visiting order
    // (nodes in depth-first pre-order; within a node triangles, then spheres).
    std::vector<int32_t> first_key(static_cast<size_t>(d.n_prims), std::numeric_limits<int32_t>::max());
    std::vector<int32_t> first_tri_slot(static_cast<size_t>(d.n_prims), -1), first_sph_slot(static_cast<size_t>(d.n_prims), -1);
    int32_t rank = 0;
    for (int32_t n = 0; n < d.n_nodes; ++n) {
        const rt_node &nd = d.nodes[n];
        for (int32_t s = nd.tri_begin; s < nd.tri_begin + nd.tri_count; ++s, ++rank) {
            int32_t p = d.tri[s].prim;
            if (first_key[p] > rank) { first_key[p] = rank; first_tri_slot[p] = s; }
        }
        for (int32_t s = nd.sph_begin; s < nd.sph_begin + nd.sph_count; ++s, ++rank) {
            int32_t p = static_cast<int32_t>(d.sph_prim[s] & ~RT_SPH_TRANSPARENT);
            if (first_key[p] > rank) { first_key[p] = rank; first_sph_slot[p] = s; }
        }
    }
    for (auto &r : octree_tri) {
        r._pad0 = first_key[r.prim];
        r._pad1[0] = r.ab[1] * r.ac[2] - r.ac[1] * r.ab[2];
        r._pad1[1] = 0;
    }
    for (int32_t s = 0; s < d.n_sph_slots; ++s) octree_sph_key[s] = first_key[d.sph_prim[s] & ~RT_SPH_TRANSPARENT];

    // one item per distinct primitive that is resident somewhere in the octree
    std::vector<Item> tri_items, sph_items;
    for (int32_t p = 0; p < d.n_prims; ++p) {
        if (first_tri_slot[p] >= 0) {
            const rt_tri_rec &r = d.tri[first_tri_slot[p]];
            Item it;
            for (int k = 0; k < 3; ++k) {
                const double v0 = r.v0[k], v1 = r.v0[k] - r.ab[k], v2 = r.v0[k] - r.ac[k];
                it.lo[k] = std::min({v0, v1, v2});
                it.hi[k] = std::max({v0, v1, v2});
                it.c[k] = 0.5 * (it.lo[k] + it.hi[k]);
            }
            it.src = first_tri_slot[p];
            tri_items.push_back(it);
        } else if (first_sph_slot[p] >= 0) {
            const rt_sph_rec &r = d.sph[first_sph_slot[p]];
            Item it;
            for (int k = 0; k < 3; ++k) { it.lo[k] = r.center[k] - r.radius; it.hi[k] = r.center[k] + r.radius; it.c[k] = r.center[k]; }
            it.src = first_sph_slot[p];
            sph_items.push_back(it);
        }
    }
    // Geometry the single-precision boxes cannot bound safely -- a NaN or infinite coordinate (which the
    // reference's tests simply never hit, but which poisons every box above it), or one beyond 10^6 --
    // keeps the whole scene on the octree walk, whose arithmetic is the reference's in every such case.
    for (const std::vector<Item> *items : {&tri_items, &sph_items})
        for (const Item &it : *items)
            for (int k = 0; k < 3; ++k)
                if (!(std::fabs(it.lo[k]) < 1e6) || !(std::fabs(it.hi[k]) < 1e6)) return;   // usable stays false
    out.tri_on_device = opt.device_min > 0 && tri_items.size() >= 2 && (long long)tri_items.size() >= opt.device_min;
    out.sph_on_device = opt.device_min > 0 && sph_items.size() >= 2 && (long long)sph_items.size() >= opt.device_min;
    for (int k = 0; k < 3; ++k) { out.lo[k] = std::numeric_limits<double>::infinity(); out.hi[k] = -out.lo[k]; }
    for (const std::vector<Item> *items : {&tri_items, &sph_items})
        for (const Item &it : *items)
            for (int k = 0; k < 3; ++k) { out.lo[k] = std::min(out.lo[k], it.lo[k]); out.hi[k] = std::max(out.hi[k], it.hi[k]); }
    if (out.tri_on_device) {
        out.tri_src.reserve(tri_items.size());
        for (const Item &it : tri_items) out.tri_src.push_back(it.src);
    } else {
        const int split_budget = opt.split_budget >= 0 ? opt.split_budget : kSplitBudgetPercent;
        split_references(tri_items, d.tri, split_budget, opt.verbose);
        Builder(tri_items, out.tri_bvh, leaf_max).run();
        out.tri.reserve(tri_items.size());
        for (const Item &it : tri_items) out.tri.push_back(octree_tri[it.src]);
        collapse(out.tri_bvh, out.tri_bvh4);
    }
    if (out.sph_on_device) {
        out.sph_src.reserve(sph_items.size());
        for (const Item &it : sph_items) out.sph_src.push_back(it.src);
    } else {
        // spheres one to a leaf unless asked otherwise: a sphere's own box in the 4-wide node is a cheap single-precision test
        // in front of its FP64 test (balls.nff at 4096 x 4096: 12.0 -> 11.6 ms; triangles are best at two, 69.7 vs 74.2 ms at one)
        Builder(sph_items, out.sph_bvh, (opt.leaf_max >= 1 && opt.leaf_max <= 4) ? leaf_max : 1).run();
        out.sph.reserve(sph_items.size());
        for (const Item &it : sph_items) {
            out.sph.push_back(d.sph[it.src]);
            out.sph_prim.push_back(d.sph_prim[it.src]);
            out.sph_key.push_back(octree_sph_key[it.src]);
        }
        collapse(out.sph_bvh, out.sph_bvh4);
    }
    // a walk needs max_stack + 1 entries of its LDS stack
    out.usable = out.tri_bvh4.max_stack + 1 <= 64 && out.sph_bvh4.max_stack + 1 <= 64 &&
                 out.tri.size() < (1u << 28) && out.sph.size() < (1u << 28);
}

}  // namespace accel
