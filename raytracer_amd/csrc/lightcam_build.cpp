// Point octree of a light camera (Octree<Vector3d*>, Octree.h:1394-1497), built by partitioning one
// array of points in place instead of filling per-node sets.
//
// Reference rules kept: root cube centred on the origin with side 2*maxDistFromZero+1
// (LightCam.h:312,325); a node subdivides iff size/2 > 1e-8 and it holds more than 10 points
// (Octree.h:1367-1368,1427); a point goes to the FIRST child (octant order 0..7) whose closed box
// contains it (Octree.h:1441-1446); `leaf` stays true unless every point found a child
// (Octree.h:1452-1453); every node keeps all of its points (Octree.h:1396), here as one
// contiguous range; a child's side is hi.x - lo.x of its box (Octree.h:1465).
#include "lightcam_build.hpp"

#include <cmath>
#include <cstring>
#include <future>

namespace lightcam {
namespace {

// The points themselves are kept in partition order (not an index list): every level reads and writes its
// ranges sequentially.
struct Work {
    std::vector<Point> pts, scratch;
    std::vector<uint8_t> which;
};

// Sub-trees with at least this many points are built by a thread of their own (down to kParallelDepth).
constexpr int32_t kParallelMin = 40000;
constexpr int kParallelDepth = 3;

// Builds the node of points [begin, end) and everything below it at the end of `nodes` (depth-first
// numbering: a node, then the sub-trees of its children in octant order) and returns its index.
int32_t build_node(Work &w, std::vector<Node> &nodes, double lox, double loy, double loz, double size, int32_t begin, int32_t end,
                   int depth)
{
    const int32_t me = static_cast<int32_t>(nodes.size());
    nodes.emplace_back();
    Node nd;
    std::memset(&nd, 0, sizeof nd);
    nd.size = size;
    nd.cx = lox + size / 2;
    nd.cy = loy + size / 2;
    nd.cz = loz + size / 2;
    nd.begin = begin;
    nd.end = end;
    nd.leaf = 1;
    nd.rest_begin = begin;
    for (int k = 0; k < 8; ++k) nd.child[k] = -1;
    const int32_t n = end - begin;
    if (size / 2 > .00000001 && n > 10) {
        double clo[8][3], chi[8][3];
        const double h = size / 2;
        for (int k = 0; k < 8; ++k) {
            clo[k][0] = (k & 1) ? nd.cx : nd.cx - size / 2;
            clo[k][1] = (k & 2) ? nd.cy : nd.cy - size / 2;
            clo[k][2] = (k & 4) ? nd.cz : nd.cz - size / 2;
            for (int d = 0; d < 3; ++d) chi[k][d] = clo[k][d] + h;
        }
        // "The first child, in octant order, whose closed box holds the point": the boxes are products of one
        // of two intervals per axis (octant bit 0 = x, 1 = y, 2 = z), so the first such child takes, on every
        // axis, the lower interval if it holds the coordinate, else the upper one; none if an axis has neither.
        const double lo0[3] = {clo[0][0], clo[0][1], clo[0][2]}, hi0[3] = {chi[0][0], chi[0][1], chi[0][2]};
        const double lo1[3] = {clo[7][0], clo[7][1], clo[7][2]}, hi1[3] = {chi[7][0], chi[7][1], chi[7][2]};
        int32_t cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int32_t i = begin; i < end; ++i) {
            const Point &p = w.pts[i];
            const double c[3] = {p.x, p.y, p.z};
            int where = 0;
            for (int d = 0; d < 3; ++d) {
                if (lo0[d] <= c[d] && c[d] <= hi0[d]) continue;
                if (lo1[d] <= c[d] && c[d] <= hi1[d]) where |= 1 << d;
                else { where = 8; break; }
            }
            w.which[i] = static_cast<uint8_t>(where);
            cnt[where]++;
        }
        if (cnt[8] == 0) nd.leaf = 0;
        int32_t off[9], pos[9];
        off[0] = begin;
        for (int k = 1; k < 9; ++k) off[k] = off[k - 1] + cnt[k - 1];
        std::memcpy(pos, off, sizeof pos);
        for (int32_t i = begin; i < end; ++i) w.scratch[pos[w.which[i]]++] = w.pts[i];   // stable
        std::memcpy(&w.pts[begin], &w.scratch[begin], sizeof(Point) * static_cast<size_t>(n));
        nd.rest_begin = off[8];
        bool threads = false;
        if (depth < kParallelDepth)
            for (int k = 0; k < 8; ++k) threads = threads || cnt[k] >= kParallelMin;
        if (!threads) {
            for (int k = 0; k < 8; ++k)
                if (cnt[k] > 0) {
                    nd.child[k] = build_node(w, nodes, clo[k][0], clo[k][1], clo[k][2], chi[k][0] - clo[k][0], off[k], off[k] + cnt[k], depth + 1);
                    if (nodes[nd.child[k]].leaf) nd.child_leaf |= 1 << k;
                }
        } else {
            // every child on a thread of its own, each into a private node list; the lists are appended in octant
            // order with their indices shifted once all are done, so the numbering is the sequential one
            std::vector<Node> sub[8];
            std::future<void> job[8];
            for (int k = 0; k < 8; ++k)
                if (cnt[k] > 0)
                    job[k] = std::async(cnt[k] >= kParallelMin / 8 ? std::launch::async : std::launch::deferred,
                                        [&w, &sub, &clo, &chi, &off, &cnt, k, depth]() {
                        sub[k].reserve(static_cast<size_t>(cnt[k]) / 2 + 16);     // a tree has about 0.4 nodes per point
                        build_node(w, sub[k], clo[k][0], clo[k][1], clo[k][2], chi[k][0] - clo[k][0], off[k], off[k] + cnt[k], depth + 1);
                    });
            for (int k = 0; k < 8; ++k) {
                if (cnt[k] == 0) continue;
                job[k].get();
                const int32_t shift = static_cast<int32_t>(nodes.size());
                for (Node s : sub[k]) {
                    for (int c = 0; c < 8; ++c) if (s.child[c] >= 0) s.child[c] += shift;
                    nodes.push_back(s);
                }
                nd.child[k] = shift;
                if (nodes[shift].leaf) nd.child_leaf |= 1 << k;
                std::vector<Node>().swap(sub[k]);
            }
        }
    }
    nodes[me] = nd;
    return me;
}

}  // namespace

void build(const std::vector<Point> &pts, double max_dist_from_zero, Tree &out)
{
    out.nodes.clear();
    out.points.clear();
    Work w;
    w.pts = pts;
    w.scratch.resize(pts.size());
    w.which.resize(pts.size());
    const double size = max_dist_from_zero * 2 + 1;
    build_node(w, out.nodes, 0 - size / 2, 0 - size / 2, 0 - size / 2, size, 0, static_cast<int32_t>(pts.size()), 0);
    out.points.swap(w.pts);
    // weight sum of every node's range (its points in range order), a slice of the node list per thread
    auto sums = [&out](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k) {
            Node &nd = out.nodes[k];
            double sum = 0;
            for (int32_t i = nd.begin; i < nd.end; ++i) sum += out.points[i].w;
            nd.wsum = sum;
        }
    };
    const size_t n_nodes = out.nodes.size();
    if (pts.size() < static_cast<size_t>(kParallelMin)) sums(0, n_nodes);
    else {
        // the top of the tree holds the long ranges: interleaved slices would balance better, contiguous ones
        // keep the code simple and the root's range is one pass either way
        constexpr size_t kSlices = 8;
        std::future<void> job[kSlices];
        for (size_t t = 0; t < kSlices; ++t)
            job[t] = std::async(std::launch::async, sums, n_nodes * t / kSlices, n_nodes * (t + 1) / kSlices);
        for (auto &j : job) j.get();
    }
}

namespace {
struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 unit(V3 a)
{
    double z = dot(a, a);
    if (z > 0) { double s = std::sqrt(z); return {a.x / s, a.y / s, a.z / s}; }
    return a;
}
}  // namespace

double length3(double x, double y, double z) { return std::sqrt((x * x + y * y) + z * z); }

void setup_frame(const double lp[3], const double sp[3], double radius, Frame &out)
{
    const V3 from = {lp[0], lp[1], lp[2]}, at = {sp[0], sp[1], sp[2]}, up = {0, .5, 1};
    const double distToSolid = std::sqrt(dot(sub(at, from), sub(at, from)));
    const double angleOfView = std::atan2(radius, distToSolid) * 2.82;
    const V3 w = unit(sub(from, at));
    const V3 u = unit(cross(up, w));
    const V3 v = unit(cross(w, u));
    const double aspectRatio = 1;            // resx / resy with a 1x1 image
    const double d = std::sqrt(dot(sub(from, at), sub(from, at)));
    const double distToEdge = d * std::tan(angleOfView / 2);
    const double pixWidth = 2 * distToEdge / 1;
    out.u[0] = u.x; out.u[1] = u.y; out.u[2] = u.z;
    out.v[0] = v.x; out.v[1] = v.y; out.v[2] = v.z;
    out.w[0] = w.x; out.w[1] = w.y; out.w[2] = w.z;
    out.e[0] = from.x; out.e[1] = from.y; out.e[2] = from.z;
    out.d = d;
    out.pix_width = pixWidth;
    out.l = -distToEdge + 0.5 * pixWidth;
    out.t = distToEdge * aspectRatio - 0.5 * pixWidth;
}

}  // namespace lightcam
