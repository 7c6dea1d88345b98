// Point octree of a light camera (Octree<Vector3d*>, Octree.h:1394-1497), built breadth-by-depth
// over an index permutation instead of per-node sets.
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

namespace lightcam {
namespace {

struct Work {
    const std::vector<Point> *src;
    std::vector<int32_t> perm, scratch;
    std::vector<uint8_t> which;
    std::vector<Node> *nodes;
};

int32_t build_node(Work &w, double lox, double loy, double loz, double size, int32_t begin, int32_t end)
{
    const int32_t me = static_cast<int32_t>(w.nodes->size());
    w.nodes->emplace_back();
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
        int32_t cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int32_t i = begin; i < end; ++i) {
            const Point &p = (*w.src)[w.perm[i]];
            int where = 8;
            for (int k = 0; k < 8; ++k)
                if (clo[k][0] <= p.x && p.x <= chi[k][0] && clo[k][1] <= p.y && p.y <= chi[k][1] && clo[k][2] <= p.z &&
                    p.z <= chi[k][2]) {
                    where = k;
                    break;
                }
            w.which[i] = static_cast<uint8_t>(where);
            cnt[where]++;
        }
        if (cnt[8] == 0) nd.leaf = 0;
        int32_t off[9], pos[9];
        off[0] = begin;
        for (int k = 1; k < 9; ++k) off[k] = off[k - 1] + cnt[k - 1];
        std::memcpy(pos, off, sizeof pos);
        for (int32_t i = begin; i < end; ++i) w.scratch[pos[w.which[i]]++] = w.perm[i];   // stable
        for (int32_t i = begin; i < end; ++i) w.perm[i] = w.scratch[i];
        nd.rest_begin = off[8];
        for (int k = 0; k < 8; ++k)
            if (cnt[k] > 0) {
                nd.child[k] = build_node(w, clo[k][0], clo[k][1], clo[k][2], chi[k][0] - clo[k][0], off[k], off[k] + cnt[k]);
                if ((*w.nodes)[nd.child[k]].leaf) nd.child_leaf |= 1 << k;
            }
    }
    (*w.nodes)[me] = nd;
    return me;
}

}  // namespace

void build(const std::vector<Point> &pts, double max_dist_from_zero, Tree &out)
{
    out.nodes.clear();
    out.points.clear();
    Work w;
    w.src = &pts;
    w.perm.resize(pts.size());
    w.scratch.resize(pts.size());
    w.which.resize(pts.size());
    w.nodes = &out.nodes;
    for (size_t i = 0; i < pts.size(); ++i) w.perm[i] = static_cast<int32_t>(i);
    const double size = max_dist_from_zero * 2 + 1;
    build_node(w, 0 - size / 2, 0 - size / 2, 0 - size / 2, size, 0, static_cast<int32_t>(pts.size()));
    out.points.resize(pts.size());
    for (size_t i = 0; i < pts.size(); ++i) out.points[i] = pts[w.perm[i]];
    for (Node &nd : out.nodes) {
        double sum = 0;
        for (int32_t i = nd.begin; i < nd.end; ++i) sum += out.points[i].w;
        nd.wsum = sum;
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
