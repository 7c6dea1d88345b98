// Host side of the light-camera caustics path (LightCam.h, Octree<Vector3d*> of Octree.h:1347-1634):
// the point octree over the impact locations of one (transparent sphere, light) pair, flattened
// for the device's density query.
#pragma once
#include <cstdint>
#include <vector>

namespace lightcam {

struct Point {            // 32 bytes
    double x, y, z, w;    // impact position and weight coef*(1-T) (LightCam.h:533-537)
};

// One node of the point octree (Octree.h:1394-1497).  Its points are points[begin..end) of the
// re-ordered point array; children keep the reference's octant order, -1 = no child.
struct Node {             // 96 bytes
    double cx, cy, cz, size;
    int32_t begin, end;
    int32_t leaf;         // the reference's `leaf` flag (true unless every point went to a child)
    int32_t child[8];
    int32_t rest_begin;   // points[rest_begin..end) belong to no child (all of the node's points when it has none)
    double wsum;          // sum of the weights of points[begin..end)
    int32_t child_leaf;   // bit c: child c exists and its `leaf` flag is set
    int32_t _pad;
};
static_assert(sizeof(Node) == 96, "lightcam::Node must be 96 bytes");

struct Tree {
    std::vector<Node> nodes;     // node 0 = root
    std::vector<Point> points;   // in node order
};

// `pts` in the order the light rays produced them; `max_dist_from_zero` as LightCam.h:548.
void build(const std::vector<Point> &pts, double max_dist_from_zero, Tree &out);

// View frame of the light camera for one (sphere, light) pair: Camera.h:420-432 and
// LightCam::computeCoordinateSystem (LightCam.h:77-115; angle in radians, 1x1 "image").
struct Frame {
    double u[3], v[3], w[3], e[3];
    double d, l, t, pix_width;
};
void setup_frame(const double light_pos[3], const double sphere_pos[3], double sphere_radius, Frame &out);

// |p| with the association the reference's vector library uses, for maxDistFromZero.
double length3(double x, double y, double z);

}  // namespace lightcam
