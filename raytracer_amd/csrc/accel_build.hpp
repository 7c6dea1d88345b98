// Internal acceleration structure derived from the flattened octree at upload time.
//
// The octree arrays of rt_scene_desc define WHAT a ray query returns (the reference's
// traversal).  For scenes where that result provably equals "nearest / any intersection
// over all primitives" (see accel_build.cpp), the device answers the same query through
// two bounding-volume hierarchies (one over triangles, one over spheres) whose leaves
// hold the same primitive records, so far fewer exact primitive tests are evaluated.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace accel {

// Two children per node, each with its own box; 128 bytes.
struct BvhNode {
    double lo[2][3];
    double hi[2][3];
    int32_t child[2];   // >= 0: index of the child node; -1: no child; else leaf code -2-(first*4+count-1)
    int32_t count[2];   // number of leaf records (0 for an inner child); informational
    int32_t _pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");

struct Bvh {
    std::vector<BvhNode> nodes;   // node 0 is the root; empty when there are no primitives
    int32_t max_depth = 0;
};

// The device walks a 4-wide form of the same tree: every node holds the boxes of up to four
// children as single-precision bounds rounded OUTWARD (lo down, hi up), so one 128-byte record
// answers two levels of the binary tree.  plane[axis][0] holds the four children's lower bounds
// on that axis, plane[axis][1] the upper bounds: a ray picks its near and far planes of an axis
// by the sign of its direction with one address bit (byte offset 32*axis + 16*side).  An unused
// child slot has an inverted box and code -1.  Codes as in BvhNode::child.
struct Bvh4Node {
    float plane[3][2][4];
    int32_t child[4];
    int32_t _pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

struct Bvh4 {
    std::vector<Bvh4Node> nodes;  // node 0 is the root; empty when there are no primitives
    int32_t max_depth = 0;        // in 4-wide nodes
    int32_t max_stack = 0;        // most entries a depth-first walk can have pending: siblings left behind on the way down
    double max_abs_coord = 0;     // largest |coordinate| of any box: scales the rounding pad of the slab test
};

struct Accel {
    Bvh tri_bvh, sph_bvh;
    Bvh4 tri_bvh4, sph_bvh4;
    std::vector<rt_tri_rec> tri;        // leaf-ordered; _pad0 = tie-break key, _pad1[0] = ab.y*ac.z - ac.y*ab.z
    std::vector<rt_sph_rec> sph;        // leaf-ordered
    std::vector<uint32_t> sph_prim;     // primitive id | RT_SPH_TRANSPARENT
    std::vector<int32_t> sph_key;       // tie-break key
    bool usable = false;                // false: keep to the octree walk
    // A hierarchy left to the device builder (lbvh_device.hpp; Options::device_min): its items as slots of the source
    // record arrays, in primitive order, and the box of all of them; the 2- and 4-wide trees and the leaf arrays above stay empty.
    bool tri_on_device = false, sph_on_device = false;
    std::vector<int32_t> tri_src, sph_src;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // box of every item of either kind (valid when anything is left to the device)
};

// Experiment knobs of the builder; the defaults are what the product uses.
struct Options {
    int leaf_max = 0;        // primitives per leaf, 1..4; 0 = default (2)
    int split_budget = -1;   // growth in percent allowed to reference splitting; < 0 = default (100), 0 = off
    bool verbose = false;    // splitter statistics on stderr
    long long device_min = 0;   // > 0: a hierarchy over at least this many primitives is left to the device builder
};

// Also fills the per-slot helper fields of `octree_tri` (copy of desc.tri): key and the
// ray-independent minor.  `octree_sph_key` receives one key per octree sphere slot.
void build(const rt_scene_desc &desc, Accel &out, std::vector<rt_tri_rec> &octree_tri,
           std::vector<int32_t> &octree_sph_key, const Options &opt = Options());

}  // namespace accel
