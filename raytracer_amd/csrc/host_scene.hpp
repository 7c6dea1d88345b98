// Host front end of the MI355X render path: .nff ingest, triangulation, octree
// construction and flattening into the arrays the device path consumes.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/raytracer_amd.h"

namespace rth {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct Primitive {
    bool sphere = false;
    bool has_vnormals = false;
    int material = 0;
    Vec3 v[3];       // triangle vertices
    Vec3 vn[3];      // patch vertex normals (normalised at ingest)
    Vec3 ab, ac, n;  // v0-v1, v0-v2, flat normal
    Vec3 c;          // sphere centre
    double r = 0;    // sphere radius
};

struct Scene {
    // file contents
    Vec3 background, from, at, up;
    double angle = 0, hither = 0;
    int resx = 0, resy = 0;
    int q_directive = 0;
    int lightcam_samples = 150;
    std::vector<Primitive> prims;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    // flattened octree
    std::vector<rt_node> nodes;
    std::vector<rt_tri_rec> tri;
    std::vector<rt_sph_rec> sph;
    std::vector<uint32_t> sph_prim;
    std::vector<uint32_t> prim_flags;
    std::vector<int32_t> prim_material;
    std::vector<double> prim_normals;

    bool load(const std::string &path, std::string &err);
    void build_octree();
    void fill_desc(rt_scene_desc &d) const;
    void camera(int rx, int ry, rt_camera &cam) const;
};

}  // namespace rth
