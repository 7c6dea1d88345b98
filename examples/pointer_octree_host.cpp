// A host that holds its scene the way the reference does -- primitives in a vector, the octree as nodes that point to
// their children and list their residents -- and renders through the C ABI of include/raytracer_amd.h.
//
// It is the compile-tested form of the change a maintainer would make to the reference's renderMultithreaded
// (Driver.cpp:355-390): flatten the pointer octree in depth-first pre-order with skip links, fill the per-primitive
// tables, copy the camera, call rt_render, write the image.  The classes below are this example's own stand-ins for the
// reference's Solid / Octree<Solid*> / Camera (only the fields the flattening reads); to have a real scene to work
// on, the pointer octree is inflated from the library's own front end and then flattened again by flatten_octree(),
// which must give back the same arrays.
//
//     pointer_octree_host input.nff output.ppm        exit code 0: arrays identical, frame written
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "raytracer_amd.h"

struct HostSolid {                 // Solid.h / Triangle.h / Sphere.h: what a primitive knows
    bool sphere = false;
    double v0[3] = {0, 0, 0}, ab[3] = {0, 0, 0}, ac[3] = {0, 0, 0};     // Triangle.h:40-46
    double center[3] = {0, 0, 0}, radius = 0;                          // Sphere.h:25-33
    double normals[12] = {0};      // flat normal, then the three vertex normals of a patch triangle
    bool has_vertex_normals = false;
    int material = 0;
    bool transparent = false;      // properties.T > 0
};

struct HostOctree {                // Octree<Solid*>, Octree.h:733-907
    double center[3] = {0, 0, 0}, radius = 0;         // boundingSphere
    int depth = 0;
    std::vector<int> solids;                          // indices into the primitive vector (the reference keeps pointers)
    std::unique_ptr<HostOctree> children[8];
};

// ---- the part a maintainer of the reference adds: pointer octree -> the arrays of rt_scene_desc ----
struct FlatScene {
    std::vector<rt_node> nodes;
    std::vector<rt_tri_rec> tri;
    std::vector<rt_sph_rec> sph;
    std::vector<uint32_t> sph_prim;
};

static void flatten_octree(const HostOctree *n, const std::vector<HostSolid> &solids, FlatScene &out)
{
    const size_t me = out.nodes.size();
    out.nodes.emplace_back();
    rt_node nd = {};
    for (int k = 0; k < 3; ++k) nd.center[k] = n->center[k];
    nd.radius = n->radius;
    nd.depth = n->depth;
    std::vector<int> res(n->solids);
    std::sort(res.begin(), res.end());                // fixes the order the reference's unordered_set leaves open
    nd.tri_begin = (int32_t)out.tri.size();
    for (int p : res)
        if (!solids[p].sphere) {
            rt_tri_rec r = {};
            for (int k = 0; k < 3; ++k) { r.v0[k] = solids[p].v0[k]; r.ab[k] = solids[p].ab[k]; r.ac[k] = solids[p].ac[k]; }
            r.prim = p;
            out.tri.push_back(r);
        }
    nd.tri_count = (int32_t)out.tri.size() - nd.tri_begin;
    nd.sph_begin = (int32_t)out.sph.size();
    for (int p : res)
        if (solids[p].sphere) {
            rt_sph_rec r = {{solids[p].center[0], solids[p].center[1], solids[p].center[2]}, solids[p].radius};
            out.sph.push_back(r);
            out.sph_prim.push_back((uint32_t)p | (solids[p].transparent ? RT_SPH_TRANSPARENT : 0u));
        }
    nd.sph_count = (int32_t)out.sph.size() - nd.sph_begin;
    out.nodes[me] = nd;
    for (int c = 0; c < 8; ++c)
        if (n->children[c]) flatten_octree(n->children[c].get(), solids, out);
    out.nodes[me].skip = (int32_t)out.nodes.size();   // first node that is not in this subtree
}

// ---- only for this example: a pointer octree to work on, inflated from the library's front end ----
static std::unique_ptr<HostOctree> inflate(const rt_scene_desc &d, int32_t &idx)
{
    const rt_node &nd = d.nodes[idx];
    std::unique_ptr<HostOctree> n(new HostOctree());
    for (int k = 0; k < 3; ++k) n->center[k] = nd.center[k];
    n->radius = nd.radius;
    n->depth = nd.depth;
    for (int32_t s = nd.tri_begin; s < nd.tri_begin + nd.tri_count; ++s) n->solids.push_back(d.tri[s].prim);
    for (int32_t s = nd.sph_begin; s < nd.sph_begin + nd.sph_count; ++s) n->solids.push_back((int)(d.sph_prim[s] & ~RT_SPH_TRANSPARENT));
    const int32_t end = nd.skip;
    ++idx;
    int c = 0;
    while (idx < end && c < 8) n->children[c++] = inflate(d, idx);      // child slots in order; which octant is not needed to flatten
    return n;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) != RT_OK) {                                               \
            std::fprintf(stderr, "%s failed: %s\n", #call, rt_last_error()); \
            return 2;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: pointer_octree_host input.nff output.ppm\n"); return 2; }
    rth_scene *scene = nullptr;
    CHECK(rth_scene_load(argv[1], &scene));
    rt_scene_desc lib;
    CHECK(rth_scene_desc(scene, &lib));

    // the host's own representation
    std::vector<HostSolid> solids((size_t)lib.n_prims);
    for (int32_t p = 0; p < lib.n_prims; ++p) {
        HostSolid &s = solids[p];
        s.sphere = (lib.prim_flags[p] & RT_PRIM_SPHERE) != 0;
        s.has_vertex_normals = (lib.prim_flags[p] & RT_PRIM_HAS_VNORMALS) != 0;
        s.material = lib.prim_material[p];
        s.transparent = lib.materials[s.material].T > 0;
        std::memcpy(s.normals, lib.prim_normals + (size_t)p * 12, sizeof s.normals);
    }
    for (int32_t k = 0; k < lib.n_tri_slots; ++k) {
        HostSolid &s = solids[lib.tri[k].prim];
        for (int c = 0; c < 3; ++c) { s.v0[c] = lib.tri[k].v0[c]; s.ab[c] = lib.tri[k].ab[c]; s.ac[c] = lib.tri[k].ac[c]; }
    }
    for (int32_t k = 0; k < lib.n_sph_slots; ++k) {
        HostSolid &s = solids[lib.sph_prim[k] & ~RT_SPH_TRANSPARENT];
        for (int c = 0; c < 3; ++c) s.center[c] = lib.sph[k].center[c];
        s.radius = lib.sph[k].radius;
    }
    int32_t idx = 0;
    std::unique_ptr<HostOctree> root = inflate(lib, idx);

    // what the reference-side patch does
    FlatScene flat;
    flatten_octree(root.get(), solids, flat);
    std::vector<uint32_t> prim_flags(solids.size());
    std::vector<int32_t> prim_material(solids.size());
    std::vector<double> prim_normals(solids.size() * 12);
    for (size_t p = 0; p < solids.size(); ++p) {
        prim_flags[p] = (solids[p].sphere ? RT_PRIM_SPHERE : 0u) | (solids[p].has_vertex_normals ? RT_PRIM_HAS_VNORMALS : 0u);
        prim_material[p] = solids[p].material;
        std::memcpy(&prim_normals[p * 12], solids[p].normals, sizeof solids[p].normals);
    }
    rt_scene_desc desc = lib;                       // materials, lights, background, Z: taken over as they are
    desc.nodes = flat.nodes.data();        desc.n_nodes = (int32_t)flat.nodes.size();
    desc.tri = flat.tri.data();            desc.n_tri_slots = (int32_t)flat.tri.size();
    desc.sph = flat.sph.data();            desc.n_sph_slots = (int32_t)flat.sph.size();
    desc.sph_prim = flat.sph_prim.data();
    desc.prim_flags = prim_flags.data();
    desc.prim_material = prim_material.data();
    desc.prim_normals = prim_normals.data();

    // the flattening must reproduce the front end's arrays
    bool same = desc.n_nodes == lib.n_nodes && desc.n_tri_slots == lib.n_tri_slots && desc.n_sph_slots == lib.n_sph_slots;
    for (int32_t i = 0; same && i < lib.n_nodes; ++i) {
        const rt_node &a = desc.nodes[i], &b = lib.nodes[i];
        same = std::memcmp(a.center, b.center, sizeof a.center) == 0 && a.radius == b.radius && a.skip == b.skip && a.tri_begin == b.tri_begin &&
               a.tri_count == b.tri_count && a.sph_begin == b.sph_begin && a.sph_count == b.sph_count && a.depth == b.depth;
    }
    for (int32_t i = 0; same && i < lib.n_tri_slots; ++i)
        same = std::memcmp(desc.tri[i].v0, lib.tri[i].v0, 9 * sizeof(double)) == 0 && desc.tri[i].prim == lib.tri[i].prim;
    for (int32_t i = 0; same && i < lib.n_sph_slots; ++i)
        same = std::memcmp(&desc.sph[i], &lib.sph[i], sizeof(rt_sph_rec)) == 0 && desc.sph_prim[i] == lib.sph_prim[i];
    if (!same) { std::fprintf(stderr, "flattened arrays differ from the front end's\n"); return 1; }

    rt_camera cam;
    CHECK(rth_scene_camera(scene, 0, 0, &cam));     // the reference copies camera.u/v/w/e, d, l, t, pixWidth, resx, resy (Camera.h:34-41)
    rt_ctx *ctx = nullptr;
    CHECK(rt_create(0, &ctx));
    CHECK(rt_upload_scene(ctx, &desc));
    rt_render_opts opts = {};
    opts.num_samples = 1;
    opts.n_parts = 1;
    std::vector<double> pixels((size_t)cam.resx * cam.resy * 3);        // the buffer of Driver.cpp:358
    rt_frame_out out = {};
    out.rgb_f64 = pixels.data();
    rt_stats stats;
    CHECK(rt_render(ctx, &cam, &opts, &out, &stats));
    CHECK(rth_write_ppm_f64(argv[2], pixels.data(), cam.resx, cam.resy));   // writeImage, Driver.cpp:712-733
    std::printf("%d solids, %d octree nodes flattened; %llu closest + %llu shadow rays in %.3f ms\n", desc.n_prims, desc.n_nodes,
                (unsigned long long)stats.closest_rays, (unsigned long long)stats.shadow_rays, stats.gpu_ms);
    rt_destroy(ctx);
    rth_scene_free(scene);
    return 0;
}
