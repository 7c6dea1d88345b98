// Command-line host with the reference's interface (Driver.cpp:91-216):
//     rt_driver [--gpus G] [--rccl] [-s N] [-a A] [-p] input.nff output.ppm
// Parses the scene, builds the octree, renders through the C ABI -- on GPU 0, or with `--gpus G` as row bands over the
// first G GPUs of the node (rt_group: the multi-GPU form of the reference's two-thread row split, Driver.cpp:377; with
// `--rccl` the bands are gathered as RCCL sends and receives instead of stores through peer access) --
// and writes a binary PPM.  Unlike the reference (Driver.cpp:727) the output argument is honoured;
// pass `hide.ppm` to get the reference's file name.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/raytracer_amd.h"

static long long now_ms()
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != RT_OK) {                                             \
            std::fprintf(stderr, "%s failed: %s\n", #call, rt_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    int32_t num_samples = 1, phong = 0;
    double aperture = 0;
    char in_path[4096], out_path[4096];
    // `--gpus G` is this host's own switch: taken out before the reference's argument scanner sees the rest
    int gpus = 1;
    bool rccl = false;
    std::vector<const char *> args;
    for (int i = 0; i < argc; ++i) {
        if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) { gpus = std::atoi(argv[++i]); continue; }
        if (std::strcmp(argv[i], "--rccl") == 0) { rccl = true; continue; }
        args.push_back(argv[i]);
    }
    if (gpus < 1) { std::fprintf(stderr, "--gpus needs a positive number\n"); return 1; }
    CHECK(rth_parse_args((int)args.size(), args.data(), &num_samples, &aperture, &phong, in_path, sizeof in_path, out_path, sizeof out_path));
    std::printf("Arguments recieved:\nNum samples = %d\nAperture = %f\nPhong = %d\n", num_samples, aperture, phong);

    long long t1 = now_ms();
    rth_scene *scene = nullptr;
    CHECK(rth_scene_load(in_path, &scene));
    rt_scene_desc desc;
    CHECK(rth_scene_desc(scene, &desc));
    rt_camera cam;
    CHECK(rth_scene_camera(scene, 0, 0, &cam));
    long long t2 = now_ms();
    std::printf("Nff parsing, polygon construction and octree construction took %lld ms\n", t2 - t1);
    std::printf("There are %d total solids in the scene.\nThere are %d octrees!\n", desc.n_prims, desc.n_nodes);

    rt_ctx *ctx = nullptr;
    rt_group *group = nullptr;
    if (gpus > 1 || rccl) {
        int visible = 0;
        CHECK(rt_device_count(&visible));
        if (gpus > visible) { std::fprintf(stderr, "--gpus %d, but only %d GPU(s) are visible\n", gpus, visible); return 1; }
        std::vector<int> ids;
        for (int i = 0; i < gpus; ++i) ids.push_back(i);
        CHECK(rt_group_create(ids.data(), gpus, &group));
        if (rccl) CHECK(rt_group_set_option(group, "rccl", 1));
        CHECK(rt_group_upload_scene(group, &desc));
    } else {
        CHECK(rt_create(0, &ctx));
        CHECK(rt_upload_scene(ctx, &desc));
    }

    rt_render_opts opts = {};
    opts.num_samples = num_samples;
    opts.aperture = aperture;
    opts.flags = phong ? RT_FLAG_PHONG_NORMALS : 0u;
    opts.n_parts = 1;
    std::vector<double> pixels(static_cast<size_t>(cam.resx) * cam.resy * 3);
    rt_frame_out out = {};
    out.rgb_f64 = pixels.data();
    rt_stats stats;
    t1 = now_ms();
    if (group) CHECK(rt_group_render(group, &cam, &opts, &out, &stats));
    else CHECK(rt_render(ctx, &cam, &opts, &out, &stats));
    t2 = now_ms();
    std::printf("Render took %lld ms (GPU %.3f ms; %llu closest + %llu shadow rays, %.1f Mrays/s)\n", t2 - t1, stats.gpu_ms,
                (unsigned long long)stats.closest_rays, (unsigned long long)stats.shadow_rays,
                (stats.closest_rays + stats.shadow_rays) / (stats.gpu_ms * 1e3));
    CHECK(rth_write_ppm_f64(out_path, pixels.data(), cam.resx, cam.resy));
    rt_destroy(ctx);
    rt_group_destroy(group);
    rth_scene_free(scene);
    return 0;
}
