// MI355X (gfx950) render path, host side: the C ABI of include/raytracer_amd.h (contexts, scene
// upload, the light cameras' set-up, and the launch sequence of one frame).
//
// A frame is rendered in batches of primary samples; a batch runs one generation of the ray tree
// per launch: k_closest(g) -> k_shadow(g) [-> k_lightcam(g)] for g = 0..5, rays and hits travelling
// between the passes in sharded queues in HBM, then k_finalize.  The kernels are in
// render_kernels.hpp, the queries they make in octree_walk.hpp (the reference's own traversal over
// the flattened octree) and hierarchy_walk.hpp (the equivalent 4-wide hierarchies).
//
// Arithmetic is FP64 in the reference's evaluation order; this file is compiled with
// -ffp-contract=off so that every +,-,*,/ and sqrt is the same IEEE operation the CPU
// performs.  Only the power of the specular term (spec_pow in shading.hpp, a few ulps from pow()) and the order in
// which a pixel's contributions are summed can differ from a CPU evaluation.
#include <hip/hip_runtime.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"
#include "accel_build.hpp"
#include "lightcam_build.hpp"
#include "render_kernels.hpp"
#include "lightcam_device.hpp"
#include "viewgrid.hpp"
#include "lbvh_device.hpp"
#include "lightcam_paths.hpp"
#include <hipcub/hipcub.hpp>

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string g_error;
int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}
}  // namespace

extern "C" void rt_set_error_string(const char *msg) { g_error = msg ? msg : ""; }
extern "C" const char *rt_last_error(void) { return g_error.c_str(); }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// ------------------------------------------------------------------ static scratch
// Work memory of rt_render_noalloc, the entry point for callers that cannot allocate: the accumulation frame, the
// queue counters and one batch's ray, hit and deferred-query queues are carved out of this array instead of
// device allocations.  It is part of the code object, so every GPU the library runs on holds its own copy until
// the process ends.  In host code the name is only a handle: rt_create asks the runtime for the array's address
// on its device (hipGetSymbolAddress) and everything else -- kernel arguments, memsets -- uses that pointer.
#define RT_NOALLOC_SCRATCH_BYTES (128u << 20)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
#define RT_MAX_DEVICES 64
namespace {
std::mutex g_scratch_mutex;
void *g_scratch_ptr[RT_MAX_DEVICES] = {};       // per device: address of that device's copy of `scratch`
}  // namespace

// Queue memory one context may take unless rt_set_memory_budget says otherwise.  The 8K x 16 spp frame by budget
// (tools/exp_budget.py): 8 GiB 77.8 ms (32 batches), 16 GiB 75.2 (16), 32 GiB 73.7 (8), 48 GiB 72.5 (4), 96 GiB 70.6 (2):
// every batch pays for the ramp and the tail of its launches again.  The default keeps to a ninth of the card.
#define RT_TOTALS_WORDS 16     // per-frame statistics rolled up on the device (k_collect_counts)
#define RT_DEFAULT_QUEUE_BUDGET ((size_t)32 << 30)
// Frames of at most this many samples run as one launch per batch (k_fused) instead of the wavefront pipeline.
#define RT_FUSED_MAX_SAMPLES (1ll << 20)
#define RT_LOCAL_MAX_SAMPLES (1ll << 22)     // frames up to this many samples run as one launch of self-contained blocks (k_local)

// Experiment switches (rt_debug_set_option); the defaults are what the product runs.
struct DebugOptions {
    int block_threads = 0;          // 256 or 1024 overrides the per-frame choice
    long long node_cache = -1;      // cap on hierarchy nodes copied into LDS
    long long tri_cache = -1;       // cap on leaf triangle records copied into LDS
    int rounds = 0;                 // rounds of resident blocks a queued pass is cut into (default 2)
    bool verbose = false;           // device properties, light-camera and splitter statistics on stderr
    int leaf_max = 0, split_budget = -1;   // hierarchy builder (accel::Options), read at the next upload
    int direct = 1;                 // the primary pass of a large batch writes its records at places that follow from the chunk's number (0: reserved from counters)
    int stack_min = 0;              // > 0: the walks' stacks get at least this many entries (to run a small scene the way a deep hierarchy runs)
    int lightcam_digits = 0;        // > 0: the two-sort builder's keys hold at most this many levels (a deeper tree is then built level by level)
    int lightcam_levels = 0;        // 1: the device builds the light cameras' trees level by level (lightcam_device.hpp), 0: with two sorts (lightcam_paths.hpp)
    int lightcam_host = 0;          // 1: the light cameras' point octrees are built by the host builder (lightcam_build.cpp)
    int fused = -1;                 // 1: always the one-launch kernel, 0: never, -1: by frame size
    int local_threads = 1024;       // block size of the block-local kernel (256, 512 or 1024)
    int local_min_chunks = 0;       // primary chunks a block of the block-local kernel gets at least (0: four)
    int whole_chunks = 2;           // the closest-hit passes of so many generations push whole chunks of records (0: packed)
    int local_rotate = -1;          // places a block's share of the primary chunks moves on per round (-1: chosen from the image width)
    int local = -1;                 // 1: always the block-local one-launch kernel where it applies, 0: never, -1: by frame size
    int units = 1;                  // 0: the primary pass never shares its chunks out on the device (every block keeps to a fixed share)
    int overlap = -1;               // 1: shadow passes on the second stream beside the next closest-hit pass, 0: one stream, -1: by batch size
    int pair = -1;                  // 1: shadow pass g and closest-hit pass g+1 in one launch, 0: a launch each, -1: default (paired)
    int device_build_stack = BVH_STACK - 1;   // a device-built hierarchy whose walk could need more pending entries is built again on the host
    long long device_build_min = 50000;   // hierarchies over at least this many primitives are built on the device (0: never); read at the next upload
    int grid = 1;                   // 0: no grids, every query walks the hierarchies; 1: lights' and eye's grids (viewgrid.hpp); 2: lights only; 3: eye only
    int grid_kernels = 1;           // 0: the passes that read the grids run in the general kernels (1024-thread blocks)
    int fuse_lights = 1;            // 0: the primary pass writes hit records for a shadow pass of its own even where it could shade them itself
    int grid_res = 0;               // bins per edge of a face of a light's cube map (0: default)
    long long grid_min_samples = -1; // the eye's grid is made for frames of at least this many samples (-1: default)
    int ring_poison = 0;            // 1: before a primary pass that hands out units, the LDS ring words of every compute unit are filled with
                                    //    matching tags and wrong unit numbers (k_poison_ring): a deterministic form of what stale LDS can hold
    int lightcam_pool = LC_POOL;    // places used of a wave's pool of pending cells in the light-camera density pass (128..LC_POOL)
};

// ------------------------------------------------------------------ host side
struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // scene
    DevScene S{};
    std::vector<void *> scene_allocs;
    bool has_scene = false;
    bool any_transparent = false;          // any material with T > 0
    bool any_secondary = false;            // any material with Ks != 0 or T > 0: ray trees go beyond the first hit
    bool any_transparent_sphere = false;
    bool any_sphere = false, any_vnormals = false;
    bool spheres_inside_root = true;       // every sphere lies inside the root cube
    bool accel_usable = false;
    long long n_btri = 0;                  // leaf triangle records of the triangle hierarchy             // hierarchies built and shallow enough for the LDS stack
    int accel_stack = 0;                   // deepest hierarchy, in nodes: the stack entries a ray can need
    // host copies needed to set up the light cameras
    std::vector<rt_light> h_lights;
    std::vector<std::array<double, 4>> h_lc_spheres;   // centre + radius of every transparent sphere, by rank
    std::vector<void *> lc_allocs;                     // node and point arrays of the cameras that exist
    void *lc_ws = nullptr; size_t lc_ws_bytes = 0;     // work space of the device-side tree builder (grow-only)
    void *lc_photon_ws = nullptr; size_t lc_photon_ws_bytes = 0;     // ... of the photon passes
    lightcam::Point *lc_packed = nullptr; size_t lc_packed_bytes = 0, lc_packed_points = 0;   // ... their packed impacts
    DevLightCam *lc_table = nullptr;                   // one entry per (transparent sphere, light) pair
    uint32_t *lc_needed = nullptr;                     // per pair: raised by the shading passes
    std::vector<uint8_t> lc_have;                      // per pair: the camera exists
    std::vector<uint32_t> h_needed;
    size_t lc_n_built = 0;
    uint32_t lc_flags = 0;                             // the normal-interpolation mode the cameras were set up for
    unsigned long long lc_rays = 0;
    int32_t lightcam_samples = 150;
    // frame resources (grow-only)
    double *accum = nullptr; size_t accum_values = 0;
    RayRec *rays[2] = {nullptr, nullptr}; size_t ray_capacity[2] = {0, 0};
    HitRec *hits[4] = {nullptr, nullptr, nullptr, nullptr}; size_t hit_capacity[4] = {0, 0, 0, 0};
    HitRec *deferred[6] = {}; size_t deferred_capacity[6] = {};   // one per generation: a batch's light-camera queries run in one launch at its end
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_c[6] = {}, ev_s[6] = {}, ev_batch[2] = {};
    uint32_t *counts = nullptr;
    unsigned long long *totals = nullptr;
    unsigned long long *h_totals = nullptr;     // pinned: the totals of the frame in flight arrive here with the frame's last command
    const double *accum_clean_ptr = nullptr;
    size_t accum_clean = 0;                     // leading values of `accum` known to be zero (k_finalize leaves them so)
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr}; size_t stage_bytes[4] = {0, 0, 0, 0};
    int64_t batch_override = 0;
    size_t queue_budget = RT_DEFAULT_QUEUE_BUDGET;
    int growth = 2;                        // ray queues hold this many times the first generation's capacity (plan_queues)
    void *scratch = nullptr;               // this device's copy of the static scratch array
    DebugOptions dbg;
    // pending async frame
    unsigned long long *pending_totals = nullptr;
    bool pending_fused = false;            // the frame in flight ran as k_fused launches
    bool pending = false;
    uint32_t pending_flags = 0;
    unsigned long long pending_primaries = 0;
    int pending_batches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_closest, ev_shadow, ev_pair;
    size_t ev_closest_used = 0, ev_shadow_used = 0, ev_pair_used = 0;
    int pending_closest_per_batch = 6;     // closest-hit launches per batch of the frame in flight (1 when passes are paired)
    // grids of what a point can see (viewgrid.hpp)
    int32_t *grid_tri_index = nullptr; int32_t n_grid_tri = 0, n_grid_sph = 0;   // the leaf triangle records to enter: one per primitive
    vg::Bin *lg_bins = nullptr; vg::Entry *lg_entries = nullptr;      // the lights' grids (made at upload; owned by the scene)
    vg::Bin *cg_bins = nullptr; vg::Entry *cg_entries = nullptr;      // the eye's grid: made when a frame's camera differs from the last one's
    size_t cg_bins_cap = 0, cg_entries_words = 0;
    rt_camera cg_cam{};
    bool cg_valid = false;
    bool cg_refused = false;              // the grid of camera cg_cam would have lists too long to be worth reading
    rt_camera last_cam{}; bool last_cam_valid = false;        // the camera of the context's last frame of this scene
    int32_t cg_tiles_x = 0;
    void *grid_ws = nullptr; size_t grid_ws_bytes = 0;        // work space of the grid builder (grow-only)
    void *direct_ws = nullptr; size_t direct_ws_bytes = 0;    // per-chunk counts and group lists of a primary pass with direct placement (grow-only)
    int cu_count = 256;
    int32_t n_materials = 0;
    size_t lds_per_cu = 64 * 1024;   // LDS of one compute unit (maxSharedMemoryPerMultiProcessor)
};

static void free_lightcams(rt_ctx *ctx)
{
    for (void *p : ctx->lc_allocs) (void)hipFree(p);
    ctx->lc_allocs.clear();
    (void)hipFree(ctx->lc_table); ctx->lc_table = nullptr;
    (void)hipFree(ctx->lc_needed); ctx->lc_needed = nullptr;
    ctx->lc_have.clear();
    ctx->lc_n_built = 0;
    ctx->S.lc_table = nullptr;
    ctx->S.lc_needed = nullptr;
    ctx->lc_rays = 0;
}

static int free_scene(rt_ctx *ctx)
{
    free_lightcams(ctx);
    (void)hipFree(ctx->lg_bins); ctx->lg_bins = nullptr;
    (void)hipFree(ctx->lg_entries); ctx->lg_entries = nullptr;
    ctx->grid_tri_index = nullptr;      // (one of scene_allocs)
    ctx->cg_valid = false;
    ctx->cg_refused = false;
    ctx->last_cam_valid = false;
    for (void *p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->has_scene = false;
    return RT_OK;
}

template <typename T>
static int upload(rt_ctx *ctx, const T *src, size_t n, const T **dst)
{
    *dst = nullptr;
    size_t bytes = sizeof(T) * (n ? n : 1);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    ctx->scene_allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, sizeof(T) * n, hipMemcpyHostToDevice));
    *dst = static_cast<const T *>(p);
    return RT_OK;
}

// The jitter of a sample is (rand() % 1000) / 1000.0 (Camera.h:99-103); the kernels form that quotient without dividing
// (thousandth() in shading.hpp), which is exact for each of the thousand numerators: checked here, once per context.
static bool rand_quotients_exact()
{
    for (int k = 0; k < 1000; ++k) {
        volatile double num = (double)k, den = 1000.0;
        const double by_division = num / den;
        if (thousandth((double)k) != by_division) return false;
    }
    return true;
}

extern "C" int rt_device_count(int *count)
{
    if (!count) return fail(RT_ERR_INVALID, "rt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

static int create_resources(rt_ctx *ctx, int device_id, const hipDeviceProp_t &prop);

extern "C" int rt_create(int device_id, rt_ctx **out)
{
    if (!out) return fail(RT_ERR_INVALID, "rt_create: null argument");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(RT_ERR_DEVICE, "rt_create: no HIP device visible; this library has no CPU path");
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID, "rt_create: device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device_id;
    // whatever has been created when a later step fails is given back (rt_destroy takes a half-made context)
    const int rc = create_resources(ctx, device_id, prop);
    if (rc != RT_OK) {
        const std::string msg = g_error;
        rt_destroy(ctx);
        return fail(rc, msg);
    }
    *out = ctx;
    return RT_OK;
}

static int create_resources(rt_ctx *ctx, int device_id, const hipDeviceProp_t &prop)
{
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    for (int k = 0; k < 6; ++k) {
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_c[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ctx->ev_s[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_batch[k], hipEventDisableTiming));
    if (!rand_quotients_exact()) return fail(RT_ERR_DEVICE, "rt_create: k / 1000 by fused multiply-adds differs from the division");
    HIP_TRY(hipEventCreate(&ctx->ev_begin));
    HIP_TRY(hipEventCreate(&ctx->ev_end));
    // the frame's totals, then the two counter blocks, in one allocation: a frame's first memset clears the totals and
    // the first batch's counters together
    HIP_TRY(hipMalloc((void **)&ctx->totals, RT_TOTALS_WORDS * sizeof(unsigned long long) + 2 * CNT_WORDS * sizeof(uint32_t)));
    ctx->counts = reinterpret_cast<uint32_t *>(ctx->totals + RT_TOTALS_WORDS);
    HIP_TRY(hipHostMalloc((void **)&ctx->h_totals, RT_TOTALS_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    {
        // a block of the traversal kernels may use all of a compute unit's LDS
#ifdef RT_STAMP
        const int lds_max = (int)ctx->lds_per_cu - 16 * 32 * 8;     // (the stamps' own LDS is static)
#else
        const int lds_max = (int)ctx->lds_per_cu;
#endif
        const void *kernels[] = {(const void *)k_closest<true, true>,  (const void *)k_closest<true, false>, (const void *)k_closest<false, true>,
                                 (const void *)k_closest<false, false>, (const void *)k_shadow<true>,       (const void *)k_shadow<false>,
                                 (const void *)k_shadow_hybrid,
                                 (const void *)k_pair<true, SHADOW_BVH>, (const void *)k_pair<true, SHADOW_HYBRID>,
                                 (const void *)k_pair<true, SHADOW_EXACT>, (const void *)k_pair<false, SHADOW_EXACT>,
                                 (const void *)k_local<SHADOW_BVH>, (const void *)k_local<SHADOW_HYBRID>,
                                 (const void *)k_fused<true, SHADOW_BVH>, (const void *)k_fused<true, SHADOW_HYBRID>,
                                 (const void *)k_fused<true, SHADOW_EXACT>, (const void *)k_fused<false, SHADOW_EXACT>,
                                 (const void *)k_poison_ring, (const void *)k_closest<true, true, true>, (const void *)k_closest<true, true, true, 1>,
                                 (const void *)k_closest<true, true, true, 2>, (const void *)k_shadow_grid};
        for (const void *k : kernels) HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    }
    if (device_id < RT_MAX_DEVICES) {
        // once per device, with that device current: where this GPU keeps its copy of the static scratch
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_id]) {
            void *p = nullptr;
            HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device_id] = p;
        }
        ctx->scratch = g_scratch_ptr[device_id];
    }
    return RT_OK;
}

extern "C" int rt_destroy(rt_ctx *ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    free_scene(ctx);
    (void)hipFree(ctx->accum);
    (void)hipFree(ctx->rays[0]);
    (void)hipFree(ctx->rays[1]);
    for (HitRec *h : ctx->hits) (void)hipFree(h);
    for (int k = 0; k < 6; ++k) (void)hipFree(ctx->deferred[k]);
    for (int k = 0; k < 6; ++k) { (void)hipEventDestroy(ctx->ev_c[k]); (void)hipEventDestroy(ctx->ev_s[k]); }
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ctx->ev_batch[k]);
    (void)hipStreamDestroy(ctx->stream2);

    (void)hipFree(ctx->totals);      // (the counters are part of it)
    (void)hipHostFree(ctx->h_totals);
    (void)hipFree(ctx->lc_ws);
    (void)hipFree(ctx->lc_photon_ws);
    (void)hipFree(ctx->lc_packed);
    (void)hipFree(ctx->cg_bins);
    (void)hipFree(ctx->cg_entries);
    (void)hipFree(ctx->grid_ws);
    (void)hipFree(ctx->direct_ws);
    for (void *p : ctx->stage) (void)hipFree(p);
    for (auto &pr : ctx->ev_closest) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : ctx->ev_shadow) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : ctx->ev_pair) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    (void)hipEventDestroy(ctx->ev_begin);
    (void)hipEventDestroy(ctx->ev_end);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

extern "C" int rt_debug_intersect(rt_ctx *ctx, int32_t kind, int32_t n, const double *geom, const double *rays, double minT,
                                  double maxT, double *out)
{
    if (!ctx || !geom || !rays || !out || n < 0 || kind < 0 || kind > 7) return fail(RT_ERR_INVALID, "rt_debug_intersect: bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t gdoubles = (size_t)n * ((kind <= 1 || kind == 4 || kind == 5) ? 9 : 4);
    double *d_geom = nullptr, *d_rays = nullptr, *d_out = nullptr;
    int rc = RT_OK;
    if (hipMalloc((void **)&d_geom, gdoubles * 8) != hipSuccess || hipMalloc((void **)&d_rays, (size_t)n * 48) != hipSuccess ||
        hipMalloc((void **)&d_out, (size_t)n * 32) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: hipMalloc failed");
    if (rc == RT_OK && (hipMemcpy(d_geom, geom, gdoubles * 8, hipMemcpyHostToDevice) != hipSuccess ||
                        hipMemcpy(d_rays, rays, (size_t)n * 48, hipMemcpyHostToDevice) != hipSuccess))
        rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: upload failed");
    if (rc == RT_OK) {
        hipLaunchKernelGGL(k_debug_intersect, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)kind, (int)n,
                           d_geom, d_rays, minT, maxT, d_out);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(out, d_out, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(RT_ERR_DEVICE, "rt_debug_intersect: kernel or download failed");
    }
    (void)hipFree(d_geom); (void)hipFree(d_rays); (void)hipFree(d_out);
    return rc;
}

#ifdef RT_COUNT
// Diagnostic builds: totals since the last call: 3 groups (primary, queued closest, shadow) x 8 words,
// then 8 words of the light-camera queries.
extern "C" int rt_debug_counters(unsigned long long *out48)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_walk_counts), 48 * sizeof(unsigned long long)));
    unsigned long long zero[48] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_counts), zero, sizeof zero));
    return RT_OK;
}
#endif

#ifdef RT_STAMP
// Diagnostic builds: the clock stamps (100 MHz ticks) the waves of the last traversal launches left; cleared by the call.
extern "C" int rt_debug_stamp_select(int kind_times_8_plus_generation)
{
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_select), &kind_times_8_plus_generation, sizeof(int)));
    return RT_OK;
}
extern "C" int rt_debug_stamps(unsigned long long *out, int n_words)
{
    HIP_TRY(hipDeviceSynchronize());
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n_words < RT_STAMP_WAVES * RT_STAMP_SLOTS ? n_words : RT_STAMP_WAVES * RT_STAMP_SLOTS);
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), bytes));
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)));
    HIP_TRY(hipMemset(p, 0, sizeof(unsigned long long) * RT_STAMP_WAVES * RT_STAMP_SLOTS));
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamp_seen)));
    HIP_TRY(hipMemset(p, 0, sizeof(unsigned int) * RT_STAMP_WAVES * RT_STAMP_SLOTS));
    return RT_OK;
}
#endif

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
extern "C" const char *rt_build_id(void) { return RT_BUILD_ID; }

extern "C" int rt_set_batch(rt_ctx *ctx, int64_t primaries)
{
    if (!ctx || primaries < 0) return fail(RT_ERR_INVALID, "rt_set_batch: bad argument");
    ctx->batch_override = primaries;
    return RT_OK;
}

// Validates every index the kernels will follow, so that a malformed scene is an error
// code here and never an out-of-bounds access on the GPU.
static int validate_scene(const rt_scene_desc *s)
{
    if (!s->nodes || s->n_nodes <= 0) return fail(RT_ERR_INVALID, "scene: no nodes");
    if (s->n_tri_slots < 0 || s->n_sph_slots < 0 || s->n_prims < 0 || s->n_materials < 0 || s->n_lights < 0)
        return fail(RT_ERR_INVALID, "scene: negative count");
    if ((s->n_tri_slots && !s->tri) || (s->n_sph_slots && (!s->sph || !s->sph_prim)))
        return fail(RT_ERR_INVALID, "scene: missing primitive records");
    if (s->n_prims && (!s->prim_flags || !s->prim_material || !s->prim_normals))
        return fail(RT_ERR_INVALID, "scene: missing per-primitive arrays");
    if ((s->n_materials && !s->materials) || (s->n_lights && !s->lights)) return fail(RT_ERR_INVALID, "scene: missing materials or lights");
    for (int32_t i = 0; i < s->n_nodes; ++i) {
        const rt_node &n = s->nodes[i];
        if (n.skip <= i || n.skip > s->n_nodes) return fail(RT_ERR_INVALID, "scene: node skip link out of order");
        if (n.tri_count < 0 || n.tri_begin < 0 || (int64_t)n.tri_begin + n.tri_count > s->n_tri_slots)
            return fail(RT_ERR_INVALID, "scene: node triangle range out of bounds");
        if (n.sph_count < 0 || n.sph_begin < 0 || (int64_t)n.sph_begin + n.sph_count > s->n_sph_slots)
            return fail(RT_ERR_INVALID, "scene: node sphere range out of bounds");
    }
    for (int32_t i = 0; i < s->n_tri_slots; ++i)
        if (s->tri[i].prim < 0 || s->tri[i].prim >= s->n_prims || (s->prim_flags[s->tri[i].prim] & RT_PRIM_SPHERE))
            return fail(RT_ERR_INVALID, "scene: triangle slot names a bad primitive");
    for (int32_t i = 0; i < s->n_sph_slots; ++i) {
        uint32_t p = s->sph_prim[i] & ~RT_SPH_TRANSPARENT;
        if (p >= (uint32_t)s->n_prims || !(s->prim_flags[p] & RT_PRIM_SPHERE))
            return fail(RT_ERR_INVALID, "scene: sphere slot names a bad primitive");
    }
    for (int32_t i = 0; i < s->n_prims; ++i)
        if (s->prim_material[i] < 0 || s->prim_material[i] >= s->n_materials)
            return fail(RT_ERR_INVALID, "scene: primitive material out of range");
    return RT_OK;
}

static int build_light_grids(rt_ctx *ctx, DevScene &D, int n_sph_records);

// ------------------------------------------------------------------ hierarchy build on the device (lbvh_device.hpp)
template <typename T>
static int scene_alloc(rt_ctx *ctx, size_t n, T **out)
{
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, sizeof(T) * (n ? n : 1)));
    ctx->scene_allocs.push_back(p);
    *out = static_cast<T *>(p);
    return RT_OK;
}

// Twelve levels of three pending siblings each: 36 stack entries and one more, which a compute unit's LDS holds for 1024 threads
// beside the tables.
#define LBVH_LEVELS_WANTED 12
struct DeviceTree {
    const accel::Bvh4Node *nodes = nullptr;
    int32_t n_nodes = 0, max_stack = 0, levels = 0;
};

// Builds the 4-wide hierarchy over the n items `src` (slots of the resident record array `tri` or `sph`) and leaves the order
// of the leaves (sorted position -> item) in *order_out (device memory inside the returned work buffer, which the caller frees
// after gathering the leaf records).
static int lbvh_build(rt_ctx *ctx, hipStream_t st, bool is_tri, const rt_tri_rec *tri, const rt_sph_rec *sph, const std::vector<int32_t> &src,
                      int leaf_max, const lbvh::Bounds &sb, DeviceTree &out, void **work_out, const int32_t **src_out, const uint32_t **order_out)
{
    const int n = (int)src.size();
    *work_out = nullptr;
    size_t sort_bytes = 0;
    {
        hipcub::DoubleBuffer<unsigned long long> dk(nullptr, nullptr);
        hipcub::DoubleBuffer<uint32_t> dv(nullptr, nullptr);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, n, 0, 63, st));
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t N = (size_t)n;
    const size_t o_src = take(N * 4), o_box = take(N * sizeof(lbvh::Box6)), o_code0 = take(N * 8), o_code1 = take(N * 8), o_ord0 = take(N * 4),
                 o_ord1 = take(N * 4), o_left = take(N * 4), o_right = take(N * 4), o_first = take(N * 4), o_last = take(N * 4), o_pin = take(N * 4),
                 o_pleaf = take(N * 4), o_ibox = take(N * sizeof(lbvh::Box6)), o_height = take(N * 4), o_arr = take(N * 4), o_q0 = take(N * 4), o_q1 = take(N * 4),
                 o_need = take(N * 4), o_ctr = take(64), o_sort = take(sort_bytes);
    unsigned char *ws = nullptr;
    HIP_TRY(hipMalloc((void **)&ws, off));
    *work_out = ws;
    int32_t *d_src = reinterpret_cast<int32_t *>(ws + o_src);
    lbvh::Box6 *boxes = reinterpret_cast<lbvh::Box6 *>(ws + o_box), *iboxes = reinterpret_cast<lbvh::Box6 *>(ws + o_ibox);
    unsigned long long *codes[2] = {reinterpret_cast<unsigned long long *>(ws + o_code0), reinterpret_cast<unsigned long long *>(ws + o_code1)};
    uint32_t *ord[2] = {reinterpret_cast<uint32_t *>(ws + o_ord0), reinterpret_cast<uint32_t *>(ws + o_ord1)};
    int32_t *left = reinterpret_cast<int32_t *>(ws + o_left), *right = reinterpret_cast<int32_t *>(ws + o_right),
            *first = reinterpret_cast<int32_t *>(ws + o_first), *last = reinterpret_cast<int32_t *>(ws + o_last),
            *pin = reinterpret_cast<int32_t *>(ws + o_pin), *pleaf = reinterpret_cast<int32_t *>(ws + o_pleaf);
    uint32_t *arrived = reinterpret_cast<uint32_t *>(ws + o_arr);
    int32_t *height = reinterpret_cast<int32_t *>(ws + o_height);
    int32_t *queue[2] = {reinterpret_cast<int32_t *>(ws + o_q0), reinterpret_cast<int32_t *>(ws + o_q1)};
    int32_t *need = reinterpret_cast<int32_t *>(ws + o_need), *ctr = reinterpret_cast<int32_t *>(ws + o_ctr);
    HIP_TRY(hipMemcpyAsync(d_src, src.data(), N * 4, hipMemcpyHostToDevice, st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (is_tri) hipLaunchKernelGGL((lbvh::k_item_boxes<true>), dim3(nb), dim3(256), 0, st, tri, sph, d_src, n, sb, boxes, codes[0], ord[0]);
    else hipLaunchKernelGGL((lbvh::k_item_boxes<false>), dim3(nb), dim3(256), 0, st, tri, sph, d_src, n, sb, boxes, codes[0], ord[0]);
    hipcub::DoubleBuffer<unsigned long long> dk(codes[0], codes[1]);
    hipcub::DoubleBuffer<uint32_t> dv(ord[0], ord[1]);
    size_t sbytes = sort_bytes;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ws + o_sort, sbytes, dk, dv, n, 0, 63, st));
    const unsigned long long *sorted_codes = dk.Current();
    const uint32_t *order = dv.Current();
    hipLaunchKernelGGL(lbvh::k_radix_tree, dim3(nb), dim3(256), 0, st, sorted_codes, n, left, right, first, last, pin, pleaf);
    HIP_TRY(hipMemsetAsync(arrived, 0, N * 4, st));
    hipLaunchKernelGGL(lbvh::k_fit_boxes, dim3(nb), dim3(256), 0, st, boxes, order, n, left, right, first, last, leaf_max, pin, pleaf, iboxes, height, arrived);
    HIP_TRY(hipGetLastError());
    // the fold, level by level from the root (binary inner node 0)
    accel::Bvh4Node *nodes = nullptr;
    int rc = scene_alloc(ctx, N, &nodes);            // a 4-wide node stands for at least one binary inner node: at most n - 1 of them
    if (rc != RT_OK) return rc;
    const int32_t zero = 0;
    HIP_TRY(hipMemcpyAsync(queue[0], &zero, 4, hipMemcpyHostToDevice, st));
    // two levels more than a perfectly even tree over the leaves would have, at most LBVH_LEVELS_WANTED
    int levels_wanted = 2;
    for (long long leaves = (n + leaf_max - 1) / leaf_max; leaves > 1; leaves = (leaves + 3) / 4) ++levels_wanted;
    if (levels_wanted > LBVH_LEVELS_WANTED) levels_wanted = LBVH_LEVELS_WANTED;
    std::vector<std::pair<int32_t, int32_t>> levels;      // (first node, nodes)
    int32_t base = 0, count = 1;
    int cur = 0;
    while (count > 0) {
        if ((size_t)base + (size_t)count > N) return fail(RT_ERR_DEVICE, "device hierarchy build: more 4-wide nodes than binary ones");
        HIP_TRY(hipMemsetAsync(ctr, 0, 8, st));
        lbvh::FoldLevel L{};
        L.queue = queue[cur]; L.count = count; L.base = base;
        L.next_queue = queue[cur ^ 1]; L.next_count = ctr; L.next_cap = n; L.overflow = ctr + 1;
        L.level = (int32_t)levels.size(); L.levels_wanted = levels_wanted;
        hipLaunchKernelGGL(lbvh::k_fold_level, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, L, left, right, first, last, iboxes, height, boxes, order,
                           leaf_max, nodes);
        int32_t h[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h, ctr, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h[1]) return fail(RT_ERR_DEVICE, "device hierarchy build: level queue overflow");
        levels.emplace_back(base, count);
        base += count;
        count = h[0];
        cur ^= 1;
        if (levels.size() > 200) return fail(RT_ERR_DEVICE, "device hierarchy build: the fold does not end");
    }
    for (size_t l = levels.size(); l-- > 0;)
        hipLaunchKernelGGL(lbvh::k_stack_need, dim3((unsigned)((levels[l].second + 255) / 256)), dim3(256), 0, st, nodes, levels[l].first, levels[l].second, need);
    int32_t need0 = 0;
    HIP_TRY(hipMemcpyAsync(&need0, need, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    out.nodes = nodes;
    out.n_nodes = base;
    out.max_stack = need0;
    out.levels = (int32_t)levels.size();
    *src_out = d_src;
    *order_out = order;
    return RT_OK;
}

extern "C" int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *s)
{
    if (!ctx || !s) return fail(RT_ERR_INVALID, "rt_upload_scene: null argument");
    int rc = validate_scene(s);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    DevScene D{};
    if ((rc = upload(ctx, s->nodes, (size_t)s->n_nodes, &D.nodes)) != RT_OK) return rc;
    accel::Accel acc;
    std::vector<rt_tri_rec> octree_tri;
    std::vector<int32_t> octree_sph_key;
    accel::Options aopt;
    aopt.leaf_max = ctx->dbg.leaf_max;
    aopt.split_budget = ctx->dbg.split_budget;
    aopt.verbose = ctx->dbg.verbose;
    aopt.device_min = ctx->dbg.device_build_min;
    const auto t_build = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (ctx->dbg.verbose)
            fprintf(stderr, "[rt] upload: %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count(), what);
    };
    accel::build(*s, acc, octree_tri, octree_sph_key, aopt);
    mark("keys, items and host-built hierarchies");
    if ((rc = upload(ctx, octree_tri.data(), octree_tri.size(), &D.tri)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->sph, (size_t)s->n_sph_slots, &D.sph)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->sph_prim, (size_t)s->n_sph_slots, &D.sph_prim)) != RT_OK) return rc;
    mark("octree records resident");
    long long n_btri = (long long)acc.tri.size(), n_bsph = (long long)acc.sph.size();
    int stack_tri = acc.tri_bvh4.max_stack, stack_sph = acc.sph_bvh4.max_stack;
    double max_abs = acc.tri_bvh4.max_abs_coord > acc.sph_bvh4.max_abs_coord ? acc.tri_bvh4.max_abs_coord : acc.sph_bvh4.max_abs_coord;
    bool usable = acc.usable;
    if (acc.tri_on_device || acc.sph_on_device) {
        // Large hierarchies are built on the device (lbvh_device.hpp) from the record arrays that are resident by now.
        lbvh::Bounds sb;
        // (one scale for the three axes: the curve's cells are cubes, and a flat scene is not cut along its thin side as finely
        // as along the others, which made the trees deeper)
        double widest = 0.0;
        for (int k = 0; k < 3; ++k) widest = std::max(widest, acc.hi[k] - acc.lo[k]);
        for (int k = 0; k < 3; ++k) {
            sb.lo[k] = acc.lo[k];
            sb.inv[k] = widest > 0.0 ? 1.0 / widest : 0.0;
            // (the boxes get a relative slack of 1e-7 and are rounded outward to single precision)
            max_abs = std::max({max_abs, std::fabs(acc.lo[k]) * 1.000001 + 1e-6, std::fabs(acc.hi[k]) * 1.000001 + 1e-6});
        }
        const int leaf_tri = (aopt.leaf_max >= 1 && aopt.leaf_max <= 4) ? aopt.leaf_max : 2;
        const int leaf_sph = (aopt.leaf_max >= 1 && aopt.leaf_max <= 4) ? aopt.leaf_max : 1;
        // A tree whose walks would need more stack than a full compute unit's LDS holds (LBVH_LEVELS_WANTED) is built again with
        // larger leaves, unless the leaf size was asked for: a million triangles at two to a leaf come out at 13 levels and run in
        // the deep-stack shape (4096 x 4096 frame 107 ms), at three to a leaf at 37 entries (86 ms; the host builder's tree 79 ms).
        auto build_fitting = [&](bool is_tri, const std::vector<int32_t> &src, int leaf, DeviceTree &t, void **work, const int32_t **d_src,
                                 const uint32_t **d_order) -> int {
            for (;;) {
                const size_t held = ctx->scene_allocs.size();
                const int brc = lbvh_build(ctx, ctx->stream, is_tri, D.tri, D.sph, src, leaf, sb, t, work, d_src, d_order);
                if (brc != RT_OK) return brc;
                const bool asked = aopt.leaf_max >= 1 && aopt.leaf_max <= 4;
                if (t.max_stack <= 3 * LBVH_LEVELS_WANTED + 1 || asked || leaf >= 4) return RT_OK;
                (void)hipFree(*work); *work = nullptr;
                while (ctx->scene_allocs.size() > held) { (void)hipFree(ctx->scene_allocs.back()); ctx->scene_allocs.pop_back(); }
                ++leaf;
            }
        };
        if (acc.tri_on_device) {
            DeviceTree t;
            void *work = nullptr;
            const int32_t *d_src = nullptr;
            const uint32_t *d_order = nullptr;
            rc = build_fitting(true, acc.tri_src, leaf_tri, t, &work, &d_src, &d_order);
            rt_tri_rec *leaves = nullptr;
            if (rc == RT_OK) rc = scene_alloc(ctx, acc.tri_src.size(), &leaves);
            if (rc == RT_OK) {
                const long long pieces = (long long)acc.tri_src.size() * 6;
                hipLaunchKernelGGL(lbvh::k_gather_tri, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, ctx->stream, D.tri, d_src, d_order,
                                   (int)acc.tri_src.size(), leaves);
                if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(RT_ERR_DEVICE, "device hierarchy build: gathering the leaf records failed");
            }
            (void)hipFree(work);
            if (rc != RT_OK) return rc;
            D.tri_nodes = t.nodes; D.n_tri_nodes = t.n_nodes; D.btri = leaves;
            n_btri = (long long)acc.tri_src.size();
            stack_tri = t.max_stack;
            if (ctx->dbg.verbose) fprintf(stderr, "[rt] triangle hierarchy on the device: %zu triangles, %d nodes in %d levels, stack %d\n", acc.tri_src.size(), t.n_nodes, t.levels, t.max_stack);
        }
        if (acc.sph_on_device) {
            DeviceTree t;
            void *work = nullptr;
            const int32_t *d_src = nullptr, *d_keys = nullptr;
            const uint32_t *d_order = nullptr;
            rc = build_fitting(false, acc.sph_src, leaf_sph, t, &work, &d_src, &d_order);
            rt_sph_rec *leaves = nullptr;
            uint32_t *leaf_prim = nullptr;
            int32_t *leaf_key = nullptr;
            const size_t ns = acc.sph_src.size();
            if (rc == RT_OK) rc = upload(ctx, octree_sph_key.data(), octree_sph_key.size(), &d_keys);
            if (rc == RT_OK) rc = scene_alloc(ctx, ns, &leaves);
            if (rc == RT_OK) rc = scene_alloc(ctx, ns, &leaf_prim);
            if (rc == RT_OK) rc = scene_alloc(ctx, ns, &leaf_key);
            if (rc == RT_OK) {
                hipLaunchKernelGGL(lbvh::k_gather_sph, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, D.sph, D.sph_prim, d_keys, d_src,
                                   d_order, (int)ns, leaves, leaf_prim, leaf_key);
                if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(RT_ERR_DEVICE, "device hierarchy build: gathering the leaf records failed");
            }
            (void)hipFree(work);
            if (rc != RT_OK) return rc;
            D.sph_nodes = t.nodes; D.n_sph_nodes = t.n_nodes; D.bsph = leaves; D.bsph_prim = leaf_prim; D.bsph_key = leaf_key;
            n_bsph = (long long)ns;
            stack_sph = t.max_stack;
            if (ctx->dbg.verbose) fprintf(stderr, "[rt] sphere hierarchy on the device: %zu spheres, %d nodes in %d levels, stack %d\n", ns, t.n_nodes, t.levels, t.max_stack);
        }
        const int cap = ctx->dbg.device_build_stack < BVH_STACK - 1 ? ctx->dbg.device_build_stack : BVH_STACK - 1;
        if ((acc.tri_on_device && stack_tri > cap) || (acc.sph_on_device && stack_sph > cap)) {
            // A curve-ordered tree over badly distributed primitives can be deeper than the walks' stacks: the host builder's
            // is balanced by construction.  (What the device built stays allocated until the next scene replaces it.)
            if (ctx->dbg.verbose) fprintf(stderr, "[rt] device-built hierarchies too deep (stack %d / %d): building on the host\n", stack_tri, stack_sph);
            aopt.device_min = 0;
            accel::build(*s, acc, octree_tri, octree_sph_key, aopt);
            n_btri = (long long)acc.tri.size(); n_bsph = (long long)acc.sph.size();
            stack_tri = acc.tri_bvh4.max_stack; stack_sph = acc.sph_bvh4.max_stack;
            max_abs = acc.tri_bvh4.max_abs_coord > acc.sph_bvh4.max_abs_coord ? acc.tri_bvh4.max_abs_coord : acc.sph_bvh4.max_abs_coord;
            usable = acc.usable;
        }
    }
    if (!acc.tri_on_device) {
        if ((rc = upload(ctx, acc.tri_bvh4.nodes.data(), acc.tri_bvh4.nodes.size(), &D.tri_nodes)) != RT_OK) return rc;
        if ((rc = upload(ctx, acc.tri.data(), acc.tri.size(), &D.btri)) != RT_OK) return rc;
        D.n_tri_nodes = (int32_t)acc.tri_bvh4.nodes.size();
    }
    if (!acc.sph_on_device) {
        if ((rc = upload(ctx, acc.sph_bvh4.nodes.data(), acc.sph_bvh4.nodes.size(), &D.sph_nodes)) != RT_OK) return rc;
        if ((rc = upload(ctx, acc.sph.data(), acc.sph.size(), &D.bsph)) != RT_OK) return rc;
        if ((rc = upload(ctx, acc.sph_prim.data(), acc.sph_prim.size(), &D.bsph_prim)) != RT_OK) return rc;
        if ((rc = upload(ctx, acc.sph_key.data(), acc.sph_key.size(), &D.bsph_key)) != RT_OK) return rc;
        D.n_sph_nodes = (int32_t)acc.sph_bvh4.nodes.size();
    }
    if (acc.tri_on_device || acc.sph_on_device)
        usable = acc.usable && stack_tri + 1 <= 64 && stack_sph + 1 <= 64 && n_btri < (1ll << 28) && n_bsph < (1ll << 28);
    D.bvh_bound = (float)(max_abs * 1.000001 + 1e-30);
    ctx->accel_usable = usable;
    ctx->n_btri = n_btri;
    ctx->accel_stack = stack_tri > stack_sph ? stack_tri : stack_sph;
    mark("hierarchies resident");
    if ((rc = upload(ctx, s->prim_flags, (size_t)s->n_prims, &D.prim_flags)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->prim_material, (size_t)s->n_prims, &D.prim_material)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->prim_normals, (size_t)s->n_prims * 12, &D.prim_normals)) != RT_OK) return rc;
    {
        // successive normalisations of each flat normal, in the reference's arithmetic (v / sqrt(v.v) when v.v > 0)
        std::vector<double> units((size_t)s->n_prims * 9, 0.0);
        for (int32_t p = 0; p < s->n_prims; ++p) {
            if (s->prim_flags[p] & RT_PRIM_SPHERE) continue;
            double v[3] = {s->prim_normals[(size_t)p * 12], s->prim_normals[(size_t)p * 12 + 1], s->prim_normals[(size_t)p * 12 + 2]};
            for (int k = 0; k < 3; ++k) {
                const double z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
                if (z > 0) { const double q = std::sqrt(z); v[0] = v[0] / q; v[1] = v[1] / q; v[2] = v[2] / q; }
                units[(size_t)p * 9 + k * 3] = v[0]; units[(size_t)p * 9 + k * 3 + 1] = v[1]; units[(size_t)p * 9 + k * 3 + 2] = v[2];
            }
        }
        if ((rc = upload(ctx, units.data(), units.size(), &D.prim_units)) != RT_OK) return rc;
        std::vector<double> n23((size_t)s->n_prims * 6);
        for (int32_t p = 0; p < s->n_prims; ++p)
            for (int k = 0; k < 6; ++k) n23[(size_t)p * 6 + k] = units[(size_t)p * 9 + 3 + k];
        if ((rc = upload(ctx, n23.data(), n23.size(), &D.prim_n23)) != RT_OK) return rc;
    }
    if ((rc = upload(ctx, s->materials, (size_t)s->n_materials, &D.materials)) != RT_OK) return rc;
    if ((rc = upload(ctx, s->lights, (size_t)s->n_lights, &D.lights)) != RT_OK) return rc;
    {
        std::vector<int32_t> rank(static_cast<size_t>(s->n_prims), -1);
        ctx->h_lc_spheres.clear();
        for (int32_t i = 0; i < s->n_sph_slots; ++i) {
            if (!(s->sph_prim[i] & RT_SPH_TRANSPARENT)) continue;
            const int32_t p = static_cast<int32_t>(s->sph_prim[i] & ~RT_SPH_TRANSPARENT);
            if (rank[p] >= 0) continue;
            rank[p] = static_cast<int32_t>(ctx->h_lc_spheres.size());
            ctx->h_lc_spheres.push_back({s->sph[i].center[0], s->sph[i].center[1], s->sph[i].center[2], s->sph[i].radius});
        }
        if ((rc = upload(ctx, rank.data(), rank.size(), &D.prim_lc_rank)) != RT_OK) return rc;
        std::vector<rt_sph_rec> tsph(ctx->h_lc_spheres.size());
        std::vector<int32_t> tsph_prim(ctx->h_lc_spheres.size());
        for (int32_t p = 0; p < s->n_prims; ++p)
            if (rank[p] >= 0) {
                const auto &c = ctx->h_lc_spheres[rank[p]];
                tsph[rank[p]] = rt_sph_rec{{c[0], c[1], c[2]}, c[3]};
                tsph_prim[rank[p]] = p;
            }
        if ((rc = upload(ctx, tsph.data(), tsph.size(), &D.tsph)) != RT_OK) return rc;
        if ((rc = upload(ctx, tsph_prim.data(), tsph_prim.size(), &D.tsph_prim)) != RT_OK) return rc;
        D.n_tsph = static_cast<int32_t>(tsph.size());
        ctx->h_lights.assign(s->lights, s->lights + s->n_lights);
    }
    {
        auto bounded = [](double x) { return std::fabs(x) < 1e100; };      // false for NaN and infinities too
        bool ok = true;
        for (int32_t i = 0; i < s->n_materials; ++i) {
            const rt_material &m = s->materials[i];
            ok = ok && bounded(m.r) && bounded(m.g) && bounded(m.b) && bounded(m.Kd) && bounded(m.Ks) && bounded(m.Shine);
        }
        for (int32_t i = 0; i < s->n_lights; ++i) {
            const rt_light &l = s->lights[i];
            ok = ok && bounded(l.intensity) && bounded(l.color[0]) && bounded(l.color[1]) && bounded(l.color[2]);
        }
        D.shade_bounded = ok ? 1 : 0;
    }
    D.lc_table = nullptr;
    D.lc_needed = nullptr;
    D.n_nodes = s->n_nodes;
    D.n_lights = s->n_lights;
    D.bg[0] = s->background[0]; D.bg[1] = s->background[1]; D.bg[2] = s->background[2];
    {
        // what the view grids enter: every primitive once (a triangle the hierarchy builder cut into pieces has a leaf
        // record per piece; the first one stands for it), then the lights' grids
        std::vector<int32_t> first_record;
        std::vector<uint8_t> seen(static_cast<size_t>(s->n_prims), 0);
        for (size_t k = 0; k < acc.tri.size(); ++k) {
            const int32_t p = acc.tri[k].prim;
            if (!seen[(size_t)p]) { seen[(size_t)p] = 1; first_record.push_back((int32_t)k); }
        }
        if (acc.tri_on_device)       // (built on the device: a leaf record per primitive, no pieces)
            for (size_t k = 0; k < acc.tri_src.size(); ++k) first_record.push_back((int32_t)k);
        const int32_t *d_index = nullptr;
        if ((rc = upload(ctx, first_record.data(), first_record.size(), &d_index)) != RT_OK) return rc;
        ctx->grid_tri_index = const_cast<int32_t *>(d_index);
        ctx->n_grid_tri = (int32_t)first_record.size();
        ctx->n_grid_sph = (int32_t)(acc.sph_on_device ? acc.sph_src.size() : acc.sph.size());
        mark("primitive, material and light tables resident");
        if ((rc = build_light_grids(ctx, D, ctx->n_grid_sph)) != RT_OK) return rc;
        mark("lights' grids");
    }
    ctx->S = D;
    ctx->lightcam_samples = s->lightcam_samples;
    ctx->n_materials = s->n_materials;
    ctx->any_transparent = false;
    ctx->any_secondary = false;
    for (int32_t i = 0; i < s->n_materials; ++i) {
        if (s->materials[i].T > 0) ctx->any_transparent = true;
        if (s->materials[i].T > 0 || s->materials[i].Ks != 0.0) ctx->any_secondary = true;
    }
    ctx->any_sphere = s->n_sph_slots > 0;
    ctx->any_vnormals = false;
    for (int32_t p = 0; p < s->n_prims; ++p) if (s->prim_flags[p] & RT_PRIM_HAS_VNORMALS) ctx->any_vnormals = true;
    ctx->any_transparent_sphere = false;
    for (int32_t i = 0; i < s->n_sph_slots; ++i) if (s->sph_prim[i] & RT_SPH_TRANSPARENT) ctx->any_transparent_sphere = true;
    // Pruning by entry distance is only result-preserving when every sphere surface point
    // lies inside the root cube (then some leaf that contains the hit point also holds the sphere).
    ctx->spheres_inside_root = true;
    const rt_node &root = s->nodes[0];
    const double half = root.radius / (std::sqrt(3.0) / 2.0) / 2.0;
    for (int32_t i = 0; i < s->n_sph_slots; ++i)
        for (int k = 0; k < 3; ++k) {
            double c = s->sph[i].center[k] - root.center[k];
            if (!(std::fabs(c) + s->sph[i].radius < half * (1.0 - 1e-9))) ctx->spheres_inside_root = false;
        }
    ctx->growth = 2;
    ctx->has_scene = true;
    return RT_OK;
}

static int local_rows_of(const rt_camera *cam, const rt_render_opts *o)
{
    int band = o->band_rows > 0 ? o->band_rows : cam->resy;
    int parts = o->n_parts > 0 ? o->n_parts : 1;
    int rows = 0;
    int n_bands = (cam->resy + band - 1) / band;
    for (int b = o->part; b < n_bands; b += parts) {
        int r0 = b * band, r1 = r0 + band;
        if (r1 > cam->resy) r1 = cam->resy;
        rows += r1 - r0;
    }
    return rows;
}

extern "C" int rt_part_rows(const rt_camera *cam, const rt_render_opts *opts, int32_t *rows)
{
    if (!cam || !opts || !rows) return fail(RT_ERR_INVALID, "rt_part_rows: null argument");
    if (cam->resx <= 0 || cam->resy <= 0) return fail(RT_ERR_INVALID, "rt_part_rows: empty frame");
    int parts = opts->n_parts > 0 ? opts->n_parts : 1;
    if (opts->part < 0 || opts->part >= parts) return fail(RT_ERR_INVALID, "rt_part_rows: part out of range");
    *rows = local_rows_of(cam, opts);
    return RT_OK;
}

// Grow-only device buffer.  A failed allocation leaves the buffer empty and reports RT_ERR_DEVICE; the caller
// may try again with a smaller request.
template <typename T>
static int ensure(T **ptr, size_t *have, size_t want)
{
    if (*have >= want && *ptr) return RT_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    hipError_t e = hipMalloc((void **)ptr, sizeof(T) * want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return fail(RT_ERR_DEVICE, std::string("hipMalloc of ") + std::to_string(sizeof(T) * want) + " bytes: " + hipGetErrorString(e));
    }
    *have = want;
    return RT_OK;
}

static std::pair<hipEvent_t, hipEvent_t> *next_event_pair(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, size_t &used)
{
    if (used == v.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        v.emplace_back(a, b);
    }
    return &v[used++];
}

// ------------------------------------------------------------------ light cameras
// A light camera belongs to a (transparent sphere, light) pair and is set up the first time a frame's shading points
// ask for it (Camera.h:417-440): the shading passes raise the pair's word in DevScene::lc_needed, and before a batch's
// density queries run, the cameras that are asked for and do not exist yet are built -- all of them together:
//   photon pass per pair (k_photons), impacts packed in ray order into one shared array (prefix sum of the per-ray
//   counts), then ONE level-synchronous build of all their point octrees (lightcam_device.hpp) into one shared node
//   array, a root per camera.
// The host builder (lightcam_build.cpp, one thread per tree) is kept behind the "lightcam_host" option.
#define LC_MAX_PAIRS (1 << 20)

template <typename T>
static int ws_ensure(T **ptr, size_t *have, size_t want)
{
    if (*ptr && *have >= want) return RT_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr; *have = 0;
    HIP_TRY(hipMalloc((void **)ptr, want ? want : 1));
    *have = want;
    return RT_OK;
}

// Builds the point octrees of `n_trees` cameras whose impacts lie in d_pts (tree t: [h_begins[t], h_begins[t+1])) on
// stream `st`.  d_pts is used as work space.  On success *nodes_out / *points_out are exact-size allocations the caller
// owns: tree t's root is node t, node ranges index *points_out.
static int build_trees_on_device(rt_ctx *ctx, hipStream_t st, lightcam::Point *d_pts, const std::vector<int32_t> &h_begins,
                                 const unsigned long long *d_max_bits, void **nodes_out, int32_t *n_nodes_out, void **points_out, int *levels_out)
{
    *nodes_out = *points_out = nullptr;
    *n_nodes_out = 0;
    const int n_trees = (int)h_begins.size() - 1;
    const int n_pts = h_begins.back();
    const size_t n = (size_t)(n_pts > 0 ? n_pts : 1);
    const size_t cap = 2 * n + 1024 + (size_t)n_trees;     // nodes; a tree has about 0.4 nodes per point
    int key_bits = 5;
    while (((size_t)1 << (key_bits - 4)) < n + 1 && key_bits < 60) ++key_bits;
    size_t sort_bytes = 0;
    {
        hipcub::DoubleBuffer<unsigned long long> dk(nullptr, nullptr);
        hipcub::DoubleBuffer<uint32_t> dv(nullptr, nullptr);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, (int)n, 0, key_bits, st));
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pts = take(n * sizeof(lightcam::Point)), o_key0 = take(n * 8), o_key1 = take(n * 8), o_idx0 = take(n * 4),
                 o_idx1 = take(n * 4), o_own0 = take(n * 4), o_own1 = take(n * 4), o_nodes = take(cap * sizeof(lightcam::Node)),
                 o_runs = take(cap * 9 * 4), o_sub = take(cap), o_act0 = take(cap * 4), o_act1 = take(cap * 4),
                 o_ctr = take(sizeof(lcdev::LevelCounters)), o_begins = take((size_t)(n_trees + 1) * 4), o_sort = take(sort_bytes);
    int rc = ws_ensure(&ctx->lc_ws, &ctx->lc_ws_bytes, off);
    if (rc != RT_OK) return rc;
    unsigned char *ws = static_cast<unsigned char *>(ctx->lc_ws);
    lightcam::Point *pts[2] = {d_pts, reinterpret_cast<lightcam::Point *>(ws + o_pts)};
    unsigned long long *key[2] = {reinterpret_cast<unsigned long long *>(ws + o_key0), reinterpret_cast<unsigned long long *>(ws + o_key1)};
    uint32_t *idx[2] = {reinterpret_cast<uint32_t *>(ws + o_idx0), reinterpret_cast<uint32_t *>(ws + o_idx1)};
    int32_t *owner[2] = {reinterpret_cast<int32_t *>(ws + o_own0), reinterpret_cast<int32_t *>(ws + o_own1)};
    lightcam::Node *nodes = reinterpret_cast<lightcam::Node *>(ws + o_nodes);
    int32_t *runs = reinterpret_cast<int32_t *>(ws + o_runs);
    uint8_t *subdiv = ws + o_sub;
    int32_t *active[2] = {reinterpret_cast<int32_t *>(ws + o_act0), reinterpret_cast<int32_t *>(ws + o_act1)};
    lcdev::LevelCounters *ctr = reinterpret_cast<lcdev::LevelCounters *>(ws + o_ctr);
    int32_t *d_begins = reinterpret_cast<int32_t *>(ws + o_begins);
    void *sort_ws = ws + o_sort;

    const unsigned pb = (unsigned)((n + 255) / 256);
    lcdev::LevelCounters h{};
    h.n_nodes = n_trees;
    HIP_TRY(hipMemcpyAsync(ctr, &h, sizeof h, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_begins, h_begins.data(), (size_t)(n_trees + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lcdev::k_roots, dim3((unsigned)((n_trees + 63) / 64)), dim3(64), 0, st, nodes, d_max_bits, d_begins, n_trees, subdiv, ctr, active[0]);
    if (n_pts > 0) hipLaunchKernelGGL(lcdev::k_fill_owner, dim3(pb), dim3(256), 0, st, owner[0], n_pts, d_begins, n_trees, subdiv);
    HIP_TRY(hipMemcpyAsync(&h, ctr, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int cur = 0, levels = 0;        // pts[cur], owner[cur], active[cur] are the current level's
    int32_t level_end = n_trees;             // nodes created so far, as of the last level read back
    std::vector<int32_t> level_first{0};     // nodes are numbered level by level: level l is [level_first[l], level_first[l + 1])
    while (h.n_next > 0) {
        const int n_active = h.n_next;
        if ((size_t)h.n_nodes + 8 * (size_t)n_active > cap) return fail(RT_ERR_CAPACITY, "light-camera trees: node workspace too small");
        hipLaunchKernelGGL(lcdev::k_clear_runs, dim3((unsigned)((n_active + 255) / 256)), dim3(256), 0, st, active[cur], n_active, runs, ctr);
        hipLaunchKernelGGL(lcdev::k_classify, dim3(pb), dim3(256), 0, st, pts[cur], owner[cur], nodes, n_pts, key[0], idx[0]);
        hipcub::DoubleBuffer<unsigned long long> dk(key[0], key[1]);
        hipcub::DoubleBuffer<uint32_t> dv(idx[0], idx[1]);
        size_t sb = sort_bytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_ws, sb, dk, dv, n_pts, 0, key_bits, st));
        const unsigned long long *key_sorted = dk.Current();
        hipLaunchKernelGGL(lcdev::k_gather, dim3(pb), dim3(256), 0, st, pts[cur], owner[cur], key_sorted, dv.Current(), n_pts, pts[cur ^ 1],
                           owner[cur ^ 1], runs);
        hipLaunchKernelGGL(lcdev::k_children, dim3((unsigned)((n_active + 63) / 64)), dim3(64), 0, st, nodes, (int)cap, active[cur], n_active, runs,
                           ctr, active[cur ^ 1], subdiv);
        hipLaunchKernelGGL(lcdev::k_next_owner, dim3(pb), dim3(256), 0, st, key_sorted, owner[cur ^ 1], nodes, subdiv, n_pts);
        HIP_TRY(hipMemcpyAsync(&h, ctr, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h.overflow) return fail(RT_ERR_CAPACITY, "light-camera trees: node workspace too small");
        level_first.push_back(level_end);
        level_end = h.n_nodes;
        cur ^= 1;
        if (++levels > 80) return fail(RT_ERR_DEVICE, "light-camera trees: the build does not terminate");
    }
    level_first.push_back(level_end);
    for (int l = (int)level_first.size() - 2; l >= 0; --l) {
        const int first = level_first[l], count = level_first[l + 1] - first;
        if (count <= 0) continue;
        hipLaunchKernelGGL(lcdev::k_finish_level, dim3((unsigned)(count < 4096 ? (count + 3) / 4 : 1024)), dim3(256), 0, st, nodes, first, count,
                           pts[cur]);
    }
    HIP_TRY(hipGetLastError());
    void *dn = nullptr, *dp = nullptr;
    HIP_TRY(hipMalloc(&dn, sizeof(lightcam::Node) * (size_t)h.n_nodes));
    if (hipMalloc(&dp, sizeof(lightcam::Point) * n) != hipSuccess) { (void)hipFree(dn); return fail(RT_ERR_DEVICE, "hipMalloc failed for light-camera trees"); }
    if (hipMemcpyAsync(dn, nodes, sizeof(lightcam::Node) * (size_t)h.n_nodes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        (n_pts > 0 && hipMemcpyAsync(dp, pts[cur], sizeof(lightcam::Point) * (size_t)n_pts, hipMemcpyDeviceToDevice, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess) {
        (void)hipFree(dn); (void)hipFree(dp);
        return fail(RT_ERR_DEVICE, "light-camera trees: copy out of the workspace failed");
    }
    *nodes_out = dn;
    *points_out = dp;
    *n_nodes_out = h.n_nodes;
    if (levels_out) *levels_out = levels;
    return RT_OK;
}

// The same trees with two sorts of the points' whole ways instead of a sort per level (lightcam_paths.hpp).  Returns
// RT_ERR_UNSUPPORTED (no message) when a tree is deeper than the keys can say: the caller then builds level by level.
static int build_trees_by_paths(rt_ctx *ctx, hipStream_t st, lightcam::Point *d_pts, const std::vector<int32_t> &h_begins,
                                const unsigned long long *d_max_bits, void **nodes_out, int32_t *n_nodes_out, void **points_out, int *levels_out)
{
    *nodes_out = *points_out = nullptr;
    *n_nodes_out = 0;
    const int n_trees = (int)h_begins.size() - 1;
    const int n_pts = h_begins.back();
    const size_t n = (size_t)(n_pts > 0 ? n_pts : 1);
    const size_t cap = 2 * n + 1024 + (size_t)n_trees;
    lcpath::Params P{};
    P.n = n_pts;
    P.n_trees = n_trees;
    while ((1ll << P.tree_bits) < (long long)n_trees) ++P.tree_bits;
    P.tree_bits = (P.tree_bits + 3) & ~3;
    P.digits = (128 - P.tree_bits) / 4;
    if (P.digits > LCP_MAX_DIGITS) P.digits = LCP_MAX_DIGITS;
    if (ctx->dbg.lightcam_digits > 0 && ctx->dbg.lightcam_digits < P.digits) P.digits = ctx->dbg.lightcam_digits;
    const int used_bits = P.tree_bits + 4 * P.digits;                  // of the 128, from the top
    const int lo_begin = used_bits > 64 ? 128 - used_bits : 64;        // lowest bit of the low word that can be set
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        hipcub::DoubleBuffer<unsigned long long> dk(nullptr, nullptr);
        hipcub::DoubleBuffer<uint32_t> dv(nullptr, nullptr);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, (int)n, 0, 64, st));
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, st));
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_pts = take(n * sizeof(lightcam::Point)), o_w0 = take(n * 8), o_w1 = take(n * 8), o_w2 = take(n * 8), o_w3 = take(n * 8),
                 o_w4 = take(n * 8), o_i0 = take(n * 4), o_i1 = take(n * 4), o_geo = take(n), o_here = take(n * 4), o_base = take(n * 4),
                 o_nodes = take(cap * sizeof(lightcam::Node)), o_roots = take((size_t)n_trees * sizeof(lightcam::Node)), o_level = take(cap),
                 o_list = take(cap * 4), o_small = take(4096), o_sub = take((size_t)n_trees), o_act = take((size_t)n_trees * 4),
                 o_begins = take((size_t)(n_trees + 1) * 4), o_sort = take(sort_bytes), o_scan = take(scan_bytes);
    int rc = ws_ensure(&ctx->lc_ws, &ctx->lc_ws_bytes, off);
    if (rc != RT_OK) return rc;
    unsigned char *ws = static_cast<unsigned char *>(ctx->lc_ws);
    lightcam::Point *pts_out = reinterpret_cast<lightcam::Point *>(ws + o_pts);
    unsigned long long *w[5] = {reinterpret_cast<unsigned long long *>(ws + o_w0), reinterpret_cast<unsigned long long *>(ws + o_w1),
                                reinterpret_cast<unsigned long long *>(ws + o_w2), reinterpret_cast<unsigned long long *>(ws + o_w3),
                                reinterpret_cast<unsigned long long *>(ws + o_w4)};
    uint32_t *ix[2] = {reinterpret_cast<uint32_t *>(ws + o_i0), reinterpret_cast<uint32_t *>(ws + o_i1)};
    uint8_t *geo = ws + o_geo;
    uint32_t *here = reinterpret_cast<uint32_t *>(ws + o_here), *base = reinterpret_cast<uint32_t *>(ws + o_base);
    lightcam::Node *nodes = reinterpret_cast<lightcam::Node *>(ws + o_nodes), *roots = reinterpret_cast<lightcam::Node *>(ws + o_roots);
    uint8_t *level_of = ws + o_level;
    int32_t *list = reinterpret_cast<int32_t *>(ws + o_list);
    // small words: [0] too deep, [1] node overflow, [64..127] nodes per level, [128..191] first of level, [192..255] cursors; LevelCounters at 1024
    uint32_t *small = reinterpret_cast<uint32_t *>(ws + o_small);
    lcdev::LevelCounters *ctr = reinterpret_cast<lcdev::LevelCounters *>(ws + o_small + 1024);
    uint8_t *subdiv = ws + o_sub;
    int32_t *active = reinterpret_cast<int32_t *>(ws + o_act);
    int32_t *d_begins = reinterpret_cast<int32_t *>(ws + o_begins);
    void *sort_ws = ws + o_sort;
    P.begins = d_begins;

    const unsigned pb = (unsigned)((n + 255) / 256);
    HIP_TRY(hipMemsetAsync(small, 0, 4096, st));
    HIP_TRY(hipMemcpyAsync(d_begins, h_begins.data(), (size_t)(n_trees + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lcdev::k_roots, dim3((unsigned)((n_trees + 63) / 64)), dim3(64), 0, st, nodes, d_max_bits, d_begins, n_trees, subdiv, ctr, active);
    HIP_TRY(hipMemcpyAsync(roots, nodes, (size_t)n_trees * sizeof(lightcam::Node), hipMemcpyDeviceToDevice, st));
    int32_t total_nodes = n_trees;
    int deepest = 0;
    if (n_pts > 0) {
        // a 128-bit sort as two stable 64-bit sorts, low word first; returns where the words and the order ended up
        auto sort128 = [&](unsigned long long *hi, unsigned long long *lo, unsigned long long *spare_a, unsigned long long *spare_b,
                           unsigned long long **hi_sorted, unsigned long long **lo_sorted, uint32_t **order) -> int {
            // `ix[0]` holds the identity on entry
            hipcub::DoubleBuffer<uint32_t> dv(ix[0], ix[1]);
            if (lo_begin < 64) {
                hipLaunchKernelGGL(lcpath::k_gather_words, dim3(pb), dim3(256), 0, st, lo, ix[0], n_pts, spare_a);       // (a copy: the sort may leave its input changed)
                hipcub::DoubleBuffer<unsigned long long> dk(spare_a, spare_b);
                size_t sb = sort_bytes;
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_ws, sb, dk, dv, n_pts, lo_begin, 64, st));
            }
            hipLaunchKernelGGL(lcpath::k_gather_words, dim3(pb), dim3(256), 0, st, hi, dv.Current(), n_pts, spare_a);
            hipcub::DoubleBuffer<unsigned long long> dk(spare_a, spare_b);
            size_t sb = sort_bytes;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sort_ws, sb, dk, dv, n_pts, 128 - used_bits > 64 ? 128 - used_bits - 64 : 0, 64, st));
            *hi_sorted = dk.Current();
            *order = dv.Current();
            unsigned long long *free_words = dk.Current() == spare_a ? spare_b : spare_a;
            hipLaunchKernelGGL(lcpath::k_gather_words, dim3(pb), dim3(256), 0, st, lo, dv.Current(), n_pts, free_words);
            *lo_sorted = free_words;
            return RT_OK;
        };
        // whole ways -> their order -> where the ways end (cleared keys, in the original order)
        hipLaunchKernelGGL(lcpath::k_paths, dim3(pb), dim3(256), 0, st, P, d_pts, roots, w[0], w[1], ix[0], geo);
        unsigned long long *hs = nullptr, *ls = nullptr;
        uint32_t *order = nullptr;
        if ((rc = sort128(w[0], w[1], w[2], w[3], &hs, &ls, &order)) != RT_OK) return rc;
        // (the whole keys in w[0], w[1] have been read for the last time: the cleared keys, in the original order, take their place)
        hipLaunchKernelGGL(lcpath::k_depths, dim3(pb), dim3(256), 0, st, P, hs, ls, order, geo, w[0], w[1], reinterpret_cast<int32_t *>(small));
        // (the first sort's results in w[2], w[3] and its order are not needed once k_depths has run: the second sort takes the same places)
        hipLaunchKernelGGL(lcpath::k_iota, dim3(pb), dim3(256), 0, st, ix[0], n_pts);
        unsigned long long *hf = nullptr, *lf = nullptr;
        uint32_t *perm = nullptr;
        if ((rc = sort128(w[0], w[1], w[2], w[3], &hf, &lf, &perm)) != RT_OK) return rc;
        hipLaunchKernelGGL(lcpath::k_place, dim3(pb), dim3(256), 0, st, P, hf, lf, perm, d_pts, pts_out, here);
        size_t sc = scan_bytes;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws + o_scan, sc, here, base, n_pts, st));
        uint32_t last[2] = {0, 0};
        int32_t flags[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(&last[0], base + (n_pts - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&last[1], here + (n_pts - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(flags, small, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags[0]) return RT_ERR_UNSUPPORTED;
        if ((size_t)n_trees + (size_t)last[0] + (size_t)last[1] > cap) return fail(RT_ERR_CAPACITY, "light-camera trees: node workspace too small");
        total_nodes = n_trees + (int32_t)(last[0] + last[1]);
        HIP_TRY(hipMemsetAsync(level_of, 0, (size_t)n_trees, st));      // (a tree without points keeps the root k_roots made, at level 0)
        hipLaunchKernelGGL(lcpath::k_nodes, dim3(pb), dim3(256), 0, st, P, hf, lf, base, roots, nodes, (int)cap, level_of, reinterpret_cast<int32_t *>(small) + 1);
        hipLaunchKernelGGL(lcpath::k_link, dim3(pb), dim3(256), 0, st, P, hf, lf, base, nodes, (int)cap);
        const unsigned nb = (unsigned)((total_nodes + 255) / 256);
        hipLaunchKernelGGL(lcpath::k_level_hist, dim3(nb), dim3(256), 0, st, level_of, total_nodes, small + 64);
        uint32_t per_level[LCP_LEVELS];
        HIP_TRY(hipMemcpyAsync(per_level, small + 64, sizeof per_level, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(flags, small, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags[1]) return fail(RT_ERR_CAPACITY, "light-camera trees: node workspace too small");
        uint32_t first_of[LCP_LEVELS];
        uint32_t run = 0;
        for (int d = 0; d < LCP_LEVELS; ++d) { first_of[d] = run; run += per_level[d]; if (per_level[d]) deepest = d; }
        if (run != (uint32_t)total_nodes) return fail(RT_ERR_DEVICE, "light-camera trees: the nodes' levels do not add up");
        HIP_TRY(hipMemcpyAsync(small + 128, first_of, sizeof first_of, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lcpath::k_level_lists, dim3(nb), dim3(256), 0, st, level_of, total_nodes, small + 128, small + 192, list);
        for (int d = deepest; d >= 0; --d) {
            const int count = (int)per_level[d];
            if (count <= 0) continue;
            hipLaunchKernelGGL(lcpath::k_finish_list, dim3((unsigned)(count < 4096 ? (count + 3) / 4 : 1024)), dim3(256), 0, st, nodes, list + first_of[d],
                               count, pts_out);
        }
    }
    HIP_TRY(hipGetLastError());
    void *dn = nullptr, *dp = nullptr;
    HIP_TRY(hipMalloc(&dn, sizeof(lightcam::Node) * (size_t)total_nodes));
    if (hipMalloc(&dp, sizeof(lightcam::Point) * n) != hipSuccess) { (void)hipFree(dn); return fail(RT_ERR_DEVICE, "hipMalloc failed for light-camera trees"); }
    if (hipMemcpyAsync(dn, nodes, sizeof(lightcam::Node) * (size_t)total_nodes, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        (n_pts > 0 && hipMemcpyAsync(dp, pts_out, sizeof(lightcam::Point) * (size_t)n_pts, hipMemcpyDeviceToDevice, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess) {
        (void)hipFree(dn); (void)hipFree(dp);
        return fail(RT_ERR_DEVICE, "light-camera trees: copy out of the workspace failed");
    }
    *nodes_out = dn;
    *points_out = dp;
    *n_nodes_out = total_nodes;
    if (levels_out) *levels_out = deepest;        // (levels that subdivide, as the level-by-level builder counts them)
    return RT_OK;
}

// The table of cameras (one entry per pair, all "not built") and the words the shading passes raise.
static int ensure_lightcam_table(rt_ctx *ctx, uint32_t lc_flags)
{
    const size_t n_pairs = ctx->h_lc_spheres.size() * ctx->h_lights.size();
    if (ctx->lc_table && ctx->lc_flags == lc_flags) return RT_OK;
    free_lightcams(ctx);
    if (n_pairs > LC_MAX_PAIRS)
        return fail(RT_ERR_CAPACITY, "rt_render: " + std::to_string(n_pairs) + " (transparent sphere, light) pairs, more than the table of " +
                                         std::to_string(LC_MAX_PAIRS) + " light cameras; RT_FLAG_NO_LIGHTCAM renders without the term");
    HIP_TRY(hipMalloc((void **)&ctx->lc_table, sizeof(DevLightCam) * n_pairs));
    if (hipMalloc((void **)&ctx->lc_needed, sizeof(uint32_t) * n_pairs) != hipSuccess) {
        (void)hipFree(ctx->lc_table); ctx->lc_table = nullptr;
        return fail(RT_ERR_DEVICE, "hipMalloc failed for the light-camera table");
    }
    HIP_TRY(hipMemset(ctx->lc_table, 0, sizeof(DevLightCam) * n_pairs));
    HIP_TRY(hipMemset(ctx->lc_needed, 0, sizeof(uint32_t) * n_pairs));
    ctx->lc_have.assign(n_pairs, 0);
    ctx->lc_n_built = 0;
    ctx->lc_flags = lc_flags;
    ctx->h_needed.resize(n_pairs);
    return RT_OK;
}

// Sets up the cameras of `pairs` (indices rank * n_lights + light), all together, and enters them in the table.
static int build_lightcams(rt_ctx *ctx, const std::vector<int32_t> &pairs, uint32_t flags, bool cull)
{
    if (pairs.empty()) return RT_OK;
    const int N = ctx->lightcam_samples;
    if (N < 1 || N > 4096) return fail(RT_ERR_INVALID, "light camera sample count (Z) must be in 1..4096");
    const size_t n_rays = (size_t)N * N, n_build = pairs.size(), n_lights = ctx->h_lights.size();
    const auto t_begin = std::chrono::steady_clock::now();
    {
        // a light ray leaves about three weighted points, and a point costs about 64 bytes with its share of the tree (twice
        // that while it is being built): the cameras of one call may take half of the free device memory
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if ((double)n_build * (double)n_rays * 3.0 * 128.0 > 0.5 * (double)free_b)
            return fail(RT_ERR_CAPACITY, "rt_render: the light cameras this frame asks for would need more than half of the free device "
                                         "memory; RT_FLAG_NO_LIGHTCAM renders without the term");
    }
    hipStream_t st = ctx->stream;
    // work space of the photon passes: 21 impact slots per ray, per-ray counts, offsets, counters (2 words per camera:
    // largest |impact| as bits, castLightRay calls), prefix-sum scratch
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n_rays, st);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_slots = take(n_rays * LC_MAX_IMPACTS * sizeof(lightcam::Point)), o_counts = take(n_rays * 4), o_cnt32 = take(n_rays * 4),
                 o_offsets = take(n_rays * 4), o_meta = take(n_build * 16), o_scan = take(scan_bytes);
    int rc = ws_ensure(&ctx->lc_photon_ws, &ctx->lc_photon_ws_bytes, off);
    if (rc != RT_OK) return rc;
    unsigned char *pw = static_cast<unsigned char *>(ctx->lc_photon_ws);
    lightcam::Point *d_slots = reinterpret_cast<lightcam::Point *>(pw + o_slots);
    int32_t *d_counts = reinterpret_cast<int32_t *>(pw + o_counts);
    uint32_t *d_cnt32 = reinterpret_cast<uint32_t *>(pw + o_cnt32), *d_offsets = reinterpret_cast<uint32_t *>(pw + o_offsets);
    unsigned long long *d_meta = reinterpret_cast<unsigned long long *>(pw + o_meta);
    void *d_scan = pw + o_scan;
    HIP_TRY(hipMemsetAsync(d_meta, 0, n_build * 16, st));

    // the impacts of all cameras of this call, packed one camera after the other (grows when the estimate was short)
    size_t packed_cap = ctx->lc_packed_points;
    if (packed_cap < n_build * n_rays * 3) {
        rc = ws_ensure(&ctx->lc_packed, &ctx->lc_packed_bytes, n_build * n_rays * 3 * sizeof(lightcam::Point));
        if (rc != RT_OK) return rc;
        packed_cap = ctx->lc_packed_points = ctx->lc_packed_bytes / sizeof(lightcam::Point);
    }
    std::vector<int32_t> begins(n_build + 1, 0);
    std::vector<DevLightCam> cams(n_build);
    const unsigned rb = (unsigned)((n_rays + 255) / 256);
    for (size_t b = 0; b < n_build; ++b) {
        const size_t pair = (size_t)pairs[b], r = pair / n_lights, li = pair % n_lights;
        const auto &sp = ctx->h_lc_spheres[r];
        const rt_light &lt = ctx->h_lights[li];
        lightcam::Frame fr;
        const double spos[3] = {sp[0], sp[1], sp[2]};
        lightcam::setup_frame(lt.pos, spos, sp[3], fr);
        PhotonParams Q{};
        Q.S = ctx->S;
        for (int k = 0; k < 3; ++k) { Q.u[k] = fr.u[k]; Q.v[k] = fr.v[k]; Q.w[k] = fr.w[k]; Q.e[k] = fr.e[k]; }
        Q.d = fr.d; Q.l = fr.l; Q.t = fr.t; Q.pix_width = fr.pix_width;
        Q.N = N;
        Q.flags = flags;
        Q.cull = cull ? 1 : 0;
        Q.slots = d_slots;
        Q.counts = d_counts;
        hipLaunchKernelGGL(k_photons, dim3(rb), dim3(256), 0, st, Q);
        hipLaunchKernelGGL(lcdev::k_impact_counts, dim3(rb), dim3(256), 0, st, d_counts, (int)n_rays, d_cnt32);
        size_t sb = scan_bytes;
        HIP_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan, sb, d_cnt32, d_offsets, (int)n_rays, st));
        uint32_t last_off = 0, last_cnt = 0;
        HIP_TRY(hipMemcpyAsync(&last_off, d_offsets + (n_rays - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&last_cnt, d_cnt32 + (n_rays - 1), 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const size_t n_pts = (size_t)last_off + last_cnt;
        if ((size_t)begins[b] + n_pts >= ((size_t)1 << 31)) return fail(RT_ERR_CAPACITY, "light cameras of one frame with more than 2^31 impacts");
        if ((size_t)begins[b] + n_pts > packed_cap) {
            // the estimate was short: a larger array, the impacts packed so far carried over
            lightcam::Point *bigger = nullptr;
            const size_t want = ((size_t)begins[b] + n_pts) * 2;
            HIP_TRY(hipMalloc((void **)&bigger, want * sizeof(lightcam::Point)));
            if (begins[b] > 0 && hipMemcpy(bigger, ctx->lc_packed, (size_t)begins[b] * sizeof(lightcam::Point), hipMemcpyDeviceToDevice) != hipSuccess) {
                (void)hipFree(bigger);
                return fail(RT_ERR_DEVICE, "light cameras: carrying the packed impacts over to a larger array failed");
            }
            (void)hipFree(ctx->lc_packed);
            ctx->lc_packed = bigger;
            ctx->lc_packed_bytes = want * sizeof(lightcam::Point);
            packed_cap = ctx->lc_packed_points = want;
        }
        hipLaunchKernelGGL(lcdev::k_pack_photons, dim3(rb), dim3(256), 0, st, d_slots, d_counts, d_offsets, (int)n_rays, LC_MAX_IMPACTS,
                           ctx->lc_packed + begins[b], d_meta + 2 * b, d_meta + 2 * b + 1);
        begins[b + 1] = begins[b] + (int32_t)n_pts;
        DevLightCam cam{};
        cam.root = (int32_t)b;
        cam.n_points = (int32_t)n_pts;
        cam.spw = fr.pix_width / N;
        for (int c = 0; c < 3; ++c) { cam.light_pos[c] = lt.pos[c]; cam.sphere_pos[c] = spos[c]; }
        cams[b] = cam;
    }
    HIP_TRY(hipGetLastError());
    void *dn = nullptr, *dp = nullptr;
    int32_t n_nodes = 0;
    int levels = 0;
    bool by_paths = false;
    std::vector<lightcam::Node> host_nodes;      // (the host builder's output)
    std::vector<lightcam::Point> host_points;
    if (!ctx->dbg.lightcam_host) {
        rc = ctx->dbg.lightcam_levels ? RT_ERR_UNSUPPORTED : build_trees_by_paths(ctx, st, ctx->lc_packed, begins, d_meta, &dn, &n_nodes, &dp, &levels);
        by_paths = rc == RT_OK;
        if (rc == RT_ERR_UNSUPPORTED) rc = build_trees_on_device(ctx, st, ctx->lc_packed, begins, d_meta, &dn, &n_nodes, &dp, &levels);
        if (rc != RT_OK) return rc;
    } else {
        // host builder: every tree on a thread of its own, the node lists joined behind the roots (root b = node b)
        const size_t n_pts = (size_t)begins[n_build];
        std::vector<lightcam::Point> all(n_pts ? n_pts : 1);
        HIP_TRY(hipStreamSynchronize(st));
        if (n_pts) HIP_TRY(hipMemcpy(all.data(), ctx->lc_packed, n_pts * sizeof(lightcam::Point), hipMemcpyDeviceToHost));
        std::vector<lightcam::Tree> trees(n_build);
        std::vector<std::thread> pool;
        for (size_t b = 0; b < n_build; ++b)
            pool.emplace_back([&, b]() {
                std::vector<lightcam::Point> pts(all.begin() + begins[b], all.begin() + begins[b + 1]);
                double max_dist = 0;
                for (const lightcam::Point &q : pts) { const double dz = lightcam::length3(q.x, q.y, q.z); if (dz > max_dist) max_dist = dz; }
                lightcam::build(pts, max_dist, trees[b]);
            });
        for (std::thread &t : pool) t.join();
        std::vector<lightcam::Node> &nodes = host_nodes;
        std::vector<lightcam::Point> &points = host_points;
        nodes.assign(n_build, lightcam::Node{});
        points.reserve(n_pts);
        for (size_t b = 0; b < n_build; ++b) {
            const int32_t shift = (int32_t)nodes.size() - 1, pshift = begins[b];       // node k > 0 of tree b becomes node shift + k
            for (size_t k = 0; k < trees[b].nodes.size(); ++k) {
                lightcam::Node nd = trees[b].nodes[k];
                for (int c = 0; c < 8; ++c) if (nd.child[c] >= 0) nd.child[c] += shift;
                nd.begin += pshift; nd.end += pshift; nd.rest_begin += pshift;
                if (k == 0) nodes[b] = nd; else nodes.push_back(nd);
            }
            points.insert(points.end(), trees[b].points.begin(), trees[b].points.end());
        }
        n_nodes = (int32_t)nodes.size();
        HIP_TRY(hipMalloc(&dn, sizeof(lightcam::Node) * nodes.size()));
        if (hipMalloc(&dp, sizeof(lightcam::Point) * (points.size() ? points.size() : 1)) != hipSuccess) { (void)hipFree(dn); return fail(RT_ERR_DEVICE, "hipMalloc failed for light-camera trees"); }
    }
    // the context owns the arrays from here on: a failure below leaves them to free_lightcams
    ctx->lc_allocs.push_back(dn);
    ctx->lc_allocs.push_back(dp);
    if (ctx->dbg.lightcam_host) {
        HIP_TRY(hipMemcpy(dn, host_nodes.data(), sizeof(lightcam::Node) * host_nodes.size(), hipMemcpyHostToDevice));
        if (!host_points.empty()) HIP_TRY(hipMemcpy(dp, host_points.data(), sizeof(lightcam::Point) * host_points.size(), hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> meta(n_build * 2);
    HIP_TRY(hipMemcpy(meta.data(), d_meta, n_build * 16, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < n_build; ++b) {
        cams[b].nodes = static_cast<const lightcam::Node *>(dn);
        cams[b].points = static_cast<const lightcam::Point *>(dp);
        cams[b].n_nodes = n_nodes;
        ctx->lc_rays += meta[2 * b + 1];
        HIP_TRY(hipMemcpy(ctx->lc_table + pairs[b], &cams[b], sizeof(DevLightCam), hipMemcpyHostToDevice));
        ctx->lc_have[(size_t)pairs[b]] = 1;
        ctx->lc_n_built++;
    }
    if (ctx->dbg.verbose)
        fprintf(stderr, "[rt] %zu light cameras set up in %.2f ms: %d impacts, %d nodes, %d levels (%s builder)\n", n_build,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), begins[n_build], n_nodes, levels,
                ctx->dbg.lightcam_host ? "host" : (by_paths ? "device, two sorts" : "device, level by level"));
    return RT_OK;
}

// Before a batch's density queries: the cameras its shading passes asked for (the stream has to drain for that; once
// every camera of the scene exists this costs nothing).  `may_build` false (the entry point that allocates nothing):
// a missing camera is an error.
static int provide_lightcams(rt_ctx *ctx, uint32_t lc_flags, bool cull, bool may_build)
{
    if (ctx->lc_n_built == ctx->lc_have.size()) return RT_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream2));
    HIP_TRY(hipMemcpy(ctx->h_needed.data(), ctx->lc_needed, ctx->h_needed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> missing;
    for (size_t k = 0; k < ctx->h_needed.size(); ++k)
        if (ctx->h_needed[k] && !ctx->lc_have[k]) missing.push_back((int32_t)k);
    if (missing.empty()) return RT_OK;
    if (!may_build)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_noalloc: this frame needs " + std::to_string(missing.size()) + " light camera(s) that are not set "
                                        "up yet (setting one up allocates); render the frame once with rt_render, or pass RT_FLAG_NO_LIGHTCAM");
    return build_lightcams(ctx, missing, lc_flags, cull);
}

// ------------------------------------------------------------------ view grids
// Builds the grids described by `builds` (viewgrid.hpp) into one pair of arrays: *bins_out holds the total_bins bin heads,
// *entries_out the lists (it ends with spare entries).  Count pass, prefix sum, fill pass, sort, heads; the stream is
// drained once in between, when the number of entries is read.  *bins_out / *entries_out are grown when too small.
#define VG_LARGE_CAP (1u << 18)
static int build_view_grids(rt_ctx *ctx, hipStream_t st, std::vector<vg::Build> &builds, size_t total_bins, vg::Bin **bins_out,
                            size_t *bins_cap, vg::Entry **entries_out, size_t *entries_words, size_t *n_entries_out, size_t max_per_bin,
                            size_t *occupied_out = nullptr)
{
    const size_t nb = builds.size();
    // Bin counts and list positions are 32-bit: an item that is not listed as large enters at most VG_LARGE_BINS bins, the
    // large ones are summed in 64 bits while they are counted, and a build that could pass 2^31 entries is not made.
    unsigned long long small_bound = 0;
    for (const vg::Build &B : builds) small_bound += (unsigned long long)VG_LARGE_BINS * (unsigned long long)(B.n_tri + B.n_sph) * (unsigned long long)B.n_faces;
    if (small_bound >= (1ull << 31)) return RT_ERR_UNSUPPORTED;
    size_t scan_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)(total_bins + 1), st));
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_bins = take((total_bins + 1) * 4), o_begin = take((total_bins + 1) * 4), o_large = take(nb * VG_LARGE_CAP * 4), o_ctr = take((nb + 1) * 4), o_total = take(16), o_scan = take(scan_bytes);
    int rc = ws_ensure(&ctx->grid_ws, &ctx->grid_ws_bytes, off);
    if (rc != RT_OK) return rc;
    unsigned char *ws = static_cast<unsigned char *>(ctx->grid_ws);
    uint32_t *bins = reinterpret_cast<uint32_t *>(ws + o_bins), *large = reinterpret_cast<uint32_t *>(ws + o_large),
             *ctr = reinterpret_cast<uint32_t *>(ws + o_ctr);
    uint32_t *begin = reinterpret_cast<uint32_t *>(ws + o_begin);
    if (*bins_cap < total_bins) {
        (void)hipFree(*bins_out); *bins_out = nullptr; *bins_cap = 0;
        HIP_TRY(hipMalloc((void **)bins_out, total_bins * sizeof(vg::Bin)));
        *bins_cap = total_bins;
    }
    HIP_TRY(hipMemsetAsync(bins, 0, (total_bins + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(ctr, 0, (nb + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(ws + o_total, 0, 16, st));
    auto launch = [&](bool fill) {
        for (size_t b = 0; b < nb; ++b) {
            vg::Build &B = builds[b];
            B.bins = bins;
            B.large = large + b * VG_LARGE_CAP;
            B.n_large = ctr + b;
            B.large_cap = VG_LARGE_CAP;
            B.overflow = ctr + nb;
            B.large_total = reinterpret_cast<unsigned long long *>(ws + o_total);
            B.entries = fill ? *entries_out : nullptr;
            const long long threads = (long long)(B.n_tri + B.n_sph) * B.n_faces;
            if (threads <= 0) continue;
            const unsigned blocks = (unsigned)((threads + 255) / 256);
            if (fill) {
                hipLaunchKernelGGL((vg::k_grid_items<true>), dim3(blocks), dim3(256), 0, st, B);
                hipLaunchKernelGGL((vg::k_grid_large<true>), dim3(2048), dim3(256), 0, st, B);
            } else {
                hipLaunchKernelGGL((vg::k_grid_items<false>), dim3(blocks), dim3(256), 0, st, B);
                hipLaunchKernelGGL((vg::k_grid_large<false>), dim3(2048), dim3(256), 0, st, B);
            }
        }
    };
    launch(false);
    HIP_TRY(hipGetLastError());
    size_t sb = scan_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(ws + o_scan, sb, bins, begin, (int)(total_bins + 1), st));
    hipLaunchKernelGGL(vg::k_grid_occupied, dim3((unsigned)((total_bins + 255) / 256)), dim3(256), 0, st, begin, (uint32_t)total_bins,
                       reinterpret_cast<uint32_t *>(ws + o_total + 8));
    uint32_t total = 0, overflow = 0, occupied = 0;
    unsigned long long large_total = 0;
    HIP_TRY(hipMemcpyAsync(&total, begin + total_bins, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&overflow, ctr + nb, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&large_total, ws + o_total, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&occupied, ws + o_total + 8, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (occupied_out) *occupied_out = occupied;
    if (overflow) return fail(RT_ERR_CAPACITY, "view grids: too many items that cover many bins");
    if (small_bound + large_total >= (1ull << 31)) return RT_ERR_UNSUPPORTED;
    // Lists this long (per bin that holds anything) are no shorter than what a walk of the hierarchy tests -- a dense scene seen
    // through few bins.  Measured: the eye's grid on 320 000 triangles and 100 000 spheres, 18 entries per tile 0.99 against 1.13 ms,
    // 50 per tile 4.64 against 4.71, 139 per tile 2.23 against 1.64; the lights' grids at 4096 x 4096, gears.nff with 19.6 per bin
    // 10.1 against 12.2 ms, tetra.nff (4096 small triangles behind one another) with 36 per bin 1.46 against 1.20.
    if (max_per_bin && (size_t)total > max_per_bin * (size_t)(occupied ? occupied : 1)) return RT_ERR_UNSUPPORTED;
    if (*entries_words < (size_t)total + 4) {
        (void)hipFree(*entries_out); *entries_out = nullptr; *entries_words = 0;
        const size_t want = (size_t)total + (size_t)total / 4 + 1024;
        HIP_TRY(hipMalloc((void **)entries_out, want * sizeof(vg::Entry)));
        *entries_words = want;
    }
    HIP_TRY(hipMemsetAsync(*entries_out + total, 0, 4 * sizeof(vg::Entry), st));        // the spare entries behind the last one
    HIP_TRY(hipMemcpyAsync(bins, begin, (total_bins + 1) * 4, hipMemcpyDeviceToDevice, st));
    launch(true);
    hipLaunchKernelGGL(vg::k_grid_sort, dim3((unsigned)((total_bins + 255) / 256)), dim3(256), 0, st, begin, (uint32_t)total_bins, *entries_out);
    hipLaunchKernelGGL(vg::k_grid_heads, dim3((unsigned)((total_bins + 255) / 256)), dim3(256), 0, st, begin, (uint32_t)total_bins, *entries_out, *bins_out);
    HIP_TRY(hipGetLastError());
    if (n_entries_out) *n_entries_out = total;
    return RT_OK;
}

// The frames of a cube map around a point: faces +x, -x, +y, -y, +z, -z; on each the two other coordinates, in axis order,
// divided by the major one (light_bin in grid_query.hpp looks bins up the same way).
static void cube_faces(vg::Build &B, int res, uint32_t first_bin)
{
    static const double W[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    static const double U[6][3] = {{0, 1, 0}, {0, 1, 0}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}, {1, 0, 0}};
    static const double V[6][3] = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};
    B.n_faces = 6;
    for (int f = 0; f < 6; ++f) {
        vg::Face &F = B.face[f];
        for (int k = 0; k < 3; ++k) { F.u[k] = U[f][k]; F.v[k] = V[f][k]; F.w[k] = W[f][k]; }
        F.x0 = F.y0 = -1.0;
        F.inv_cell = 0.5 * res;
        F.nx = F.ny = res;
        F.first_bin = first_bin + (uint32_t)f * (uint32_t)res * (uint32_t)res;
        F._pad = 0;
    }
}

// The grids of what each light can see, made once per scene (the hierarchies' leaf records are resident already).
static int build_light_grids(rt_ctx *ctx, DevScene &D, int n_sph_records)
{
    D.lg_bins = nullptr; D.lg_entries = nullptr; D.lg_res = 0;
    const int mode = ctx->dbg.grid;
    if (!(mode == 1 || mode == 2) || !ctx->accel_usable || D.n_lights < 1 || D.n_lights > 16) return RT_OK;
    // 256 bins to a face's edge; 512 for a scene of few primitives, whose lists then take a millisecond more to make and are a
    // fifth shorter (8K teapot frame 36.7 -> 35.5 ms, balls at 4096 x 4096 5.74 -> 5.31; small frames the same) -- for the 72 k
    // triangles of gears the finer grids would take 51 ms instead of 19
    int res = ctx->dbg.grid_res > 0 ? ctx->dbg.grid_res : ((long long)ctx->n_grid_tri + n_sph_records <= 8192 ? 512 : 256);
    if (res < 8) res = 8;
    if (res > 1024) res = 1024;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<vg::Build> builds((size_t)D.n_lights);
    const size_t per_light = (size_t)6 * res * res;
    for (int l = 0; l < D.n_lights; ++l) {
        vg::Build &B = builds[(size_t)l];
        std::memset(&B, 0, sizeof B);
        for (int k = 0; k < 3; ++k) B.origin[k] = ctx->h_lights[(size_t)l].pos[k];
        // a shadow query ends .01 before the light (Camera.h:404): what it can report lies at least .01 from the light, at a
        // depth of at least .01 / sqrt(3) on the face of its major axis
        B.z_near = 0.005;
        B.margin = 2.0 / 65536.0;
        cube_faces(B, res, (uint32_t)(per_light * (size_t)l));
        B.n_tri = ctx->n_grid_tri;
        B.n_sph = n_sph_records;
        B.tri_index = ctx->grid_tri_index;
        B.tri = D.btri;
        B.sph = D.bsph;
    }
    size_t bins_cap = 0, entries_words = 0, n_entries = 0, occupied = 0;
    int rc = build_view_grids(ctx, ctx->stream, builds, per_light * (size_t)D.n_lights, &ctx->lg_bins, &bins_cap, &ctx->lg_entries,
                              &entries_words, &n_entries, ctx->dbg.grid_res > 0 ? 0 : 24, &occupied);      // (a grid asked for by size is made whatever its lists)
    if (rc != RT_OK) {
        // (no grids: the hierarchies answer; not an error of the upload)
        (void)hipFree(ctx->lg_bins); ctx->lg_bins = nullptr;
        (void)hipFree(ctx->lg_entries); ctx->lg_entries = nullptr;
        if (rc == RT_ERR_CAPACITY || rc == RT_ERR_UNSUPPORTED) return RT_OK;
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    D.lg_bins = ctx->lg_bins;
    D.lg_entries = ctx->lg_entries;
    D.lg_res = res;
    if (ctx->dbg.verbose)
        fprintf(stderr, "[rt] light grids: %d lights x 6 x %d^2 bins, %zu entries (%.2f per bin, %.2f per bin that holds any), %.2f ms\n", D.n_lights, res, n_entries,
                (double)n_entries / (double)(per_light * (size_t)D.n_lights), (double)n_entries / (double)(occupied ? occupied : 1),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return RT_OK;
}

// The grid of what the eye can see, in tiles of 8 x 8 pixels of the whole frame; kept until the camera or the scene changes.
static int provide_camera_grid(rt_ctx *ctx, const rt_camera *cam, int n_sph_records)
{
    if (ctx->cg_valid && std::memcmp(&ctx->cg_cam, cam, sizeof(rt_camera)) == 0) return RT_OK;
    if (ctx->cg_refused && std::memcmp(&ctx->cg_cam, cam, sizeof(rt_camera)) == 0) return RT_ERR_UNSUPPORTED;
    ctx->cg_valid = false;
    ctx->cg_refused = false;
    const int tx = (cam->resx + 7) / 8, ty = (cam->resy + 7) / 8;
    std::vector<vg::Build> builds(1);
    vg::Build &B = builds[0];
    std::memset(&B, 0, sizeof B);
    for (int k = 0; k < 3; ++k) B.origin[k] = cam->e[k];
    // A sample's direction is p u + q v - d w with p = l + (column +- half a pixel) pix_width, q = t - (row +- half a pixel)
    // pix_width (Camera.h:145-188): projected on the plane at distance d along -w, x = p and y = -q, and the tile of pixel
    // (row, column) is the bin (row / 8, column / 8) of a grid that starts half a pixel before pixel (0, 0).
    vg::Face &F = B.face[0];
    for (int k = 0; k < 3; ++k) { F.u[k] = cam->d * cam->u[k]; F.v[k] = -cam->d * cam->v[k]; F.w[k] = -cam->w[k]; }
    F.x0 = cam->l - 0.5 * cam->pix_width;
    F.y0 = -cam->t - 0.5 * cam->pix_width;
    F.inv_cell = 1.0 / (8.0 * cam->pix_width);
    F.nx = tx; F.ny = ty;
    F.first_bin = 0;
    B.n_faces = 1;
    const double ext_x = 8.0 * cam->pix_width * tx, ext_y = 8.0 * cam->pix_width * ty;
    B.margin = (ext_x > ext_y ? ext_x : ext_y) / 65536.0;
    {
        // a hit lies at least minT = .01 along a unit direction (Camera.h:322): its depth along -w is at least .01 times the
        // cosine of the widest angle a sample makes with the axis
        const double px = std::fmax(std::fabs(F.x0), std::fabs(F.x0 + ext_x)), py = std::fmax(std::fabs(F.y0), std::fabs(F.y0 + ext_y));
        const double cosine = std::fabs(cam->d) / std::sqrt(cam->d * cam->d + px * px + py * py);
        B.z_near = 0.005 * cosine;
    }
    if (!(B.z_near > 0) || !(cam->pix_width > 0) || !(cam->d > 0)) return fail(RT_ERR_UNSUPPORTED, "camera grid: degenerate camera");
    B.n_tri = ctx->n_grid_tri;
    B.n_sph = n_sph_records;
    B.tri_index = ctx->grid_tri_index;
    B.tri = ctx->S.btri;
    B.sph = ctx->S.bsph;
    size_t n_entries = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    int rc = build_view_grids(ctx, ctx->stream, builds, (size_t)tx * ty, &ctx->cg_bins, &ctx->cg_bins_cap, &ctx->cg_entries,
                              &ctx->cg_entries_words, &n_entries, ctx->dbg.grid_min_samples >= 0 ? 0 : 48);
    if (rc == RT_ERR_UNSUPPORTED) { ctx->cg_cam = *cam; ctx->cg_refused = true; }      // (not asked again while the camera stays)
    if (rc != RT_OK) return rc;
    ctx->cg_cam = *cam;
    ctx->cg_tiles_x = tx;
    ctx->cg_valid = true;
    if (ctx->dbg.verbose) {
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[rt] camera grid: %d x %d tiles, %zu entries (%.2f per tile), %.2f ms\n", tx, ty, n_entries,
                (double)n_entries / ((double)tx * ty), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    }
    return RT_OK;
}

// ------------------------------------------------------------------ frame planning
// Sizes of the sharded queues of one batch.  A block pushes into shard blockIdx % N_SHARDS and the 64-item chunks
// of a pass are dealt to the blocks in a fixed pattern, so what a shard can receive in a generation follows from
// what one wave can be handed, times the waves that feed the shard: with those capacities a push cannot overflow
// as long as the generation's input fits its own queue.  The input of generation g+1 is at most fan_out times the
// hits of generation g (reflection, plus refraction when a material is transparent).  Carried through six
// generations that bound is 32 times the primary count, which no real scene approaches (the reference's scenes
// stay below 1.3 rays per primary in every generation), so the ray queues are capped at `growth` times the
// first generation's capacity.  A push beyond a capacity is dropped and raises the overflow word; rt_sync reports
// it (RT_ERR_CAPACITY), the context doubles `growth`, and rt_render renders the frame again by itself.
struct QueuePlan {
    long long batch_chunks = 0;
    int max_blocks = 0;               // grid of the queued passes
    long long blocks0 = 0;            // grid of a full primary pass
    long long hit_cap[6] = {}, ray_cap[7] = {};      // records per shard
    long long need_rays[2] = {0, 0}, need_hits = 0;  // records per buffer
    uint32_t unit_chunks = 0;         // the primary pass shares its chunks out in units of this many (0: fixed shares)
    int hit_buffers = 1;
    long long hit_stride = sizeof(HitRec);   // bytes per hit record: 64 when every hit of the scene is a flat triangle's
    long long hit_slots() const { return (need_hits * hit_stride + (long long)sizeof(HitRec) - 1) / (long long)sizeof(HitRec); }   // in whole HitRec
    bool deferred = false;
    long long defer_mult = 1;         // lights: the deferred queue of a generation holds every (hit, light) pair
    size_t deferred_records(int g) const { return (size_t)(hit_cap[g] * defer_mult * N_SHARDS); }
    size_t bytes() const
    {
        size_t b = (size_t)(need_rays[0] + need_rays[1]) * sizeof(RayRec) + (size_t)hit_slots() * sizeof(HitRec) * hit_buffers;
        if (deferred)
            for (int g = 0; g < 6; ++g) b += deferred_records(g) * sizeof(HitRec);
        return b;
    }
};

static void plan_queues(QueuePlan &q, long long batch_chunks, int waves_per_block, int max_blocks, long long fan_out, long long growth,
                        bool units_allowed = true)
{
    q.batch_chunks = batch_chunks;
    q.max_blocks = max_blocks;
    long long blocks0 = (batch_chunks + waves_per_block - 1) / waves_per_block;
    if (blocks0 > max_blocks) blocks0 = max_blocks;
    q.blocks0 = blocks0;
    const long long waves_q = (long long)max_blocks * waves_per_block;
    const long long wps_q = ((max_blocks + N_SHARDS - 1) / N_SHARDS) * (long long)waves_per_block;
    const long long w0 = blocks0 * waves_per_block, wps0 = ((blocks0 + N_SHARDS - 1) / N_SHARDS) * waves_per_block;
    q.ray_cap[0] = 0;
    q.hit_cap[0] = ((batch_chunks + w0 - 1) / w0) * 64 * wps0;
    // A primary pass with many chunks per wave hands them out in units while it runs (k_closest); a unit's records go to
    // the shard given by its number, so a shard holds at most its share of the units whichever blocks worked on them.
    q.unit_chunks = 0;
    if (units_allowed && batch_chunks >= 16 * w0 && blocks0 == max_blocks) {
        q.unit_chunks = 32;      // 16 to 128 measure alike (8K teapot frame, 2 to 8 batches: within 0.5 %), 256 is 2-4 % slower
        const long long units = (batch_chunks + q.unit_chunks - 1) / q.unit_chunks;
        q.hit_cap[0] = ((units + N_SHARDS - 1) / N_SHARDS) * q.unit_chunks * 64;
    }
    const long long ray_limit = q.hit_cap[0] * (growth < 1 ? 1 : growth);
    q.ray_cap[1] = q.hit_cap[0] * fan_out < ray_limit ? q.hit_cap[0] * fan_out : ray_limit;
    for (int g = 1; g < 6; ++g) {
        const long long c_in = N_SHARDS * ((q.ray_cap[g] + 63) / 64);
        q.hit_cap[g] = ((c_in + waves_q - 1) / waves_q) * 64 * wps_q;
        q.ray_cap[g + 1] = q.hit_cap[g] * fan_out < ray_limit ? q.hit_cap[g] * fan_out : ray_limit;
    }
    q.need_rays[0] = q.need_rays[1] = 64;
    q.need_hits = 0;
    for (int g = 1; g < 6; ++g)
        if (q.ray_cap[g] * N_SHARDS > q.need_rays[g & 1]) q.need_rays[g & 1] = q.ray_cap[g] * N_SHARDS;
    for (int g = 0; g < 6; ++g)
        if (q.hit_cap[g] * N_SHARDS > q.need_hits) q.need_hits = q.hit_cap[g] * N_SHARDS;
}

// Equal batches instead of full ones and a short last one (a short batch costs its twelve launches all the same);
// a batch is a whole number of 8x8-pixel blocks (spp chunks each).
static long long even_batches(long long n_chunks, long long batch_chunks, long long spp)
{
    if (batch_chunks > (1ll << 25)) batch_chunks = 1ll << 25;      // samples of a batch are numbered in 32 bits
    batch_chunks = batch_chunks / spp * spp;
    if (batch_chunks > n_chunks) batch_chunks = n_chunks;
    if (batch_chunks < spp) batch_chunks = spp;
    const long long n_batches = (n_chunks + batch_chunks - 1) / batch_chunks;
    const long long even = (n_chunks + n_batches - 1) / n_batches;
    return (even + spp - 1) / spp * spp;
}

// The buffers a frame works in besides its inputs and outputs: the FP64 accumulation frame, the ray, hit and
// deferred-query queues of one batch, the queue counters.  rt_render / rt_render_async take them from the
// context (device allocations that only grow); rt_render_noalloc carves them out of the device's static scratch.
struct FrameBuffers {
    double *accum = nullptr;
    RayRec *rays[2] = {nullptr, nullptr};
    HitRec *hits[4] = {nullptr, nullptr, nullptr, nullptr};
    HitRec *deferred[6] = {};
    uint32_t *counts = nullptr;                 // 2 * CNT_WORDS
    unsigned long long *totals = nullptr;       // RT_TOTALS_WORDS
};

static int buffers_from_context(rt_ctx *ctx, const QueuePlan &q, size_t n_pixels, FrameBuffers &B)
{
    int rc;
    if ((rc = ensure(&ctx->accum, &ctx->accum_values, n_pixels * 3)) != RT_OK) return rc;
    if ((rc = ensure(&ctx->rays[0], &ctx->ray_capacity[0], (size_t)q.need_rays[0])) != RT_OK) return rc;
    if ((rc = ensure(&ctx->rays[1], &ctx->ray_capacity[1], (size_t)q.need_rays[1])) != RT_OK) return rc;
    for (int k = 0; k < q.hit_buffers; ++k)
        if ((rc = ensure(&ctx->hits[k], &ctx->hit_capacity[k], (size_t)q.hit_slots())) != RT_OK) return rc;
    if (q.deferred)
        for (int g = 0; g < 6; ++g)
            if ((rc = ensure(&ctx->deferred[g], &ctx->deferred_capacity[g], q.deferred_records(g))) != RT_OK) return rc;
    B.accum = ctx->accum;
    B.rays[0] = ctx->rays[0]; B.rays[1] = ctx->rays[1];
    for (int k = 0; k < 4; ++k) B.hits[k] = ctx->hits[k];
    for (int g = 0; g < 6; ++g) B.deferred[g] = ctx->deferred[g];
    B.counts = ctx->counts;
    B.totals = ctx->totals;
    return RT_OK;
}

static void release_queues(rt_ctx *ctx)
{
    for (int k = 0; k < 2; ++k) { (void)hipFree(ctx->rays[k]); ctx->rays[k] = nullptr; ctx->ray_capacity[k] = 0; }
    for (int k = 0; k < 4; ++k) { (void)hipFree(ctx->hits[k]); ctx->hits[k] = nullptr; ctx->hit_capacity[k] = 0; }
    for (int g = 0; g < 6; ++g) { (void)hipFree(ctx->deferred[g]); ctx->deferred[g] = nullptr; ctx->deferred_capacity[g] = 0; }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t scratch_need(const QueuePlan &q, size_t n_pixels)
{
    size_t b = align256(n_pixels * 3 * sizeof(double)) + align256(2 * CNT_WORDS * sizeof(uint32_t)) + align256(RT_TOTALS_WORDS * sizeof(unsigned long long));
    b += align256((size_t)q.need_rays[0] * sizeof(RayRec)) + align256((size_t)q.need_rays[1] * sizeof(RayRec));
    b += align256((size_t)q.hit_slots() * sizeof(HitRec));
    if (q.deferred)
        for (int g = 0; g < 6; ++g) b += align256(q.deferred_records(g) * sizeof(HitRec));
    return b;
}
static void buffers_from_scratch(unsigned char *base, const QueuePlan &q, size_t n_pixels, FrameBuffers &B)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base + off; off += align256(bytes); return p; };
    B.accum = reinterpret_cast<double *>(take(n_pixels * 3 * sizeof(double)));
    B.counts = reinterpret_cast<uint32_t *>(take(2 * CNT_WORDS * sizeof(uint32_t)));
    B.totals = reinterpret_cast<unsigned long long *>(take(RT_TOTALS_WORDS * sizeof(unsigned long long)));
    B.rays[0] = reinterpret_cast<RayRec *>(take((size_t)q.need_rays[0] * sizeof(RayRec)));
    B.rays[1] = reinterpret_cast<RayRec *>(take((size_t)q.need_rays[1] * sizeof(RayRec)));
    B.hits[0] = reinterpret_cast<HitRec *>(take((size_t)q.hit_slots() * sizeof(HitRec)));
    if (q.deferred)
        for (int g = 0; g < 6; ++g) B.deferred[g] = reinterpret_cast<HitRec *>(take(q.deferred_records(g) * sizeof(HitRec)));
}

// How a frame is put on the device.
enum FrameMode {
    FRAME_HOST_OUT,      // rt_render with host output buffers (staged)
    FRAME_DEVICE_OUT,    // rt_render / rt_render_async with device output buffers
    FRAME_NOALLOC        // rt_render_noalloc: device outputs, work buffers from the static scratch, no allocation of any kind
};

static int enqueue_frame(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *o, const rt_frame_out *out, FrameMode mode)
{
    if (!ctx || !cam || !o || !out) return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (!ctx->has_scene) return fail(RT_ERR_INVALID, "rt_render: no scene uploaded");
    if (cam->resx <= 0 || cam->resy <= 0) return fail(RT_ERR_INVALID, "rt_render: empty frame");
    if (o->num_samples < 1 || o->num_samples > 64) return fail(RT_ERR_INVALID, "rt_render: num_samples must be in 1..64");
    const int parts = o->n_parts > 0 ? o->n_parts : 1;
    const int band = o->band_rows > 0 ? o->band_rows : cam->resy;
    if (o->part < 0 || o->part >= parts) return fail(RT_ERR_INVALID, "rt_render: part out of range");
    if (ctx->pending) return fail(RT_ERR_INVALID, "rt_render: a frame is already in flight; call rt_sync first");
    const bool noalloc = mode == FRAME_NOALLOC;
    const bool outputs_on_device = mode != FRAME_HOST_OUT;
    if (noalloc && (o->flags & RT_FLAG_TIMING))
        return fail(RT_ERR_UNSUPPORTED, "rt_render_noalloc: RT_FLAG_TIMING creates events and is not available here");
    if (noalloc && !ctx->scratch) return fail(RT_ERR_DEVICE, "rt_render_noalloc: the static scratch of this device was not located");
    HIP_TRY(hipSetDevice(ctx->device));

    const int local_rows = local_rows_of(cam, o);
    const int N = o->num_samples;
    const size_t n_pixels = (size_t)local_rows * cam->resx;
    if (n_pixels >= (size_t)1 << 31) return fail(RT_ERR_INVALID, "rt_render: more than 2^31 pixels in one part");

    const bool may_cull = !(o->flags & RT_FLAG_NO_CULL) && ctx->spheres_inside_root;
    // light cameras: one per (transparent sphere, light) pair and normal-interpolation mode (the light rays see it), set up
    // when a frame first asks for them
    const uint32_t lc_flags = o->flags & RT_FLAG_PHONG_NORMALS;
    {
        const bool want_lc = !(o->flags & RT_FLAG_NO_LIGHTCAM) && ctx->any_transparent_sphere && !ctx->h_lights.empty();
        if (want_lc && !(ctx->lc_table && ctx->lc_flags == lc_flags)) {
            if (noalloc)
                return fail(RT_ERR_UNSUPPORTED, "rt_render_noalloc: this scene's table of light cameras does not exist yet for these flags "
                                                "(making it allocates); render one frame with rt_render first, or pass RT_FLAG_NO_LIGHTCAM");
            int rc = ensure_lightcam_table(ctx, lc_flags);
            if (rc != RT_OK) return rc;
        }
        ctx->S.lc_table = want_lc ? ctx->lc_table : nullptr;
        ctx->S.lc_needed = want_lc ? ctx->lc_needed : nullptr;
    }
    FrameParams P{};
    P.S = ctx->S;
    if (o->flags & RT_FLAG_NO_LIGHTCAM) P.S.lc_table = nullptr;
    P.cam = *cam;
    P.N = N;
    P.aperture = o->aperture;
    P.flags = o->flags;
    P.seed = o->seed;
    P.band_rows = band;
    P.n_parts = parts;
    P.part = o->part;
    P.local_rows = local_rows;
    P.blocks_x = (cam->resx + 7) / 8;
    P.div_spp = make_fastdiv((uint32_t)(N * N));
    P.div_blocks_x = make_fastdiv((uint32_t)P.blocks_x);
    P.div_n = make_fastdiv((uint32_t)N);
    P.div_band = make_fastdiv((uint32_t)band);
    P.spw = cam->pix_width / N;
    P.div_lights = make_fastdiv((uint32_t)(P.S.n_lights > 0 ? P.S.n_lights : 1));
    P.cull_closest = may_cull ? 1 : 0;
    P.cull_shadow = (may_cull && !ctx->any_transparent_sphere) ? 1 : 0;

    // ---- batch size: primary ordering is 8x8-pixel blocks, N*N samples per pixel -> (N*N) 64-sample chunks per block
    const long long spp = (long long)N * N;
    const long long blocks_y = ((long long)local_rows + 7) / 8;
    const long long n_chunks = blocks_y * P.blocks_x * spp;
    const long long default_batch = ctx->any_transparent ? (1ll << 24) : (1ll << 28);
    const long long wanted_batch_chunks = ((ctx->batch_override > 0 ? ctx->batch_override : default_batch) + 63) / 64;
    long long batch_chunks = even_batches(n_chunks, wanted_batch_chunks, spp);
    if (ctx->batch_override > 0) {
        // an explicit batch size is taken as it is (whole blocks), not evened out
        long long b = wanted_batch_chunks;
        if (b > (1ll << 25)) b = 1ll << 25;
        b = b / spp * spp;
        if (b > n_chunks) b = n_chunks;
        if (b < spp) b = spp;
        batch_chunks = b;
    }

    // ---- launch shape
    // The hierarchies answer "nearest / any hit over all primitives"; that equals the reference's
    // walk when every sphere is inside the root cube, and for shadow queries only when no
    // transparent sphere exists (checkOccluded3 is then order-free).
    bool bvh_closest_ok = may_cull && ctx->accel_usable;
    // the builder's bound on pending siblings, plus one (BVH_STACK is the cap the builder checks against)
    int stack_entries = ctx->accel_stack + 1;
    if (stack_entries < ctx->dbg.stack_min) stack_entries = ctx->dbg.stack_min;
    if (stack_entries > BVH_STACK) stack_entries = BVH_STACK;
    if (stack_entries < 4) stack_entries = 4;
    // Block size.  Large batches run one 1024-thread block per compute unit (one LDS copy of the top of the scene
    // per compute unit); small ones, which live on latency and on many independent blocks, 256-thread blocks.
    // The one-launch kernel (k_fused: a lane follows its sample's whole tree) is chosen for frames that cannot fill the
    // chip pass by pass AND whose trees end at the first hit -- no material reflects or refracts.  Measured at 512x512:
    // tetra-3 0.249 -> 0.063 ms.  With deeper trees a wave lasts as long as its deepest sample (six generations of
    // closest hit plus shadow queries, each a whole walk long), which is what the wavefront pipeline's passes also add
    // up to, and the regrouping of the survivors into full waves is lost: teapot-3 0.40 -> 0.38 ms, refract 0.47 -> 0.45,
    // balls 0.69 -> 0.86.  The "fused" option forces either path.
    const bool use_fused = !noalloc && local_rows > 0 &&
                           (ctx->dbg.fused == 1 || (ctx->dbg.fused < 0 && (long long)n_pixels * spp <= RT_FUSED_MAX_SAMPLES &&
                                                    !ctx->any_secondary && !(o->flags & RT_FLAG_KEEP_ZERO_WEIGHT)));
    // The block-local one-launch kernel (k_local) takes frames of one batch that are too small to fill the chip pass by
    // pass, in scenes whose trees go on after the first hit and whose shadow queries take the hierarchies.
    bool want_local = !noalloc && !use_fused && local_rows > 0 && may_cull && ctx->accel_usable && batch_chunks >= n_chunks &&
                            (ctx->dbg.local == 1 || (ctx->dbg.local < 0 && n_chunks * 64 <= RT_LOCAL_MAX_SAMPLES));
    int block = want_local ? ctx->dbg.local_threads : (noalloc || use_fused || batch_chunks * 64 <= (1ll << 22)) ? 256 : RT_BLOCK;
    // k_local: primary chunks per block (one run per resident wave slot) and the records a block's ray list can hold
    // (with two rays per hit, `growth` times the first generation, as in plan_queues); beside its counters the block keeps
    // a word per 64 records of the ray lists of generations 1..5 in LDS
    long long local_cpb = 0, local_cap = 0, local_grid = 0;
    size_t local_lds = 0;
    auto plan_local = [&](int per_cu) {
        // every compute unit gets a block as soon as there are `min_chunks` primary chunks for each (a frame of 4800 chunks
        // cut into runs of whole rounds of waves was 150 blocks of 32 chunks on a chip of 256 compute units)
        const long long resident = (long long)ctx->cu_count * per_cu;
        const long long min_chunks = ctx->dbg.local_min_chunks > 0 ? ctx->dbg.local_min_chunks : 4;
        local_grid = (n_chunks + min_chunks - 1) / min_chunks;
        if (local_grid > resident) local_grid = resident;
        local_cpb = (n_chunks + local_grid - 1) / local_grid;
        local_cap = local_cpb * 64 * (ctx->any_transparent ? (long long)ctx->growth : 1ll);
        local_lds = (size_t)(CNT_WORDS + 32 + 5 * (local_cap / 64)) * sizeof(uint32_t);
    };
    if (!use_fused && (ctx->dbg.block_threads == 256 || ctx->dbg.block_threads == 1024)) block = ctx->dbg.block_threads;
    if (want_local) plan_local(1024 / block);
    // Self-contained blocks whose stacks and lists do not fit a compute unit together give way to a launch per pass (a
    // 500 000-triangle terrain, 37 entries: 1024 x 1024 frame 6.45 ms in blocks of 256 with three to a compute unit, 4.98 pass by pass).
    if (want_local && ctx->dbg.local < 0 && (size_t)stack_entries * sizeof(int) * 1024 + 8192 + local_lds * (size_t)(1024 / block) > ctx->lds_per_cu) {
        want_local = false;
        local_lds = 0; local_cpb = local_cap = local_grid = 0;
        block = (noalloc || use_fused || batch_chunks * 64 <= (1ll << 22)) ? 256 : RT_BLOCK;
        if (!use_fused && (ctx->dbg.block_threads == 256 || ctx->dbg.block_threads == 1024)) block = ctx->dbg.block_threads;
    }
    // A hierarchy so deep that the stacks of a full compute unit (1024 threads) do not fit its LDS beside the tables (and the
    // lists of k_local) is walked in blocks of 256, as many to a compute unit as their stacks leave room for (three at 40
    // entries): fewer waves in flight, but the hierarchy and not the octree walk, which on a scene large enough to have such
    // a hierarchy is hundreds of times slower.
    const size_t stack_bytes_per_thread = (size_t)stack_entries * sizeof(int);
    const size_t lds_beside_stacks = 8192 + (want_local && local_lds <= 24 * 1024 ? local_lds * (size_t)(1024 / block) : 0);
    const bool deep = may_cull && ctx->accel_usable && !use_fused && stack_bytes_per_thread * 1024 + lds_beside_stacks > ctx->lds_per_cu;
    int fit_per_cu = 1024 / block;
    if (deep && !(ctx->dbg.block_threads == 256 || ctx->dbg.block_threads == 1024)) {
        block = 256;
        fit_per_cu = 4;
        for (;;) {
            if (want_local) plan_local(fit_per_cu);
            const size_t per_block = stack_bytes_per_thread * 256 + 4096 + (want_local && local_lds <= 24 * 1024 ? local_lds : 0);
            if (fit_per_cu == 1 || per_block * (size_t)fit_per_cu <= ctx->lds_per_cu) break;
            --fit_per_cu;
        }
    }
    const int blocks_per_cu = fit_per_cu, waves_per_block = block / 64;
    bool local_fits = want_local && local_lds <= 24 * 1024;       // (deep queues after overflows: the launch-per-pass path takes over)
    P.S.stack_shift = block == 256 ? 8 : (block == 512 ? 9 : 10);
    // LDS per block: the stacks, the material and light tables, then as many top-of-tree nodes as fit
    // beside them when the compute unit holds 1024 threads (the register budget allows four waves per SIMD).
    // A tree so deep that its stacks alone do not fit keeps to the octree walk.
    const size_t lds_block = ctx->lds_per_cu / blocks_per_cu;      // LDS a block may take when a compute unit is full
    size_t lds_stack = (size_t)stack_entries * block * sizeof(int);
    if (lds_stack + 8192 + (local_fits ? local_lds : 0) > ctx->lds_per_cu) bvh_closest_ok = false;     // (a block may take a whole compute unit's LDS)
    if (!bvh_closest_ok) lds_stack = 0;
    const bool bvh_shadow_ok = bvh_closest_ok && !ctx->any_transparent_sphere;
    // every hit record is the 64-byte kind when the scene has no sphere and no triangle is shaded with
    // interpolated normals; the records then lie back to back
    const bool all_flat = !ctx->any_sphere && !((o->flags & RT_FLAG_PHONG_NORMALS) && ctx->any_vnormals);
    // With a few transparent spheres a shadow ray is first tested against those alone: a ray that touches none
    // of them has an order-free answer and takes the hierarchies; only the others walk the octree.
    const bool hybrid_shadow = bvh_closest_ok && ctx->any_transparent_sphere && P.S.n_tsph <= MAX_HYBRID_SPHERES &&
                               !(o->flags & RT_FLAG_NO_CULL);
    P.S.mat_cached = ctx->n_materials < 64 ? ctx->n_materials : 64;
    P.S.light_cached = P.S.n_lights < 16 ? P.S.n_lights : 16;
    const size_t lds_tables = (size_t)(P.S.mat_cached + P.S.light_cached) * 64;
    {
        const size_t per_block = lds_block;
        const size_t fixed = lds_stack + lds_tables + 128 + 512 + (local_fits ? local_lds : 0);
        long long room = per_block > fixed ? (long long)((per_block - fixed) / (NODE_LDS_WORDS * 4)) : 0;
        if (ctx->dbg.node_cache >= 0 && ctx->dbg.node_cache < room) room = ctx->dbg.node_cache;
        // the triangle hierarchy first, the sphere hierarchy takes what is left; split evenly when both are large
        long long nt = P.S.n_tri_nodes, ns = P.S.n_sph_nodes;
        if (!bvh_closest_ok) nt = ns = 0;
        if (nt + ns > room) {
            const long long half = room / 2;
            if (nt <= half) ns = room - nt;
            else if (ns <= half) nt = room - ns;
            else { nt = half; ns = room - half; }
        }
        P.S.tri_cached = (int32_t)nt;
        P.S.sph_cached = (int32_t)ns;
        // what is left after the nodes takes leaf triangle records (80 bytes each)
        long long left = (long long)(per_block > fixed ? per_block - fixed : 0) - (nt + ns) * NODE_LDS_WORDS * 4;
        long long tc = bvh_closest_ok && left > 0 ? left / 80 : 0;
        if (tc > ctx->n_btri) tc = ctx->n_btri;
        if (ctx->dbg.tri_cache >= 0 && ctx->dbg.tri_cache < tc) tc = ctx->dbg.tri_cache;
        P.S.btri_cached = (int32_t)tc;
        P.S.lds_cache_word = (int32_t)(lds_stack / sizeof(int));
        P.S.lds_tri_word = P.S.lds_cache_word + (P.S.tri_cached + P.S.sph_cached) * NODE_LDS_WORDS;
        P.S.lds_table_word = P.S.lds_tri_word + P.S.btri_cached * 20;
        P.S.lds_cam_word = P.S.lds_table_word + (P.S.mat_cached + P.S.light_cached) * 16;
        P.S.lds_ticket_word = P.S.lds_cam_word + 32;
    }
    // every traversal kernel gets the same layout (the octree-walk variants use the tables only)
    const size_t lds_bvh = lds_stack + (size_t)(P.S.tri_cached + P.S.sph_cached) * NODE_LDS_WORDS * sizeof(int) +
                           (size_t)P.S.btri_cached * 80 + lds_tables + 128 + 32;
    bool use_local = local_fits && bvh_closest_ok && (bvh_shadow_ok || hybrid_shadow);      // (and its queues within the memory budget, below)
    const int lds_counts_word = P.S.lds_ticket_word + 8;                 // k_local: the block's own queue counters
    const size_t lds_local = lds_bvh + local_lds;
    const size_t lds_closest = lds_bvh;
    const size_t lds_shadow = lds_bvh;
    const bool use_lightcam = P.S.lc_table != nullptr && !bvh_shadow_ok && P.S.n_lights > 0;
    // The eye's grid (viewgrid.hpp), for pinhole frames.  Making it costs about a millisecond, more than a small frame takes,
    // and it is kept while the camera stays: a frame of at least 4 M samples gets it at once, a smaller one when its camera
    // is the one of the frame before (a camera at rest; one that moves from frame to frame walks the hierarchies).
    P.cg_bins = nullptr; P.cg_entries = nullptr; P.cg_tiles_x = (cam->resx + 7) / 8;
    {
        const long long cg_min = ctx->dbg.grid_min_samples >= 0 ? ctx->dbg.grid_min_samples : (1ll << 22);
        const bool same_camera = ctx->last_cam_valid && std::memcmp(&ctx->last_cam, cam, sizeof(rt_camera)) == 0;
        ctx->last_cam = *cam;
        ctx->last_cam_valid = true;
        if (bvh_closest_ok && !noalloc && !use_fused && local_rows > 0 && (ctx->dbg.grid == 1 || ctx->dbg.grid == 3) && !(o->aperture > 0) &&
            ((long long)n_pixels * spp >= cg_min || (same_camera && ctx->dbg.grid_min_samples < 0))) {
            const int grc = provide_camera_grid(ctx, cam, ctx->n_grid_sph);
            if (grc == RT_OK) { P.cg_bins = ctx->cg_bins; P.cg_entries = ctx->cg_entries; P.cg_tiles_x = ctx->cg_tiles_x; }
            else if (grc != RT_ERR_CAPACITY && grc != RT_ERR_UNSUPPORTED) return grc;
        }
    }
    // The kernels that only read the grids (k_closest<.., GRID>, k_shadow_grid) run in blocks of RT_GRID_BLOCK threads; their
    // LDS holds the leaf triangle records and the tables, no traversal stacks and no nodes.
    FrameParams Pg = P;      // (its output pointers follow below, once P has them)
    size_t lds_grid = 0;
    const int grid_blocks_per_cu = RT_GRID_BLOCK <= 512 ? 1024 / RT_GRID_BLOCK : (RT_GRID_BLOCK <= 640 ? 2 : 1);
    {
        const size_t per_block = ctx->lds_per_cu / grid_blocks_per_cu;
        const size_t fixed = lds_tables + 128 + 512;
        long long tc = per_block > fixed ? (long long)((per_block - fixed) / 80) : 0;
        if (tc > ctx->n_btri) tc = ctx->n_btri;
        if (ctx->dbg.tri_cache >= 0 && ctx->dbg.tri_cache < tc) tc = ctx->dbg.tri_cache;
        Pg.S.tri_cached = Pg.S.sph_cached = 0;
        Pg.S.stack_shift = RT_GRID_BLOCK == 256 ? 8 : (RT_GRID_BLOCK == 512 ? 9 : 10);      // (log2 of the waves per block + 6, for the chunk tickets)
        Pg.S.lds_cache_word = 0;
        Pg.S.btri_cached = (int32_t)tc;
        Pg.S.lds_tri_word = 0;
        Pg.S.lds_table_word = Pg.S.btri_cached * 20;
        Pg.S.lds_cam_word = Pg.S.lds_table_word + (Pg.S.mat_cached + Pg.S.light_cached) * 16;
        Pg.S.lds_ticket_word = Pg.S.lds_cam_word + 32;
        lds_grid = (size_t)Pg.S.btri_cached * 80 + lds_tables + 128 + 32;
    }
    const bool grid_kernels = ctx->dbg.grid_kernels != 0 && block == RT_BLOCK && !noalloc;

    // ---- queues
    const int rounds = ctx->dbg.rounds > 0 ? ctx->dbg.rounds : 2;        // rounds of resident blocks a queued pass is cut into
    int max_blocks = ctx->cu_count * rounds * blocks_per_cu;
    const int max_blocks_256 = ctx->cu_count * 8;                        // the same for kernels with 256-thread blocks
#ifndef LC_GRID_WAVES
#define LC_GRID_WAVES 16
#endif
    const int lightcam_blocks = ctx->cu_count * LC_GRID_WAVES * 64 / LC_BLOCK;      // sixteen waves of density queries per compute unit
    const long long fan_out = ctx->any_transparent ? 2 : 1;
    QueuePlan QP;
    FrameBuffers B;
    bool overlap = false;
    // Paired passes (k_pair): the shadow pass of a generation and the closest-hit pass of the next one share a launch.
    // Taken for batches of up to 16 M samples, where a launch's ramp and tail are a large part of it (8K x 16 spp frame
    // in 16 M batches 81.4 -> 79.1 ms; 512 x 512 frames: teapot-3 0.435 -> 0.389 ms, balls 0.723 -> 0.643, refract
    // 0.504 -> 0.459, against the two-stream overlap they replace).  Larger batches fill the chip for long enough
    // that the paired kernel's few extra register spills cost more than the launches it saves (256 M batches: 70.2 -> 73.4 ms).
    // (decided for the batch size the memory budget leaves: see the loop below)
    bool use_pair = false;
    auto pair_for = [&](long long chunks) {
        return !noalloc && !use_fused && P.S.n_lights > 0 && (ctx->dbg.pair == 1 || (ctx->dbg.pair < 0 && chunks * 64 <= (1ll << 24)));
    };
    long long fused_defer_cap = 0;
    LocalPlan LP{};
    if (use_local) {
        LP.chunks_per_block = (uint32_t)local_cpb;
        // about 0.38 of an image row of 8x8-pixel blocks, odd: a block's successive chunks then lie in columns spread over the
        // whole width, whatever the grid's relation to the row length is
        LP.rotate = ctx->dbg.local_rotate >= 0 ? (uint32_t)ctx->dbg.local_rotate : ((uint32_t)(P.blocks_x * 0.382 + 0.5) | 1u);
        const size_t bytes = (size_t)(local_grid * local_cap) * (7 * sizeof(RayRec) + 6 * (all_flat ? 64 : sizeof(HitRec)));
        if (bytes > ctx->queue_budget) use_local = false;        // the launch-per-pass path cuts the frame into batches
    }
    if (use_fused) {
        // no ray or hit queues; only the deferred light-camera queries leave the kernel (when that queue is full they are
        // answered in place, so its size is a matter of speed, not of correctness)
        if (batch_chunks > (1ll << 25) / spp * spp) batch_chunks = (1ll << 25) / spp * spp;
        if (batch_chunks < spp) batch_chunks = spp;
        int rc;
        if ((rc = ensure(&ctx->accum, &ctx->accum_values, n_pixels * 3)) != RT_OK) return rc;
        if (use_lightcam) {
            const long long bs = batch_chunks < n_chunks ? batch_chunks : n_chunks;
            fused_defer_cap = (bs * 64 * ctx->growth * (P.S.n_lights > 0 ? P.S.n_lights : 1) + N_SHARDS - 1) / N_SHARDS;
            if (fused_defer_cap < 1024) fused_defer_cap = 1024;
            if ((rc = ensure(&ctx->deferred[0], &ctx->deferred_capacity[0], (size_t)(fused_defer_cap * N_SHARDS))) != RT_OK) return rc;
        }
        B.accum = ctx->accum;
        B.deferred[0] = ctx->deferred[0];
        B.counts = ctx->counts;
        B.totals = ctx->totals;
        } else if (use_local) {
        const long long cap = local_cap;
        LP.ray_cap = LP.hit_cap = (uint32_t)cap;
        const size_t hit_stride = all_flat ? 64 : sizeof(HitRec);
        int rc;
        if ((rc = ensure(&ctx->accum, &ctx->accum_values, n_pixels * 3)) != RT_OK) return rc;
        if ((rc = ensure(&ctx->rays[0], &ctx->ray_capacity[0], (size_t)(7 * local_grid * cap))) != RT_OK) return rc;
        LP.rays = ctx->rays[0];
        // a hit slice per generation: shadow chunks are taken whenever a wave has nothing to trace, so a generation's hits
        // may still be waiting when the generation after the next is being traced
        if ((rc = ensure(&ctx->hits[0], &ctx->hit_capacity[0], ((size_t)(6 * local_grid * cap) * hit_stride + sizeof(HitRec) - 1) / sizeof(HitRec))) != RT_OK) return rc;
        LP.hits = ctx->hits[0];
        LP.lds_counts_word = lds_counts_word;
        LP.frame_counts = ctx->counts;
        if (use_lightcam) {
            // postponed light-camera evaluations of all generations in one sharded list (as the lane-per-sample kernel keeps it)
            fused_defer_cap = (n_chunks * 64 * ctx->growth * (P.S.n_lights > 0 ? P.S.n_lights : 1) + N_SHARDS - 1) / N_SHARDS;
            if (fused_defer_cap < 1024) fused_defer_cap = 1024;
            if ((rc = ensure(&ctx->deferred[0], &ctx->deferred_capacity[0], (size_t)(fused_defer_cap * N_SHARDS))) != RT_OK) return rc;
            LP.deferred = ctx->deferred[0];
            LP.defer_shard_cap = (uint32_t)fused_defer_cap;
            B.deferred[0] = ctx->deferred[0];
        }
        B.accum = ctx->accum;
        B.counts = ctx->counts;
        B.totals = ctx->totals;
        } else if (local_rows > 0 && !noalloc) {
        // Overlap pays when a pass cannot fill the chip by itself (small frames: each pass is one
        // ray's latency long); big batches already saturate it with one persistent grid, and there
        // the extra ordering constraints only cost, so they run on one stream.
        // With overlap, four hit buffers: [batch parity][generation parity], so that the shadow pass of
        // one generation (second stream) can run beside the closest-hit pass of the next one; on one
        // stream a single buffer serves every pass.
        // The queues of a batch must fit the context's memory budget, and the allocations must succeed: the batch is
        // halved until both hold.
        // Over budget, the frame is cut into as many batches as the budget allows if bytes go with the batch size,
        // and at least one more than now; when an allocation fails, the batch is halved.
        for (;;) {
            use_pair = pair_for(batch_chunks);
            overlap = !use_pair && batch_chunks * 64 <= (1ll << 22) && block == 256;
            if (ctx->dbg.overlap >= 0 && !use_pair) overlap = ctx->dbg.overlap != 0;
            plan_queues(QP, batch_chunks, waves_per_block, max_blocks, fan_out, ctx->growth, ctx->dbg.units != 0);
            QP.hit_buffers = overlap ? 4 : (use_pair ? 2 : 1);
            QP.hit_stride = all_flat ? 64 : (long long)sizeof(HitRec);
            QP.deferred = use_lightcam;
            QP.defer_mult = P.S.n_lights > 0 ? P.S.n_lights : 1;
            const bool smallest = batch_chunks <= spp;
            const size_t plan_bytes = QP.bytes();
            const bool over_budget = plan_bytes > ctx->queue_budget && !smallest;
            int rc = RT_OK;
            if (over_budget) rc = RT_ERR_CAPACITY;
            else if (QP.need_rays[0] >= (1ll << 32) || QP.need_rays[1] >= (1ll << 32) || QP.need_hits >= (1ll << 32)) rc = RT_ERR_CAPACITY;
            else {
                rc = buffers_from_context(ctx, QP, n_pixels, B);
                if (rc != RT_OK && !smallest) release_queues(ctx);      // start over from nothing at the smaller size
            }
            if (rc == RT_OK) break;
            if (smallest) return rc == RT_ERR_CAPACITY ? fail(RT_ERR_CAPACITY, "rt_render: the queues of the smallest batch exceed the memory budget") : rc;
            long long next = (batch_chunks + 1) / 2;
            if (over_budget && ctx->batch_override <= 0) {
                // the number of batches the budget allows if bytes go with the batch size, and at least one more than now
                const long long now = (n_chunks + batch_chunks - 1) / batch_chunks;
                long long want = (long long)((double)now * (double)plan_bytes / (double)ctx->queue_budget) + 1;
                if (want <= now) want = now + 1;
                next = (n_chunks + want - 1) / want;
                if (next >= batch_chunks) next = batch_chunks - spp;
            }
            batch_chunks = even_batches(n_chunks, next, spp);
        }

    } else if (local_rows > 0) {
        // The static scratch holds the accumulation frame, the counters and one batch's queues.  Among the grids of
        // queued passes that fit, the one that keeps most waves busy wins: a smaller grid has smaller per-shard floors.
        long long best_score = -1;
        QueuePlan best;
        const int grids[] = {ctx->cu_count * 4, ctx->cu_count * 2, ctx->cu_count, ctx->cu_count / 2, N_SHARDS};
        for (int grid : grids) {
            if (grid < 1) continue;
            long long bc = even_batches(n_chunks, wanted_batch_chunks, spp);
            for (;;) {
                QueuePlan q;
                plan_queues(q, bc, waves_per_block, grid, fan_out, ctx->growth, ctx->dbg.units != 0);
                q.hit_buffers = 1;
                q.deferred = use_lightcam;
                q.defer_mult = P.S.n_lights > 0 ? P.S.n_lights : 1;
                if (scratch_need(q, n_pixels) <= RT_NOALLOC_SCRATCH_BYTES && q.need_hits < (1ll << 32)) {
                    const long long waves = (long long)grid * waves_per_block;
                    const long long score = (bc < waves ? bc : waves) * 4 + (bc >= waves ? 1 : 0) * (bc / waves < 1000 ? bc / waves : 1000);
                    if (score > best_score) { best_score = score; best = q; }
                    break;
                }
                if (bc <= spp) break;
                bc = even_batches(n_chunks, (bc + 1) / 2, spp);
            }
        }
        if (best_score < 0)
            return fail(RT_ERR_CAPACITY, "rt_render_noalloc: this frame does not fit the static scratch of " +
                                             std::to_string(RT_NOALLOC_SCRATCH_BYTES >> 20) + " MiB (" + std::to_string(n_pixels * 24 >> 20) +
                                             " MiB of it for the accumulation frame alone); render it in parts or use rt_render");
        QP = best;
        batch_chunks = QP.batch_chunks;
        max_blocks = QP.max_blocks;
        buffers_from_scratch(static_cast<unsigned char *>(ctx->scratch), QP, n_pixels, B);
    } else {
        B.totals = noalloc ? reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(ctx->scratch)) : ctx->totals;
    }
    const long long *hit_cap = QP.hit_cap, *ray_cap = QP.ray_cap;
    const long long blocks0 = QP.blocks0;

    hipStream_t st = ctx->stream;
    HIP_TRY(hipEventRecord(ctx->ev_begin, st));
    // (the first batch's counter block lies right behind the context's totals: one memset for both)
    const bool counts0_clear = !noalloc && local_rows > 0 && B.totals == ctx->totals && B.counts == ctx->counts;
    HIP_TRY(hipMemsetAsync(B.totals, 0, RT_TOTALS_WORDS * sizeof(unsigned long long) + (counts0_clear ? CNT_WORDS * sizeof(uint32_t) : 0), st));
    ctx->ev_closest_used = ctx->ev_shadow_used = ctx->ev_pair_used = 0;
    ctx->pending_closest_per_batch = use_pair ? 1 : 6;
    ctx->pending_flags = o->flags;
    ctx->pending_batches = 0;
    ctx->pending_primaries = (unsigned long long)n_pixels * (unsigned long long)(N * N);
    ctx->pending_totals = B.totals;
    ctx->pending_fused = use_fused || use_local;

    if (local_rows > 0) {
        // k_finalize leaves the values it has read at zero, so only a frame that finds the buffer new, larger than the last
        // frame's or left behind by a failed call clears it first
        if (!(B.accum == ctx->accum && ctx->accum_clean_ptr == ctx->accum && ctx->accum_clean >= n_pixels * 3))
            HIP_TRY(hipMemsetAsync(B.accum, 0, n_pixels * 3 * sizeof(double), st));
        ctx->accum_clean = 0;
        P.accum = B.accum;

        // outputs: device pointers are used directly, host pointers go through staging buffers
        void *dst[4] = {out->rgb_f64, out->rgb_u8, out->hit_id, out->lightcam_mask};
        const size_t bytes[4] = {n_pixels * 3 * sizeof(double), n_pixels * 3, n_pixels * sizeof(int32_t), n_pixels};
        void *dev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 4; ++k) {
            if (!dst[k]) continue;
            if (outputs_on_device) dev[k] = dst[k];
            else {
                if (ctx->stage_bytes[k] < bytes[k]) {
                    if (ctx->stage[k]) HIP_TRY(hipFree(ctx->stage[k]));
                    ctx->stage[k] = nullptr; ctx->stage_bytes[k] = 0;
                    HIP_TRY(hipMalloc(&ctx->stage[k], bytes[k]));
                    ctx->stage_bytes[k] = bytes[k];
                }
                dev[k] = ctx->stage[k];
            }
        }
        P.hit_id = static_cast<int32_t *>(dev[2]);
        P.lc_mask = static_cast<uint8_t *>(dev[3]);
        Pg.accum = P.accum; Pg.hit_id = P.hit_id; Pg.lc_mask = P.lc_mask;
        if (dev[3]) HIP_TRY(hipMemsetAsync(dev[3], 0, bytes[3], st));

        // Two streams.  A: counters memset, closest-hit passes C0..C5 of every batch, in order.
        // B: shadow/shading passes S0..S5 and the batch's counter roll-up.  S_g waits for C_g; C_g+2
        // waits for S_g (it re-uses S_g's hit buffer); a batch's memset waits for the roll-up of
        // the batch two before it (same counter block).  Pixel accumulation is atomic, so the
        // order in which the passes finish does not matter.
        hipStream_t sb = overlap ? ctx->stream2 : st;
        // The closest-hit passes of the first generations push whole chunks (QueueSet::whole_chunks).  Where a hit can spawn
        // two rays the queues of the later generations are sized by expected growth, not by the worst case, and fillers
        // could overflow them: only the primary pass then.  Not in the entry point that allocates nothing, whose queues
        // are cut to what the scratch array holds.
        const int whole_generations = noalloc ? 0 : (ctx->any_transparent && ctx->dbg.whole_chunks > 1 ? 1 : ctx->dbg.whole_chunks);
        const bool timing = (o->flags & RT_FLAG_TIMING) != 0;
        long long batch_no = 0;
        if (use_fused) {
            const void *kfn = !bvh_closest_ok ? (const void *)k_fused<false, SHADOW_EXACT>
                              : bvh_shadow_ok ? (const void *)k_fused<true, SHADOW_BVH>
                              : hybrid_shadow ? (const void *)k_fused<true, SHADOW_HYBRID>
                                              : (const void *)k_fused<true, SHADOW_EXACT>;
            int resident = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kfn, 256, lds_bvh));
            if (resident < 1) resident = 1;
            if (resident > 8) resident = 8;
            for (long long t0 = 0; t0 < n_chunks; t0 += batch_chunks, ++batch_no) {
                const long long t1 = (t0 + batch_chunks < n_chunks) ? t0 + batch_chunks : n_chunks;
                uint32_t *counts = B.counts;
                if (!(batch_no == 0 && counts0_clear)) HIP_TRY(hipMemsetAsync(counts, 0, CNT_WORDS * sizeof(uint32_t), st));
                QueueSet Q{};
                Q.deferred = B.deferred[0];
                Q.counts = counts;
                Q.hit_shard_cap = (uint32_t)fused_defer_cap;
                Q.defer_shard_cap = (uint32_t)fused_defer_cap;
                Q.generation = 0;
                Q.chunk_begin = t0;
                Q.chunk_end = t1;
                Q.block_row0 = (t0 / (N * N)) / P.blocks_x;
                Q.block_col0 = (int32_t)((t0 / (N * N)) % P.blocks_x);
                Q.totals = B.totals;
                long long blocks = (t1 - t0 + 3) / 4;
                if (blocks > (long long)ctx->cu_count * resident) blocks = (long long)ctx->cu_count * resident;
                std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
                if (timing && !(ev = next_event_pair(ctx->ev_closest, ctx->ev_closest_used)))
                    return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                if (!bvh_closest_ok) hipLaunchKernelGGL((k_fused<false, SHADOW_EXACT>), dim3((unsigned)blocks), dim3(256), lds_bvh, st, P, Q);
                else if (bvh_shadow_ok) hipLaunchKernelGGL((k_fused<true, SHADOW_BVH>), dim3((unsigned)blocks), dim3(256), lds_bvh, st, P, Q);
                else if (hybrid_shadow) hipLaunchKernelGGL((k_fused<true, SHADOW_HYBRID>), dim3((unsigned)blocks), dim3(256), lds_bvh, st, P, Q);
                else hipLaunchKernelGGL((k_fused<true, SHADOW_EXACT>), dim3((unsigned)blocks), dim3(256), lds_bvh, st, P, Q);
                if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                if (use_lightcam) {
                    int lrc = provide_lightcams(ctx, lc_flags, may_cull, true);
                    if (lrc != RT_OK) return lrc;
                    LightcamQueues L{};
                    L.deferred[0] = B.deferred[0];
                    L.shard_cap[0] = (uint32_t)fused_defer_cap;
                    L.counts = counts;
                    L.overflow = &counts[CNT_OVERFLOW];
                    L.next_chunk = &counts[CNT_LCWORK];
                    L.pool_cap = ctx->dbg.lightcam_pool;
                    hipLaunchKernelGGL(k_lightcam, dim3((unsigned)lightcam_blocks), dim3(LC_BLOCK), 0, st, P, L);
                }
                hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, st, counts, B.totals, P.S.n_lights);
                ctx->pending_batches++;
            }
            HIP_TRY(hipEventRecord(ctx->ev_batch[0], st));
        } else if (use_local) {
            uint32_t *counts = B.counts;
            if (!counts0_clear) HIP_TRY(hipMemsetAsync(counts, 0, CNT_WORDS * sizeof(uint32_t), st));
            QueueSet Q{};
            Q.counts = counts;
            Q.hit_stride = all_flat ? 64u : 128u;
            Q.whole_chunks = (uint32_t)whole_generations;      // (k_local: the number of generations)
            Q.chunk_begin = 0;
            Q.chunk_end = n_chunks;
            Q.block_row0 = 0;
            Q.block_col0 = 0;
            std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
            if (timing && !(ev = next_event_pair(ctx->ev_closest, ctx->ev_closest_used))) return fail(RT_ERR_DEVICE, "hipEventCreate failed");
            if (ev) HIP_TRY(hipEventRecord(ev->first, st));
            if (bvh_shadow_ok) hipLaunchKernelGGL((k_local<SHADOW_BVH>), dim3((unsigned)local_grid), dim3(block), lds_local, st, P, Q, LP);
            else hipLaunchKernelGGL((k_local<SHADOW_HYBRID>), dim3((unsigned)local_grid), dim3(block), lds_local, st, P, Q, LP);
            if (ev) HIP_TRY(hipEventRecord(ev->second, st));
            if (use_lightcam) {
                int lrc = provide_lightcams(ctx, lc_flags, may_cull, true);
                if (lrc != RT_OK) return lrc;
                LightcamQueues L{};
                L.deferred[0] = B.deferred[0];
                L.shard_cap[0] = (uint32_t)fused_defer_cap;
                L.counts = counts;
                L.overflow = &counts[CNT_OVERFLOW];
                L.next_chunk = &counts[CNT_LCWORK];
                L.pool_cap = ctx->dbg.lightcam_pool;
                hipLaunchKernelGGL(k_lightcam, dim3((unsigned)lightcam_blocks), dim3(LC_BLOCK), 0, st, P, L);
            }
            ctx->pending_batches++;      // (k_finalize rolls the counters up)
            batch_no = 1;
            HIP_TRY(hipEventRecord(ctx->ev_batch[0], st));
        } else
        for (long long t0 = 0; t0 < n_chunks; t0 += batch_chunks, ++batch_no) {
            const long long t1 = (t0 + batch_chunks < n_chunks) ? t0 + batch_chunks : n_chunks;
            const int par = (int)(batch_no & 1);
            uint32_t *counts = B.counts + (size_t)par * CNT_WORDS;
            if (batch_no >= 2) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[par], 0));
            if (!(batch_no == 0 && counts0_clear)) HIP_TRY(hipMemsetAsync(counts, 0, CNT_WORDS * sizeof(uint32_t), st));
            // (diagnostic, option "ring_poison": see k_poison_ring)
            auto poison_ring = [&](const QueueSet &Q, uint32_t *cnt, bool grid_kernel) {
                if (!ctx->dbg.ring_poison || !Q.unit_chunks) return;
                const unsigned resident = (unsigned)(ctx->cu_count * (grid_kernel ? grid_blocks_per_cu : blocks_per_cu));
                hipLaunchKernelGGL(k_poison_ring, dim3(resident), dim3(grid_kernel ? RT_GRID_BLOCK : block), grid_kernel ? lds_grid : lds_closest, st,
                                   (grid_kernel ? Pg.S.lds_ticket_word : P.S.lds_ticket_word) + 4, cnt + CNT_ARRIVED, resident);
            };
            // Direct placement (QueueSet::direct_counts): the primary pass of a batch that is handed out in units asks no counter
            // for places -- a quarter of such a pass was the wait for them -- and a small kernel lists the groups of 64 records
            // that hold anything for the two passes that read them (shadow pass 0, closest-hit pass 1).
            const long long batch_now = t1 - t0;
            const uint32_t dfan = ctx->any_transparent ? 2u : 1u;
            const bool direct = ctx->dbg.direct != 0 && !noalloc && !overlap && QP.unit_chunks != 0 &&
                                (long long)hit_cap[0] * N_SHARDS >= batch_now * 64 && (long long)ray_cap[1] * N_SHARDS >= batch_now * 64 * dfan &&
                                (long long)hit_cap[0] * N_SHARDS < (1ll << 32) && (long long)ray_cap[1] * N_SHARDS < (1ll << 32);
            uint32_t *d_counts = nullptr, *d_hit_list = nullptr, *d_ray_list = nullptr;
            if (direct) {
                const size_t need = (size_t)batch_now * 4 * (2 + dfan) + 1024;
                int drc = ws_ensure(&ctx->direct_ws, &ctx->direct_ws_bytes, need);
                if (drc != RT_OK) return drc;
                d_counts = static_cast<uint32_t *>(ctx->direct_ws);
                d_hit_list = d_counts + ((batch_now + 63) & ~63ll);
                d_ray_list = d_hit_list + ((batch_now + 63) & ~63ll);
            }
            // ... and where the primary pass is the grid-only kernel and the shadow pass would be too, the primary pass deals with
            // the lights at its hits itself: no hit records, no shadow pass for generation 0 (QueueSet::fused_lights)
            const bool fuse0 = direct && ctx->dbg.fuse_lights != 0 && grid_kernels && P.cg_bins && bvh_closest_ok && bvh_shadow_ok && P.S.lg_bins &&
                               P.S.n_lights > 0;
            auto apply_direct = [&](QueueSet &Q, int g) {
                if (!direct) return;
                if (g == 0) {
                    Q.direct_counts = d_counts; Q.direct_fan = dfan; Q.hit_list = d_hit_list;
                    Q.fused_lights = fuse0 ? 1u : 0u;
                    Q.hit_shard_cap = (uint32_t)(hit_cap[0] * N_SHARDS);
                    Q.out_shard_cap = (uint32_t)(ray_cap[1] * N_SHARDS);
                } else if (g == 1) {
                    Q.ray_list = d_ray_list;
                    Q.in_shard_cap = (uint32_t)(ray_cap[1] * N_SHARDS);
                }
            };
            auto compact_direct = [&](hipStream_t s_) {
                if (direct)
                    hipLaunchKernelGGL(k_compact_direct, dim3((unsigned)((batch_now + 255) / 256)), dim3(256), 0, s_, d_counts, (uint32_t)batch_now, dfan, 0, counts,
                                       d_hit_list, d_ray_list);
            };
            auto queue_set = [&](int g) {
                QueueSet Q{};
                Q.rays_in = B.rays[g & 1];
                Q.rays_out = B.rays[(g + 1) & 1];
                Q.hits = B.hits[g & 1];
                Q.deferred = B.deferred[g];
                Q.counts = counts;
                Q.in_shard_cap = (uint32_t)ray_cap[g];
                Q.out_shard_cap = (uint32_t)(g < 5 ? ray_cap[g + 1] : 0);
                Q.hit_shard_cap = (uint32_t)hit_cap[g];
                Q.defer_shard_cap = (uint32_t)(hit_cap[g] * QP.defer_mult);
                Q.hit_stride = all_flat ? 64u : 128u;
                Q.generation = g;
                Q.unit_chunks = g == 0 ? QP.unit_chunks : 0u;
                Q.whole_chunks = g < whole_generations ? 1u : 0u;
                Q.unit_mult = 0;
                if (g == 0 && QP.unit_chunks) {
                    // Units are handed out in the order (k * m) mod n, m near 0.618 n and sharing no factor with n:
                    // units worked on at the same time then lie all over the batch's rows, so the parts of the image
                    // where every sample hits (and writes its records) are mixed with the parts where few do, instead of
                    // the whole chip sweeping through them together (8K teapot frame in one batch: primary pass 24.2 ->
                    // 20.9 ms, in two batches 21.3 -> 20.8 ms).
                    const unsigned long long nu = (unsigned long long)((t1 - t0 + QP.unit_chunks - 1) / QP.unit_chunks);
                    unsigned long long a = (unsigned long long)((double)nu * 0.6180339887) | 1ull;
                    auto gcd = [](unsigned long long x, unsigned long long y) { while (y) { unsigned long long r = x % y; x = y; y = r; } return x; };
                    while (a > 1 && gcd(a, nu) != 1) a += 2;
                    Q.unit_mult = (uint32_t)(a % (nu ? nu : 1)) | (ctx->dbg.ring_poison ? 0x80000000u : 0u);
                }
                Q.chunk_begin = t0;
                Q.chunk_end = t1;
                Q.block_row0 = (t0 / (N * N)) / P.blocks_x;
                Q.block_col0 = (int32_t)((t0 / (N * N)) % P.blocks_x);
                apply_direct(Q, g);
                return Q;
            };
            if (use_pair) {
                // C0, then S_g + C_g+1 for g = 0..4, then S5: seven launches on one stream, hit buffers alternating by generation
                std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
                {
                    const QueueSet Q = queue_set(0);
                    if (timing && !(ev = next_event_pair(ctx->ev_closest, ctx->ev_closest_used))) return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                    long long blocks = (t1 - t0 + waves_per_block - 1) / waves_per_block;
                    if (blocks > blocks0) blocks = blocks0;
                    const bool gk = grid_kernels && P.cg_bins && Q.unit_chunks && bvh_closest_ok;
                    poison_ring(Q, counts, gk);
                    if (gk) {
                        long long gb = (t1 - t0 + RT_GRID_BLOCK / 64 - 1) / (RT_GRID_BLOCK / 64);
                        if (gb > (long long)ctx->cu_count * grid_blocks_per_cu * rounds) gb = (long long)ctx->cu_count * grid_blocks_per_cu * rounds;
                        if (!Q.fused_lights) hipLaunchKernelGGL((k_closest<true, true, true>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                        else if (all_flat) hipLaunchKernelGGL((k_closest<true, true, true, 2>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                        else hipLaunchKernelGGL((k_closest<true, true, true, 1>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                    } else
                    if (bvh_closest_ok) hipLaunchKernelGGL((k_closest<true, true>), dim3((unsigned)blocks), dim3(block), lds_closest, st, P, Q);
                    else hipLaunchKernelGGL((k_closest<true, false>), dim3((unsigned)blocks), dim3(block), lds_closest, st, P, Q);
                    compact_direct(st);
                    if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                }
                for (int g = 0; g < 5; ++g) {
                    const QueueSet Qs = queue_set(g), Qc = queue_set(g + 1);
                    ev = nullptr;
                    if (timing && !(ev = next_event_pair(ctx->ev_pair, ctx->ev_pair_used))) return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                    if (!bvh_closest_ok) hipLaunchKernelGGL((k_pair<false, SHADOW_EXACT>), dim3((unsigned)max_blocks), dim3(block), lds_bvh, st, P, Qs, Qc);
                    else if (bvh_shadow_ok) hipLaunchKernelGGL((k_pair<true, SHADOW_BVH>), dim3((unsigned)max_blocks), dim3(block), lds_bvh, st, P, Qs, Qc);
                    else if (hybrid_shadow) hipLaunchKernelGGL((k_pair<true, SHADOW_HYBRID>), dim3((unsigned)max_blocks), dim3(block), lds_bvh, st, P, Qs, Qc);
                    else hipLaunchKernelGGL((k_pair<true, SHADOW_EXACT>), dim3((unsigned)max_blocks), dim3(block), lds_bvh, st, P, Qs, Qc);
                    if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                }
                {
                    const QueueSet Q = queue_set(5);
                    ev = nullptr;
                    if (timing && !(ev = next_event_pair(ctx->ev_shadow, ctx->ev_shadow_used))) return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                    if (bvh_shadow_ok) hipLaunchKernelGGL((k_shadow<true>), dim3((unsigned)max_blocks), dim3(block), lds_shadow, st, P, Q);
                    else if (hybrid_shadow) hipLaunchKernelGGL(k_shadow_hybrid, dim3((unsigned)max_blocks), dim3(block), lds_shadow, st, P, Q);
                    else hipLaunchKernelGGL((k_shadow<false>), dim3((unsigned)max_blocks), dim3(block), lds_shadow, st, P, Q);
                    if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                }
                HIP_TRY(hipEventRecord(ctx->ev_c[5], st));
            } else
            for (int g = 0; g < 6; ++g) {
                QueueSet Q{};
                Q.rays_in = B.rays[g & 1];
                Q.rays_out = B.rays[(g + 1) & 1];
                Q.hits = B.hits[overlap ? par * 2 + (g & 1) : 0];
                Q.deferred = B.deferred[g];
                Q.counts = counts;
                Q.in_shard_cap = (uint32_t)ray_cap[g];
                Q.out_shard_cap = (uint32_t)(g < 5 ? ray_cap[g + 1] : 0);
                Q.hit_shard_cap = (uint32_t)hit_cap[g];
                Q.defer_shard_cap = (uint32_t)(hit_cap[g] * QP.defer_mult);
                Q.hit_stride = all_flat ? 64u : 128u;
                Q.generation = g;
                Q.unit_chunks = g == 0 ? QP.unit_chunks : 0u;
                Q.whole_chunks = g < whole_generations ? 1u : 0u;
                Q.unit_mult = 0;
                if (g == 0 && QP.unit_chunks) {
                    // Units are handed out in the order (k * m) mod n, m near 0.618 n and sharing no factor with n:
                    // units worked on at the same time then lie all over the batch's rows, so the parts of the image
                    // where every sample hits (and writes its records) are mixed with the parts where few do, instead of
                    // the whole chip sweeping through them together (8K teapot frame in one batch: primary pass 24.2 ->
                    // 20.9 ms, in two batches 21.3 -> 20.8 ms).
                    const unsigned long long nu = (unsigned long long)((t1 - t0 + QP.unit_chunks - 1) / QP.unit_chunks);
                    unsigned long long a = (unsigned long long)((double)nu * 0.6180339887) | 1ull;
                    auto gcd = [](unsigned long long x, unsigned long long y) { while (y) { unsigned long long r = x % y; x = y; y = r; } return x; };
                    while (a > 1 && gcd(a, nu) != 1) a += 2;
                    Q.unit_mult = (uint32_t)(a % (nu ? nu : 1)) | (ctx->dbg.ring_poison ? 0x80000000u : 0u);
                }
                Q.chunk_begin = t0;
                Q.chunk_end = t1;
                Q.block_row0 = (t0 / (N * N)) / P.blocks_x;
                Q.block_col0 = (int32_t)((t0 / (N * N)) % P.blocks_x);
                apply_direct(Q, g);
                if (g >= 2 && P.S.n_lights > 0) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_s[g - 2], 0));
                std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
                if (timing && !(ev = next_event_pair(ctx->ev_closest, ctx->ev_closest_used)))
                    return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                if (ev) HIP_TRY(hipEventRecord(ev->first, st));
                if (g == 0) {
                    long long blocks = (t1 - t0 + waves_per_block - 1) / waves_per_block;
                    if (blocks > blocks0) blocks = blocks0;
                    const bool gk = grid_kernels && P.cg_bins && Q.unit_chunks && bvh_closest_ok;
                    poison_ring(Q, counts, gk);
                    if (gk) {
                        long long gb = (t1 - t0 + RT_GRID_BLOCK / 64 - 1) / (RT_GRID_BLOCK / 64);
                        if (gb > (long long)ctx->cu_count * grid_blocks_per_cu * rounds) gb = (long long)ctx->cu_count * grid_blocks_per_cu * rounds;
                        if (!Q.fused_lights) hipLaunchKernelGGL((k_closest<true, true, true>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                        else if (all_flat) hipLaunchKernelGGL((k_closest<true, true, true, 2>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                        else hipLaunchKernelGGL((k_closest<true, true, true, 1>), dim3((unsigned)gb), dim3(RT_GRID_BLOCK), lds_grid, st, Pg, Q);
                    } else
                    if (bvh_closest_ok) hipLaunchKernelGGL((k_closest<true, true>), dim3((unsigned)blocks), dim3(block), lds_closest, st, P, Q);
                    else hipLaunchKernelGGL((k_closest<true, false>), dim3((unsigned)blocks), dim3(block), lds_closest, st, P, Q);
                    compact_direct(st);
                } else {
                    if (bvh_closest_ok) hipLaunchKernelGGL((k_closest<false, true>), dim3((unsigned)max_blocks), dim3(block), lds_closest, st, P, Q);
                    else hipLaunchKernelGGL((k_closest<false, false>), dim3((unsigned)max_blocks), dim3(block), lds_closest, st, P, Q);
                }
                if (ev) HIP_TRY(hipEventRecord(ev->second, st));
                HIP_TRY(hipEventRecord(ctx->ev_c[g], st));
                if (P.S.n_lights > 0) {
                    HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_c[g], 0));
                    const bool no_hits = g == 0 && fuse0;      // (generation 0 has no hit records: nothing to launch, nothing to time)
                    ev = nullptr;
                    if (timing && !no_hits && !(ev = next_event_pair(ctx->ev_shadow, ctx->ev_shadow_used)))
                        return fail(RT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev) HIP_TRY(hipEventRecord(ev->first, sb));
                    if (no_hits) { }
                    else if (bvh_shadow_ok && grid_kernels && P.S.lg_bins)
                        hipLaunchKernelGGL(k_shadow_grid, dim3((unsigned)(ctx->cu_count * grid_blocks_per_cu * rounds)), dim3(RT_GRID_BLOCK), lds_grid, sb, Pg, Q);
                    else if (bvh_shadow_ok) hipLaunchKernelGGL((k_shadow<true>), dim3((unsigned)max_blocks), dim3(block), lds_shadow, sb, P, Q);
                    else if (hybrid_shadow) hipLaunchKernelGGL(k_shadow_hybrid, dim3((unsigned)max_blocks), dim3(block), lds_shadow, sb, P, Q);
                    else hipLaunchKernelGGL((k_shadow<false>), dim3((unsigned)max_blocks), dim3(block), lds_shadow, sb, P, Q);
                    if (ev) HIP_TRY(hipEventRecord(ev->second, sb));
                    HIP_TRY(hipEventRecord(ctx->ev_s[g], sb));
                }
            }
            HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_c[5], 0));
            if (use_lightcam) {
                // the cameras this batch's shading points asked for, then the density queries of all six generations in one
                // launch, after the last shadow pass
                int lrc = provide_lightcams(ctx, lc_flags, may_cull, !noalloc);
                if (lrc != RT_OK) return lrc;
                LightcamQueues L{};
                for (int g = 0; g < 6; ++g) { L.deferred[g] = B.deferred[g]; L.shard_cap[g] = (uint32_t)(hit_cap[g] * QP.defer_mult); }
                L.counts = counts;
                L.overflow = &counts[CNT_OVERFLOW];
                L.next_chunk = &counts[CNT_LCWORK];
                L.pool_cap = ctx->dbg.lightcam_pool;
                hipLaunchKernelGGL(k_lightcam, dim3((unsigned)lightcam_blocks), dim3(LC_BLOCK), 0, sb, P, L);
            }
            hipLaunchKernelGGL(k_collect_counts, dim3(1), dim3(64), 0, sb, counts, B.totals, P.S.n_lights);
            HIP_TRY(hipEventRecord(ctx->ev_batch[par], sb));
            ctx->pending_batches++;
        }
        // the frame is complete when the second stream has drained
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[0], 0));
        if (batch_no >= 2 && !use_fused && !use_local) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_batch[1], 0));
        HIP_TRY(hipGetLastError());
        {
            const long long n_values = (long long)n_pixels * 3;
            long long blocks = (n_values + 255) / 256;
            if (blocks > max_blocks_256) blocks = max_blocks_256;
            hipLaunchKernelGGL(k_finalize, dim3((unsigned)blocks), dim3(256), 0, st, B.accum, n_values, N * N,
                               static_cast<double *>(dev[0]), static_cast<uint8_t *>(dev[1]),
                               use_local ? (const uint32_t *)B.counts : (const uint32_t *)nullptr, B.totals, ctx->h_totals, (int)P.S.n_lights);
            HIP_TRY(hipGetLastError());
            if (B.accum == ctx->accum) { ctx->accum_clean = (size_t)n_values; ctx->accum_clean_ptr = ctx->accum; }
        }
        if (!outputs_on_device)
            for (int k = 0; k < 4; ++k)
                if (dst[k]) HIP_TRY(hipMemcpyAsync(dst[k], dev[k], bytes[k], hipMemcpyDeviceToHost, st));
    }
    if (local_rows <= 0)     // (else k_finalize has handed them over)
        HIP_TRY(hipMemcpyAsync(ctx->h_totals, B.totals, RT_TOTALS_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(ctx->ev_end, st));
    ctx->pending = true;
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx *ctx, rt_stats *stats)
{
    if (!ctx) return fail(RT_ERR_INVALID, "rt_sync: null argument");
    if (!ctx->pending) return fail(RT_ERR_INVALID, "rt_sync: no frame in flight");
    HIP_TRY(hipSetDevice(ctx->device));
    ctx->pending = false;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long tot[RT_TOTALS_WORDS];
    std::memcpy(tot, ctx->h_totals, sizeof tot);      // copied by the frame's last command
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        tot[0] += ctx->pending_primaries;   // generation 0 is counted on the host
        tot[2] = ctx->pending_primaries;
        stats->closest_rays = tot[0];
        stats->lightcam_rays = ctx->lc_rays;
        stats->shadow_rays = tot[1];
        for (int g = 0; g < 6; ++g) { stats->rays_per_generation[g] = tot[2 + g]; stats->hits_per_generation[g] = tot[8 + g]; }
        stats->queue_overflow = tot[15] ? 1 : 0;
        stats->batches = ctx->pending_batches;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        stats->gpu_ms = ms;
        if (ctx->pending_flags & RT_FLAG_TIMING) {
            double c = 0, s = 0;
            double prim = 0;
            uint64_t prim_n = 0;
            for (size_t i = 0; i < ctx->ev_closest_used; ++i) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, ctx->ev_closest[i].first, ctx->ev_closest[i].second));
                c += m;
                if (ctx->pending_fused || i % (size_t)ctx->pending_closest_per_batch == 0) { prim += m; ++prim_n; }   // generation 0 first
            }
            stats->primary_ms = prim;
            stats->primary_launches = prim_n;
            for (size_t i = 0; i < ctx->ev_shadow_used; ++i) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, ctx->ev_shadow[i].first, ctx->ev_shadow[i].second));
                s += m;
            }
            double pr = 0;
            for (size_t i = 0; i < ctx->ev_pair_used; ++i) {
                float m = 0;
                HIP_TRY(hipEventElapsedTime(&m, ctx->ev_pair[i].first, ctx->ev_pair[i].second));
                pr += m;
            }
            stats->pair_ms = pr;
            stats->pair_launches = ctx->ev_pair_used;
            stats->closest_ms = c;
            stats->shadow_ms = s;
            stats->other_ms = ms - c - s - pr;
            stats->closest_launches = ctx->ev_closest_used;
            stats->shadow_launches = ctx->ev_shadow_used;
        }
    }
    if (tot[15]) {
        // the next frame of this scene gets queues twice as deep (up to the worst case, 32 rays per primary in a generation)
        if (ctx->growth < 32) ctx->growth *= 2;
        return fail(RT_ERR_CAPACITY, "rt_render: a ray queue overflowed; the frame is incomplete (the context has doubled its queue depth "
                                     "for the next frame)");
    }
    return RT_OK;
}

extern "C" int rt_render_async(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out)
{
    if (out && !out->on_device) return fail(RT_ERR_INVALID, "rt_render_async: outputs must be device memory");
    return enqueue_frame(ctx, cam, opts, out, FRAME_DEVICE_OUT);
}

static void abandon_frame(rt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    ctx->pending = false;
}

extern "C" int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out,
                         rt_stats *stats)
{
    // A frame whose ray tree outgrew the queues is rendered again with deeper ones (rt_sync has doubled the depth);
    // five doublings reach the worst case, which cannot overflow.
    for (int attempt = 0;; ++attempt) {
        int rc = enqueue_frame(ctx, cam, opts, out, (out && out->on_device) ? FRAME_DEVICE_OUT : FRAME_HOST_OUT);
        if (rc != RT_OK) { abandon_frame(ctx); return rc; }
        rc = rt_sync(ctx, stats);
        if (rc != RT_ERR_CAPACITY || attempt >= 5) return rc;
    }
}

extern "C" int rt_render_noalloc(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out,
                                 rt_stats *stats)
{
    if (out && !out->on_device) return fail(RT_ERR_INVALID, "rt_render_noalloc: outputs must be device memory");
    int rc = enqueue_frame(ctx, cam, opts, out, FRAME_NOALLOC);
    if (rc != RT_OK) { abandon_frame(ctx); return rc; }
    return rt_sync(ctx, stats);
}

extern "C" int rt_noalloc_scratch_bytes(size_t *bytes)
{
    if (!bytes) return fail(RT_ERR_INVALID, "rt_noalloc_scratch_bytes: null argument");
    *bytes = RT_NOALLOC_SCRATCH_BYTES;
    return RT_OK;
}

extern "C" int rt_set_memory_budget(rt_ctx *ctx, size_t bytes)
{
    if (!ctx) return fail(RT_ERR_INVALID, "rt_set_memory_budget: null argument");
    ctx->queue_budget = bytes ? bytes : RT_DEFAULT_QUEUE_BUDGET;
    return RT_OK;
}

extern "C" int rt_debug_set_option(rt_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(RT_ERR_INVALID, "rt_debug_set_option: null argument");
    const std::string n(name);
    if (n == "block_threads") ctx->dbg.block_threads = (int)value;
    else if (n == "node_cache") ctx->dbg.node_cache = value;
    else if (n == "tri_cache") ctx->dbg.tri_cache = value;
    else if (n == "rounds") ctx->dbg.rounds = (int)value;
    else if (n == "verbose") ctx->dbg.verbose = value != 0;
    else if (n == "leaf_max") ctx->dbg.leaf_max = (int)value;
    else if (n == "split_budget") ctx->dbg.split_budget = (int)value;
    else if (n == "stack_min") ctx->dbg.stack_min = (int)value;
    else if (n == "direct") ctx->dbg.direct = (int)value;
    else if (n == "lightcam_digits") { ctx->dbg.lightcam_digits = (int)value; free_lightcams(ctx); }
    else if (n == "lightcam_levels") { ctx->dbg.lightcam_levels = (int)value; free_lightcams(ctx); }
    else if (n == "lightcam_host") { ctx->dbg.lightcam_host = (int)value; free_lightcams(ctx); }
    else if (n == "fused") ctx->dbg.fused = (int)value;
    else if (n == "units") ctx->dbg.units = (int)value;
    else if (n == "local") ctx->dbg.local = (int)value;
    else if (n == "local_threads") ctx->dbg.local_threads = (value == 256 || value == 512) ? (int)value : 1024;
    else if (n == "local_min_chunks") ctx->dbg.local_min_chunks = value < 0 ? 0 : (int)(value & 0xffff);
    else if (n == "whole_chunks") ctx->dbg.whole_chunks = value < 0 ? 2 : (value > 6 ? 6 : (int)value);
    else if (n == "local_rotate") ctx->dbg.local_rotate = value < 0 ? -1 : (int)(value & 0xffff);
    else if (n == "overlap") ctx->dbg.overlap = (int)value;
    else if (n == "pair") ctx->dbg.pair = (int)value;
    else if (n == "ring_poison") ctx->dbg.ring_poison = value != 0;
    else if (n == "grid") { ctx->dbg.grid = (int)value; ctx->cg_valid = false; ctx->cg_refused = false; }          // (lights' grids: read at the next upload)
    else if (n == "grid_res") ctx->dbg.grid_res = (int)value;
    else if (n == "grid_kernels") ctx->dbg.grid_kernels = (int)value;
    else if (n == "fuse_lights") ctx->dbg.fuse_lights = (int)value;
    else if (n == "device_build_min") ctx->dbg.device_build_min = value;
    else if (n == "device_build_stack") ctx->dbg.device_build_stack = (int)value;
    else if (n == "grid_min_samples") { ctx->dbg.grid_min_samples = value; ctx->cg_refused = false; }
    else if (n == "lightcam_pool") ctx->dbg.lightcam_pool = value < 128 ? 128 : (value > LC_POOL ? LC_POOL : (int)value);
    else if (n == "queue_growth") ctx->growth = value < 1 ? 1 : (value > 32 ? 32 : (int)value);
    else return fail(RT_ERR_INVALID, "rt_debug_set_option: unknown option '" + n + "'");
    return RT_OK;
}
