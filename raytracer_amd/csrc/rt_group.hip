// Several GPUs of one node behind one call: the multi-GPU form of the render boundary.
//
// The reference splits a frame over two host threads by interleaving rows (Driver.cpp:377, 735-769).  Here the frame
// is cut into bands of `band_rows` rows dealt round-robin to the members of a group (band b -> member b % N), every
// member is an rt_ctx on its own GPU with the scene replicated, and one host thread per member renders its part into a
// compact buffer on its device.  The only exchange is the end-of-frame gather, and it has no step of its own: as soon as
// a member's part is rendered, a kernel on that member's own stream writes its rows straight into their places of the
// frame on the first member's device, through peer access over xGMI (every peer has its own link to that device, so the
// members' writes run side by side, and a member that finishes early does not wait for the others).  Only where two
// devices cannot reach each other directly does a part go through a copy on the first device.  By default no collective
// library is involved: a gather to one root over point-to-point links is N-1 independent transfers.  With the option
// "rccl" the parts travel as grouped ncclSend / ncclRecv instead (librccl is loaded when the option is first used; one
// communicator per member, made with ncclCommInitAll, so the members' devices must differ): every member sends its part to
// the first, which receives them into buffers of its own and places the rows from there.  Not run on more than one GPU
// so far: the tests drive groups of several contexts on one device, and the RCCL form as a group of one.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raytracer_amd.h"

extern "C" void rt_set_error_string(const char *msg);

namespace {
int gfail(int code, const std::string &msg)
{
    rt_set_error_string(msg.c_str());
    return code;
}
#define GHIP(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return gfail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Rows of one part, compact and ascending, into their places in the frame: band b of the part is band
// (b * n_parts + part) of the frame.  One block per row; whole words when the row allows it.
__global__ void k_place_rows(const unsigned char *src, unsigned char *dst, int rows, long long row_bytes, int band_rows, int n_parts, int part)
{
    const int r = (int)blockIdx.x;
    if (r >= rows) return;
    const int band = r / band_rows, within = r - band * band_rows;
    const long long grow = (long long)(band * n_parts + part) * band_rows + within;
    const unsigned char *s = src + (long long)r * row_bytes;
    unsigned char *d = dst + grow * row_bytes;
    if ((row_bytes & 3) == 0 && (((unsigned long long)s | (unsigned long long)d) & 3ull) == 0) {
        const unsigned int *s4 = (const unsigned int *)s;
        unsigned int *d4 = (unsigned int *)d;
        for (long long i = threadIdx.x; i < row_bytes / 4; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (long long i = threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = s[i];
    }
}

struct Buffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    int ensure(int dev, size_t want)
    {
        if (ptr && bytes >= want && device == dev) return RT_OK;
        GHIP(hipSetDevice(dev));
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; bytes = 0; device = dev;
        GHIP(hipMalloc(&ptr, want ? want : 1));
        bytes = want;
        return RT_OK;
    }
    void release()
    {
        if (ptr) { (void)hipSetDevice(device); (void)hipFree(ptr); }
        ptr = nullptr; bytes = 0;
    }
};
}  // namespace

// The few entry points of librccl the gather uses, resolved at run time (the library is large, and a process that also
// runs PyTorch has one loaded already: the same one is then used).
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::vector<void *> comm;                     // per member
    bool load(std::string *why)
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!lib) { *why = std::string("librccl not found: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) *why = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) { lib = nullptr; return false; }
        return true;
    }
};
#define RCCL_UINT8 1      // ncclUint8

struct rt_group {
    std::vector<int> devices;
    Rccl rccl;
    bool use_rccl = false;                        // option "rccl": the parts travel as ncclSend / ncclRecv
    std::vector<rt_ctx *> ctx;
    hipStream_t stream = nullptr;                 // on the first member's device: the frame's way to host memory; staged parts
    std::vector<hipStream_t> mstream;             // per member, on its device: the placement of its rows
    std::vector<char> direct;                     // per member: its device can write the first device's memory
    bool force_staged = false;                    // diagnostic: every member's part goes through the copy on the first device
    // per member and output kind (f64, u8, hit id, mask): the part on the member's device, and (only for a member without
    // direct access) its copy on the first device
    std::vector<Buffer> part[4], staged[4];
    Buffer frame[4];                              // frame-order outputs on the first device when the caller's buffers are host memory
};

static const size_t kPixelBytes[4] = {3 * sizeof(double), 3, sizeof(int32_t), 1};

extern "C" int rt_group_destroy(rt_group *g)
{
    if (!g) return RT_OK;
    for (size_t i = 0; i < g->rccl.comm.size(); ++i)
        if (g->rccl.comm[i]) { (void)hipSetDevice(g->devices[i]); (void)g->rccl.CommDestroy(g->rccl.comm[i]); }
    for (rt_ctx *c : g->ctx) (void)rt_destroy(c);
    for (int k = 0; k < 4; ++k) {
        for (Buffer &b : g->part[k]) b.release();
        for (Buffer &b : g->staged[k]) b.release();
        g->frame[k].release();
    }
    if (g->stream) { (void)hipSetDevice(g->devices[0]); (void)hipStreamDestroy(g->stream); }
    for (size_t i = 0; i < g->mstream.size(); ++i)
        if (g->mstream[i]) { (void)hipSetDevice(g->devices[i]); (void)hipStreamDestroy(g->mstream[i]); }
    delete g;
    return RT_OK;
}

extern "C" int rt_group_create(const int *device_ids, int n, rt_group **out)
{
    if (!device_ids || n < 1 || n > 64 || !out) return gfail(RT_ERR_INVALID, "rt_group_create: bad argument");
    rt_group *g = new rt_group();
    g->devices.assign(device_ids, device_ids + n);
    for (int i = 0; i < n; ++i) {
        rt_ctx *c = nullptr;
        int rc = rt_create(device_ids[i], &c);
        if (rc != RT_OK) { rt_group_destroy(g); return rc; }
        g->ctx.push_back(c);
    }
    for (int k = 0; k < 4; ++k) { g->part[k].resize(n); g->staged[k].resize(n); }
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
        rt_group_destroy(g);
        return gfail(RT_ERR_DEVICE, "rt_group_create: no stream on the first device");
    }
    // every member's device gets a stream for placing its rows, and access to the first device's memory (member -> first
    // device is the direction the rows travel); a member that cannot have it goes through a staged copy instead
    g->mstream.assign((size_t)n, nullptr);
    g->direct.assign((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&g->mstream[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
            rt_group_destroy(g);
            return gfail(RT_ERR_DEVICE, "rt_group_create: no stream on device " + std::to_string(device_ids[i]));
        }
        if (device_ids[i] == device_ids[0]) { g->direct[(size_t)i] = 1; continue; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(device_ids[0], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->direct[(size_t)i] = 1;
            (void)hipGetLastError();
        }
        // (the staged path copies with hipMemcpyPeerAsync, which needs no access of either side)
    }
    *out = g;
    return RT_OK;
}

extern "C" int rt_group_set_option(rt_group *g, const char *name, int64_t value)
{
    if (!g || !name) return gfail(RT_ERR_INVALID, "rt_group_set_option: null argument");
    if (std::string(name) == "staged") { g->force_staged = value != 0; return RT_OK; }
    if (std::string(name) == "rccl") {
        if (value == 0) { g->use_rccl = false; return RT_OK; }
        if (g->rccl.comm.empty()) {
            std::string why;
            if (!g->rccl.load(&why)) return gfail(RT_ERR_UNSUPPORTED, "rt_group_set_option: rccl: " + why);
            for (size_t i = 0; i < g->devices.size(); ++i)
                for (size_t j = 0; j < i; ++j)
                    if (g->devices[i] == g->devices[j])
                        return gfail(RT_ERR_UNSUPPORTED, "rt_group_set_option: rccl: a communicator takes a device once, and this group lists device " +
                                                             std::to_string(g->devices[i]) + " twice");
            std::vector<void *> comm(g->devices.size(), nullptr);
            const int e = g->rccl.CommInitAll(comm.data(), (int)g->devices.size(), g->devices.data());
            if (e != 0) return gfail(RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(e));
            g->rccl.comm = comm;
        }
        g->use_rccl = true;
        return RT_OK;
    }
    return gfail(RT_ERR_INVALID, std::string("rt_group_set_option: unknown option '") + name + "'");
}

extern "C" int rt_group_size(const rt_group *g, int *n)
{
    if (!g || !n) return gfail(RT_ERR_INVALID, "rt_group_size: null argument");
    *n = (int)g->ctx.size();
    return RT_OK;
}

// Runs fn(member) on one host thread per member and returns the first failure (its message becomes the caller's).
template <typename F>
static int for_each_member(rt_group *g, F fn)
{
    const size_t n = g->ctx.size();
    std::vector<int> rc(n, RT_OK);
    std::vector<std::string> msg(n);
    auto body = [&](size_t i) {
        rc[i] = fn(i);
        if (rc[i] != RT_OK) msg[i] = rt_last_error();
    };
    if (n == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; ++i) th.emplace_back(body, i);
        for (std::thread &t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i)
        if (rc[i] != RT_OK) return gfail(rc[i], "member " + std::to_string(i) + " (device " + std::to_string(g->devices[i]) + "): " + msg[i]);
    return RT_OK;
}

extern "C" int rt_group_upload_scene(rt_group *g, const rt_scene_desc *scene)
{
    if (!g || !scene) return gfail(RT_ERR_INVALID, "rt_group_upload_scene: null argument");
    return for_each_member(g, [&](size_t i) { return rt_upload_scene(g->ctx[i], scene); });
}

extern "C" int rt_group_render(rt_group *g, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out, rt_stats *stats)
{
    if (!g || !cam || !opts || !out) return gfail(RT_ERR_INVALID, "rt_group_render: null argument");
    if (cam->resx <= 0 || cam->resy <= 0) return gfail(RT_ERR_INVALID, "rt_group_render: empty frame");
    const int n = (int)g->ctx.size();
    const int band = opts->band_rows > 0 ? opts->band_rows : 16;
    void *dst[4] = {out->rgb_f64, out->rgb_u8, out->hit_id, out->lightcam_mask};
    std::vector<int32_t> rows((size_t)n, 0);
    std::vector<rt_render_opts> mo((size_t)n, *opts);
    for (int i = 0; i < n; ++i) {
        mo[i].band_rows = band; mo[i].n_parts = n; mo[i].part = i;
        int rc = rt_part_rows(cam, &mo[i], &rows[i]);
        if (rc != RT_OK) return rc;
        for (int k = 0; k < 4; ++k)
            if (dst[k] && (rc = g->part[k][i].ensure(g->devices[i], (size_t)rows[i] * cam->resx * kPixelBytes[k])) != RT_OK) return rc;
    }
    // where the rows go: the caller's device buffers, or frame buffers on the first device on the way to host memory
    const int d0 = g->devices[0];
    unsigned char *frame[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t row_bytes[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) {
        if (!dst[k]) continue;
        row_bytes[k] = (size_t)cam->resx * kPixelBytes[k];
        frame[k] = static_cast<unsigned char *>(dst[k]);
        if (!out->on_device) {
            int rc = g->frame[k].ensure(d0, (size_t)cam->resy * row_bytes[k]);
            if (rc != RT_OK) return rc;
            frame[k] = static_cast<unsigned char *>(g->frame[k].ptr);
        }
        for (int i = 0; i < n; ++i)
            if ((!g->direct[(size_t)i] || g->force_staged || g->use_rccl) && rows[i] > 0) {
                int rc = g->staged[k][i].ensure(d0, (size_t)rows[i] * row_bytes[k]);
                if (rc != RT_OK) return rc;
            }
    }
    // every member renders its part on a host thread of its own and, when it is done, puts its rows into the frame
    std::vector<rt_stats> ms((size_t)n);
    int rc = for_each_member(g, [&](size_t i) {
        rt_frame_out po = {};
        po.rgb_f64 = dst[0] ? g->part[0][i].ptr : nullptr;
        po.rgb_u8 = dst[1] ? g->part[1][i].ptr : nullptr;
        po.hit_id = dst[2] ? g->part[2][i].ptr : nullptr;
        po.lightcam_mask = dst[3] ? g->part[3][i].ptr : nullptr;
        po.on_device = 1;
        int mrc = rt_render(g->ctx[i], cam, &mo[i], &po, &ms[i]);
        if (mrc != RT_OK) return mrc;
        if (g->use_rccl) {
            // Every member sends its part to the first; the first receives all of them (its own too) into its buffers and places
            // the rows from there.  A member's calls form one group, so none of them waits for another on the host; on the
            // devices the sends meet their receives when the first member's render is done.
            Rccl &R = g->rccl;
            GHIP(hipSetDevice(g->devices[i]));
            hipStream_t st = g->mstream[i];
            int e = R.GroupStart();
            for (int k = 0; k < 4 && e == 0; ++k) {
                if (!dst[k]) continue;
                if (rows[i] > 0) e = R.Send(g->part[k][i].ptr, (size_t)rows[i] * row_bytes[k], RCCL_UINT8, 0, R.comm[i], st);
                if (i == 0)
                    for (int j = 0; j < n && e == 0; ++j)
                        if (rows[j] > 0) e = R.Recv(g->staged[k][j].ptr, (size_t)rows[j] * row_bytes[k], RCCL_UINT8, j, R.comm[0], st);
            }
            const int e2 = R.GroupEnd();
            if (e != 0 || e2 != 0) return gfail(RT_ERR_DEVICE, std::string("rccl gather: ") + R.GetErrorString(e != 0 ? e : e2));
            if (i == 0)
                for (int k = 0; k < 4; ++k)
                    for (int j = 0; j < n && dst[k]; ++j)
                        if (rows[j] > 0)
                            hipLaunchKernelGGL(k_place_rows, dim3((unsigned)rows[j]), dim3(256), 0, st, static_cast<const unsigned char *>(g->staged[k][j].ptr),
                                               frame[k], rows[j], (long long)row_bytes[k], band, n, j);
            GHIP(hipGetLastError());
            GHIP(hipStreamSynchronize(st));
            return (int)RT_OK;
        }
        if (rows[i] == 0) return mrc;
        const bool direct = g->direct[i] != 0 && !g->force_staged;
        // direct: the member's own device writes the rows to the first device's memory; else the part is copied to the first
        // device and placed there
        GHIP(hipSetDevice(direct ? g->devices[i] : d0));
        hipStream_t st = direct ? g->mstream[i] : g->stream;
        for (int k = 0; k < 4; ++k) {
            if (!dst[k]) continue;
            const unsigned char *src = static_cast<const unsigned char *>(g->part[k][i].ptr);
            if (!direct) {
                GHIP(hipMemcpyPeerAsync(g->staged[k][i].ptr, d0, g->part[k][i].ptr, g->devices[i], (size_t)rows[i] * row_bytes[k], st));
                src = static_cast<const unsigned char *>(g->staged[k][i].ptr);
            }
            hipLaunchKernelGGL(k_place_rows, dim3((unsigned)rows[i]), dim3(256), 0, st, src, frame[k], rows[i], (long long)row_bytes[k], band, n, (int)i);
        }
        GHIP(hipGetLastError());
        GHIP(hipStreamSynchronize(st));
        return (int)RT_OK;
    });
    if (rc != RT_OK) return rc;
    if (!out->on_device) {
        GHIP(hipSetDevice(d0));
        for (int k = 0; k < 4; ++k)
            if (dst[k]) GHIP(hipMemcpyAsync(dst[k], frame[k], (size_t)cam->resy * row_bytes[k], hipMemcpyDeviceToHost, g->stream));
        GHIP(hipStreamSynchronize(g->stream));
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int i = 0; i < n; ++i) {
            stats->closest_rays += ms[i].closest_rays;
            stats->shadow_rays += ms[i].shadow_rays;
            for (int q = 0; q < 6; ++q) { stats->rays_per_generation[q] += ms[i].rays_per_generation[q]; stats->hits_per_generation[q] += ms[i].hits_per_generation[q]; }
            if (ms[i].gpu_ms > stats->gpu_ms) stats->gpu_ms = ms[i].gpu_ms;       // the members run side by side
            // (with RT_FLAG_TIMING: kernel times and launches summed over the members, which run side by side -- their sum is
            // device time, not a share of gpu_ms)
            stats->closest_ms += ms[i].closest_ms; stats->shadow_ms += ms[i].shadow_ms; stats->pair_ms += ms[i].pair_ms;
            stats->primary_ms += ms[i].primary_ms; stats->other_ms += ms[i].other_ms;
            stats->closest_launches += ms[i].closest_launches; stats->shadow_launches += ms[i].shadow_launches;
            stats->pair_launches += ms[i].pair_launches; stats->primary_launches += ms[i].primary_launches;
            stats->batches += ms[i].batches;
            stats->queue_overflow |= ms[i].queue_overflow;
        }
        stats->lightcam_rays = ms[0].lightcam_rays;      // every member builds the same light cameras
    }
    return RT_OK;
}
