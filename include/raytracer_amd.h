/*
 * raytracer_amd.h -- C ABI of the MI355X render path.
 *
 * The reference (MasonSchuckman/Raytracer) has no plugin or FFI interface; the
 * boundary this library replaces is the internal C++ call
 *     renderMultithreaded(Camera)                    Driver.cpp:355-390
 * i.e. the two-thread loop
 *     pixels[(i*resx+j)*3+c] = camera.castRay(i,j,root,solids,lights)[c] * 255
 *                                                    Driver.cpp:735-769, Camera.h:112
 * with its inputs (the Camera fields of Camera.h:19-46, the globals `root`,
 * `solids`, `lights`, `resx`, `resy` of Driver.cpp:38-46) passed as plain arrays.
 *
 * Two groups of entry points:
 *   rth_*  host front end: .nff dialect -> triangulated primitives -> octree ->
 *          flattened arrays (replaces parseInput Driver.cpp:426-691 and
 *          setupOctree Driver.cpp:224-250 for callers that do not already hold
 *          the scene in memory).
 *   rt_*   device path: upload, render, read back.
 *
 * All structs are plain C, all sizes in elements unless named *_bytes.  Every
 * function returns RT_OK (0) or a negative error code; rt_last_error() gives the
 * message of the calling thread's last failure.  A context is bound to one GPU
 * and is used from one host thread at a time.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument or inconsistent scene arrays */
    RT_ERR_DEVICE = -2,       /* HIP runtime error, or no MI355X-class device */
    RT_ERR_IO = -3,           /* file could not be opened / parsed */
    RT_ERR_CAPACITY = -4,     /* a ray queue overflowed (rt_sync), or a request exceeds a fixed capacity */
    RT_ERR_UNSUPPORTED = -5
};

/* Material of the last `f` line (ObjProps, Solid.h:27-36). */
typedef struct rt_material {
    double r, g, b, Kd, Ks, Shine, T, ior;
} rt_material;                                   /* 64 bytes */

/* Light (Light.h:6-20); intensity = 1/sqrt(#lights) (Driver.cpp:685-688). */
typedef struct rt_light {
    double pos[3];
    double color[3];
    double intensity;
    double _pad;
} rt_light;                                      /* 64 bytes */

/* One octree node (Octree<Solid*>, Octree.h:733-907) in depth-first pre-order,
 * children in the reference's order 0..7.  `skip` is the pre-order index of the
 * first node that is NOT in this node's subtree, so a traversal needs no stack:
 * descend = index+1, prune = skip.  The node's resident primitives are two slot
 * ranges, one into the triangle records and one into the sphere records. */
typedef struct rt_node {
    double center[3];          /* bounding-sphere centre = cube centre (Octree.h:791-793) */
    double radius;             /* size*sqrt(3)/2 (Octree.h:796) */
    int32_t skip;
    int32_t tri_begin, tri_count;
    int32_t sph_begin, sph_count;
    int32_t depth;
    int32_t _pad[2];
} rt_node;                                       /* 64 bytes, one cache line on the host */

/* Triangle record: the per-ray operands of Triangle::intersect (Triangle.h:91-158):
 * v0 and the precomputed edges ab = v0-v1, ac = v0-v2 (Triangle.h:40-46). */
typedef struct rt_tri_rec {
    double v0[3], ab[3], ac[3];
    int32_t prim;              /* primitive id = index in parse/triangulation order */
    int32_t _pad0;
    double _pad1[2];
} rt_tri_rec;                                    /* 96 bytes */

/* Sphere record: operands of Sphere::intersect (Sphere.h:35-77). */
typedef struct rt_sph_rec {
    double center[3];
    double radius;
} rt_sph_rec;                                    /* 32 bytes */

#define RT_PRIM_SPHERE 1u          /* else triangle */
#define RT_PRIM_HAS_VNORMALS 2u    /* triangle came from a `pp` patch (Triangle.h:13) */
#define RT_SPH_TRANSPARENT 0x80000000u   /* flag bit in sph_prim[]: material T > 0 */

/* Flattened scene.  Arrays are caller-owned host memory, copied by rt_upload_scene. */
typedef struct rt_scene_desc {
    const rt_node *nodes;        int32_t n_nodes;
    const rt_tri_rec *tri;       int32_t n_tri_slots;
    const rt_sph_rec *sph;       int32_t n_sph_slots;
    const uint32_t *sph_prim;    /* per sphere slot: primitive id | RT_SPH_TRANSPARENT */
    int32_t n_prims;
    const uint32_t *prim_flags;  /* per primitive: RT_PRIM_* */
    const int32_t *prim_material;/* per primitive: index into materials */
    const double *prim_normals;  /* per primitive 12 doubles: flat normal, then vertex normals
                                    n0,n1,n2 (zero unless RT_PRIM_HAS_VNORMALS); spheres: zero */
    const rt_material *materials; int32_t n_materials;
    const rt_light *lights;      int32_t n_lights;
    double background[3];
    int32_t lightcam_samples;    /* `Z` directive, default 150 (Driver.cpp:41,505-508) */
    int32_t _pad;
} rt_scene_desc;

/* Camera after computeCoordinateSystem (Camera.h:64-97). */
typedef struct rt_camera {
    double u[3], v[3], w[3], e[3];
    double d, l, t, pix_width;
    int32_t resx, resy;
} rt_camera;

#define RT_FLAG_PHONG_NORMALS 1u   /* `-p`: interpolate patch normals (Triangle.h:144-146) */
#define RT_FLAG_NO_CULL       2u   /* visit exactly the nodes the reference visits */
#define RT_FLAG_KEEP_ZERO_WEIGHT 4u/* trace reflection rays whose weight is exactly 0, as the
                                      reference does (Camera.h:305-306); they change no pixel */
#define RT_FLAG_TIMING        8u   /* record per-kernel-class HIP-event times into rt_stats */
#define RT_FLAG_NO_LIGHTCAM  16u   /* leave the light-camera term out: a point lit only through transparent
                                      spheres counts as shadowed (it is still marked in lightcam_mask) */

typedef struct rt_render_opts {
    int32_t num_samples;       /* N of `-s N`; samples per pixel = N*N (Camera.h:123-125) */
    double aperture;           /* `-a A`; depth of field when > 0 (Camera.h:114) */
    uint32_t flags;            /* RT_FLAG_* */
    uint32_t seed;             /* jitter / lens sampling seed (used when N>1 or aperture>0) */
    /* Row partition for multi-GPU: rows are cut into bands of `band_rows` rows, band b belongs
     * to part (b % n_parts).  This call renders the rows of `part`.  n_parts=1 renders all. */
    int32_t band_rows, n_parts, part;
    int32_t _pad;
} rt_render_opts;

/* Output buffers hold only the rows of this part, in ascending row order
 * (rt_part_rows tells how many).  Any pointer may be NULL. */
typedef struct rt_frame_out {
    void *rgb_f64;             /* rows*resx*3 doubles: clamp(colour,0,1)*255 (Driver.cpp:758-760) */
    void *rgb_u8;              /* rows*resx*3 bytes: (int) truncation (Driver.cpp:722) */
    void *hit_id;              /* rows*resx int32: primitive hit by the first sample's primary ray, -1 = none */
    void *lightcam_mask;       /* rows*resx bytes: 1 where a shading point was shadowed only by transparent
                                  spheres (the light-camera branch, Camera.h:406-451) */
    int32_t on_device;         /* 1: pointers are device memory of the context's GPU; 0: host memory */
    int32_t _pad;
} rt_frame_out;

typedef struct rt_stats {
    uint64_t closest_rays;     /* calls equivalent to Camera::castRay(ray), Camera.h:318 */
    uint64_t shadow_rays;      /* calls equivalent to checkOccluded3 from Camera.h:404 */
    uint64_t lightcam_rays;    /* light-camera rays (LightCam.h:491) traced to build the cameras in use;
                                  reported separately, not part of the two counts above */
    uint64_t rays_per_generation[6];
    uint64_t hits_per_generation[6];
    double gpu_ms;             /* whole frame, HIP events on the context's stream */
    double closest_ms, shadow_ms, other_ms;   /* only with RT_FLAG_TIMING */
    uint64_t closest_launches, shadow_launches;
    int32_t batches;
    int32_t queue_overflow;
    double primary_ms;         /* only with RT_FLAG_TIMING: the share of closest_ms spent in generation 0 */
    uint64_t primary_launches; /* (ray generation fused with the closest-hit query; one launch per batch) */
    double pair_ms;            /* only with RT_FLAG_TIMING: launches that run the shadow pass of one generation together with */
    uint64_t pair_launches;    /* the closest-hit pass of the next (counted here, not in closest_ms / shadow_ms) */
} rt_stats;

typedef struct rt_ctx rt_ctx;
typedef struct rth_scene rth_scene;

/* ---- device path ---- */
const char *rt_last_error(void);
/* Identity of the sources this library was built from (16 hex digits); measurements kept beside the code name it. */
const char *rt_build_id(void);
int rt_device_count(int *count);
int rt_create(int device_id, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);
int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *scene);
int rt_part_rows(const rt_camera *cam, const rt_render_opts *opts, int32_t *rows);
/* Renders one frame (replaces renderMultithreaded, Driver.cpp:355-390, minus writeImage).
 * Synchronous: returns after the outputs are complete. */
int rt_render(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out, rt_stats *stats);
/* Same, but only enqueues the work on the context's stream; outputs must be device memory.
 * rt_sync waits and fills `stats`. */
int rt_render_async(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out);
int rt_sync(rt_ctx *ctx, rt_stats *stats);
/* Primary batch size (primary samples per pipeline pass); 0 restores the default. */
int rt_set_batch(rt_ctx *ctx, int64_t primaries);
/* Device memory the ray/hit queues of this context may take (bytes; 0 restores the default).  A frame whose
 * batch would need more runs in smaller batches; results do not depend on it. */
int rt_set_memory_budget(rt_ctx *ctx, size_t bytes);

/* Queue depth.  The queues of a batch hold, per generation of the ray tree, a fixed multiple of the batch's primary
 * samples (2 at first) instead of the 32-fold worst case of a scene in which every hit both reflects and refracts.
 * If a frame outgrows them, rt_sync returns RT_ERR_CAPACITY (the outputs are incomplete) and the context doubles
 * the multiple for its next frame; rt_render does that re-rendering by itself and only returns a complete frame. */

/* rt_render for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc,
 * processes whose memory is all handed out).  It has no counterpart in the reference.  It is a blocking call on the
 * context's own streams (it synchronises and reads a few words back), so it cannot be recorded into a HIP graph
 * by stream capture; what it guarantees is that no device or host-pinned memory is allocated or freed.  Same arguments and results as
 * rt_render with device output buffers (out->on_device must be 1), but it calls no hipMalloc/hipFree and creates
 * no events: accumulation frame, counters and queues live in a static device array of rt_noalloc_scratch_bytes()
 * (128 MiB) that every GPU holds once per process.  Limits, each an error and never a fallback:
 *   - RT_ERR_CAPACITY when the part's pixels * 24 bytes plus the smallest batch's queues exceed the array
 *     (about 5 M pixels; render larger frames in parts, see rt_render_opts);
 *   - RT_ERR_CAPACITY from a queue overflow is returned as it is (no automatic second attempt);
 *   - RT_ERR_UNSUPPORTED with RT_FLAG_TIMING, and when the scene's light cameras are not built yet (building
 *     allocates): render one frame with rt_render first or pass RT_FLAG_NO_LIGHTCAM.
 * All contexts of one GPU share the array and it keeps the last call's contents: calls that overlap in time on one
 * GPU race; calls one after the other, from any context, are fine. */
int rt_render_noalloc(rt_ctx *ctx, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out,
                      rt_stats *stats);
int rt_noalloc_scratch_bytes(size_t *bytes);

/* Diagnostic: experiment switches of one context, none of which changes a result.  Names: "block_threads" (256 or
 * 1024, 0 = per-frame choice), "node_cache", "tri_cache" (caps on what is copied into LDS, -1 = none), "rounds",
 * "verbose", "queue_growth" (1..32); "pair", "overlap", "fused", "local" (1 forces, 0 forbids, -1 leaves to the frame's
 * size: paired passes, passes on two streams, the lane-per-sample kernel, the self-contained blocks), "local_threads"
 * (256, 512, 1024), "local_rotate" (places a block's share of the primary chunks moves on per round, -1 = by image
 * width), "local_min_chunks" (primary chunks a self-contained block gets at least, 0 = four), "whole_chunks" (how many
 * generations of closest-hit passes push whole chunks of records, 0..6, -1 = default), "units" (0: the primary pass keeps fixed shares per block),
 * "direct" (0: the primary pass takes its places in the queues from counters instead of from the chunk's number),
 * "fuse_lights" (0: the primary pass leaves the lights at its hits to a shadow pass of its own), "grid", "grid_res",
 * "grid_min_samples", "grid_kernels" (the view grids: which are made, how fine, from which frame size, and whether the
 * kernels that only read them are used), "stack_min", "ring_poison", "lightcam_levels", "lightcam_digits",
 * "lightcam_host" (1: light-camera trees
 * built on the host), "lightcam_pool" (128..1536: places used of the pool of pending cells that a wave's 64 density
 * queries share; a full pool slows the pass down, it loses nothing); and for the next rt_upload_scene "leaf_max"
 * (1..4), "split_budget" (percent), "device_build_min" and "device_build_stack" (primitives from which a hierarchy is
 * built on the device, and the stack need beyond which that result is discarded for the host builder's). */
int rt_debug_set_option(rt_ctx *ctx, const char *name, int64_t value);

/* ---- several GPUs of one node ----
 * The reference splits the frame over two host threads, rows interleaved (Driver.cpp:377, 735-769).  A group is that
 * split over GPUs: one context per listed device (a device may be listed more than once), the scene replicated, the
 * frame cut into bands of opts->band_rows rows (16 when 0) dealt round-robin to the members, one host thread per
 * member; as soon as a member's part is rendered, its own device writes the rows into their places of the frame on the
 * first device through peer access (where two devices cannot reach each other directly: a copy to the first device,
 * placed there).  rt_group_render takes rt_render's arguments (opts->n_parts and opts->part are ignored) and fills
 * frame-order outputs: device memory of the first listed device when out->on_device, else host memory.  stats: ray
 * counts summed, gpu_ms the slowest member's, kernel times and launches (RT_FLAG_TIMING) summed over the members.
 * rt_group_set_option: "staged" 1 sends every member's part through the copy on the first device (diagnostic);
 * "rccl" 1 sends the parts as grouped ncclSend / ncclRecv to the first member instead (librccl is loaded at that moment, one
 * communicator per member: the listed devices must differ; RT_ERR_UNSUPPORTED when they do not or the library is missing). */
typedef struct rt_group rt_group;
int rt_group_create(const int *device_ids, int n, rt_group **out);
int rt_group_destroy(rt_group *g);
int rt_group_size(const rt_group *g, int *n);
int rt_group_upload_scene(rt_group *g, const rt_scene_desc *scene);
int rt_group_render(rt_group *g, const rt_camera *cam, const rt_render_opts *opts, const rt_frame_out *out, rt_stats *stats);
int rt_group_set_option(rt_group *g, const char *name, int64_t value);

/* Diagnostic: runs the kernels' own primitive tests on n (primitive, ray) pairs, one per GPU thread, so that
 * each can be compared with the reference's function in isolation.
 *   kind 0: Triangle::intersect (Triangle.h:91-158), the plain sequence of three divisions;
 *   kind 1: the same test in the form the traversal uses (rejections decided before dividing);
 *   kind 2: Sphere::intersect (Sphere.h:35-77), and the hit record's normal (Solid.h:119);
 *   kind 3: the node bounding-sphere test (Sphere.h:155-204), `maxT` being the sentinel it returns on a miss;
 *   kind 4: the triangle test as shadow queries use it (hit or not, no quotient formed): t is 0 for a hit, maxT for a miss;
 *   kind 5: the triangle test as nearest-hit queries use it, with `maxT` as the distance of the nearest hit so far
 *           (the test's own upper limit is 1e10): a hit beyond `maxT` may be reported as 1e10;
 *   kind 6: the power of the specular term, pow(max(rdir.v, 0), Shine) (Solid.h:166-169), as the kernels form it: t is the power.
 *   kind 7: a vector's normalisation, v / sqrt(v.v) (Eigen's normalized(), e.g. Camera.h:196, Solid.h:119-122), as the kernels
 *           form it: the unit vector comes back in the three doubles after t.
 * geom: 9 doubles per item (v0, v1, v2) for kinds 0/1/4/5, 4 doubles (centre, radius) for kinds 2/3, 4 doubles (base, exponent,
 * two unused) for kind 6, 4 doubles (x, y, z, unused) for kind 7.
 * rays: 6 doubles per item (origin, direction).  out: 4 doubles per item: t (or maxT), then the unit normal
 * for a sphere hit, else zeros.  All pointers are host memory. */
int rt_debug_intersect(rt_ctx *ctx, int32_t kind, int32_t n, const double *geom, const double *rays, double minT,
                       double maxT, double *out);

/* ---- host front end ---- */
/* Parses an .nff file (dialect of Driver.cpp:426-691), ear-clips polygons and patches
 * (Polygon.h:107-205, PolygonalPatch.h:78-157), builds and flattens the octree. */
int rth_scene_load(const char *nff_path, rth_scene **out);
int rth_scene_free(rth_scene *s);
int rth_scene_desc(const rth_scene *s, rt_scene_desc *out);           /* pointers stay owned by `s` */
/* Camera for the scene's viewpoint; resx/resy <= 0 take the file's `resolution`. */
int rth_scene_camera(const rth_scene *s, int32_t resx, int32_t resy, rt_camera *out);
int rth_scene_resolution(const rth_scene *s, int32_t *resx, int32_t *resy);
/* Triangle vertices (9 doubles) or sphere centre+radius (4 doubles) of primitive `prim`. */
int rth_scene_prim_geometry(const rth_scene *s, int32_t prim, double *out9);
/* writeImage (Driver.cpp:712-733): binary P6 from value*255 doubles, (int) truncation. */
int rth_write_ppm_f64(const char *path, const double *rgb_f64, int32_t resx, int32_t resy);
int rth_write_ppm_u8(const char *path, const uint8_t *rgb_u8, int32_t resx, int32_t resy);
/* Argument scanner with the reference's behaviour (Driver.cpp:167-216, 771-783). */
int rth_parse_args(int argc, const char *const *argv, int32_t *num_samples, double *aperture, int32_t *phong,
                   char *input_path, size_t input_cap, char *output_path, size_t output_cap);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
