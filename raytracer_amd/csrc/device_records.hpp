// Records and parameter blocks shared by the kernels of the render path: what lives in HBM
// (ray and hit queues, the scene tables, the light-camera trees) and what a launch is handed.
#pragma once
#include <cstdint>

#include "../../include/raytracer_amd.h"
#include "accel_build.hpp"
#include "lightcam_build.hpp"
#include "viewgrid.hpp"

struct alignas(16) RayRec {      // 80 bytes
    double o[3];
    double d[3];
    double A;                    // product of T along the chain of refraction calls
    double coef;                 // coefOfReflection carried by the call (Camera.h:241,305)
    uint32_t pixel;              // local pixel index
    uint32_t meta;               // bits 0..7: remaining loop iterations, bit 8: first ray of a call
    uint32_t _pad[2];
};

struct alignas(16) HitRec {      // 128 bytes; a hit on a flat triangle writes the first 64 only (whole half lines: a store of 80 bytes
                                 // of a line was measured slower than one of all 128)
    double P[3];                 // impactPos
    double v[3];                 // unit vector from the hit point to the ray origin (Solid.h:156)
    double weight;               // factor applied to the local colour
    uint32_t pixel;              // bit 31 (HIT_FLAT): a triangle shaded with its own normal -- n and nn are the primitive's entry of
                                 // prim_n23 and its material comes from prim_material; nothing beyond this word is stored
    int32_t prim;                // the solid that was hit (never occludes itself)
    int32_t light, cam;          // only in the deferred light-camera queue: light index, camera index
    int32_t mat;                 // material index of `prim`
    int32_t _pad;
    double n[3];                 // hr.normal normalised a second time, as computeLocalColor does (Camera.h:382)
    double nn[3];                // ... and a third time, as computeSpecularComp does (Solid.h:153)
};
#define HIT_FLAT 0x80000000u
#define HIT_NONE 0x7FFFFFFFu     // pixel word of a hit record that only fills a chunk up
#define RAY_NONE 0xFFFFFFFFu     // pixel word of such a ray record
static_assert(sizeof(HitRec) == 128, "HitRec must be 128 bytes");

struct DevScene {
    const rt_node *nodes;
    const rt_tri_rec *tri;
    const rt_sph_rec *sph;
    const uint32_t *sph_prim;
    const uint32_t *prim_flags;
    const int32_t *prim_material;
    const double *prim_normals;
    const double *prim_units;             // per primitive 9 doubles: the flat normal normalised once, twice and three times more
                                          // (Solid.h:119, Camera.h:382, Solid.h:153), constants of a flat triangle; zeros for spheres
    const double *prim_n23;               // per primitive 6 doubles: the last two of those (what the shading of a flat triangle reads)
    const rt_material *materials;
    const rt_light *lights;
    int32_t n_nodes, n_lights;
    double bg[3];
    // derived hierarchies (accel_build.hpp); leaf records share the layout of the octree slots
    const accel::Bvh4Node *tri_nodes;
    const rt_tri_rec *btri;
    const accel::Bvh4Node *sph_nodes;
    const rt_sph_rec *bsph;
    const uint32_t *bsph_prim;
    const int32_t *bsph_key;
    int32_t n_tri_nodes, n_sph_nodes;
    int32_t tri_cached, sph_cached;       // the first so many nodes of each hierarchy are copied into LDS by every block
    int32_t stack_shift;                  // log2 of the block size: entry k of a thread's traversal stack is k << stack_shift words on
    int32_t lds_cache_word;               // first LDS word of the node cache (after the traversal stacks)
    int32_t btri_cached, lds_tri_word;    // the first so many leaf triangle records (80 bytes each: what a test reads) are in LDS too, from this word on
    int32_t mat_cached, light_cached;     // the first so many materials / lights are copied into LDS as well
    int32_t lds_table_word;               // first LDS word of those tables (materials, then lights; 16 words each)
    int32_t lds_cam_word;                 // first LDS word of the camera's 16 doubles (u, v, w, e, d, l, t, pix_width)
    int32_t lds_ticket_word;              // LDS word from which the waves of a block draw their next chunk
    float bvh_bound;                      // largest |coordinate| of any hierarchy box
    int32_t shade_bounded;                // every material and light coefficient is finite and below 1e100 in magnitude: a Phong
                                          //   factor of zero then makes the whole term zero (no infinity times zero)
    // grids of what each light can see (viewgrid.hpp, grid_query.hpp): light l's bins are lg_bins[l * 6 * lg_res^2 ...], a bin's
    // list lies in lg_entries; null: shadow queries walk the hierarchies
    const vg::Bin *lg_bins;
    const vg::Entry *lg_entries;
    int32_t lg_res;
    // light cameras (LightCam.h): one per (transparent sphere, light) pair, index rank*n_lights+light
    const int32_t *prim_lc_rank;          // per primitive: rank among transparent spheres, or -1
    const rt_sph_rec *tsph;               // the transparent spheres, by rank, and their primitive ids: the only
    const int32_t *tsph_prim;             //   primitives that make a shadow query depend on the visiting order
    int32_t n_tsph;
    const struct DevLightCam *lc_table;   // one entry per pair (n_nodes == 0: that camera is not built yet); null when the term is off
    uint32_t *lc_needed;                  // per pair: set by the shading passes when a point needs that camera
};

struct DevLightCam {
    const lightcam::Node *nodes;          // node and point arrays are shared by the cameras that were built together:
    const lightcam::Point *points;        // `root` is this camera's root in `nodes`, node ranges index `points`
    int32_t n_nodes, n_points;            // (n_nodes: of the shared array; 0 = camera not built)
    int32_t root, _pad;
    double spw;                           // sub-pixel width of the light camera (LightCam.h:223)
    double light_pos[3], sphere_pos[3];
};

// Largest block of the traversal kernels.  Large frames run one such block per compute unit (16 waves, four
// per SIMD): the LDS copies of the top of the hierarchies, of the leaf records and of the tables then exist
// once per compute unit instead of once per 256 threads, which leaves room for four times as much of the
// scene.  Small frames, which live on latency and on many independent blocks, run 256-thread blocks.
#define RT_BLOCK 1024
// Blocks of the kernels that read the view grids only (no hierarchy walk compiled in, no traversal stacks in LDS).  (Ten waves
// per block, two blocks to a compute unit, five waves per SIMD: measured slower -- the kernels do not fit 96 vector registers
// without spilling, 8K frame 61.3 ms against 46.9 with sixteen waves of 128 registers; 512-thread blocks 48.1.)
#ifndef RT_GRID_BLOCK
#define RT_GRID_BLOCK 1024
#endif
#ifndef RT_GRID_WAVES
#define RT_GRID_WAVES 4
#endif

// Division of 32-bit unsigned numbers by a divisor fixed per launch, as a multiplication (Granlund and
// Montgomery's round-up method): for d >= 2 with l = ceil(log2 d), magic = floor(2^32 (2^l - d) / d) + 1 and
// q = (t + ((n - t) >> 1)) >> (l - 1), t = high word of magic * n, is floor(n / d) for every 32-bit n.
struct FastDiv {
    uint32_t d, magic, shift;
};
inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f{d, 0u, 0u};
    if (d < 2) return f;
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < d) ++l;
    f.magic = static_cast<uint32_t>((((1ull << l) - d) << 32) / d + 1);
    f.shift = l - 1;
    return f;
}

// The parameters of a frame, passed to the kernels by value.  (Passed as a pointer to a copy in device memory instead,
// so that fields are fetched where they are used rather than all at kernel entry: the scalar-register spills of the
// traversal kernels fall from 50-100 to under 10, but without `restrict` every field is fetched again inside the walks
// -- the 8K frame 70 -> 84 ms -- and with it the spills and the time are what they are by value, plus a copy per frame.)
struct FrameParams {
    DevScene S;
    rt_camera cam;
    int32_t N;                   // samples per axis
    double aperture;
    double spw;                  // sub-pixel width cam.pix_width / N (Camera.h:158), the same for every sample: divided once, on the host
    uint32_t flags, seed;
    int32_t band_rows, n_parts, part, local_rows;
    int32_t blocks_x;            // 8x8-pixel blocks per row of blocks (primary sample ordering)
    FastDiv div_spp, div_blocks_x, div_n, div_band;   // by N*N, blocks_x, N, band_rows
    FastDiv div_lights;          // by the number of lights (at least 1)
    int32_t cull_closest, cull_shadow;
    double *accum;               // local_rows*resx*3
    int32_t *hit_id;             // local_rows*resx or null
    uint8_t *lc_mask;            // local_rows*resx or null
    // grid of what the eye can see, in tiles of 8 x 8 pixels of the whole frame (viewgrid.hpp); null: primary rays walk the hierarchies
    const vg::Bin *cg_bins;
    const vg::Entry *cg_entries;
    int32_t cg_tiles_x;
};

// Queues are cut into N_SHARDS independent shards (a block pushes into shard blockIdx % N_SHARDS),
// so that no single counter word serialises the pushes of the whole chip; a consumer walks the
// shards' 64-ray chunks as one list.
#define N_SHARDS 64
#define CNT_RAYS(g) ((g) * N_SHARDS)             // g = 0..6
#define CNT_HITS(g) ((7 + (g)) * N_SHARDS)       // g = 0..5
#define CNT_DEFER(g) ((13 + (g)) * N_SHARDS)     // g = 0..5: deferred light-camera evaluations
#define CNT_PAD_HITS(g) ((19 + (g)) * N_SHARDS)  // g = 0..5: of the hit records of generation g, how many are fillers (QueueSet::whole_chunks)
#define CNT_PAD_RAYS(g) ((25 + (g)) * N_SHARDS)  // g = 1..6: ... and of the ray records of generation g
#define CNT_OVERFLOW (32 * N_SHARDS)
#define CNT_UNITS (32 * N_SHARDS + 4)            // units of work handed out by a primary pass that shares its chunks out on the device
#define CNT_LCWORK (32 * N_SHARDS + 5)           // 64-query chunks handed out by the light-camera density pass
#define CNT_ARRIVED (32 * N_SHARDS + 6)          // blocks of the diagnostic launch that fills the LDS ring words (k_poison_ring)
#define CNT_FUSED_HITS (32 * N_SHARDS + 7)       // primary hits whose lights were dealt with in the primary pass itself (QueueSet::fused_lights)
#define CNT_WORDS (32 * N_SHARDS + 16)

struct QueueSet {
    RayRec *rays_in;
    RayRec *rays_out;
    HitRec *hits;
    HitRec *deferred;            // light-camera evaluations postponed by the shadow pass (same shard size as hits)
    uint32_t *counts;
    uint32_t in_shard_cap, out_shard_cap, hit_shard_cap;   // records per shard
    uint32_t defer_shard_cap;    // ... of the deferred queue: hit_shard_cap times the number of lights
    uint32_t hit_stride;         // bytes from one hit record to the next: 128, or 64 when every hit of the scene is a flat triangle's
    int32_t generation;
    uint32_t unit_chunks;        // PRIMARY only: chunks per unit of work drawn from CNT_UNITS (a power of two), or 0: every block keeps to its own chunks
    uint32_t unit_mult;          // ... the k-th unit handed out is number (k * unit_mult) mod (number of units); 0: number k
                                 //     (bit 31, diagnostic: unit numbers are handed over late, see k_poison_ring)
    int64_t chunk_begin, chunk_end;  // PRIMARY only: 64-sample chunks of the primary ordering; a batch begins at a
                                     // block of 8x8 pixels (chunk_begin is a multiple of N*N)
    int64_t block_row0;              // PRIMARY only: row and column, in 8x8-pixel blocks, of the batch's first block
    int32_t block_col0;
    uint32_t whole_chunks;           // closest-hit passes: every wave that pushes hit or ray records takes whole chunks of 64 places and fills
                                     //   the rest with records marked empty (pixel HIT_NONE / RAY_NONE): a chunk of the passes that read
                                     //   them is then the output of ONE wave, neighbouring samples, not the tail of one wave's push and the
                                     //   head of another's from elsewhere in the image
    uint32_t *defer_counts;          // k_local only: the shard counters of the deferred queue (the frame's, in memory, where the
                                     //   block's other counters are in LDS); null: counts + CNT_DEFER(generation)
    uint32_t *commit;                // k_local only: per 64 records of rays_out, how many have been written (a word in LDS each)
    unsigned long long *totals;      // the one-launch kernel only: the frame's ray statistics (16 words)
    // Records at places that follow from the chunk's number (a primary pass that takes no place from a counter, and the
    // passes that read what it wrote):
    uint32_t *direct_counts;         // PRIMARY, or null: chunk c of the batch writes its hits to records c * 64 ... of `hits` and its
                                     //   rays to records c * 64 * direct_fan ... of `rays_out` (fillers up to whole groups of 64 as with
                                     //   whole_chunks), and leaves hits | rays << 8 (| hits shaded in place << 16, see fused_lights) in direct_counts[c]
    uint32_t direct_fan;             //   groups of 64 ray records a primary chunk can fill (1, or 2 where a hit can spawn two rays)
    uint32_t fused_lights;           // PRIMARY with direct_counts, grid-only kernel: the lights' terms of a primary hit are formed in the primary
                                     //   pass itself (shade_primary_hit) and no hit record is written; such hits are counted in bits 16..23
                                     //   of direct_counts[c] and reach the statistics through CNT_FUSED_HITS
    const uint32_t *hit_list;        // a pass that reads `hits`, or null: its k-th group of 64 records is group hit_list[k] of the buffer
                                     //   (k_compact_direct lists the groups that hold anything and puts their number, times 64, in shard
                                     //   0's counter; the other shards are empty)
    const uint32_t *ray_list;        // the same for `rays_in`
};

// A found primitive is named by its record slot: a triangle slot, or a sphere slot with this bit set.
#define SLOT_SPHERE 0x40000000
