// Sharded ray/hit queues (wave-aggregated pushes, chunked reads) and the run-reduced accumulation of
// pixel contributions.
#pragma once
#include "device_math.hpp"
#include "device_records.hpp"

DEV uint32_t fast_div(uint32_t n, const FastDiv &f)
{
    if (f.d < 2) return n;          // the divisor is the same for every lane
    const uint32_t t = __umulhi(f.magic, n);
    return (t + ((n - t) >> 1)) >> f.shift;
}

DEV int global_row(const FrameParams &P, int local_row)
{
    int band = (int)fast_div((uint32_t)local_row, P.div_band);
    int within = local_row - band * P.band_rows;
    return (band * P.n_parts + P.part) * P.band_rows + within;
}

// Number of set bits of `mask` below this lane's own (two instructions made for it).
DEV uint32_t rank_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave-aggregated queue push into this block's shard: one atomic per wave, lanes take
// consecutive slots.  Returns the record index in the whole queue.
DEV uint32_t wave_reserve(bool want, uint32_t *shard_counters, uint32_t shard_cap, uint32_t *overflow, bool *ok)
{
    const uint32_t shard = blockIdx.x & (N_SHARDS - 1);
    unsigned long long mask = __ballot(want);
    uint32_t n = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (n) {
        int leader = __ffsll((long long)mask) - 1;
        int lane = (int)(threadIdx.x & 63);
        if (lane == leader) base = atomicAdd(&shard_counters[shard], n);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);      // (the leader is the same for every lane: no trip through LDS)
    }
    uint32_t rank = rank_below(mask);
    uint32_t slot = base + rank;
    *ok = want && slot < shard_cap;
    if (want && slot >= shard_cap) *overflow = 1u;
    return shard * shard_cap + slot;
}

// View of a sharded queue as one list of 64-item chunks.  Lane l keeps shard l's item count and
// the index of its first chunk; chunk c belongs to the last shard whose first chunk is <= c.
struct ShardMap {
    uint32_t items;      // this lane's shard: number of items
    uint32_t first;      // this lane's shard: first chunk index
    uint32_t total;      // chunks in all shards
};
DEV ShardMap shard_map(const uint32_t *shard_counters, uint32_t shard_cap, uint32_t items_per_record)
{
    const int lane = (int)(threadIdx.x & 63);
    ShardMap m;
    uint32_t n = shard_counters[lane];
    if (n > shard_cap) n = shard_cap;
    m.items = n * items_per_record;
    uint32_t chunks = (m.items + 63u) >> 6;
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    m.first = incl - chunks;
    m.total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    return m;
}
// For chunk c (same in all lanes): shard, and this lane's item index inside the shard (or -1).
DEV int shard_locate(const ShardMap &m, uint32_t c, int *shard_out, int pos_in_chunk = -1)
{
    const unsigned long long le = __ballot(m.first <= c && m.items > 0);
    const int shard = 63 - __clzll((long long)le);
    // (the shard is the same for every lane: its words come by a lane read, not by a trip through LDS)
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)m.first, shard & 63);
    const uint32_t items = (uint32_t)__builtin_amdgcn_readlane((int)m.items, shard & 63);
    const uint32_t idx = (c - first) * 64u + (uint32_t)(pos_in_chunk >= 0 ? pos_in_chunk : (int)(threadIdx.x & 63));
    *shard_out = shard;
    return idx < items ? (int)idx : -1;
}

// The (hit, light) items of a shadow pass as a list of chunks in which every chunk holds ONE light: a shard's hits in
// groups of 64, and per group one chunk for each light.  The 64 rays of a wave then all run towards the same light --
// one bundle -- where items dealt out light by light within a hit gave every wave as many interleaved bundles as there
// are lights.  `items` is the shard's hit count here.
DEV ShardMap shard_map_by_light(const uint32_t *shard_counters, uint32_t shard_cap, uint32_t n_lights)
{
    const int lane = (int)(threadIdx.x & 63);
    ShardMap m;
    uint32_t n = shard_counters[lane];
    if (n > shard_cap) n = shard_cap;
    m.items = n;
    uint32_t chunks = ((n + 63u) >> 6) * n_lights;
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    m.first = incl - chunks;
    m.total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (n_lights == 0) m.items = 0;
    return m;
}
// For chunk c (same in all lanes) of such a list: shard, light (the same for the whole wave), and this lane's hit index
// inside the shard (or -1).
DEV int shard_locate_by_light(const ShardMap &m, uint32_t c, const FastDiv &div_lights, int *shard_out, int *light_out)
{
    const unsigned long long le = __ballot(m.first <= c && m.items > 0);
    const int shard = 63 - __clzll((long long)le);
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)m.first, shard & 63);
    const uint32_t hits = (uint32_t)__builtin_amdgcn_readlane((int)m.items, shard & 63);
    const uint32_t rel = c - first;
    const uint32_t group = fast_div(rel, div_lights);
    *light_out = (int)(rel - group * div_lights.d);
    const uint32_t idx = group * 64u + (uint32_t)(threadIdx.x & 63);
    *shard_out = shard;
    return idx < hits ? (int)idx : -1;
}

// Adds w*c to a pixel.  Lanes that hold the same pixel in a contiguous run of lanes are summed
// first and the run's first lane issues the (memory-side) FP64 atomics.  `in_queue` lanes take
// part in runs even when they contribute nothing (`valid` false), so a shadowed light does not
// split the run of its hit.
DEV double lane_down(double v, int byte_address)
{
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_address, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_address, __double2loint(v)));
}
DEV void accum_add_runs(double *accum, bool in_queue, bool valid, uint32_t pixel, double w, d3 c)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t key = in_queue ? pixel : 0xFFFFFFFFu;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool head = (lane == 0) || prev != key;
    // the last lane of this lane's run: the one before the next run head above it (runs are contiguous)
    const unsigned long long heads = __ballot(head);
    const unsigned long long above = heads & ~((2ull << lane) - 1ull);
    const int run_last = above ? __ffsll((unsigned long long)above) - 2 : 63;
    double vx = valid ? w * c.x : 0.0, vy = valid ? w * c.y : 0.0, vz = valid ? w * c.z : 0.0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const bool same = lane + off <= run_last;
        if (__ballot(same) == 0) break;      // no partner at this distance, none farther
        const int from = ((lane + off) & 63) << 2;      // (a lane that wraps around has no partner)
        const double ox = lane_down(vx, from), oy = lane_down(vy, from), oz = lane_down(vz, from);
        if (same) { vx += ox; vy += oy; vz += oz; }
    }
    if (in_queue && head && (vx != 0.0 || vy != 0.0 || vz != 0.0)) {
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 0], vx);
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 1], vy);
        unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 2], vz);
    }
}

// The same when every lane of the queue adds the same amount w*c (the background to the samples of a chunk that hit
// nothing): a run's sum is its length times that amount, no partial sums travel between lanes.
DEV void accum_add_equal(double *accum, bool in_queue, uint32_t pixel, double w, d3 c)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t key = in_queue ? pixel : 0xFFFFFFFFu;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool head = (lane == 0) || prev != key;
    const unsigned long long heads = __ballot(head);
    const unsigned long long above = heads & ~((2ull << lane) - 1ull);
    const int run_last = above ? __ffsll((unsigned long long)above) - 2 : 63;
    if (in_queue && head) {
        const double n = (double)(run_last - lane + 1);
        const double vx = n * (w * c.x), vy = n * (w * c.y), vz = n * (w * c.z);
        if (vx != 0.0 || vy != 0.0 || vz != 0.0) {
            unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 0], vx);
            unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 1], vy);
            unsafeAtomicAdd(&accum[(size_t)pixel * 3 + 2], vz);
        }
    }
}

// Queue records move as whole 16-byte pieces, all pieces of a record issued together: the compiler,
// left to the individual fields, splits a record into a dozen narrower accesses placed where each
// field happens to be used, and the memory pipeline charges per instruction, not per byte
// (8K x 16 spp frame: 108 -> 97 ms from the stores alone).  (Non-temporal stores were tried for
// these write-once records and are far slower: 178 ms.)
//
// Layout.  The records of a queue lie in groups of 64: piece k of the 64 records of a group side by side (1 KiB), then
// piece k + 1 of all of them, and so on.  A wave pushes records into consecutive places and reads 64 consecutive ones,
// so every store and load instruction of a record's piece then moves one or two contiguous runs of up to a kilobyte --
// whole cache lines -- where records kept whole (64, 80 or 128 bytes each) made every such instruction touch 64
// separate lines for 16 bytes of each.  Shard capacities and block slices are multiples of 64 records.
#define REC_PLANE 1024       // bytes from piece k of a record to its piece k + 1
typedef double rec_v2d __attribute__((ext_vector_type(2)));
DEV rec_v2d rec_pack(double a, double b) { rec_v2d v; v.x = a; v.y = b; return v; }
DEV rec_v2d rec_pack_words(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    rec_v2d v;
    v.x = __hiloint2double((int)b, (int)a);
    v.y = __hiloint2double((int)d, (int)c);
    return v;
}
// piece 0 of record `rec` of a queue whose records have `pieces` pieces (the queue starts at a multiple of 64 records)
DEV char *rec_at(const void *queue, size_t rec, uint32_t pieces)
{
    return (char *)queue + (rec >> 6) * ((size_t)pieces * REC_PLANE) + (rec & 63u) * 16u;
}
// pieces first .. first + N - 1 of the record whose piece 0 is at `rec0`
template <int N>
DEV void rec_store(char *rec0, int first, const rec_v2d (&q)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) *(__attribute__((address_space(1))) rec_v2d *)(rec0 + (first + k) * REC_PLANE) = q[k];
}
DEV rec_v2d rec_load(const char *rec0, int k)
{
    return *(const __attribute__((address_space(1))) rec_v2d *)(rec0 + k * REC_PLANE);
}
// piece k of a table entry or of a record that is kept whole (pieces back to back)
DEV rec_v2d tab_load(const void *src, int k)
{
    return ((const __attribute__((address_space(1))) rec_v2d *)src)[k];
}
template <int N>
DEV void tab_store(void *dst, const rec_v2d (&q)[N])
{
    __attribute__((address_space(1))) rec_v2d *p = (__attribute__((address_space(1))) rec_v2d *)dst;
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = q[k];
}
