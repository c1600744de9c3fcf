// rtx_shade.hip -- the frame: RayTracer::trace_rays (src/lib.rs:257-277) and the
// Whitted recursion shade_intersection (src/lib.rs:201-255) flattened into
// bounded-depth ray queues.
//
// Color clamps on every operation (src/scene/material.rs:17-27,65-84), so a
// node's colour needs its children's FINAL colours: throughput-weighted
// accumulation is not equivalent. The recursion therefore becomes
//   forward, level 0..D-1 : trace queue -> shade_a (hit geometry, node alloc)
//                           -> shadow rays (one task per node and light)
//                           -> shade_b (lights summed in scene order, children spawned)
//   backward, level D-1..0: combine ((direct+ambient) + ks*refl) + refr*kt, each
//                           child writing its final colour into its parent's slot.
// Every count lives in device memory (Ctl); the host only reads the size of the
// next queue once per level to stop early.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <utility>

#include <hip/hip_ext.h>

#include "rtx_coop.h"
#include "rtx_shading.h"

namespace {

struct Ctl {
    uint32_t n_rays[2];
    uint32_t n_nodes;
    uint32_t pool_base;
    uint32_t overflow;
    uint32_t spec_miss;   // levels were enqueued without asking how many the ray tree has, and it had more (see LevelHint)
    uint32_t pad0[1];
    uint32_t pad[1];
    uint32_t unit_head[2][8][32];   // per-shard work counters (closest, shadow), one 128-byte line each
    uint32_t level_start[RTX_MAX_DEPTH + 2];
    uint32_t level_rays[RTX_MAX_DEPTH + 2];   // rays traced at each level of the batch (level 0: its primary rays)
    unsigned long long stat[7];    // primary, shadow, reflect, refract, skipped, nodes, panic
    unsigned long long saved[7];   // snapshot at batch start, restored when a batch is re-rendered
};
enum { ST_PRIMARY = 0, ST_SHADOW, ST_REFLECT, ST_REFRACT, ST_SKIPPED, ST_NODES, ST_PANIC };

struct Work {
    RayQueue q[2];
    uint32_t cap_rays;
    float4 *nP, *nN, *nV, *nC;   // per-level node arrays: (point, mat0) (normal, flags) (ray dir: only for nodes with a specular term or children) (cd: written and read only where it is a blend of vertex materials)
    // Shadow terms, [n_slots][cap_rays]: one slot per sample of shadow_scalar's loop (lib.rs:113-135).
    // Light j owns slots first_slot(j) .. + n_j - 1, n_j = n_samples for an area light, else 1,
    // first_slot(j) = j + area_before(j) * (n_samples - 1).
    float* shadow;
    float4 *pL, *pC0, *pC1;      // retained pool: (local colour, mat0), reflected child colour, refracted child colour
    uint32_t* pLink;             // retained pool: parent link
    uint64_t* pKey;              // retained pool: sampling key of the node; nullptr without area lights
    // Hits per chunk of 2048 rays (counted by the closest-hit kernel) and their exclusive prefix
    // (k_scan_chunks): node slots follow ray order, whatever order the workgroups run in.
    uint32_t *chunk_cnt, *chunk_base;
    // per work unit (64 rays) of the closest-hit launch: which of its rays hit something. Rays that missed write no hit
    // record and k_shade_a reads none for them (in the 1080p bench frame 78 % of the primary rays leave the scene).
    unsigned long long* hit_mask;
    uint32_t cap_pool;
    uint32_t n_samples;          // RayTracer::num_samples, at least 1 here
    uint32_t area_off;           // num_samples == 0: area lights trace nothing and contribute nothing
    Ctl* ctl;
};

// Where a level-0 link (ray index of the rank) lies in the image.
struct PixMap {
    uint32_t level0, rank, world, tiles_x, width;
    uint32_t frames, rays_per_frame;   // a batch of several frames: link = frame * rays_per_frame + the frame's ray index
};
RTX_D uint32_t first_slot(const DevLight& l, uint32_t j, uint32_t n_samples) { return j + l.area_before * (n_samples - 1u); }

struct CamSetup {   // RayTracer fields after setup_camera (lib.rs:69-84) plus the two plane vectors (lib.rs:86-94)
    V3 center, cam_pos, hplane, vplane;
    float inv_w, inv_h;
    uint32_t w, h;
};

CamSetup make_cam(const rtx_camera& c, uint32_t w, uint32_t h) {
    const float SCALE = 10000.0f;  // lib.rs:22
    CamSetup cs;
    const V3 view = v3(c.view_dir[0], c.view_dir[1], c.view_dir[2]);
    const V3 up = v3(c.ortho_up[0], c.ortho_up[1], c.ortho_up[2]);
    const V3 pos = v3(c.pos[0], c.pos[1], c.pos[2]);
    V3 right = vcross(view, up);
    V3 pup = vcross(right, view);
    right = vnormalize(right);
    pup = vnormalize(pup);
    const float vfov = c.vertical_fov;
    const float hfov = c.vertical_fov * ((float)w / (float)h);
    cs.cam_pos = pos;
    cs.center = pos + vmult(view, SCALE);
    cs.vplane = vmult(pup, tanf(vfov / 2.0f) * SCALE);
    cs.hplane = vmult(right, tanf(hfov / 2.0f) * SCALE);
    cs.inv_w = 1.0f / (float)w;
    cs.inv_h = 1.0f / (float)h;
    cs.w = w;
    cs.h = h;
    return cs;
}

// RayTracer::compute_ray (lib.rs:96-103); the missing `- camera_pos` is the reference's behaviour.
RTX_D V3 compute_ray_dir(const CamSetup& c, float x, float y) {
    x = x * c.inv_w;
    y = y * c.inv_h;
    const V3 dx = vmult(c.hplane, 2.0f * x - 1.0f);
    const V3 dy = vmult(c.vplane, 2.0f * y - 1.0f);
    return vnormalize(c.center + dx + dy);
}

RTX_D uint64_t lanemask_lt() {
    const int lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}
// Wave-aggregated allocation: one atomic per wavefront. Must be called by all lanes of the wave.
RTX_D uint32_t wave_alloc(uint32_t* counter, bool want, uint32_t count_each = 1) {
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (uint32_t)__popcll(m) * count_each);
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & lanemask_lt()) * count_each;
}
RTX_D void wave_count(unsigned long long* counter, bool flag) {
    const uint64_t m = __ballot(flag);
    if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(counter, (unsigned long long)__popcll(m));
}

// Kernels with workgroup barriers leave early as a WHOLE workgroup: the flag can be raised by
// another workgroup of the same launch, so one thread reads it for all.
RTX_D bool block_sees_overflow(const Ctl* ctl) {
    __shared__ uint32_t s_overflow;
    if (threadIdx.x == 0) s_overflow = ctl->overflow;
    __syncthreads();
    return s_overflow != 0;
}

// Up to kCamsByValue cameras of a frame or batch travel as arguments of its first kernel (a copy from pinned memory
// costs a stream that renders one small frame after the other 20-40 us per frame: a hand-over to the copy engine and back).
constexpr uint32_t kCamsByValue = 8;
struct CamBlock {
    CamSetup cam[kCamsByValue];
};
// first: the first batch of a frame starts the frame's counters from zero.
__global__ void k_batch_init(Ctl* ctl, uint32_t n, int rollback, int first, CamSetup* cams_dst, uint32_t n_cams, CamBlock cams) {
    for (uint32_t k = 0; k < n_cams; ++k) cams_dst[k] = cams.cam[k];
    for (int k = 0; k < 7; ++k) {
        if (first) ctl->stat[k] = 0;
        if (rollback) ctl->stat[k] = ctl->saved[k];
        else ctl->saved[k] = ctl->stat[k];
    }
    ctl->overflow = 0;
    ctl->spec_miss = 0;
    for (int k = 0; k < (int)RTX_MAX_DEPTH + 2; ++k) ctl->level_rays[k] = 0;
    ctl->level_rays[0] = n;
    for (int k = 0; k < 8; ++k) ctl->unit_head[0][k][0] = ctl->unit_head[1][k][0] = 0;
    ctl->n_rays[0] = n;
    ctl->n_rays[1] = 0;
    ctl->n_nodes = 0;
    ctl->pool_base = 0;
    ctl->level_start[0] = 0;
}

// Primary rays of one batch. Ray g of a rank lives in tile (g / 64) of that rank's
// interleaved tile list and is pixel (g % 8, g / 8 % 8) of the tile: one
// wavefront = one 8x8-pixel tile, so its 64 rays are coherent. The level-0 kernels
// compute the ray from its index (same operations, same bits) instead of reading it
// from a queue; `enabled` = 0 for deeper levels.
struct PrimaryGen {
    CamSetup cam;
    uint64_t first;
    uint32_t rank, world, tiles_x, n_tiles;
    uint32_t enabled;
    // A batch of `frames` > 1 frames (one camera each, cams[frames] in device memory): ray g belongs to frame
    // g / rays_per_frame; rays_per_frame is the same multiple of 64 on every rank (tiles_per_rank * 64).
    uint32_t frames, rays_per_frame;
    const CamSetup* cams;
};
RTX_D void primary_ray_of(const CamSetup& cam, const PrimaryGen& pg, uint64_t g, uint64_t gl, float4& o4, float4& d4) {
    const uint64_t tile = (gl >> 6) * pg.world + pg.rank;
    const uint32_t lane = (uint32_t)(gl & 63);
    const uint32_t x = (uint32_t)(tile % pg.tiles_x) * RTX_TILE + (lane & 7u);
    const uint32_t y = (uint32_t)(tile / pg.tiles_x) * RTX_TILE + (lane >> 3);
    const bool valid = tile < pg.n_tiles && x < cam.w && y < cam.h;
    V3 d = v3(0.0f, 0.0f, 0.0f);
    if (valid) d = compute_ray_dir(cam, (float)x, (float)(cam.h - y - 1));   // lib.rs:263
    o4 = make_float4(cam.cam_pos.x, cam.cam_pos.y, cam.cam_pos.z, u2f((uint32_t)g));
    d4 = make_float4(d.x, d.y, d.z, u2f(valid ? (RTX_RAY_VALID | RTX_RAY_VACUUM) : 0u));
}
// lds_cams: the workgroup's copy of pg.cams in LDS, or nullptr (the traversal kernel asks for a camera per work unit,
// and most units of a frame with sky in it are done after a step or two: the camera's latency is a large part of them).
// Three separate calls, so that each reads its camera from one known address space (kernel arguments, LDS, memory).
// CAMS = false: the caller knows that there is no camera array (one frame, camera in the kernel arguments): the
// traversal kernel is compiled once that way, because the mere possibility of an array costs its per-unit set-up a
// quarter of the launch (the camera then lives in registers picked from three sources instead of in scalar operands).
template <bool CAMS = true>
RTX_D void primary_ray(const PrimaryGen& pg, uint32_t i, float4& o4, float4& d4, const CamSetup* lds_cams = nullptr) {
    const uint64_t g = pg.first + i;
    uint64_t gl = g;
    uint32_t frame = 0;
    if (pg.frames > 1) {   // first == 0 and g < 2^32 (checked by the host)
        frame = (uint32_t)g / pg.rays_per_frame;
        gl = (uint32_t)g - frame * pg.rays_per_frame;
    }
    // A wavefront's 64 rays belong to one frame (rays_per_frame is a multiple of 64): the index is uniform.
    frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)frame);
    if (!CAMS) {
        primary_ray_of(pg.cam, pg, g, gl, o4, d4);
        return;
    }
    // (the camera is copied into registers first: handing the core a reference would make the compiler keep one
    // pointer for the three sources, and a pointer to a kernel argument puts the whole argument block into scratch)
    CamSetup cam;
    if (lds_cams) cam = lds_cams[frame];
    else if (pg.cams) cam = pg.cams[frame];   // cameras handed in by the caller (device array)
    else cam = pg.cam;                        // one frame: its camera is a kernel argument
    primary_ray_of(cam, pg, g, gl, o4, d4);
}
// The same rays written to queue 0 (rtx_primary_hits reads them back).
__global__ __launch_bounds__(RTX_BLOCK) void k_raygen(PrimaryGen pg, Work wk, uint32_t n) {
    const uint32_t i = blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (i < n) {
        float4 o4, d4;
        primary_ray(pg, i, o4, d4);
        wk.q[0].o[i] = o4;
        wk.q[0].d[i] = d4;
    }
}

// Dynamic distribution of work in units of 64 items (one wavefront's worth). The grid
// is sized to the device, and a wavefront that finishes its unit takes the next one, so
// wavefronts with cheap rays do not idle while others work through expensive ones. The
// unit range is cut into 8 shards with one counter each (its own 128-byte line); a
// workgroup starts on shard blockIdx % 8 -- workgroups b and b+8 usually share an XCD, so
// a shard's rays stay in one L2 -- and moves on to the other shards when it runs dry.
// One word would serve only ~90 atomics per microsecond; eight shards do not limit a launch
// (32 shards measured ~10 % slower on the 1080p and 4K frames: a shard's rays spread over more of the image).
// Cost-ordered units. A launch ends with its slowest units, and those are slow because
// their rays are (a few rays grazing the surface take ten times the steps of the rest), so
// the launch is shortest when the expensive units start first. What a unit costs is not
// known before it ran, but the previous frame knows: every unit records its duration, one
// workgroup of the shading launch that follows (order_units) sorts the units by it, and the next
// launch of the same kind hands units out in that order whenever its unit count equals the
// recorded one -- a re-render, the next sample of the same view, or the next frame of a
// slowly moving camera. Any permutation of the units gives the same image; with no valid
// order the units go out in index order.
// Costs are kept per GROUP of 8 consecutive units (neighbouring tiles cost about the same):
// sorting a frame's few thousand groups takes one small workgroup a few microseconds, and a
// group stays together on one shard, which keeps its nodes in one XCD's L2.
#ifndef RTX_GROUP_LOG2
#define RTX_GROUP_LOG2 3
#endif
constexpr uint32_t kGroupLog2 = RTX_GROUP_LOG2, kGroup = 1u << kGroupLog2;
struct UnitOrder {
    uint32_t n_cost;    // unit count of the launch that last added to cost[] (0: none, or too many for cap)
    uint32_t n_order;   // unit count whose groups order[] is a permutation of (0: none)
    uint32_t cap;       // capacity in groups
    uint32_t fine_units; // per shard: the first fine_units units go out one per atomic, the cheap rest eight per atomic
    RTX_D uint32_t* cost() { return reinterpret_cast<uint32_t*>(this + 1); }          // [cap] summed durations, 10 ns ticks
    RTX_D uint32_t* order() { return reinterpret_cast<uint32_t*>(this + 1) + cap; }   // [cap] groups, most expensive first
};

struct UnitQueue {
    uint32_t* heads;        // [8][32]
    uint32_t n_units, n_groups, per_shard, shard, probe;
    uint32_t n_pos = 0, n_ordered = 0;   // sorted positions to hand out (the larger of the two group counts); groups the order knows
    uint32_t res_k = 0, res_n = 0, fine = 0xffffffffu;   // COARSE: next position of the current shard's counter that this wavefront holds, how many it holds
    const uint32_t* order;  // sorted position -> group, or nullptr
    UnitOrder* uo;
    RTX_D void init(uint32_t* h, uint64_t n_items, UnitOrder* uo_) {
        heads = h;
        n_units = (uint32_t)((n_items + 63) >> 6);
        n_groups = (n_units + kGroup - 1) >> kGroupLog2;
        per_shard = (n_units + 7) >> 3;
        shard = blockIdx.x & 7u;
        probe = 0;
        uo = uo_;
        // The order of the previous launch serves this one when it has the same number of units -- or nearly the same:
        // the shadow launch of the next frame of a moving camera has a few units more or fewer (its units follow the
        // hits), and its costs still lie where they lay. The order is a permutation of the groups it knew; groups past
        // those go out at the end in index order, groups it knew that no longer exist are skipped.
        order = nullptr;
        if (uo && n_units > 0 && uo->n_order > 0) {
            n_ordered = (uo->n_order + kGroup - 1) >> kGroupLog2;
            const uint32_t lo_g = min(n_groups, n_ordered), hi_g = max(n_groups, n_ordered);
            if (n_ordered <= uo->cap && hi_g - lo_g <= (hi_g >> 3)) {
                order = uo->order();
                n_pos = hi_g;
            }
        }
        fine = order ? uo->fine_units : 0xffffffffu;
    }
    // Called once per launch by every wavefront after init (only one thread writes).
    RTX_D void announce() {
        if (uo && blockIdx.x == 0 && threadIdx.x == 0) uo->n_cost = (n_groups <= uo->cap) ? n_units : 0u;
    }
    RTX_D void record(uint32_t unit, unsigned long long t0) {
        if (uo && (threadIdx.x & 63) == 0 && (unit >> kGroupLog2) < uo->cap)
            atomicMax(&uo->cost()[unit >> kGroupLog2], (uint32_t)(wall_clock64() - t0));   // a group is as urgent as its slowest unit: its units go out together
    }
    // Wave-uniform: next unit for this wavefront, or 0xffffffff when every shard is empty.
    // COARSE: units past the expensive ones of a shard (their groups cost under 1/16 of the dearest group: rays that
    // leave the scene at once) are reserved eight per atomic, so that the end of a launch, when nothing else is left,
    // is not one atomic round trip per unit and wavefront.
    template <bool COARSE = false>
    RTX_D uint32_t next() {
        uint32_t unit = 0xffffffffu;
        if ((threadIdx.x & 63) == 0) {
            while (probe < 8) {
                if (order) {   // shard sh owns the groups at sorted positions sh, sh + 8, ...: every shard starts with expensive ones
                    const uint32_t sh = (shard + probe) & 7u;
                    uint32_t k;
                    if (COARSE && res_n) {
                        k = res_k++;
                        --res_n;
                    } else {
                        const uint32_t take = (COARSE && res_k > fine) ? 8u : 1u;   // res_k: one past the last position seen on this shard
                        k = atomicAdd(&heads[sh * 32], take);
                        if (COARSE) {
                            res_k = k + 1u;
                            res_n = take - 1u;
                        }
                    }
                    const uint32_t gpos = (k >> kGroupLog2) * 8u + sh;
                    if (gpos < n_pos) {
                        const uint32_t g = gpos < n_ordered ? order[gpos] : gpos;   // beyond what the order knew: the group itself
                        const uint32_t u = g * kGroup + (k & (kGroup - 1u));
                        if (u < n_units) { unit = u; break; }
                        continue;   // a group that no longer exists, or past the end of the last, partial group
                    }
                    if (COARSE) res_k = res_n = 0;   // this shard is empty: the next one starts one unit at a time
                } else {
                    const uint32_t sh = (shard + probe) & 7u;
                    const uint32_t lo = sh * per_shard, hi = min(lo + per_shard, n_units);
                    if (lo < hi) {
                        const uint32_t u = lo + atomicAdd(&heads[sh * 32], 1u);
                        if (u < hi) { unit = u; break; }
                    }
                }
                ++probe;
            }
        }
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    }
};

// Sorts the groups of the last launch by recorded cost, most expensive first, and clears the
// costs for the next recording: a stable counting sort into 16 classes (two per octave below
// the largest cost), so groups of similar cost keep their index order. Work for ONE workgroup
// of 256 threads; the shading launch that follows a traversal launch carries one workgroup more
// than its shading needs, and that workgroup does nothing else: no launch of its own, and the
// sort runs beside the shading.
constexpr int kOrderThreads = RTX_BLOCK, kOrderClasses = 16;
// Half-octave code of a cost: 2 * floor(log2 c) + the bit below the leading one (0 for c <= 1), at most 63.
RTX_D uint32_t cost_code(uint32_t c) {
    c = max(c, 1u);
    const int e = 31 - __clz((int)c);
    return (uint32_t)(2 * e) + ((e > 0) ? ((c >> (e - 1)) & 1u) : 0u);
}
RTX_D void order_units(UnitOrder* uo) {
    __shared__ uint32_t offs[kOrderClasses * kOrderThreads];   // [class][thread]: count, then start position
    __shared__ uint32_t wave_sum[kOrderThreads / 64];
    __shared__ uint32_t s_max;
    const uint32_t n_units = uo->n_cost;
    const uint32_t n = (n_units + kGroup - 1) >> kGroupLog2;
    const uint32_t t = threadIdx.x;
    if (n_units == 0 || n > uo->cap) {   // block-uniform
        if (t == 0) uo->n_order = 0;
        __syncthreads();
        return;
    }
    uint32_t* cost = uo->cost();
    uint32_t* order = uo->order();
    const uint32_t per = (((n + kOrderThreads - 1) / kOrderThreads) + 3u) & ~3u;   // a multiple of 4: a thread's costs are whole 16-byte words
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);   // thread t sorts the groups [lo, hi)
    if (t == 0) s_max = 0;
    for (int k = 0; k < kOrderClasses; ++k) offs[k * kOrderThreads + t] = 0;
    __syncthreads();
    // A thread's costs are read once, four per load and all loads independent, and kept as one-byte codes when they
    // fit its registers (up to 16384 groups: four 1080p frames): the sort runs beside a shading launch that may be
    // short (tens of microseconds for one frame), so its own latency counts. cost[] is padded to a multiple of 4 entries (see unit_order).
    constexpr uint32_t kHeld = 16;   // registers: 4 codes each
    uint32_t held[kHeld];
    const bool fits = per <= 4 * kHeld;
    // More groups than that (eight 1080p frames have 32 400) are read twice, again four per load with independent loads --
    // once for the largest cost, once more (from the L2) for the classes, which are four bits and go eight to a word that is
    // parked in the place of the first of its eight costs (registers for them would be the shading blocks' registers too,
    // LDS their LDS: both cost the launch a resident workgroup). A walk of one dependent 4-byte load per group, thread and
    // pass -- what this path replaces -- took 165 us for the eight-frame batch: longer than the shading beside it.
    const uint32_t words4 = per / 4u, words8 = (per + 7u) / 8u;
    uint32_t mx = 0;
    if (!fits) {
#pragma unroll 4
        for (uint32_t k = 0; k < words4; ++k) {
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            if (lo + 4 * k < hi) c = *reinterpret_cast<const uint4*>(cost + lo + 4 * k);
            // entries past hi inside the last word: cost[] is padded with zeros, whose code is 0
            mx = max(max(mx, cost_code(c.x)), max(cost_code(c.y), max(cost_code(c.z), cost_code(c.w))));
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kHeld; ++k) {
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            if (lo + 4 * k < hi) c = *reinterpret_cast<const uint4*>(cost + lo + 4 * k);
            held[k] = cost_code(c.x) | (cost_code(c.y) << 8) | (cost_code(c.z) << 16) | (cost_code(c.w) << 24);
        }
#pragma unroll
        for (uint32_t k = 0; k < kHeld; ++k)
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if (lo + 4 * k + b < hi) mx = max(mx, (held[k] >> (8 * b)) & 0xffu);
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    const int top = (int)(s_max | 1u);   // class 0 = the upper half of the largest cost's octave
    auto cls = [top](uint32_t code) { return min(kOrderClasses - 1, top - (int)code); };
    if (!fits) {
#pragma unroll 2
        for (uint32_t r = 0; r < words8; ++r) {
            uint4 c0 = make_uint4(0u, 0u, 0u, 0u), c1 = c0;
            if (lo + 8 * r < hi) c0 = *reinterpret_cast<const uint4*>(cost + lo + 8 * r);
            if (lo + 8 * r + 4 < hi) c1 = *reinterpret_cast<const uint4*>(cost + lo + 8 * r + 4);
            const uint32_t word = (uint32_t)cls(cost_code(c0.x)) | ((uint32_t)cls(cost_code(c0.y)) << 4) | ((uint32_t)cls(cost_code(c0.z)) << 8) |
                                  ((uint32_t)cls(cost_code(c0.w)) << 12) | ((uint32_t)cls(cost_code(c1.x)) << 16) | ((uint32_t)cls(cost_code(c1.y)) << 20) |
                                  ((uint32_t)cls(cost_code(c1.z)) << 24) | ((uint32_t)cls(cost_code(c1.w)) << 28);
            if (lo + 8 * r < hi) cost[lo + 8 * r] = word;   // parked where its first cost was: the costs are spent, and cleared below
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b)
                if (lo + 8 * r + b < hi) ++offs[((word >> (4 * b)) & 0xfu) * kOrderThreads + t];
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kHeld; ++k)
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if (lo + 4 * k + b < hi) ++offs[cls((held[k] >> (8 * b)) & 0xffu) * kOrderThreads + t];
    }
    __syncthreads();
    // Exclusive scan of the class-major, thread-minor sequence in place; thread t scans the
    // 16 consecutive entries [16 t, 16 t + 16).
    uint32_t sum = 0;
    for (int k = 0; k < kOrderClasses; ++k) {
        const uint32_t c = offs[t * kOrderClasses + k];
        offs[t * kOrderClasses + k] = sum;
        sum += c;
    }
    uint32_t incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((int)(t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) wave_sum[t >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (uint32_t w = 0; w < (t >> 6); ++w) base += wave_sum[w];
    for (int k = 0; k < kOrderClasses; ++k) offs[t * kOrderClasses + k] += base;
    __syncthreads();
    // groups in the eight dearest half-octave classes (cost >= 1/16 of the dearest): where class 8 starts in `order`
    static_assert(kOrderClasses == 16, "class 8 is four octaves below the dearest group");
    const uint32_t dear_groups = (t == 0) ? offs[8 * kOrderThreads] : 0u;
    if (!fits) {
        for (uint32_t r = 0; r < words8; ++r) {
            const uint32_t word = (lo + 8 * r < hi) ? cost[lo + 8 * r] : 0u;
#pragma unroll
            for (uint32_t b = 0; b < 8; ++b)
                if (lo + 8 * r + b < hi) order[offs[((word >> (4 * b)) & 0xfu) * kOrderThreads + t]++] = lo + 8 * r + b;
            if (lo + 8 * r < hi) *reinterpret_cast<uint4*>(cost + lo + 8 * r) = make_uint4(0u, 0u, 0u, 0u);
            if (lo + 8 * r + 4 < hi) *reinterpret_cast<uint4*>(cost + lo + 8 * r + 4) = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kHeld; ++k) {
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if (lo + 4 * k + b < hi) order[offs[cls((held[k] >> (8 * b)) & 0xffu) * kOrderThreads + t]++] = lo + 4 * k + b;
            if (lo + 4 * k < hi) *reinterpret_cast<uint4*>(cost + lo + 4 * k) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    if (t == 0) {
        // groups in the eight dearest half-octave classes (cost >= 1/16 of the dearest) come first in `order`
        uo->fine_units = ((dear_groups + 7u) / 8u) * kGroup;   // per shard: sorted positions are dealt round-robin to 8 shards
        uo->n_order = n_units;
    }
    __syncthreads();
}

// k_shade_a turns hits into shading nodes in chunks of RTX_BLOCK * kShadeItems consecutive rays
// (= 1 << kChunkUnitsLog2 work units of the closest-hit kernel, which counts the hits per chunk).
#ifndef RTX_SHADE_ITEMS
#define RTX_SHADE_ITEMS 4
#endif
constexpr int kShadeItems = RTX_SHADE_ITEMS;
constexpr int kChunkUnitsLog2 = (kShadeItems == 8) ? 5 : (kShadeItems == 4) ? 4 : (kShadeItems == 2) ? 3 : 2;
static_assert(RTX_BLOCK * kShadeItems == 64 << kChunkUnitsLog2, "a chunk of k_shade_a is a power-of-two number of units");

// Closest hit for every ray of a queue (Scene::intersects, scene/mod.rs:181-201):
// one ray per lane (variant 0, the default).
template <bool SPH, bool ALWAYS = false, bool CAMS = false>
__global__ __launch_bounds__(RTX_BLOCK) RTX_WAVES_ATTR(closest_waves(SPH)) void k_trace_queue(DevScene s, RayQueue q, Ctl* ctl, int qi, PrimaryGen pg, UnitOrder* uo, uint32_t* chunk_cnt,
                                                                                                                    unsigned long long* hit_mask) {
    __shared__ int lds[RTX_STACK_LDS * RTX_BLOCK];
    __shared__ CoopLds coop;
    constexpr uint32_t kTop = top_nodes(closest_waves(SPH));
    __shared__ uint4 top_lds[kTop * 4];
    TravStack st;
    st.set(lds, s);
    // the cameras of a batch of frames, once per workgroup (up to kCamLds of them; more are read from memory per unit)
    constexpr uint32_t kCamLds = CAMS ? 16 : 1;
    __shared__ CamSetup cam_lds[kCamLds];
    const bool cams_in_lds = CAMS && pg.enabled && pg.cams && pg.frames <= kCamLds;
    if (CAMS && cams_in_lds)
        for (uint32_t k = threadIdx.x; k < pg.frames * (uint32_t)(sizeof(CamSetup) / 4); k += RTX_BLOCK)
            reinterpret_cast<uint32_t*>(cam_lds)[k] = reinterpret_cast<const uint32_t*>(pg.cams)[k];
    TopCache tc;
    tc.load(top_lds, s, kTop);   // ends with a workgroup barrier
    if (ctl->overflow) return;   // the batch is being abandoned
    const uint32_t n = ctl->n_rays[qi];
    WaveStamp ws;
    ws.begin(s, 0);
    UnitQueue uq;
    uq.init(&ctl->unit_head[0][0][0], n, uo);
    uq.announce();
    for (uint32_t unit = uq.next<true>(); unit != 0xffffffffu; unit = uq.next<true>()) {
        const unsigned long long unit_start = wall_clock64();
        ++ws.units;
        const uint32_t i = unit * 64 + (threadIdx.x & 63);
#ifdef RTX_COUNT_STEPS
        st.steps = 0;
        st.leaves = 0;
        const unsigned long long unit_t0 = wall_clock64();
#endif
        float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d4 = o4;
        if (i < n) {
            if (pg.enabled) {
                primary_ray<CAMS>(pg, i, o4, d4, cams_in_lds ? cam_lds : nullptr);
            } else {
                o4 = q.o[i];
                d4 = q.d[i];
            }
        }
        int id = -1;
        float t = 0.0f;
        coop_trace<false, SPH, ALWAYS>(s, tc, (f2u(d4.w) & RTX_RAY_VALID) != 0, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), 0.0f, st, coop,
                          s.tune[0], t, id);
        const uint64_t hits = __ballot(i < n && id >= 0);
        if (hit_mask) {   // the reader goes by the unit's mask: only hits are stored
            if (i < n && id >= 0) q.hit[i] = make_uint2((uint32_t)id, f2u(t));
            if ((threadIdx.x & 63) == 0) hit_mask[unit] = hits;
        } else if (i < n) {
            q.hit[i] = make_uint2((uint32_t)id, f2u(id >= 0 ? t : 0.0f));
        }
        if (chunk_cnt && hits && (threadIdx.x & 63) == 0) atomicAdd(&chunk_cnt[unit >> kChunkUnitsLog2], (uint32_t)__popcll(hits));
        uq.record(unit, unit_start);
#ifdef RTX_COUNT_STEPS
        ws.add_steps(st.steps);
        if (s.stamps) {   // per-unit record after the two wave tables: (start, duration | max iterations << 24 | sum << 44)
            uint32_t ms = st.steps + st.leaves, ml = st.steps + st.leaves;   // max and sum over lanes of the lane's iterations
            for (int o = 32; o > 0; o >>= 1) {
                ms = max(ms, (uint32_t)__shfl_xor(ms, o));
                ml += (uint32_t)__shfl_xor(ml, o);
            }
            if ((threadIdx.x & 63) == 0 && unit < (1u << 19)) {
                unsigned long long* ur = s.stamps + (size_t)2 * (s.spill_stride / 64) * 4 + (size_t)unit * 2;
                ur[0] = unit_t0;
                ur[1] = (wall_clock64() - unit_t0) | ((unsigned long long)ms << 24) | ((unsigned long long)ml << 44);
            }
        }
#endif
    }
#ifdef RTX_COUNT_STEPS
    ws.prof = ((st.t_mem >> 8) & 0x1fffff) | (((st.t_alu >> 8) & 0x1fffff) << 21) | (((st.t_don >> 8) & 0x1fffff) << 42);
#endif
    ws.end();
}

// Where level-0 colours go: the rank's packed tile slab (pixel slot = ray index) or, for a
// single-GPU frame, straight into the row-major image.
struct OutMap {
    uint8_t* rgb;
    uint32_t direct_w;   // 0: tile slab; otherwise the image width (row-major output)
    uint32_t tiles_x;
    // a batch of several frames: slabs follow each other at rays_per_frame pixels (so a link is its own slab index),
    // row-major images at frame_pixels
    uint32_t frames, rays_per_frame, frame_pixels;
    RTX_D size_t pixel(uint32_t link) const {
        if (!direct_w) return link;
        size_t base = 0;
        if (frames > 1) {
            const uint32_t frame = link / rays_per_frame;
            link -= frame * rays_per_frame;
            base = (size_t)frame * frame_pixels;
        }
        const uint32_t tile = link >> 6, lane = link & 63u;
        const uint32_t x = (tile % tiles_x) * RTX_TILE + (lane & 7u), y = (tile / tiles_x) * RTX_TILE + (lane >> 3);
        return base + (size_t)y * direct_w + x;
    }
};
// `direct + ambient + reflective + refractive` (lib.rs:254) with reflective = ks * child
// (lib.rs:234) and refractive = child.mult(kt) (lib.rs:244-245); `local` = direct + ambient.
// A missing child is a zero colour, which yields the same value as Color::new().
RTX_D Col combine_colour(const DevScene& s, uint32_t mat0, Col local, Col child_refl, Col child_refr) {
    const rtx_material* m = material_or_null(s, mat0);
    const Col ks = m ? col_raw(m->specular) : col_new();
    const float kt = m ? m->transparency : 0.0f;
    const Col reflective = ks * child_refl;
    const Col refractive = col_mult(child_refr, kt);
    return (local + reflective) + refractive;
}
RTX_D void store_pixel(const OutMap& out, uint32_t link, Col c) {   // Color::as_pixel, material.rs:57-63
    uint8_t* px = out.rgb + out.pixel(link) * 3;
    px[0] = col_u8(c.r);
    px[1] = col_u8(c.g);
    px[2] = col_u8(c.b);
}

// Forward step 1: every hit becomes a shading node (shade_intersection is only
// entered for hits: lib.rs:264-271, 232-236, 243-247). A workgroup takes a chunk of
// 256 * kShadeItems consecutive rays; the chunk's first node slot is the number of hits in
// the chunks before it (k_scan_chunks over the counts the traversal kernel took), and inside
// the chunk slots follow the ray order, so the node order is the ray order: no atomics, and
// the same frame gives the same order every time.

// Exclusive prefix of the per-chunk hit counts (one workgroup of 1024 threads: a batch of 16 Mi rays has 16 Ki chunks);
// sets the level's node count and clears the counts for the next launch. One CU does all of it, so what counts is how
// many cache lines its accesses touch: thread t takes the chunks t, t + 1024, ... (a wavefront reads and writes whole
// lines; a thread walking 16 consecutive chunks made every access of a wavefront 64 lines wide, 21 us for 16 Ki
// chunks), rows of 1024 chunks are scanned inside the wavefronts with shuffles, and the 16 x 16 (row, wavefront)
// totals are scanned once by the first wavefront.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_scan_chunks(Ctl* ctl, int qi, uint32_t* __restrict__ cnt, uint32_t* __restrict__ base) {
    constexpr uint32_t kRows = 16, kWaves = kScanThreads / 64;
    __shared__ uint32_t tot[kRows * kWaves];
    __shared__ uint32_t wave_sum[kWaves];
    const uint32_t n_chunks = (ctl->n_rays[qi] + RTX_BLOCK * kShadeItems - 1) / (RTX_BLOCK * kShadeItems);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (n_chunks <= kRows * kScanThreads) {   // block-uniform
        uint32_t v[kRows], inc[kRows];
#pragma unroll
        for (uint32_t j = 0; j < kRows; ++j) {
            const uint32_t c = j * kScanThreads + t;
            v[j] = (c < n_chunks) ? cnt[c] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kRows; ++j) {
            uint32_t x = v[j];
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o);
                if ((int)lane >= o) x += y;
            }
            inc[j] = x;
            if (lane == 63) tot[j * kWaves + wave] = x;
        }
        __syncthreads();
        if (wave == 0) {   // the 256 totals in (row, wavefront) order = chunk order: four per lane
            const uint32_t a0 = tot[4 * lane], a1 = tot[4 * lane + 1], a2 = tot[4 * lane + 2], a3 = tot[4 * lane + 3];
            const uint32_t sum = a0 + a1 + a2 + a3;
            uint32_t x = sum;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x, o);
                if ((int)lane >= o) x += y;
            }
            const uint32_t ex = x - sum;
            tot[4 * lane] = ex;
            tot[4 * lane + 1] = ex + a0;
            tot[4 * lane + 2] = ex + a0 + a1;
            tot[4 * lane + 3] = ex + a0 + a1 + a2;
            if (lane == 63) ctl->n_nodes = x;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < kRows; ++j) {
            const uint32_t c = j * kScanThreads + t;
            if (c < n_chunks) {
                base[c] = tot[j * kWaves + wave] + inc[j] - v[j];
                cnt[c] = 0;
            }
        }
        return;
    }
    // larger batches ("batch_pixels" above the default): a thread walks its consecutive chunks
    const uint32_t per = (n_chunks + kScanThreads - 1) / kScanThreads;
    const uint32_t lo = min(t * per, n_chunks), hi = min(lo + per, n_chunks);
    uint32_t sum = 0;
    for (uint32_t c = lo; c < hi; ++c) sum += cnt[c];
    uint32_t incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if ((int)lane >= o) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wave; ++w) run += wave_sum[w];
    for (uint32_t c = lo; c < hi; ++c) {
        base[c] = run;
        run += cnt[c];
        cnt[c] = 0;
    }
    if (t == kScanThreads - 1) ctl->n_nodes = run;
}
// Level 0 with row-major output also paints the pixels of primary rays that missed
// (bmp::Image::new's black), so the image needs no clearing pass.
__global__ __launch_bounds__(RTX_BLOCK) void k_shade_a(DevScene s, Work wk, int qi, PixMap pm, PrimaryGen pg, OutMap out,
                                                       UnitOrder* sort_uo) {
    __shared__ uint32_t cnt[kShadeItems][RTX_BLOCK / 64];
    Ctl* ctl = wk.ctl;
    // With costs to sort the launch has one workgroup more than the shading needs: the first one sorts the costs of the
    // closest-hit launch just finished and does nothing else, so the sort (4-30 us for one to eight 1080p frames) runs
    // beside the shading instead of in front of one workgroup's share of it.
    uint32_t bid = blockIdx.x, nb = gridDim.x;
    if (sort_uo) {
        if (blockIdx.x == 0) {
            order_units(sort_uo);
            return;
        }
        --bid;
        --nb;
    }
    if (block_sees_overflow(ctl)) return;
    const uint32_t n = ctl->n_rays[qi];
    const uint32_t pool_base = ctl->pool_base;
    const RayQueue q = wk.q[qi];
    const uint32_t chunk = RTX_BLOCK * kShadeItems;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt = lanemask_lt();
    for (uint32_t c0 = bid * chunk; c0 < n; c0 += nb * chunk) {   // block-uniform trip count
        // slots of this chunk's hits: prefix of the chunks before it (one address for the workgroup, asked for before
        // the hit records so that the two waits overlap)
        const uint32_t chunk_first = wk.chunk_base[c0 / chunk];
        uint2 hit[kShadeItems];
        uint32_t pre[kShadeItems];
#pragma unroll
        for (int k = 0; k < kShadeItems; ++k) {
            const uint32_t i = c0 + k * RTX_BLOCK + threadIdx.x;
            hit[k] = make_uint2(0xffffffffu, 0u);
            if ((i & ~63u) < n && ((wk.hit_mask[i >> 6] >> lane) & 1ull)) hit[k] = q.hit[i];   // i's unit = this wavefront's 64 rays
            const uint64_t m = __ballot((int)hit[k].x >= 0);
            pre[k] = (uint32_t)__popcll(m & lt);
            if (lane == 0) cnt[k][w] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        uint32_t run = chunk_first;
        uint32_t panic = 0;   // nodes of this thread where the reference would have panicked
#pragma unroll
        for (int k = 0; k < kShadeItems; ++k) {
            uint32_t before = 0, row = 0;
#pragma unroll
            for (int j = 0; j < RTX_BLOCK / 64; ++j) {
                const uint32_t c = cnt[k][j];
                if (j < w) before += c;
                row += c;
            }
            const uint32_t node = run + before + pre[k];
            run += row;
            if (pg.enabled && out.direct_w) {   // block-uniform: black for the primary rays that missed
                // A wavefront's 64 rays are one 8x8 tile and 8 consecutive lanes one row of it: 24 bytes of the
                // image. A row that missed completely is written as six words, any other miss as three bytes.
                const uint32_t i = c0 + k * RTX_BLOCK + threadIdx.x;
                bool black = false;
                size_t pix = 0;
                if (i < n && (int)hit[k].x < 0) {
                    const uint32_t g = (uint32_t)pg.first + i;   // the ray's link, as in OutMap::pixel
                    const uint32_t frame = (out.frames > 1) ? g / out.rays_per_frame : 0u;
                    const uint32_t fl = g - frame * out.rays_per_frame;
                    const uint32_t x = ((fl >> 6) % out.tiles_x) * RTX_TILE + (fl & 7u), y = ((fl >> 6) / out.tiles_x) * RTX_TILE + ((fl >> 3) & 7u);
                    black = x < pg.cam.w && y < pg.cam.h;   // the frames of a batch have one size
                    pix = (size_t)frame * out.frame_pixels + (size_t)y * out.direct_w + x;
                }
                const uint32_t row = (uint32_t)(__ballot(black) >> (lane & 56)) & 0xffu;
                uint8_t* first = out.rgb + (pix - (uint32_t)(lane & 7)) * 3;
                if (row == 0xffu && ((uintptr_t)first & 3u) == 0) {
                    if ((lane & 7) < 6) reinterpret_cast<uint32_t*>(first)[lane & 7] = 0u;
                } else if (black) {
                    uint8_t* px = out.rgb + pix * 3;
                    px[0] = px[1] = px[2] = 0;
                }
            }
            if ((int)hit[k].x < 0) continue;
            if (node >= wk.cap_rays || pool_base + node >= wk.cap_pool) {
                ctl->overflow = 1;
                continue;
            }
            const uint32_t i = c0 + k * RTX_BLOCK + threadIdx.x;
            float4 o4, d4;
            if (pg.enabled) {
                primary_ray(pg, i, o4, d4);
            } else {
                o4 = q.o[i];
                d4 = q.d[i];
            }
            const V3 o = v3(o4.x, o4.y, o4.z), d = v3(d4.x, d4.y, d4.z);
            const HitGeom g = hit_geometry(s, (int)hit[k].x, o, d, u2f(hit[k].y), true);
            panic += g.panic ? 1u : 0u;
            const uint32_t slot = pool_base + node;
            // The ray's direction enters the node's colour through the specular term and the children only. A material
            // with ks = 0, kt = 0 and a moderate exponent has neither (rj . v <= 1 + a few ulp, so powf stays finite and
            // ks * powf is exactly zero, see direct_lightning): such a node does without the direction. (|rj| is the length
            // of the light's direction: a directional light whose `dir` is longer than 1 -- the reference uses it as
            // given, scene/mod.rs:52-55 -- can drive powf to infinity, and 0 * inf is a NaN the reference's pixel shows.)
            const rtx_material* gm = material_or_null(s, g.mat0);
            const float gq = (gm ? gm->shininess : 0.0f) * 128.0f;
            const bool plain = (!gm || (gm->specular[0] == 0.0f && gm->specular[1] == 0.0f && gm->specular[2] == 0.0f && gm->transparency == 0.0f)) &&
                               gq >= 0.0f && gq <= 65536.0f && s.unit_light_dirs != 0u;
            const uint32_t nflags = (f2u(d4.w) & RTX_RAY_VACUUM) | (g.panic ? 4u : 0u) | (g.blended ? 8u : 0u) | (plain ? 0u : 16u);
            wk.nP[node] = make_float4(g.p.x, g.p.y, g.p.z, u2f(g.mat0));
            wk.nN[node] = make_float4(g.n.x, g.n.y, g.n.z, u2f(nflags));
            if (!plain) wk.nV[node] = make_float4(d.x, d.y, d.z, 0.0f);
            if (g.blended) wk.nC[node] = make_float4(g.cd.r, g.cd.g, g.cd.b, 0.0f);   // any other colour is the material's: k_shade_b looks it up
            wk.pLink[slot] = f2u(o4.w);
            if (wk.pKey) {   // the node's sampling key: hash of the pixel, extended along the path through the ray tree
                const uint32_t link = f2u(o4.w);
                uint64_t key;
                if (pm.level0) {
                    const uint32_t fl = (pm.frames > 1) ? link % pm.rays_per_frame : link;   // the key names the pixel, not the frame
                    const uint32_t tile = (fl >> 6) * pm.world + pm.rank, ln = fl & 63u;
                    const uint32_t x = (tile % pm.tiles_x) * RTX_TILE + (ln & 7u), y = (tile / pm.tiles_x) * RTX_TILE + (ln >> 3);
                    key = rng_mix64((uint64_t)y * pm.width + x);
                } else {
                    key = rng_push(wk.pKey[link & 0x7fffffffu], (link >> 31) ? kKeyRefract : kKeyReflect);
                }
                wk.pKey[slot] = key;
            }
        }
        if (panic) atomicAdd(&ctl->stat[ST_PANIC], (unsigned long long)panic);   // rare
        __syncthreads();
    }
}

// Forward step 2: RayTracer::shadow_scalar (lib.rs:105-141). One lane takes one sample of
// the outermost call's loop for one (node, light): a task is (slot, node), ordered
// slot-major so that a wavefront's rays converge on one light. The task stores the
// sample's term; k_shade_b adds a light's terms in sample order and divides by n.
// ANY: scenes without transparent materials, the first occluder ends the search.
// Otherwise: closest hit, and a transparent hit recurses (lib.rs:129) -- for an area light
// that inner call loops over n samples of its own. The recursion runs in the lane with an
// explicit stack of frames (origin, key, sample number, partial sum, kt: 8 words) parked
// in the top of the lane's spill column, so sums and products are formed in the
// reference's order. AREA: some light is an area light (keys are derived, see rng_push).
constexpr int kFrameWords = 8;
template <bool AREA>
struct ShadowTask {
    const DevScene* s;
    Work wk;
    uint32_t nn, depth, pool_base;
    TravStack* st;
    uint32_t slot, node, j;   // current task
    uint32_t n;          // samples per shadow_scalar call for this light
    uint32_t rem;        // remaining recursion depth of the call whose sample is being traced
    int frames;          // suspended calls
    uint64_t ks;         // key of the sample being traced
    unsigned long long rays;

    RTX_D void store(float v) { wk.shadow[(size_t)slot * wk.cap_rays + node] = v; }
    RTX_D int* frame_word(int f, int w) {
        return st->spill + (size_t)(s->spill_entries - 1 - (uint32_t)(f * kFrameWords + w)) * st->spill_stride;
    }
    // Sets up the shadow ray of task `item`; false when the light is skipped (fattj <= 0, lib.rs:218).
    RTX_D bool start(uint32_t item, V3& o, V3& d, float& dl) {
        slot = item / nn;
        node = item - slot * nn;
        j = slot;
        uint32_t smp = 0;
        if (AREA) {
            j = 0;
            for (uint32_t k = 0; k < s->n_lights; ++k) {
                const uint32_t first = k + s->lights[k].area_before * (wk.n_samples - 1u);
                if (first <= slot) {
                    j = k;
                    smp = slot - first;
                }
            }
        }
        const DevLight light = s->lights[j];
        n = (AREA && light.type == RTX_LIGHT_AREA) ? wk.n_samples : 1u;
        if (AREA && light.type == RTX_LIGHT_AREA && wk.area_off) {
            store(0.0f);
            return false;
        }
        const float4 p4 = wk.nP[node], n4 = wk.nN[node];
        const V3 point = v3(p4.x, p4.y, p4.z), normal = v3(n4.x, n4.y, n4.z);
        uint64_t kl = 0;
        if (AREA) kl = rng_push(wk.pKey[pool_base + node], kKeyLight + j);
        if (!(calculate_fattj(light, point, AREA ? rng_push(kl, 0) : 0ull) > 0.0f)) {
            store(0.0f);
            return false;
        }
        o = point + vmult(normal, 0.0001f);                   // lib.rs:111
        ks = AREA ? rng_push(rng_push(kl, 1), kKeyShadowSample + smp) : 0ull;
        d = light_dir(light, o, AREA ? rng_push(ks, 1) : 0ull);
        dl = (light.type == RTX_LIGHT_DIRECTIONAL) ? -1.0f : vdistance(o, light_pos(light, AREA ? rng_push(ks, 2) : 0ull));
        rem = depth;
        frames = 0;
        ++rays;
        return true;
    }
    RTX_D void finish_any(bool occluded) { store(occluded ? 0.0f : 1.0f); }
    // Closest-hit form: (id, t) is the closest hit of the ray (o, d). True: continue with the new ray in o/d.
    RTX_D bool finish_closest(int id, float t, V3& o, V3& d) {
        const DevLight light = s->lights[j];
        float term = 1.0f;                                    // Missed, lib.rs:133
        if (id >= 0) {
            const HitGeom g = hit_geometry(*s, id, o, d, t, false);
            const rtx_material* m = material_or_null(*s, g.mat0);
            const float kt = m ? m->transparency : 0.0f;
            if (kt == 0.0f) {                                 // lib.rs:120-128
                term = (light.type == RTX_LIGHT_DIRECTIONAL)
                           ? 0.0f
                           : ((vdistance(o, g.p) > vdistance(o, light_pos(light, AREA ? rng_push(ks, 2) : 0ull))) ? 1.0f : 0.0f);
            } else if (rem - 1u == 0u) {
                term = kt * 0.0f;                             // depth exhausted: the inner call returns Color::new()
            } else {                                          // kt * shadow_scalar(hit, n, depth - 1), lib.rs:129
                const V3 ori = g.p + vmult(g.n, 0.0001f);
                const uint64_t key = AREA ? rng_push(ks, 3) : 0ull;
                *frame_word(frames, 0) = (int)f2u(ori.x);
                *frame_word(frames, 1) = (int)f2u(ori.y);
                *frame_word(frames, 2) = (int)f2u(ori.z);
                *frame_word(frames, 3) = (int)(uint32_t)key;
                *frame_word(frames, 4) = (int)(uint32_t)(key >> 32);
                *frame_word(frames, 5) = 0;
                *frame_word(frames, 6) = (int)f2u(0.0f);
                *frame_word(frames, 7) = (int)f2u(kt);
                ++frames;
                --rem;
                ks = AREA ? rng_push(key, kKeyShadowSample) : 0ull;
                o = ori;
                d = light_dir(light, o, AREA ? rng_push(ks, 1) : 0ull);
                ++rays;
                return true;
            }
        }
        while (frames > 0) {                                  // the sample's term goes to the innermost suspended call
            const int f = frames - 1;
            const float shade = u2f((uint32_t)*frame_word(f, 6)) + term;
            const uint32_t smp = (uint32_t)*frame_word(f, 5) + 1u;
            if (smp < n) {                                    // next sample of that call
                *frame_word(f, 5) = (int)smp;
                *frame_word(f, 6) = (int)f2u(shade);
                const uint64_t key = (uint64_t)(uint32_t)*frame_word(f, 3) | ((uint64_t)(uint32_t)*frame_word(f, 4) << 32);
                ks = AREA ? rng_push(key, kKeyShadowSample + smp) : 0ull;
                o = v3(u2f((uint32_t)*frame_word(f, 0)), u2f((uint32_t)*frame_word(f, 1)), u2f((uint32_t)*frame_word(f, 2)));
                d = light_dir(light, o, AREA ? rng_push(ks, 1) : 0ull);
                ++rays;
                return true;
            }
            term = u2f((uint32_t)*frame_word(f, 7)) * clamp01(shade / (float)n);   // kt * Color::init(shade / n).r
            --frames;
            ++rem;
        }
        store(term);
        return false;
    }
};

template <bool ANY, bool AREA, bool SPH, bool ALWAYS = false>
__global__ __launch_bounds__(RTX_BLOCK) RTX_WAVES_ATTR(shadow_waves(ANY, AREA, SPH)) void k_shadow(DevScene s, Work wk, uint32_t depth, uint32_t n_slots, UnitOrder* uo) {
    __shared__ int lds[RTX_STACK_LDS * RTX_BLOCK];
    __shared__ CoopLds coop;
    constexpr uint32_t kTop = top_nodes(shadow_waves(ANY, AREA, SPH));
    __shared__ uint4 top_lds[kTop * 4];
    TravStack st;
    st.set(lds, s);
    TopCache tc;
    tc.load(top_lds, s, kTop);
    Ctl* ctl = wk.ctl;
    if (ctl->overflow) return;
    ShadowTask<AREA> task;
    task.s = &s;
    task.wk = wk;
    task.nn = min(ctl->n_nodes, wk.cap_rays);
    task.pool_base = ctl->pool_base;
    task.depth = depth;
    task.st = &st;
    task.rays = 0;
    const uint64_t total = min((uint64_t)task.nn * n_slots, (uint64_t)0xfffffff0u);
    WaveStamp ws;
    ws.begin(s, 1);
    UnitQueue uq;
    uq.init(&ctl->unit_head[1][0][0], total, uo);
    uq.announce();
#ifdef RTX_COUNT_STEPS
    unsigned long long dq0 = __builtin_readcyclecounter();
#endif
    for (uint32_t unit = uq.next(); unit != 0xffffffffu; unit = uq.next()) {
        const unsigned long long unit_start = wall_clock64();
#ifdef RTX_COUNT_STEPS
        const unsigned long long su0 = __builtin_readcyclecounter();
        st.t_deq += su0 - dq0;
#endif
        ++ws.units;
        const uint64_t k = (uint64_t)unit * 64 + (threadIdx.x & 63);
        V3 o = v3(0.0f, 0.0f, 0.0f), d = o;
        float dl = -1.0f, t;
        int id;
        bool live = k < total && task.start((uint32_t)k, o, d, dl);
#ifdef RTX_COUNT_STEPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st.t_setup += __builtin_readcyclecounter() - su0;
#endif
        if (ANY) {
            coop_trace<true, SPH, ALWAYS>(s, tc, live, o, d, dl, st, coop, s.tune[0], t, id);
            if (live) task.finish_any(id != 0);
        } else {
            while (__ballot(live)) {   // all lanes stay in the loop: the traversal is cooperative
                coop_trace<false, SPH, ALWAYS>(s, tc, live, o, d, 0.0f, st, coop, s.tune[0], t, id);
                if (live) live = task.finish_closest(id, t, o, d);
            }
        }
        uq.record(unit, unit_start);
#ifdef RTX_COUNT_STEPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        dq0 = __builtin_readcyclecounter();
#endif
    }
#ifdef RTX_COUNT_STEPS
    // five shares of the wavefront's life in units of 256 shader cycles, 12 bits each: fetch wait, compute, giving work, taking units, ray set-up
    ws.prof = ((st.t_mem >> 8) & 0xfff) | (((st.t_alu >> 8) & 0xfff) << 12) | (((st.t_don >> 8) & 0xfff) << 24) |
              (((st.t_deq >> 8) & 0xfff) << 36) | (((st.t_setup >> 8) & 0xfff) << 48);
#endif
    ws.end();
    unsigned long long rays = task.rays;
    for (int o = 32; o > 0; o >>= 1) rays += __shfl_xor(rays, o);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(&ctl->stat[ST_SHADOW], rays);
}

// Forward step 3: local colour and children. `finalize`: the frame has a single level (no
// material reflects or refracts, or depth 1), so a node's colour is final: it goes straight to
// the image and the level bookkeeping of k_next_level is done here.
__global__ __launch_bounds__(RTX_BLOCK) void k_shade_b(DevScene s, Work wk, int qo, uint32_t node_depth, uint32_t n_samples,
                                                       OutMap out, int finalize, UnitOrder* sort_uo) {
    Ctl* ctl = wk.ctl;
    uint32_t bid = blockIdx.x, nb = gridDim.x;
    if (sort_uo) {   // the launch's first workgroup only sorts (costs of the shadow launch just finished), see k_shade_a
        if (blockIdx.x == 0) {
            order_units(sort_uo);
            return;
        }
        --bid;
        --nb;
    }
    if (block_sees_overflow(ctl)) return;
    const uint32_t nn = ctl->n_nodes;
    const uint32_t pool_base = ctl->pool_base;   // node k of this level keeps slot pool_base + k of the pool (k_shade_a)
    if (finalize && bid == 0 && threadIdx.x == 0) ctl->stat[ST_NODES] += min(nn, wk.cap_rays);
    const RayQueue q = wk.q[qo];
    const uint32_t stride = nb * RTX_BLOCK;
    const bool spawn = node_depth > 1;   // children are shaded at node_depth - 1
    __shared__ uint32_t s_wave_rays[RTX_BLOCK / 64], s_ray_base, s_stat[2];
    if (threadIdx.x < 2) s_stat[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c_refl = 0, c_refr = 0;
    for (uint32_t base = bid * RTX_BLOCK; base < nn; base += stride) {
        const uint32_t node = base + threadIdx.x;
        bool refl = false, refr = false;
        V3 ro0, rd0, ro1, rd1;
        uint32_t slot = 0;
        if (node < nn) {
            const float4 p4 = wk.nP[node], n4 = wk.nN[node];
            const V3 point = v3(p4.x, p4.y, p4.z), normal = v3(n4.x, n4.y, n4.z);
            const uint32_t mat0 = f2u(p4.w), nflags = f2u(n4.w);
            V3 dir = v3(0.0f, 0.0f, 0.0f);   // not stored for nodes whose colour and children do not depend on it (k_shade_a)
            if (nflags & 16u) {
                const float4 v4 = wk.nV[node];
                dir = v3(v4.x, v4.y, v4.z);
            }
            slot = pool_base + node;
            const rtx_material* m = material_or_null(s, mat0);
            Col cd = m ? col_raw(m->diffuse) : col_new();   // as hit_geometry sets it, unless the colour is a blend of vertex materials
            if (nflags & 8u) {
                const float4 c4 = wk.nC[node];
                cd = Col{c4.x, c4.y, c4.z};
            }
            const float kt = m ? m->transparency : 0.0f;
            const float shininess = m ? m->shininess : 0.0f;
            const Col ks = m ? col_raw(m->specular) : col_new();
            const Col ka = m ? col_raw(m->ambient) : col_new();
            const Col ambient_light = col_mult(cd * ka, 1.0f - kt);   // lib.rs:143-145
            Col direct_light = col_new();
            const uint64_t key = wk.pKey ? wk.pKey[slot] : 0ull;
            // Without area lights slot j is light j and every slot was written (0 for a light that is skipped): the terms
            // of up to four lights are asked for here, together, instead of one wait per light inside the loop.
            const bool early_terms = s.n_area_lights == 0 && s.n_lights <= 4;   // uniform
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
            if (early_terms) {
                const float* term = wk.shadow + node;
                if (s.n_lights > 0) t0 = term[0];
                if (s.n_lights > 1) t1 = term[(size_t)wk.cap_rays];
                if (s.n_lights > 2) t2 = term[(size_t)2 * wk.cap_rays];
                if (s.n_lights > 3) t3 = term[(size_t)3 * wk.cap_rays];
            }
            for (uint32_t j = 0; j < s.n_lights; ++j) {              // lib.rs:216-228, scene order
                const DevLight light = s.lights[j];
                const bool area = light.type == RTX_LIGHT_AREA;
                const uint64_t kl = area ? rng_push(key, kKeyLight + j) : 0ull;
                const float fattj = calculate_fattj(light, point, area ? rng_push(kl, 0) : 0ull);
                if (fattj > 0.0f) {
                    const uint32_t n = area ? n_samples : 1u;         // lib.rs:219-222
                    const float* term = wk.shadow + (size_t)first_slot(light, j, wk.n_samples) * wk.cap_rays + node;
                    float shade = 0.0f;                               // lib.rs:112-138
                    if (early_terms) shade += (j == 0) ? t0 : (j == 1) ? t1 : (j == 2) ? t2 : t3;
                    else for (uint32_t smp = 0; smp < n; ++smp) shade += term[(size_t)smp * wk.cap_rays];
                    shade = clamp01(n != 1u ? shade / (float)n : shade);
                    direct_light = direct_light + direct_lightning(light, point, normal, dir, cd, ks, kt, shininess,
                                                                   Col{shade, shade, shade}, fattj, n, area ? rng_push(kl, 2) : 0ull);
                }
            }
            const Col local = direct_light + ambient_light;           // first '+' of lib.rs:254
            if (finalize) {
                store_pixel(out, wk.pLink[slot], combine_colour(s, mat0, local, col_new(), col_new()));
            } else {
                wk.pL[slot] = make_float4(local.r, local.g, local.b, u2f(mat0));
                wk.pC0[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                wk.pC1[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (col_scalar(ks) > 0.0f) {                              // lib.rs:230
                refl = true;
                reflective_ray(dir, point, normal, ro0, rd0);
            }
            if (kt > 0.0f) refr = refractive_ray(dir, point, normal, (nflags & RTX_RAY_VACUUM) != 0, ro1, rd1);   // lib.rs:241-242
        }
        c_refl += refl ? 1u : 0u;
        c_refr += refr ? 1u : 0u;
        if (!spawn) continue;   // block-uniform: the children would be shaded at depth 0 (counted as skipped below)
        // One reservation per workgroup and pass for all its new rays (a per-wavefront atomic on the
        // queue counter serialises: ~90 same-address atomics per microsecond): reflected rays of
        // wavefront 0, refracted rays of wavefront 0, reflected rays of wavefront 1, ...
        const uint64_t m0 = __ballot(refl), m1 = __ballot(refr);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) s_wave_rays[wv] = (uint32_t)(__popcll(m0) + __popcll(m1));
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int j = 0; j < RTX_BLOCK / 64; ++j) total += s_wave_rays[j];
            s_ray_base = total ? atomicAdd(&ctl->n_rays[qo], total) : 0u;
        }
        __syncthreads();
        uint32_t wave_base = s_ray_base;
        for (int j = 0; j < wv; ++j) wave_base += s_wave_rays[j];
        const uint64_t lt = lanemask_lt();
        const uint32_t i0 = wave_base + (uint32_t)__popcll(m0 & lt);
        const uint32_t i1 = wave_base + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1 & lt);
        if (refl && i0 >= wk.cap_rays) { ctl->overflow = 1; refl = false; }
        if (refl) {
            q.o[i0] = make_float4(ro0.x, ro0.y, ro0.z, u2f(slot));
            q.d[i0] = make_float4(rd0.x, rd0.y, rd0.z, u2f(RTX_RAY_VALID | RTX_RAY_VACUUM));
        }
        if (refr && i1 >= wk.cap_rays) { ctl->overflow = 1; refr = false; }
        if (refr) {   // refracted rays are never "in vacuum" again (intersection.rs:77-78)
            q.o[i1] = make_float4(ro1.x, ro1.y, ro1.z, u2f(slot | 0x80000000u));
            q.d[i1] = make_float4(rd1.x, rd1.y, rd1.z, u2f(RTX_RAY_VALID));
        }
        __syncthreads();   // s_wave_rays is reused by the next pass
    }
    // ray statistics: one atomic per workgroup and counter
    for (int o = 32; o > 0; o >>= 1) {
        c_refl += __shfl_xor(c_refl, o);
        c_refr += __shfl_xor(c_refr, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_stat[0], c_refl);
        atomicAdd(&s_stat[1], c_refr);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_stat[0]) atomicAdd(&ctl->stat[ST_REFLECT], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&ctl->stat[ST_REFRACT], (unsigned long long)s_stat[1]);
        if (!spawn && (s_stat[0] + s_stat[1])) atomicAdd(&ctl->stat[ST_SKIPPED], (unsigned long long)s_stat[0] + s_stat[1]);
    }
}

// last_enqueued: no further level follows although the frame's depth would allow one (the host enqueued as many
// levels as the previous frame used, plus one): rays left for it mean that the frame has to be rendered again.
__global__ void k_next_level(Ctl* ctl, int q_consumed, uint32_t level, uint32_t cap_rays, int last_enqueued) {
    ctl->stat[ST_NODES] += min(ctl->n_nodes, cap_rays);
    ctl->pool_base += min(ctl->n_nodes, cap_rays);
    ctl->level_start[level + 1] = ctl->pool_base;
    ctl->n_nodes = 0;
    for (int k = 0; k < 8; ++k) ctl->unit_head[0][k][0] = ctl->unit_head[1][k][0] = 0;
    ctl->n_rays[q_consumed] = 0;
    if (ctl->n_rays[q_consumed ^ 1] > cap_rays) ctl->n_rays[q_consumed ^ 1] = cap_rays;
    ctl->level_rays[level + 1] = ctl->n_rays[q_consumed ^ 1];
    if (last_enqueued && ctl->n_rays[q_consumed ^ 1] != 0) ctl->spec_miss = 1;
}

// Backward pass, one level per launch from the deepest up: a node's final colour goes into
// its parent's child slot, or, at level 0, into the image.
__global__ __launch_bounds__(RTX_BLOCK) void k_combine(DevScene s, Work wk, uint32_t level, OutMap out) {
    const Ctl* ctl = wk.ctl;
    if (ctl->overflow) return;
    const uint32_t lo = ctl->level_start[level], hi = ctl->level_start[level + 1];
    for (uint32_t slot = lo + blockIdx.x * RTX_BLOCK + threadIdx.x; slot < hi; slot += gridDim.x * RTX_BLOCK) {
        const float4 l4 = wk.pL[slot], c0 = wk.pC0[slot], c1 = wk.pC1[slot];
        const Col c = combine_colour(s, f2u(l4.w), Col{l4.x, l4.y, l4.z}, Col{c0.x, c0.y, c0.z}, Col{c1.x, c1.y, c1.z});
        const uint32_t link = wk.pLink[slot];
        if (level == 0) {
            store_pixel(out, link, c);
        } else {
            const uint32_t parent = link & 0x7fffffffu;
            float4* dst = (link >> 31) ? wk.pC1 : wk.pC0;
            dst[parent] = make_float4(c.r, c.g, c.b, 0.0f);
        }
    }
}

// Gathered slabs [world][frames][tiles_per_rank][64][3] -> row-major images [frames][h][w][3].
__global__ __launch_bounds__(RTX_BLOCK) void k_assemble(uint32_t w, uint32_t h, uint32_t world, uint32_t tiles_x,
                                                        uint64_t tiles_per_rank, uint32_t frames, const uint8_t* __restrict__ slabs,
                                                        uint8_t* __restrict__ rgb) {
    const uint64_t fpix = (uint64_t)w * h, npix = fpix * frames;
    for (uint64_t p = (uint64_t)blockIdx.x * RTX_BLOCK + threadIdx.x; p < npix; p += (uint64_t)gridDim.x * RTX_BLOCK) {
        const uint64_t frame = p / fpix, fp = p - frame * fpix;
        const uint32_t x = (uint32_t)(fp % w), y = (uint32_t)(fp / w);
        const uint64_t tile = (uint64_t)(y / RTX_TILE) * tiles_x + x / RTX_TILE;
        const uint64_t src = (((tile % world) * frames + frame) * tiles_per_rank + tile / world) * 64 + (y % RTX_TILE) * 8 + x % RTX_TILE;
        rgb[p * 3 + 0] = slabs[src * 3 + 0];
        rgb[p * 3 + 1] = slabs[src * 3 + 1];
        rgb[p * 3 + 2] = slabs[src * 3 + 2];
    }
}

__global__ __launch_bounds__(RTX_BLOCK) void k_primary_out(RayQueue q, uint64_t first, uint32_t n, uint32_t w, uint32_t h,
                                                           uint32_t tiles_x, int32_t* __restrict__ prim,
                                                           float* __restrict__ t, float* __restrict__ dir) {
    const uint32_t i = blockIdx.x * RTX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 d4 = q.d[i];
    if (!(f2u(d4.w) & RTX_RAY_VALID)) return;
    const uint32_t tile = (uint32_t)((first + i) >> 6), lane = (uint32_t)((first + i) & 63);
    const uint32_t x = (tile % tiles_x) * RTX_TILE + (lane & 7u), y = (tile / tiles_x) * RTX_TILE + (lane >> 3);
    const size_t p = (size_t)y * w + x;
    const uint2 hit = q.hit[i];
    if (prim) prim[p] = (int)hit.x;
    if (t) t[p] = ((int)hit.x >= 0) ? u2f(hit.y) : 0.0f;
    if (dir) { dir[p * 3] = d4.x; dir[p * 3 + 1] = d4.y; dir[p * 3 + 2] = d4.z; }
}

template <typename T>
int walloc(rtx_context* ctx, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, std::string("hipMalloc(workspace): ") + hipGetErrorString(e));
    ctx->work_allocs.push_back(p);
    *out = (T*)p;
    return RTX_OK;
}

int ensure_workspace(rtx_context* ctx, uint64_t batch, uint32_t slots, Work* out) {
    static_assert(sizeof(Work) < 512, "Work is passed by value");
    slots = std::max<uint32_t>(slots, 1);
    const bool keys = ctx->ds.n_area_lights != 0;
    if (ctx->work_batch >= batch && ctx->work_slots >= slots && ctx->work_keys == keys && ctx->work_blob.size() == sizeof(Work)) {
        std::memcpy(out, ctx->work_blob.data(), sizeof(Work));
        return RTX_OK;
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);   // frames in flight still use the old workspace
    rtx_clear_graphs(ctx);
    for (void* p : ctx->work_allocs) (void)hipFree(p);
    ctx->work_allocs.clear();
    ctx->work_batch = 0;
    Work wk{};
    const uint64_t cap_rays64 = std::min<uint64_t>(2 * batch, 0x7fffffffull);
    const uint64_t cap_pool64 = std::min<uint64_t>((uint64_t)ctx->opt_pool_factor * batch, 0x7fffffffull);
    wk.cap_rays = (uint32_t)cap_rays64;
    wk.cap_pool = (uint32_t)cap_pool64;
#define W_TRY(x) do { int rc_ = (x); if (rc_ != RTX_OK) return rc_; } while (0)
    for (int k = 0; k < 2; ++k) {
        W_TRY(walloc(ctx, &wk.q[k].o, wk.cap_rays));
        W_TRY(walloc(ctx, &wk.q[k].d, wk.cap_rays));
        W_TRY(walloc(ctx, &wk.q[k].hit, wk.cap_rays));
    }
    W_TRY(walloc(ctx, &wk.nP, wk.cap_rays));
    W_TRY(walloc(ctx, &wk.nN, wk.cap_rays));
    W_TRY(walloc(ctx, &wk.nV, wk.cap_rays));
    W_TRY(walloc(ctx, &wk.nC, wk.cap_rays));
    W_TRY(walloc(ctx, &wk.shadow, (size_t)wk.cap_rays * slots));
    W_TRY(walloc(ctx, &wk.pL, wk.cap_pool));
    W_TRY(walloc(ctx, &wk.pC0, wk.cap_pool));
    W_TRY(walloc(ctx, &wk.pC1, wk.cap_pool));
    W_TRY(walloc(ctx, &wk.pLink, wk.cap_pool));
    {
        const size_t chunks = (size_t)wk.cap_rays / (RTX_BLOCK * kShadeItems) + 2;
        W_TRY(walloc(ctx, &wk.chunk_cnt, chunks));
        W_TRY(walloc(ctx, &wk.chunk_base, chunks));
        W_TRY(walloc(ctx, &wk.hit_mask, (size_t)wk.cap_rays / 64 + 2));
        if (hipMemsetAsync(wk.chunk_cnt, 0, chunks * sizeof(uint32_t), ctx->stream) != hipSuccess)
            return rtx_fail(ctx, RTX_ERR_HIP, "hipMemsetAsync(chunk counts) failed");
    }
    if (keys) W_TRY(walloc(ctx, &wk.pKey, wk.cap_pool));
    W_TRY(walloc(ctx, &wk.ctl, 1));
#undef W_TRY
    ctx->work_batch = batch;
    ctx->work_slots = slots;
    ctx->work_keys = keys;
    ctx->work_blob.assign((const char*)&wk, (const char*)&wk + sizeof(Work));
    *out = wk;
    return RTX_OK;
}

// Waits for the stream by polling: a blocking hipStreamSynchronize wakes up tens of
// microseconds after the stream has drained, which is a large part of a sub-millisecond frame.
// The poll gives up after kWaitSeconds (a hung kernel must not pin a host core for ever): hipErrorNotReady is then
// returned and reported as RTX_ERR_HIP by the caller.
constexpr double kWaitSeconds = 300.0;
template <typename Query>
inline hipError_t poll_until_ready(Query query) {
    std::chrono::steady_clock::time_point start;
    for (uint64_t spins = 0;; ++spins) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        if ((spins & 0xfff) == 0xfff) {
            const auto now = std::chrono::steady_clock::now();
            if (spins == 0xfff) start = now;
            else if (std::chrono::duration<double>(now - start).count() > kWaitSeconds) return hipErrorNotReady;
        }
    }
}
inline hipError_t wait_stream(hipStream_t st) {
    return poll_until_ready([st]() { return hipStreamQuery(st); });
}

inline uint32_t blocks_for(uint64_t n, uint32_t cap) {
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + RTX_BLOCK - 1) / RTX_BLOCK, cap));
}

struct EventPool {
    std::vector<hipEvent_t>* pool;
    size_t used = 0;
    hipEvent_t next() {
        if (used == pool->size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool->push_back(e);
        }
        return (*pool)[used++];
    }
};

// Remembers which share of a frame's closest-hit rays hit something (sizes the next shadow launch).
void note_hit_fraction(rtx_context* ctx, const Ctl& hc, uint64_t primary) {
    const double closest = (double)primary + (double)hc.stat[ST_REFLECT] + (double)hc.stat[ST_REFRACT] - (double)hc.stat[ST_SKIPPED];
    if (closest >= 1.0) {
        ctx->hit_fraction = (float)std::min(1.0, std::max(1.0 / 256.0, (double)hc.stat[ST_NODES] / closest));
        ctx->hit_fraction_known = true;
    }
}

// Counters and event times of a finished frame.
void fill_stats(const Ctl& hc, uint64_t primary, hipEvent_t f0, hipEvent_t f1,
                const std::vector<std::pair<hipEvent_t, hipEvent_t>>& closest_events,
                const std::vector<std::pair<hipEvent_t, hipEvent_t>>& shadow_events, uint32_t n_batches, rtx_render_stats* stats) {
    rtx_render_stats rs{};
    rs.primary_rays = primary;
    rs.shadow_rays = hc.stat[ST_SHADOW];
    rs.reflect_rays = hc.stat[ST_REFLECT];
    rs.refract_rays = hc.stat[ST_REFRACT];
    rs.skipped_rays = hc.stat[ST_SKIPPED];
    rs.shaded_nodes = hc.stat[ST_NODES];
    rs.panic_pixels = hc.stat[ST_PANIC];
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, f0, f1);
    rs.render_ms = ms;
    auto sum_ms = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs) {
        float total = 0.0f;
        for (auto& p : evs) {
            float e = 0.0f;
            if (p.first && p.second) (void)hipEventElapsedTime(&e, p.first, p.second);   // null: kernel timing is off
            total += e;
        }
        return total;
    };
    rs.closest_ms = sum_ms(closest_events);
    rs.shadow_ms = sum_ms(shadow_events);
    (void)hipGetLastError();   // a failed time query must not surface in the caller's next HIP call
    rs.closest_launches = (uint32_t)closest_events.size();
    rs.shadow_launches = (uint32_t)shadow_events.size();
    rs.batches = n_batches;
    *stats = rs;
}

// Workgroups of a traversal launch. A launch is bounded by the spill area (at most tune[2] * CUs
// workgroups) and sized so that a wavefront expects `quarter_units / 4` work units: a launch with
// few units runs faster on fewer wavefronts (every step of a ray waits less for the CU's address
// unit and caches, and the expensive-first unit order needs more than one unit per wavefront).
uint32_t trav_blocks(const rtx_context* ctx, uint64_t items_upper, double expected_items, uint32_t quarter_units, int resident,
                     int root_rule = 0) {
    // `resident`: workgroups per CU the kernel is compiled for (closest_waves / shadow_waves)
    uint32_t max_per_cu = std::min<uint32_t>((uint32_t)std::min<int32_t>(std::max<int32_t>(ctx->ds.tune[2], 1), 8), (uint32_t)resident);
    // A seventh workgroup per CU pays when every wavefront still gets a long run of units (a batch of frames); a launch of
    // one 1080p frame -- eight units per wavefront at seven workgroups -- ends later with it than without.
    if (max_per_cu > 6 && expected_items / 64.0 < 24.0 * (double)ctx->num_cus * 7.0 * (RTX_BLOCK / 64)) max_per_cu = 6;
    const double waves = expected_items / 64.0 * 4.0 / (double)std::max<uint32_t>(quarter_units, 1);
    double per_cu = std::ceil(waves / ((double)ctx->num_cus * (RTX_BLOCK / 64)));
    // Launches of one small frame: a unit takes longer the more wavefronts share the CU, and a launch lasts about (units per
    // wavefront) x (time per unit), so the best number of resident workgroups grows with the square root of the work, not
    // with the work. Measured frame by frame on the bench scene (workgroups per CU: ms per frame).
    // root_rule 1, shadow launches (units of rays that will be traced): 7 200 units (960x540): 2 0.2406, 3 0.2357;
    //   14 400 (1360x768): 3 0.308, 4 0.305, 6 0.326;  29 000 (1920x1080): 4 0.4395, 5 0.439, 6 0.4455.
    // root_rule 2, primary rays, with the rays that will hit something as the work (sky costs next to nothing) once a frame
    //   has told their share: 8 100 units (960x540): 1 0.236, 2 0.2295;  16 300: 2 0.306, 3 0.304, 4 0.320;  32 400 (1080p):
    //   2 0.460, 3 0.439, 4 0.443, 6 0.467;  57 600 (1440p): 3 0.6215, 4 0.612, 5 0.612, 6 0.622;  129 600 (4K): 4 1.108, 6 1.097.
    // Larger launches are capped by what the kernel keeps resident either way.
    const double root = std::sqrt(expected_items / 64.0 / ((double)ctx->num_cus * (RTX_BLOCK / 64)));
    if (root_rule == 1) per_cu = std::min(per_cu, std::floor(root + 0.5));
    if (root_rule == 2 && ctx->hit_fraction_known) per_cu = std::ceil(root * std::sqrt((double)ctx->hit_fraction));
    const uint32_t wgs_per_cu = (uint32_t)std::min<double>(std::max<double>(per_cu, 1.0), (double)max_per_cu);
    const uint64_t cap = (uint64_t)ctx->num_cus * wgs_per_cu;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((items_upper + RTX_BLOCK - 1) / RTX_BLOCK, cap));
}
// shade: the hits will be shaded, so they are counted per chunk and the node slots laid out (k_scan_chunks).
// t0 / t1 (optional) take the traversal kernel's own start and end times: events attached to the
// launch, because events recorded between launches each cost the stream several idle microseconds.
void launch_closest(const rtx_context* ctx, const Work& wk, int qi, uint32_t n_upper, const PrimaryGen& pg, UnitOrder* uo,
                    bool shade, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    const dim3 grid(trav_blocks(ctx, n_upper, (double)n_upper, (uint32_t)ctx->ds.tune[1] & 0xffu, closest_waves(general_kernels(ctx->ds)), pg.enabled ? 2 : 0)),
        block(RTX_BLOCK);
    uint32_t* counts = shade ? wk.chunk_cnt : nullptr;
    unsigned long long* masks = shade ? wk.hit_mask : nullptr;
    // one binary per (spheres, always-tested triangles, camera array): see primary_ray
#define RTX_LAUNCH_CLOSEST(SPH_, ALW_, CAMS_)                                                                                              \
    do {                                                                                                                                    \
        if (t0 || t1) hipExtLaunchKernelGGL((k_trace_queue<SPH_, ALW_, CAMS_>), grid, block, 0, st, t0, t1, 0, ctx->ds, wk.q[qi], wk.ctl, qi, pg, uo, counts, masks); \
        else hipLaunchKernelGGL((k_trace_queue<SPH_, ALW_, CAMS_>), grid, block, 0, st, ctx->ds, wk.q[qi], wk.ctl, qi, pg, uo, counts, masks);   \
    } while (0)
    const bool cams = pg.enabled && pg.cams != nullptr;
    if (ctx->ds.n_always) {
        if (cams) RTX_LAUNCH_CLOSEST(true, true, true); else RTX_LAUNCH_CLOSEST(true, true, false);
    } else if (ctx->ds.n_spheres) {
        if (cams) RTX_LAUNCH_CLOSEST(true, false, true); else RTX_LAUNCH_CLOSEST(true, false, false);
    } else {
        if (cams) RTX_LAUNCH_CLOSEST(false, false, true); else RTX_LAUNCH_CLOSEST(false, false, false);
    }
#undef RTX_LAUNCH_CLOSEST
    if (shade) hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(kScanThreads), 0, st, wk.ctl, qi, wk.chunk_cnt, wk.chunk_base);
}
template <bool ANY, bool AREA>
void launch_shadow_kernel(dim3 grid, hipStream_t st, hipEvent_t t0, hipEvent_t t1, const DevScene& s, const Work& wk, uint32_t depth,
                          uint32_t n_slots, UnitOrder* uo) {
    if (s.n_always) {
        if (t0 || t1) hipExtLaunchKernelGGL((k_shadow<ANY, AREA, true, true>), grid, dim3(RTX_BLOCK), 0, st, t0, t1, 0, s, wk, depth, n_slots, uo);
        else hipLaunchKernelGGL((k_shadow<ANY, AREA, true, true>), grid, dim3(RTX_BLOCK), 0, st, s, wk, depth, n_slots, uo);
    } else if (s.n_spheres) {
        if (t0 || t1) hipExtLaunchKernelGGL((k_shadow<ANY, AREA, true>), grid, dim3(RTX_BLOCK), 0, st, t0, t1, 0, s, wk, depth, n_slots, uo);
        else hipLaunchKernelGGL((k_shadow<ANY, AREA, true>), grid, dim3(RTX_BLOCK), 0, st, s, wk, depth, n_slots, uo);
    } else {
        if (t0 || t1) hipExtLaunchKernelGGL((k_shadow<ANY, AREA, false>), grid, dim3(RTX_BLOCK), 0, st, t0, t1, 0, s, wk, depth, n_slots, uo);
        else hipLaunchKernelGGL((k_shadow<ANY, AREA, false>), grid, dim3(RTX_BLOCK), 0, st, s, wk, depth, n_slots, uo);
    }
}
void launch_shadow(const rtx_context* ctx, const Work& wk, uint32_t depth, uint64_t tasks_upper, uint32_t n_slots, bool any,
                   UnitOrder* uo, hipStream_t st, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr) {
    const DevScene& s = ctx->ds;
    const bool area = s.n_area_lights != 0;
    // the share of rays that hit something is not known before the launch: the last frame's share stands in
    const dim3 grid(trav_blocks(ctx, tasks_upper, (double)tasks_upper * ctx->hit_fraction, ((uint32_t)ctx->ds.tune[1] >> 8) & 0xffu,
                                shadow_waves(any, area, general_kernels(s)), 1));
    if (any && area) launch_shadow_kernel<true, true>(grid, st, t0, t1, s, wk, depth, n_slots, uo);
    else if (any) launch_shadow_kernel<true, false>(grid, st, t0, t1, s, wk, depth, n_slots, uo);
    else if (area) launch_shadow_kernel<false, true>(grid, st, t0, t1, s, wk, depth, n_slots, uo);
    else launch_shadow_kernel<false, false>(grid, st, t0, t1, s, wk, depth, n_slots, uo);
}

// The cost / order buffer of one (kind, batch) pair of level-0 launches; nullptr when the
// option is off or the allocation fails (the launch then uses index order).
UnitOrder* unit_order(rtx_context* ctx, uint32_t kind, uint32_t batch_index, uint64_t cap_groups) {
    if (!ctx->opt_cost_order || batch_index >= 64 || cap_groups == 0 || cap_groups > (1u << 26)) return nullptr;
    const uint64_t cap_units = (cap_groups + 3u) & ~(uint64_t)3u;   // array length: order_units reads and clears costs four at a time
    const uint64_t key = ((uint64_t)kind << 40) | ((uint64_t)batch_index << 32) | cap_groups;
    for (auto& e : ctx->order_bufs)
        if (e.first == key) return (UnitOrder*)e.second;
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(UnitOrder) + 2 * cap_units * sizeof(uint32_t)) != hipSuccess) return nullptr;
    UnitOrder hdr{0u, 0u, (uint32_t)cap_units, 0u};
    if (hipMemsetAsync(p, 0, sizeof(UnitOrder) + 2 * cap_units * sizeof(uint32_t), ctx->stream) != hipSuccess ||   // costs are sums
        hipMemcpyAsync(p, &hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    ctx->order_bufs.emplace_back(key, p);
    return (UnitOrder*)p;
}

}  // namespace

uint64_t rtx_tiles_per_rank(uint32_t width, uint32_t height, uint32_t world) {
    if (world == 0) return 0;
    const uint64_t tiles = (uint64_t)((width + RTX_TILE - 1) / RTX_TILE) * ((height + RTX_TILE - 1) / RTX_TILE);
    return (tiles + world - 1) / world;
}

void rtx_clear_graphs(rtx_context* ctx) {
    ctx->level_hints.clear();   // same occasions: another scene, another tree, other options
    if (ctx->graphs.empty()) {
        ++ctx->graph_epoch;
        return;
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);   // a recorded frame may still be running
    for (auto& g : ctx->graphs) (void)hipGraphExecDestroy(g.exec);
    ctx->graphs.clear();
    ++ctx->graph_epoch;
}

namespace {
rtx_context::LevelHint* find_hint(rtx_context* ctx, const uint64_t key[4]) {
    for (auto& hnt : ctx->level_hints)
        if (std::memcmp(hnt.key, key, sizeof(hnt.key)) == 0) return &hnt;
    return nullptr;
}
// After a batch that rendered completely: what its levels held.
void note_levels(rtx_context* ctx, const uint64_t key[4], const Ctl& hc, uint32_t levels) {
    rtx_context::LevelHint* hnt = find_hint(ctx, key);
    if (!hnt) {
        if (ctx->level_hints.size() >= 64) ctx->level_hints.clear();
        ctx->level_hints.emplace_back();
        hnt = &ctx->level_hints.back();
        std::memcpy(hnt->key, key, sizeof(hnt->key));
    }
    hnt->levels_used = 0;
    for (uint32_t l = 0; l < RTX_MAX_DEPTH + 2; ++l) {
        hnt->rays[l] = hc.level_rays[l];
        if (l < levels && hc.level_rays[l] > 0) hnt->levels_used = l + 1;
    }
}
}  // namespace

int rtx_render_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t depth, uint32_t samples, uint32_t rank,
                    uint32_t world, uint8_t* d_rgb_tiles, rtx_render_stats* stats, bool direct, int async_slot,
                    uint32_t n_frames, const rtx_camera* cameras) {
    const DevScene& s = ctx->ds;
    if (n_frames == 0 || n_frames > RTX_MAX_BATCH_FRAMES)
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "a batch holds 1 to RTX_MAX_BATCH_FRAMES frames");
    rtx_context::AsyncSlot* slot = (async_slot >= 0) ? &ctx->async_slots[async_slot] : nullptr;
    hipStream_t st = ctx->stream;
    if (depth > RTX_MAX_DEPTH) {
        // The reference accepts any depth (main.rs:65). Without a reflective or refractive material nothing recurses
        // (one level of rays, lib.rs:230,241; shadow rays stop at the first hit, lib.rs:120-128), so the value is moot.
        if (ctx->has_secondary || s.has_transparent)
            return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "depth above RTX_MAX_DEPTH (32) for a scene with reflective or transparent materials");
        depth = RTX_MAX_DEPTH;
    }
    // num_samples only multiplies area-light rays (lib.rs:219-222)
    const uint32_t n_samples = s.n_area_lights ? std::max<uint32_t>(samples, 1) : 1;
    const uint64_t n_slots64 = (uint64_t)s.n_lights + (uint64_t)s.n_area_lights * (n_samples - 1);
    if (n_slots64 > 4096) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "more than 4096 shadow samples per hit");
    const uint32_t n_slots = (uint32_t)n_slots64;
    const uint32_t tiles_x = (w + RTX_TILE - 1) / RTX_TILE, tiles_y = (h + RTX_TILE - 1) / RTX_TILE;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    if (n_tiles >= (1ull << 25)) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "image too large");
    const uint64_t tiles_per_rank = rtx_tiles_per_rank(w, h, world);
    const uint64_t tiles_local = (rank < n_tiles) ? (n_tiles - rank + world - 1) / world : 0;
    // A batch of several frames gives every frame the same number of ray slots on every rank (tiles_per_rank * 64;
    // slots past a rank's own tiles hold invalid rays), so that a ray's index is also its place in the frames' slabs.
    const uint64_t rays_per_frame = (n_frames > 1) ? tiles_per_rank * 64 : tiles_local * 64;
    const uint64_t rays_local = rays_per_frame * n_frames;
    uint64_t primary = 0;   // pixels of this rank's tiles that lie inside the image
    {
        const uint32_t edge_w = w - (tiles_x - 1) * RTX_TILE, edge_h = h - (tiles_y - 1) * RTX_TILE;
        for (uint64_t tile = rank; tile < n_tiles; tile += world) {
            const uint32_t tw = (tile % tiles_x == tiles_x - 1) ? edge_w : RTX_TILE;
            const uint32_t th = (tile / tiles_x == tiles_y - 1) ? edge_h : RTX_TILE;
            primary += (uint64_t)tw * th;
        }
        primary *= n_frames;
    }
    // direct: d_rgb_tiles is a row-major [h][w][3] image (single GPU only); otherwise the rank's tile slab
    if (direct && world != 1) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "row-major output needs world == 1");
    // Without reflective or refractive materials no child is ever spawned: one level is exact.
    const uint32_t levels = ctx->has_secondary ? depth : std::min<uint32_t>(depth, 1);
    // Row-major frames are painted completely by the kernels (k_shade_a paints the misses);
    // tile slabs have padding that no ray covers, and a depth-0 frame shades nothing.
    const bool clear_first = !direct || levels == 0;
    const OutMap out_map{d_rgb_tiles, direct ? w : 0u, tiles_x, n_frames, (uint32_t)rays_per_frame, w * h};
    if (!ctx->pinned) {
        void* p = nullptr;
        RTX_HIP(ctx, hipHostMalloc(&p, 8 * 4096, hipHostMallocDefault));   // [0] counters, [1] next-level count, [2..3] counters of the frames in flight, [4..6] cameras of a batch
        ctx->pinned = p;
    }
    static_assert(sizeof(CamSetup) * RTX_MAX_BATCH_FRAMES <= 4096, "a batch's cameras must fit one pinned page");
    Ctl* host_ctl = (Ctl*)((char*)ctx->pinned + (slot ? (size_t)(2 + async_slot) * 4096 : 0));
    uint32_t* host_next = (uint32_t*)((char*)ctx->pinned + 4096);
    static_assert(sizeof(Ctl) <= 4096, "Ctl must fit the pinned staging area");

    uint64_t batch = std::min<uint64_t>(std::max<int64_t>(ctx->opt_batch_pixels, 64), std::max<uint64_t>(rays_local, 64));
    batch = (batch + 63) & ~63ull;
    // shadow tasks of one level (rays x slots) are indexed with 32 bits; this also keeps the
    // shadow-term array of a many-sample frame near 1 GiB
    batch = std::max<uint64_t>(64, std::min<uint64_t>(batch, ((1ull << 27) / std::max<uint32_t>(n_slots, 1)) & ~63ull));
    Work wk;
    int rc = ensure_workspace(ctx, batch, n_slots, &wk);
    if (rc != RTX_OK) return rc;
    wk.n_samples = n_samples;
    wk.area_off = (s.n_area_lights && samples == 0) ? 1u : 0u;
    if (ctx->shade_a_resident == 0) {   // workgroups of k_shade_a a CU keeps resident: asked once, before anything is recorded
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shade_a, RTX_BLOCK, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        (void)hipGetLastError();
        ctx->shade_a_resident = (uint32_t)per_cu;
    }

    // One frame: its camera is an argument of the level-0 kernels (the scene's, or the caller's). A batch of frames: the
    // cameras travel through a pinned page into a small device array that the level-0 kernels read (one page per frame
    // in flight). A RECORDED launch sequence (below) bakes kernel arguments in, so a frame that can be recorded takes
    // the array as well when the caller hands in a camera: a camera path then replays one recording instead of
    // recording every frame.
    const bool single_level = levels == 1;   // k_shade_b writes the pixels, no backward pass
    const bool graphable = slot && ctx->opt_graphs && !ctx->opt_kernel_timing && ctx->hit_fraction_known && single_level && rays_local > 0 &&
                           rays_local <= batch;
    const CamSetup cam = make_cam((cameras && n_frames == 1) ? cameras[0] : ctx->camera, w, h);
    const CamSetup* d_cams = nullptr;
    if (n_frames > 1 && (rays_local > batch || rays_local >= (1ull << 32)))
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "the frames of a batch must fit one ray batch (fewer frames, or a larger batch_pixels)");
    if (n_frames > 1 || (cameras && graphable)) {
        if (!ctx->cam_dev) RTX_HIP(ctx, hipMalloc(&ctx->cam_dev, 3 * 4096));
        const int page = slot ? 1 + async_slot : 0;
        CamSetup* host_cams = (CamSetup*)((char*)ctx->pinned + (size_t)(4 + page) * 4096);
        // (page 0 belongs to the synchronous calls, which end with a wait for the stream: its last copy has run)
        for (uint32_t f = 0; f < n_frames; ++f) host_cams[f] = make_cam(cameras ? cameras[f] : ctx->camera, w, h);
        d_cams = (const CamSetup*)((char*)ctx->cam_dev + (size_t)page * 4096);
    }
    const bool any_shadow = !s.has_transparent;
    EventPool ev{slot ? &slot->events : &ctx->events};
    hipEvent_t f0 = ev.next(), f1 = ev.next();
    if (!f0 || !f1) return rtx_fail(ctx, RTX_ERR_HIP, "hipEventCreate failed");
    std::vector<std::pair<hipEvent_t, hipEvent_t>> closest_events, shadow_events;

    // A frame that needs no host decision between its launches (one batch, one level, async entry point) is
    // recorded once as a graph and replayed afterwards.
    // (not with kernel timing: events recorded by a graph cannot be read back as times)
    // (and not before a frame has told how large the shadow launch should be)
    rtx_context::FrameGraph* fg = nullptr;
    struct CaptureGuard {   // a stream must not be left in capture mode on an error path
        hipStream_t st;
        bool active = false;
        ~CaptureGuard() {
            if (active) {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(st, &g);
                if (g) (void)hipGraphDestroy(g);
            }
        }
    } capture{st};
    rtx_context::FrameGraph fresh{};
    if (graphable) {
        const uint32_t n0 = (uint32_t)rays_local;
        // buffers the launches use are created before recording
        (void)unit_order(ctx, 0, 0, (wk.cap_rays / 64 + kGroup) / kGroup);
        if (s.n_lights > 0) (void)unit_order(ctx, 1, 0, ((uint64_t)wk.cap_rays * n_slots / 64 + kGroup) / kGroup);
        const uint64_t key[8] = {((uint64_t)w << 32) | h, ((uint64_t)depth << 32) | samples, ((uint64_t)rank << 32) | world,
                                 (uint64_t)(uintptr_t)d_rgb_tiles, ((uint64_t)(direct ? 1 : 0) << 32) | (uint32_t)async_slot,
                                 ((uint64_t)trav_blocks(ctx, n0, (double)n0, (uint32_t)ctx->ds.tune[1] & 0xffu, closest_waves(general_kernels(s)), 2) << 32) |
                                     trav_blocks(ctx, (uint64_t)n0 * n_slots, (double)n0 * n_slots * ctx->hit_fraction, ((uint32_t)ctx->ds.tune[1] >> 8) & 0xffu,
                                                 shadow_waves(any_shadow, s.n_area_lights != 0, general_kernels(s)), 1),
                                 (uint64_t)(uintptr_t)wk.ctl ^ (d_cams ? 0x9e3779b97f4a7c15ull : 0ull), (uint64_t)ctx->graph_epoch ^ ((uint64_t)n_frames << 56)};
        for (auto& g : ctx->graphs)
            if (std::memcmp(g.key, key, sizeof(key)) == 0) fg = &g;
        if (fg) {
            RTX_HIP(ctx, hipEventRecord(f0, st));
            RTX_HIP(ctx, hipGraphLaunch(fg->exec, st));
            RTX_HIP(ctx, hipEventRecord(f1, st));
            slot->deferred = true;
            slot->f0 = f0;
            slot->f1 = f1;
            slot->a = slot->b = slot->c = slot->d = nullptr;
            slot->has_shadow = fg->has_shadow;
            slot->primary = primary;
            return RTX_OK;
        }
        std::memcpy(fresh.key, key, sizeof(key));
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            capture.active = true;
        } else {   // no graphs on this context from now on
            (void)hipGetLastError();
            ctx->opt_graphs = 0;
        }
    }
    const bool recording = capture.active;
    if (!recording) RTX_HIP(ctx, hipEventRecord(f0, st));
    if (clear_first)
        RTX_HIP(ctx, hipMemsetAsync(d_rgb_tiles, 0, (direct ? (size_t)w * h * 3 : (size_t)tiles_per_rank * 64 * 3) * n_frames, st));
    // The cameras reach the device array inside the first kernel's arguments when they are few and the launches are
    // not being recorded (a recording must read the cameras of the batch it is replayed for: a copy from the pinned
    // page, whose contents are those of replay time); otherwise as that copy.
    const bool cams_by_value = d_cams && n_frames <= kCamsByValue && !graphable;
    CamBlock cam_block{};
    if (cams_by_value) {
        const int page = slot ? 1 + async_slot : 0;
        std::memcpy(cam_block.cam, (char*)ctx->pinned + (size_t)(4 + page) * 4096, sizeof(CamSetup) * n_frames);
    } else if (d_cams) {
        const int page = slot ? 1 + async_slot : 0;
        RTX_HIP(ctx, hipMemcpyAsync((void*)d_cams, (char*)ctx->pinned + (size_t)(4 + page) * 4096, sizeof(CamSetup) * n_frames, hipMemcpyHostToDevice, st));
    }
    uint32_t n_batches = 0;
    uint64_t cur_batch = batch;
    int rollback = 0;
    bool no_hints = !ctx->opt_level_hints || ctx->render_without_hints;   // (the second: a frame being rendered again after a miss)
    ctx->render_without_hints = false;
    for (uint64_t first = 0; first < rays_local;) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(cur_batch, rays_local - first);
        ++n_batches;
        // Several levels without a word from the device in between, when the last frame of this shape left a hint.
        const uint64_t hint_key[4] = {((uint64_t)w << 32) | h, ((uint64_t)depth << 32) | samples,
                                      ((uint64_t)rank << 48) | ((uint64_t)world << 32) | n_frames, first};
        const rtx_context::LevelHint* hint = (levels > 1 && !no_hints && cur_batch == batch) ? find_hint(ctx, hint_key) : nullptr;
        const uint32_t levels_enq = hint ? std::min<uint32_t>(levels, hint->levels_used + 1) : levels;
        auto hinted_rays = [&](uint32_t level) {   // launch sizing only: the kernels go by the device's own counts
            const uint64_t r = (uint64_t)hint->rays[level] + hint->rays[level] / 4 + 8192;
            return (uint32_t)std::min<uint64_t>(r, wk.cap_rays);
        };
        hipLaunchKernelGGL(k_batch_init, dim3(1), dim3(1), 0, st, wk.ctl, n, rollback, n_batches == 1 ? 1 : 0, (CamSetup*)d_cams,
                           (cams_by_value && n_batches == 1) ? n_frames : 0u, cam_block);
        rollback = 0;
        const PrimaryGen gen{cam, first, rank, world, tiles_x, (uint32_t)n_tiles, 1u, n_frames, (uint32_t)rays_per_frame, d_cams};   // level 0: rays come from their index
        PrimaryGen no_gen = gen;
        no_gen.enabled = 0;
        int qi = 0;
        uint32_t n_rays = n, levels_done = 0;
        if (levels == 0)   // depth 0: primary rays are traced, every pixel stays black (lib.rs:203-205)
            launch_closest(ctx, wk, 0, n_rays, gen, nullptr, false, st);
        // k_shade_a walks its chunks in a loop with a workgroup-wide wait per chunk, so its time goes with
        // (rounds of resident workgroups) x (chunks per workgroup). Measured on the eight-frame batch (16 200 chunks, 5
        // workgroups per CU resident): 187 / 167 / 151 / 195 / 168 us with 3 / 4 / 5 / 6 / 8 workgroups per CU and 166 with
        // 4096 workgroups -- one round wins when a workgroup has many chunks to even out; with 6 chunks per workgroup (a
        // four-frame batch) the 4096 win, because workgroups that start as others end balance the load. k_shade_b, whose
        // passes are independent, keeps the 4096 (one round of resident workgroups measured 195 us against 141).
        const uint32_t shade_resident_a = (uint32_t)ctx->num_cus * ctx->shade_a_resident, shade_cap_b = 4096u;
        for (uint32_t level = 0; level < levels_enq; ++level) {
            const bool no_kt = !ctx->opt_kernel_timing;   // closest_ms / shadow_ms stay 0
            hipEvent_t a = no_kt ? nullptr : ev.next(), b = no_kt ? nullptr : ev.next();
            if (!no_kt && (!a || !b)) return rtx_fail(ctx, RTX_ERR_HIP, "hipEventCreate failed");
            const PrimaryGen& pg = (level == 0) ? gen : no_gen;
            // level-0 launches of unchanged size reuse the unit costs of the previous frame (see UnitOrder)
            const bool ordered = level == 0 && cur_batch == batch;
            UnitOrder* uo0 = ordered ? unit_order(ctx, 0, n_batches - 1, (wk.cap_rays / 64 + kGroup) / kGroup) : nullptr;
            UnitOrder* uo1 = nullptr;
            launch_closest(ctx, wk, qi, n_rays, pg, uo0, true, st, a, b);
            closest_events.emplace_back(a, b);
            const uint64_t shade_chunks = ((uint64_t)n_rays + RTX_BLOCK * kShadeItems - 1) / (RTX_BLOCK * kShadeItems);
            const uint32_t shade_cap_a = (shade_chunks >= 8ull * shade_resident_a) ? shade_resident_a : 4096u;
            hipLaunchKernelGGL(k_shade_a, dim3(blocks_for((n_rays + kShadeItems - 1) / kShadeItems, shade_cap_a) + (uo0 ? 1u : 0u)), dim3(RTX_BLOCK), 0, st, s, wk, qi,
                               PixMap{level == 0 ? 1u : 0u, rank, world, tiles_x, w, n_frames, (uint32_t)rays_per_frame}, pg, out_map, uo0);
            if (s.n_lights > 0) {
                hipEvent_t c = no_kt ? nullptr : ev.next(), d = no_kt ? nullptr : ev.next();
                if (!no_kt && (!c || !d)) return rtx_fail(ctx, RTX_ERR_HIP, "hipEventCreate failed");
                const uint64_t tasks = (uint64_t)n_rays * n_slots;
                uo1 = ordered ? unit_order(ctx, 1, n_batches - 1, ((uint64_t)wk.cap_rays * n_slots / 64 + kGroup) / kGroup) : nullptr;
                launch_shadow(ctx, wk, depth - level, tasks, n_slots, any_shadow, uo1, st, c, d);
                shadow_events.emplace_back(c, d);
            }
            hipLaunchKernelGGL(k_shade_b, dim3(blocks_for(n_rays, shade_cap_b) + (uo1 ? 1u : 0u)), dim3(RTX_BLOCK), 0, st, s, wk, qi ^ 1, depth - level, samples,
                               out_map, single_level ? 1 : 0, uo1);
            if (single_level) break;
            hipLaunchKernelGGL(k_next_level, dim3(1), dim3(1), 0, st, wk.ctl, qi, level, wk.cap_rays,
                               (hint && level + 1 == levels_enq && levels_enq < levels) ? 1 : 0);
            levels_done = level + 1;
            if (hint) {   // no question asked: the next level is sized from the last frame's count
                if (level + 1 < levels_enq) {
                    n_rays = hinted_rays(level + 1);
                    qi ^= 1;
                }
            } else if (level + 1 < levels) {
                RTX_HIP(ctx, hipMemcpyAsync(host_next, &wk.ctl->n_rays[qi ^ 1], 4, hipMemcpyDeviceToHost, st));
                RTX_HIP(ctx, wait_stream(st));
                const uint32_t next = *host_next;
                if (next == 0) break;
                n_rays = std::min(next, wk.cap_rays);
                qi ^= 1;
            }
        }
        for (uint32_t level = levels_done; level-- > 0;) {
            const uint64_t upper = (level == 0) ? n : wk.cap_rays;
            hipLaunchKernelGGL(k_combine, dim3(blocks_for(upper, 4096)), dim3(RTX_BLOCK), 0, st, s, wk, level, out_map);
        }
        // one read-back and one synchronisation per batch: counters, overflow flag (and, after
        // the last batch, the frame-end event)
        const bool last_batch = first + n >= rays_local;
        RTX_HIP(ctx, hipMemcpyAsync(host_ctl, wk.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, st));
        if (recording) {   // close the recording, keep it, and run it as this frame
            hipGraph_t graph = nullptr;
            capture.active = false;
            hipError_t ge = hipStreamEndCapture(st, &graph);
            if (ge == hipSuccess) ge = hipGraphInstantiate(&fresh.exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (ge == hipSuccess) {
                RTX_HIP(ctx, hipEventRecord(f0, st));
                ge = hipGraphLaunch(fresh.exec, st);
            }
            if (ge != hipSuccess) {   // render the frame the plain way, now and from now on
                (void)hipGetLastError();
                if (fresh.exec) (void)hipGraphExecDestroy(fresh.exec);
                ctx->opt_graphs = 0;
                return rtx_render_impl(ctx, w, h, depth, samples, rank, world, d_rgb_tiles, stats, direct, async_slot, n_frames, cameras);
            }
            fresh.has_shadow = !shadow_events.empty();
            if (ctx->graphs.size() >= 32) rtx_clear_graphs(ctx);   // a caller cycling through many buffers: start over
            ctx->graphs.push_back(fresh);
        }
        if (last_batch) RTX_HIP(ctx, hipEventRecord(f1, st));
        if (slot && (single_level || hint) && n_batches == 1 && last_batch) {
            // Nothing in this frame needs the host: leave it in flight, rtx_render_wait collects it.
            slot->deferred = true;
            slot->f0 = f0;
            slot->f1 = f1;
            slot->a = closest_events[0].first;
            slot->b = closest_events[0].second;
            slot->has_shadow = !shadow_events.empty();
            if (slot->has_shadow) {
                slot->c = shadow_events[0].first;
                slot->d = shadow_events[0].second;
            }
            slot->primary = primary;
            slot->multi_level = !single_level;
            if (slot->multi_level) {
                slot->closest_events = closest_events;
                slot->shadow_events = shadow_events;
                std::memcpy(slot->hint_key, hint_key, sizeof(hint_key));
                slot->levels = levels;
            }
            return RTX_OK;
        }
        RTX_HIP(ctx, wait_stream(st));
        if (host_ctl->spec_miss && !host_ctl->overflow) {
            // The ray tree is deeper than the last frame's: the same rays again, level by level (counters rolled back).
            no_hints = true;
            rollback = 1;
            --n_batches;
            closest_events.clear();
            shadow_events.clear();
            ev.used = 2;   // f0 and f1 stay
            continue;
        }
        if (!host_ctl->overflow && levels > 1 && cur_batch == batch) note_levels(ctx, hint_key, *host_ctl, levels);
        if (host_ctl->overflow) {
            // The ray tree of this batch outgrew the queues: render the same pixels again in
            // smaller batches (the workspace keeps its size, so relative capacity doubles).
            if (cur_batch <= 64)
                return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "ray queues overflow even for one tile; raise the pool_factor option");
            if (n_frames > 1) {
                // A batch is one pass by construction (a ray's index is its place in the frames' slabs), so it cannot be
                // halved; its frames can be rendered one after the other, though, each in as many passes as it needs.
                rtx_render_stats total{};
                const size_t frame_bytes = direct ? (size_t)w * h * 3 : (size_t)tiles_per_rank * 64 * 3;
                for (uint32_t fr = 0; fr < n_frames; ++fr) {
                    rtx_render_stats one{};
                    const rtx_camera cam = cameras ? cameras[fr] : ctx->camera;
                    const int rc = rtx_render_impl(ctx, w, h, depth, samples, rank, world, d_rgb_tiles + fr * frame_bytes, &one, direct, -1, 1, &cam);
                    if (rc != RTX_OK) return rc;
                    total.primary_rays += one.primary_rays; total.shadow_rays += one.shadow_rays; total.reflect_rays += one.reflect_rays;
                    total.refract_rays += one.refract_rays; total.skipped_rays += one.skipped_rays; total.shaded_nodes += one.shaded_nodes;
                    total.panic_pixels += one.panic_pixels; total.render_ms += one.render_ms; total.closest_ms += one.closest_ms;
                    total.shadow_ms += one.shadow_ms; total.closest_launches += one.closest_launches; total.shadow_launches += one.shadow_launches;
                    total.batches += one.batches;
                }
                if (stats) *stats = total;
                return RTX_OK;
            }
            cur_batch = std::max<uint64_t>(64, (cur_batch / 2 + 63) & ~63ull);
            rollback = 1;   // k_batch_init restores the ray counters of the abandoned attempt
            continue;
        }
        first += n;
    }
    if (rays_local == 0) {
        RTX_HIP(ctx, hipEventRecord(f1, st));
        RTX_HIP(ctx, hipStreamSynchronize(st));
        std::memset(host_ctl, 0, sizeof(Ctl));
    }
    RTX_HIP(ctx, hipGetLastError());
    note_hit_fraction(ctx, *host_ctl, primary);
    if (stats) fill_stats(*host_ctl, primary, f0, f1, closest_events, shadow_events, n_batches, stats);
    return RTX_OK;
}

// Enqueues a frame; at most two frames may be waiting to be collected.
int rtx_render_async_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t depth, uint32_t samples, uint32_t rank,
                          uint32_t world, uint8_t* d_rgb_tiles, bool direct, uint32_t n_frames, const rtx_camera* cameras) {
    if (ctx->async_count >= 2) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "two frames are already in flight: collect one with rtx_render_wait");
    const int idx = (int)((ctx->async_head + ctx->async_count) & 1u);
    rtx_context::AsyncSlot& slot = ctx->async_slots[idx];
    slot.used = true;
    slot.deferred = false;
    slot.w = w; slot.h = h; slot.depth = depth; slot.samples = samples; slot.rank = rank; slot.world = world;
    slot.out = d_rgb_tiles;
    slot.direct = direct;
    slot.n_frames = n_frames;
    slot.cameras.assign(cameras ? cameras : nullptr, cameras ? cameras + n_frames : nullptr);
    const int rc = rtx_render_impl(ctx, w, h, depth, samples, rank, world, d_rgb_tiles, &slot.stats, direct, idx, n_frames, cameras);
    if (rc != RTX_OK) {
        slot.used = false;
        return rc;
    }
    ++ctx->async_count;
    return RTX_OK;
}

// Collects the oldest frame in flight.
int rtx_render_wait_impl(rtx_context* ctx, rtx_render_stats* stats) {
    if (ctx->async_count == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "no frame is in flight");
    const int idx = (int)(ctx->async_head & 1u);
    rtx_context::AsyncSlot& slot = ctx->async_slots[idx];
    ctx->async_head ^= 1u;
    --ctx->async_count;
    slot.used = false;
    if (slot.deferred) {
        {   // poll the frame's own end event: a later frame may already be queued behind it
            hipEvent_t done = slot.f1;
            const hipError_t e = poll_until_ready([done]() { return hipEventQuery(done); });
            if (e != hipSuccess)
                return rtx_fail(ctx, RTX_ERR_HIP, e == hipErrorNotReady ? std::string("the frame did not finish within the wait limit")
                                                                         : std::string("hipEventQuery: ") + hipGetErrorString(e));
        }
        const Ctl* hc = (const Ctl*)((char*)ctx->pinned + (size_t)(2 + idx) * 4096);
        if (hc->overflow || hc->spec_miss) {
            // rare: the frame's ray tree outgrew the queues, or was deeper than the hint it was enqueued with; render it
            // again the patient way (a later frame may be in flight behind it on the same stream: that one keeps its
            // own results, the stream is in order)
            ctx->render_without_hints = true;
            return rtx_render_impl(ctx, slot.w, slot.h, slot.depth, slot.samples, slot.rank, slot.world, slot.out, stats, slot.direct, -1,
                                   slot.n_frames, slot.cameras.empty() ? nullptr : slot.cameras.data());
        }
        note_hit_fraction(ctx, *hc, slot.primary);
        if (slot.multi_level) {
            note_levels(ctx, slot.hint_key, *hc, slot.levels);
            if (stats) fill_stats(*hc, slot.primary, slot.f0, slot.f1, slot.closest_events, slot.shadow_events, 1, stats);
            return RTX_OK;
        }
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ce{{slot.a, slot.b}}, se;
        if (slot.has_shadow) se.emplace_back(slot.c, slot.d);
        if (stats) fill_stats(*hc, slot.primary, slot.f0, slot.f1, ce, se, 1, stats);
        return RTX_OK;
    }
    if (stats) *stats = slot.stats;
    return RTX_OK;
}

int rtx_primary_impl(rtx_context* ctx, uint32_t w, uint32_t h, int32_t* d_prim, float* d_t, float* d_dir) {
    hipStream_t st = ctx->stream;
    const uint32_t tiles_x = (w + RTX_TILE - 1) / RTX_TILE, tiles_y = (h + RTX_TILE - 1) / RTX_TILE;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    if (n_tiles >= (1ull << 25)) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "image too large");
    const uint64_t rays = n_tiles * 64;
    uint64_t batch = std::min<uint64_t>(std::max<int64_t>(ctx->opt_batch_pixels, 64), std::max<uint64_t>(rays, 64));
    batch = (batch + 63) & ~63ull;
    Work wk;
    int rc = ensure_workspace(ctx, batch, ctx->ds.n_lights, &wk);
    if (rc != RTX_OK) return rc;
    const CamSetup cam = make_cam(ctx->camera, w, h);
    for (uint64_t first = 0; first < rays; first += batch) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(batch, rays - first);
        hipLaunchKernelGGL(k_batch_init, dim3(1), dim3(1), 0, st, wk.ctl, n, 0, 1, (CamSetup*)nullptr, 0u, CamBlock{});
        const PrimaryGen gen{cam, first, 0u, 1u, tiles_x, (uint32_t)n_tiles, 1u};
        hipLaunchKernelGGL(k_raygen, dim3(blocks_for(n, 1u << 22)), dim3(RTX_BLOCK), 0, st, gen, wk, n);   // k_primary_out reads the directions
        launch_closest(ctx, wk, 0, n, gen, nullptr, false, st);
        hipLaunchKernelGGL(k_primary_out, dim3(blocks_for(n, 1u << 22)), dim3(RTX_BLOCK), 0, st, wk.q[0], first, n, w, h, tiles_x,
                           d_prim, d_t, d_dir);
    }
    RTX_HIP(ctx, hipStreamSynchronize(st));
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

int rtx_assemble_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t world, const uint8_t* d_slabs, uint8_t* d_rgb, uint32_t n_frames) {
    const uint32_t tiles_x = (w + RTX_TILE - 1) / RTX_TILE;
    hipLaunchKernelGGL(k_assemble, dim3(blocks_for((uint64_t)w * h * n_frames, 8192)), dim3(RTX_BLOCK), 0, ctx->stream, w, h, world, tiles_x,
                       rtx_tiles_per_rank(w, h, world), n_frames, d_slabs, d_rgb);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

#ifdef RTX_DIAG
// Diagnostic build: reads and clears the lane-occupancy counters of rtx_coop.h (2 x 32 values: closest, any-hit).
extern "C" int rtx_diag_counters(unsigned long long* out) {
    unsigned long long zero[2][32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), sizeof(zero)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_diag), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif
