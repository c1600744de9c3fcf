// rtx_persist.h -- persistent-wavefront BVH traversal.
//
// A fixed grid of wavefronts stays resident and pulls work from a queue:
//   * each wavefront reserves kChunk consecutive items with one atomic and hands
//     them to its lanes as they fall idle, so a lane that finishes early does
//     not wait for the slowest ray of its original group of 64 (ballot +
//     popcount give every idle lane its rank in the refill);
//   * traversal alternates a node phase (every lane that holds an inner node
//     steps until no lane in the wavefront does) and a leaf phase (lanes test
//     the primitives of the leaf they reached), which keeps lanes in the same
//     instruction stream instead of interleaving box and primitive tests;
//   * per-lane stack in LDS with a global spill area (see TravStack).
// What is computed per ray is unchanged (see rtx_traverse.h): the same primitive
// tests, the same order-independent closest-hit rule, conservative boxes.
#pragma once
#include "rtx_traverse.h"

constexpr int kTravDone = 0x7fffffff;   // lane finished its ray, result not yet handed to the job
constexpr int kTravIdle = 0x7ffffffe;   // lane has no ray
constexpr uint32_t kChunk = 512;        // items reserved per atomic
constexpr int kRefillMin = 20;          // refill when at least this many lanes are idle

RTX_D bool trav_is_inner(int cur) { return (uint32_t)cur < 0x7ffffff0u; }

RTX_D uint64_t lanemask_lt64() {
    const int lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

typedef RayBox RayBoxF;
#define make_rayboxf make_raybox
#define box_hit_f box_hit

// Per-lane traversal state.
struct Trav {
    V3 o, d;
    RayBoxF rb;
    float best_t;     // closest: best t so far; any-hit: the culling limit
    int best_id;      // closest: best id; any-hit: 1 once an occluder was found
    float dist_light; // any-hit only
    int cur, sp;
};

template <bool ANY>
RTX_D void trav_begin(const DevScene& s, Trav& tr, V3 o, V3 d, float dist_light) {
    tr.o = o;
    tr.d = d;
    tr.sp = 0;
    tr.dist_light = dist_light;
    if (ANY) {
        tr.best_id = 0;
        tr.best_t = __builtin_inff();
        if (dist_light >= 0.0f) {
            // hits beyond the light never occlude; vdistance(o, o + d*t) carries an absolute error of a few ulp of |o|
            const float len = vlength(d);
            const float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
            if (len > 0.0f) tr.best_t = ((dist_light + 4e-6f * m) / len) * 1.0001f + 1e-6f;
        }
    } else {
        tr.best_t = __builtin_inff();
        tr.best_id = 0x7fffffff;
    }
    if (!s.uses_bvh || !in_safe_region(s, o)) {
        // every primitive, as the reference loop does
        if (ANY) {
            for (uint32_t i = 0; i < s.n_prims; ++i)
                if (any_record(s.recs, i, o, d, dist_light)) { tr.best_id = 1; break; }
        } else {
            for (uint32_t i = 0; i < s.n_prims; ++i) test_record(s.recs, i, o, d, tr.best_t, tr.best_id);
        }
        tr.cur = kTravDone;
        return;
    }
    tr.rb = make_rayboxf(s, o, d);
    tr.cur = 0;
}

RTX_D void trav_node_step(const BvhNode* __restrict__ nodes, Trav& tr, TravStack& st) {
    tr.cur = node4_step(nodes[tr.cur].a, nodes[tr.cur].b, nodes[tr.cur].c, nodes[tr.cur].d, tr.rb, tr.o, tr.best_t, st, tr.sp, 0, kTravDone);
}

template <bool ANY>
RTX_D void trav_leaf_step(const PrimRec* __restrict__ recs, Trav& tr, TravStack& st) {
    const uint32_t leaf = ~(uint32_t)tr.cur;
    const uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
    bool stop = false;
    for (uint32_t k = 0; k < cnt; ++k) {
        const float4 q0 = recs[first + k].q0, q1 = recs[first + k].q1, q2 = recs[first + k].q2;
        float t;
        bool hit;
        if (f2u(q1.w) == 0u) hit = tri_intersect(tr.o, tr.d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), &t);
        else hit = sphere_intersect(tr.o, tr.d, v3(q0.x, q0.y, q0.z), q1.x, &t);
        if (ANY) {
            if (hit && occluding_hit(tr.o, tr.d, t, tr.dist_light)) { stop = true; break; }
        } else {
            const int id = (int)f2u(q0.w);
            if (hit && (t < tr.best_t || (t == tr.best_t && id < tr.best_id))) {
                tr.best_t = t;
                tr.best_id = id;
            }
        }
    }
    if (ANY && stop) {
        tr.best_id = 1;
        tr.cur = kTravDone;
        return;
    }
    tr.cur = (tr.sp > 0) ? st.get(--tr.sp) : kTravDone;
}

// The engine. Job provides, per lane:
//   bool start(uint32_t item, V3& o, V3& d, float& dist_light)
//        false: the item needs no ray (the job has already stored its result)
//   bool finish(const Trav& tr, V3& o, V3& d, float& dist_light)
//        called when the ray is done; true: continue this lane with the new ray in o/d
template <bool ANY, class Job>
RTX_D void persistent_trace(const DevScene& s, Job& job, TravStack& st, uint32_t n_items, uint32_t* head, uint32_t first_dynamic) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt64();
    // wave-uniform: the first chunk is assigned statically (wave w takes chunk w), so a launch
    // does not begin with thousands of atomics on one word
    const uint32_t wave_id = (blockIdx.x * RTX_BLOCK + threadIdx.x) >> 6;
    uint32_t chunk_next = min(wave_id * kChunk, n_items), chunk_end = min(chunk_next + kChunk, n_items);
    bool more = true;                         // wave-uniform: the queue may still hold items
    const int refill_min = s.tune[0], alpha16 = s.tune[1];
    Trav tr;
    tr.cur = kTravIdle;
    tr.sp = 0;
    for (;;) {
        uint64_t idle = __ballot(tr.cur == kTravIdle);
        if (more && __popcll(idle) >= refill_min) {
            for (;;) {
                idle = __ballot(tr.cur == kTravIdle);
                const uint32_t need = (uint32_t)__popcll(idle);
                if (need == 0) break;
                if (chunk_next == chunk_end) {
                    // skip the atomic when the counter already shows an empty queue (cheap L2 read)
                    uint32_t base = first_dynamic + __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (base < n_items) {
                        if (lane == 0) base = first_dynamic + atomicAdd(head, kChunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    }
                    if (base >= n_items) {
                        more = false;
                        break;
                    }
                    chunk_next = base;
                    chunk_end = min(base + kChunk, n_items);
                }
                const uint32_t take = min(need, chunk_end - chunk_next);
                const uint32_t rank = (uint32_t)__popcll(idle & lt);
                if (tr.cur == kTravIdle && rank < take) {
                    V3 o, d;
                    float dl = -1.0f;
                    if (job.start(chunk_next + rank, o, d, dl)) trav_begin<ANY>(s, tr, o, d, dl);
                }
                chunk_next += take;
            }
        }
        const uint64_t inner = __ballot(trav_is_inner(tr.cur));
        const uint64_t leaf = __ballot(tr.cur < 0);
        const int ni = __popcll(inner), nl = __popcll(leaf);
        if (ni == 0 && nl == 0 && __ballot(tr.cur == kTravDone) == 0) {
            if (!more) break;
            continue;
        }
        // Phase vote: box steps are cheap and leaf steps expensive, so lanes that reached a
        // leaf wait until enough of the wavefront has one too (alpha16 = 0: until all have).
        if (ni != 0 && ni * 16 >= nl * alpha16) {
            if (trav_is_inner(tr.cur)) trav_node_step(s.nodes, tr, st);
        } else if (nl != 0) {
            if (tr.cur < 0) trav_leaf_step<ANY>(s.recs, tr, st);
        }
        // finished rays
        if (tr.cur == kTravDone) {
            V3 o, d;
            float dl = -1.0f;
            if (job.finish(tr, o, d, dl)) trav_begin<ANY>(s, tr, o, d, dl);
            else tr.cur = kTravIdle;
        }
    }
}
