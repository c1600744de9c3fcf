// rtx_internal.h -- device-side data layout and the context behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_math.h"

// ---------------------------------------------------------------------------
// Layout in HBM
//
// PrimRec (48 B, three 16-B quads): what the traversal kernels fetch per
// candidate primitive. One record is one contiguous 48-B read per lane, so a
// leaf of k primitives is k*48 contiguous bytes.
//   triangle: q0 = (v0.x, v0.y, v0.z, bits(prim_id))
//             q1 = (e1.x, e1.y, e1.z, bits(0))      e1 = v1 - v0
//             q2 = (e2.x, e2.y, e2.z, 0)            e2 = v2 - v0
//   sphere:   q0 = (c.x,  c.y,  c.z,  bits(prim_id))
//             q1 = (r, 0, 0, bits(1))
// In BVH mode records are stored in Morton order (leaf ranges are contiguous);
// in exhaustive mode in primitive-id order.
//
// BvhNode (64 B, four quads): a 4-wide node with quantised child boxes; one traversal
// step covers two levels of the binary tree for one 64-byte fetch.
//   a = (bits(origin.x), bits(origin.y), bits(origin.z), ex | ey<<8 | ez<<16 | n_children<<24 | child_spheres<<27 | spheres<<31)
//       origin = lower corner of the node's box; e* = biased f32 exponent of the per-axis
//       grid step (step = 2^(e-127)); n_children = 2..4, children are packed first;
//       spheres = 1 when some child's subtree holds a sphere; child_spheres: bit k = child k's does
//   b = (qlo_x[4], qlo_y[4], qlo_z[4], qhi_x[4])   one byte per child: plane = origin + q * step
//   c = (qhi_y[4], qhi_z[4], ref[0], ref[1])
//   d = (ref[2], ref[3], rmin[0] | rmin[1]<<16, rmin[2] | rmin[3]<<16)
//       rmin = smallest sphere radius below the child as f16 rounded down (0x7c00 = no sphere):
//       sphere children are widened per ray at traversal time (kSphereK, rtx_traverse.h)
// qlo is rounded down and qhi up, so a child's quantised box contains its padded box.
// ref >= 0: index of another BvhNode; ref < 0: ~ref = (first << 3) | (count - 1),
// a leaf of `count` consecutive PrimRecs starting at `first`; count is always 1 (RTX_LEAF_MAX), the traversal reads `first` only.
// ---------------------------------------------------------------------------
struct PrimRec {
    float4 q0, q1, q2;
};
struct BvhNode {
    uint4 a, b, c, d;
};
static_assert(sizeof(PrimRec) == 48, "PrimRec must be 48 bytes");
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

#define RTX_LEAF_MAX 1          // primitives per leaf: the fetch of a leaf reference IS its one record (leaves of 2-8 records behind a loop were slower)
#define RTX_EXHAUSTIVE_MAX 16   // scenes with at most this many primitives skip the BVH
#define RTX_TILE 8              // primary rays are generated in 8x8-pixel tiles (one wavefront)

struct DevLight {  // 48 B
    float a[3];
    uint32_t type;
    float b[3];
    uint32_t area_before;   // number of area lights with a smaller index (shadow-term slots, see Work::shadow)
    float color[3];
    float pad1;
};

// Everything a kernel needs to know about the scene; passed by value.
struct DevScene {
    uint32_t n_prims, n_tris, n_spheres, n_lights, n_materials;
    uint32_t uses_bvh;
    uint32_t n_nodes;           // traversal nodes (0 without a tree)
    uint32_t has_transparent;   // some material has transparency != 0
    uint32_t n_area_lights;     // lights of type RTX_LIGHT_AREA (each takes num_samples shadow rays per hit)
    uint32_t unit_light_dirs;   // every light direction the shading sees has length <= 1: point and area lights always (get_dir normalises, scene/mod.rs:35-59), a directional light when its `dir` is no longer than that (rtx_scene_set)
    // shading-side arrays, indexed through prim_ref by reference primitive id
    const uint32_t* prim_ref;   // [n_prims] bit 31 set = sphere; low 31 bits = index into its type's arrays
    const float* tri_pos;       // [n_tris][9]
    const float* tri_nrm;       // [n_tris][9] or nullptr
    const uint32_t* tri_mat;    // [n_tris][3]
    const uint8_t* tri_flags;   // [n_tris]
    const float4* sph;          // [n_spheres] (c.xyz, r)
    const uint32_t* sph_mat;    // [n_spheres]
    const rtx_material* mats;   // [n_materials]
    // What shading a hit reads first, gathered into ONE aligned 64-byte record per reference primitive id (one cache
    // line per hit instead of one each from prim_ref, tri_pos, tri_flags and tri_mat):
    //   triangle: (p0, ref) (p1, mat 0) (p2, mat 1) (mat 2, -, -, -)   sphere: (centre, ref) (radius, -, -, mat)
    //   ref = the prim_ref word with the triangle's flags in bits 28-29 (fewer than 2^28 primitives exist)
    const uint4* shade_recs;    // [n_prims][4]
    const DevLight* lights;     // [n_lights]
    // traversal-side arrays
    const PrimRec* recs;        // [n_prims]
    const BvhNode* nodes;       // [n_nodes] in level order (node 0 is the root), one slot per node
    // Primitives that every ray tests before it enters the tree, copies of their records. Triangles without an area to
    // speak of (two edges within 1e-3 rad of parallel, or a zero edge): the determinant of poly.rs:149-153 is rounding noise
    // for them, so the reference's test can report a hit for a ray that passes nowhere near the triangle, which no box can
    // promise to contain. And primitives with a coordinate or radius that is not a number, infinite or beyond 1e18: their
    // tests answer to overflow (an infinite sphere reports Hit(inf) for every ray), and their boxes would ruin the tree's.
    const PrimRec* always;      // [n_always]
    uint32_t n_always;
    // Rays whose origin lies outside the cube |o - safe_c|_inf <= safe_r take the
    // exhaustive loop: the sphere padding of the BVH is only proven inside it.
    float safe_c[3];
    float safe_r;
    float pad_rel;              // relative padding constant of the box test
    int* spill;                 // [spill_entries][spill_stride] overflow of the per-lane traversal stacks
    uint32_t spill_stride;      // thread slots = largest traversal grid * 256
    uint32_t spill_entries;
    int32_t tune[4];            // traversal tuning: [0] busy lanes below which a wavefront shares work, [1] work units a wavefront should expect, in quarters (byte 0 closest hit, byte 1 shadow), [2] most workgroups per CU, [3] unused
    unsigned long long* stamps; // diagnostics: [2 kinds][spill_stride / 64 wavefronts][4], or nullptr
};

// Work queue of rays: three arrays of 16-/8-byte elements so that every access
// of a wavefront is one fully coalesced 1 KiB (or 512 B) transaction.
//   o[i]   = (origin.xyz, bits(link))   link: level 0 = pixel slot; deeper = parent pool slot | (branch << 31)
//   d[i]   = (dir.xyz,    bits(flags))  flags bit0: ray is in vacuum (src/ray.rs:9); bit1: ray is valid
//   hit[i] = (prim_id, bits(t))         written by the traversal kernel (when the launch keeps a hit mask per work
//                                       unit for the shading pass: for rays that hit something only)
struct RayQueue {
    float4* o;
    float4* d;
    uint2* hit;
};
#define RTX_RAY_VACUUM 1u
#define RTX_RAY_VALID 2u

// Device allocations that several contexts of one GPU may share (rtx_clone): the scene arrays and
// the built tree. Freed when the last context holding them lets go.
struct DevStore {
    std::vector<void*> ptrs;
    ~DevStore() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

struct rtx_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    bool has_scene = false, built = false;

    // host copies of small things
    rtx_camera camera{};
    std::vector<rtx_light> lights;
    DevScene ds{};
    rtx_build_stats build_stats{};
    float scene_min[3]{}, scene_max[3]{};

    // device allocations: scene arrays and tree (shared with clones), traversal scratch of this context (stack overflow area, stamps)
    std::shared_ptr<DevStore> scene_store;
    std::shared_ptr<DevStore> build_store;
    std::vector<void*> trav_allocs;
    // staging for the entry points that take host buffers (rtx_render, rtx_intersect_batch*, rtx_primary_hits):
    // one buffer that only grows, so calls of a steady size allocate nothing after the first
    void* io_buf = nullptr;
    size_t io_size = 0;
    // tile-sharded frames over RCCL (rtx_comm.hip): communicator of this rank, gather buffer of the root
    void* comm = nullptr;
    uint32_t comm_rank = 0, comm_world = 0;
    void* gather_buf = nullptr;
    size_t gather_size = 0;
    void* comm_status = nullptr;   // one device word: the root tells the senders whether it can receive (rtx_gather_frames)
    uint32_t gather_agreed_root = 0xFFFFFFFFu;   // (root, bytes) every rank knows the root's gather buffer to be good for
    size_t gather_agreed_size = 0;
    // render workspace (grown on demand)
    std::vector<void*> work_allocs;
    uint64_t work_batch = 0;      // pixels per batch the workspace was sized for
    uint32_t work_slots = 0;      // shadow-term slots per node the workspace was sized for
    bool work_keys = false;       // the workspace holds per-node sampling keys (area lights)

    std::vector<char> work_blob;  // the Work struct of rtx_shade.hip describing work_allocs
    std::vector<hipEvent_t> events;
    void* pinned = nullptr;       // page-locked staging for counter read-backs and the cameras of a batch of frames
    void* cam_dev = nullptr;      // device copy of those cameras (a page for the synchronous path, one per frame in flight)
    bool has_secondary = false;   // some material reflects or refracts (src/lib.rs:230,241)
    uint32_t shade_a_resident = 0;   // workgroups of k_shade_a a CU keeps resident (asked once; sizes its launches)

    // knobs (rtx_set_option)
    int64_t opt_force_mode = 0;   // 0 auto, 1 exhaustive, 2 bvh
    int64_t opt_batch_pixels = 1 << 24;
    int64_t opt_pool_factor = 4;  // retained shading nodes per batch pixel
    uint32_t num_cus = 256;
    int32_t opt_tune[4] = {RTX_DEFAULT_TUNE0, RTX_DEFAULT_TUNE1, RTX_DEFAULT_TUNE2, RTX_LEAF_MAX};
    void* build_arena = nullptr;        // temporaries of rtx_build, kept for the next build when small
    size_t build_arena_size = 0;
    bool hit_fraction_known = false;   // a frame of the current scene has been collected
    float hit_fraction = 1.0f;   // share of the last frame's closest-hit rays that hit something
    uint32_t trav_spill_need = 0;   // stack entries beyond the LDS part that a traversal of the built tree can need
    int64_t opt_stamps = 0;
    int64_t opt_verify_build = 0;   // rtx_build re-checks every box of the refit (tests)

    // Frames in flight (rtx_render_*_async / rtx_render_wait): a frame whose kernels need no host
    // decision (one batch, one level) is only enqueued; its counters land in the slot's pinned
    // block and are read when the frame is collected. Up to two frames wait in the ring.
    struct AsyncSlot {
        bool used = false, deferred = false;
        rtx_render_stats stats{};                 // ready when !deferred
        std::vector<hipEvent_t> events;           // this slot's own events (frames in flight must not share them)
        hipEvent_t f0 = nullptr, f1 = nullptr, a = nullptr, b = nullptr, c = nullptr, d = nullptr;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> closest_events, shadow_events;   // a frame of several levels left in flight
        uint64_t hint_key[4] = {0, 0, 0, 0};
        bool multi_level = false;
        uint32_t levels = 0;
        uint64_t primary = 0;
        bool has_shadow = false;
        // arguments, to render the frame again if its queues overflowed
        uint32_t w = 0, h = 0, depth = 0, samples = 0, rank = 0, world = 1;
        uint8_t* out = nullptr;
        bool direct = false;
        uint32_t n_frames = 1;
        std::vector<rtx_camera> cameras;          // empty: the scene's camera
    };
    AsyncSlot async_slots[2];
    // What the last frame of a given shape told about its ray tree: how many levels had rays, and how many each.
    // The next frame of that shape enqueues that many levels (plus one) without asking the device in between -- one
    // wait per frame instead of one per level -- sizes their launches from the counts, and is rendered again level by
    // level if its ray tree turned out deeper (Ctl::spec_miss).
    struct LevelHint {
        uint64_t key[4];                       // (w, h) (depth, samples) (rank, world, frames) (first ray of the batch)
        uint32_t levels_used;
        uint32_t rays[RTX_MAX_DEPTH + 2];
    };
    std::vector<LevelHint> level_hints;
    int64_t opt_level_hints = 1;
    bool render_without_hints = false;   // the next rtx_render_impl call asks the device level by level (a frame rendered again after a miss)
    uint32_t async_head = 0, async_count = 0;     // oldest slot, frames waiting
    // Recorded launch sequences (hipGraph) of single-batch, single-level frames rendered through the async entry
    // points with kernel timing off: replaying one costs the host one call instead of about ten. Keyed by everything the launches depend on.
    struct FrameGraph {
        uint64_t key[8];
        hipGraphExec_t exec;
        bool has_shadow;
    };
    std::vector<FrameGraph> graphs;
    uint64_t graph_epoch = 0;
    int64_t opt_graphs = 1;
    int64_t opt_kernel_timing = 1;   // events on the traversal launches (closest_ms / shadow_ms of the statistics)
    int64_t opt_cost_order = 1;   // level-0 traversal launches hand out their units most expensive first (costs of the previous frame)
    std::vector<std::pair<uint64_t, void*>> order_bufs;   // (kind, batch, capacity) -> UnitOrder
};

// -- helpers shared by the translation units ---------------------------------
int rtx_fail(rtx_context* ctx, int status, const std::string& msg);
#define RTX_HIP(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return rtx_fail((ctx), RTX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// rtx_bvh.hip
int rtx_bvh_build(rtx_context* ctx);
int rtx_alloc_traversal_scratch(rtx_context* ctx);   // this context's stack overflow area (and stamps) for the built tree
int rtx_build_shade_recs(rtx_context* ctx, const DevScene& s, uint4* d_recs);   // s: counts and shading arrays set
int rtx_build_prim_ref(rtx_context* ctx, uint32_t n_tris, uint32_t n_spheres, const uint32_t* d_tri_index, const uint32_t* d_sph_index,
                       const uint8_t* d_tri_flags, uint32_t* d_ref, uint32_t* d_status /*[2]: bad permutation, some vertex normal*/);
// rtx_trace.hip
int rtx_launch_intersect(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                         int32_t* d_prim, float* d_t, int mode /*0 auto,1 exhaustive,2 bvh*/);
int rtx_noalloc_init_device(int device, std::string* err);   // once per device, with that device current (rtx_create)
int rtx_launch_intersect_noalloc(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                                 int32_t* d_prim, float* d_t);
// rtx_shade.hip
// n_frames > 1: a batch of frames rendered as one set of launches, one camera each (cameras[n_frames]; nullptr: the
// scene's camera); the output holds the frames' images, or the frames' tile slabs, one after the other.
int rtx_render_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t depth, uint32_t samples,
                    uint32_t rank, uint32_t world, uint8_t* d_rgb_tiles, rtx_render_stats* stats, bool direct,
                    int async_slot = -1, uint32_t n_frames = 1, const rtx_camera* cameras = nullptr);
int rtx_render_async_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t depth, uint32_t samples,
                          uint32_t rank, uint32_t world, uint8_t* d_rgb_tiles, bool direct, uint32_t n_frames = 1,
                          const rtx_camera* cameras = nullptr);
void rtx_clear_graphs(rtx_context* ctx);   // after anything that changes what a recorded frame would launch
int rtx_render_wait_impl(rtx_context* ctx, rtx_render_stats* stats);
int rtx_primary_impl(rtx_context* ctx, uint32_t w, uint32_t h, int32_t* d_prim, float* d_t, float* d_dir);
int rtx_assemble_impl(rtx_context* ctx, uint32_t w, uint32_t h, uint32_t world, const uint8_t* d_slabs, uint8_t* d_rgb, uint32_t n_frames = 1);
