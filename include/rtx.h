/*
 * rtx.h -- C ABI of librtx_hip.so, the MI355X (gfx950) implementation of
 * rust-raytracer's ray/scene intersection + Whitted shading hot path.
 *
 * Boundary being replaced (reference, /root/reference):
 *   trait IntersectableScene { get_camera, get_lights, intersects }   src/scene/mod.rs:148-154
 *   RayTracer::init / set_scene / trace_rays                          src/lib.rs:55,64,257
 * A per-ray `intersects` across FFI + PCIe is unusable, so the boundary is per
 * frame (rtx_render*) plus a batched form of `intersects` (rtx_intersect_batch*).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a caller-owned buffer.
 *   - every function returns an rtx_status; nothing aborts or throws across the
 *     ABI (the reference panics instead: src/lib.rs:72,275).
 *   - one rtx_context per host thread / GPU; calls on one context are
 *     serialised by the caller.
 *   - primitive id = index into the reference's `Scene.primitives`
 *     (src/scene/mod.rs:159): file order of blocks with every poly_set reversed
 *     (src/scene/parser/mod.rs:333-338). A miss is -1.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 2u

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID_ARG = 1,   /* NULL pointer, zero size, inconsistent descriptor */
    RTX_ERR_NO_DEVICE = 2,     /* no HIP device / device index out of range */
    RTX_ERR_HIP = 3,           /* a HIP runtime call failed; see rtx_last_error */
    RTX_ERR_NO_SCENE = 4,      /* render/intersect before rtx_scene_set (reference: panic, src/lib.rs:72,275) */
    RTX_ERR_NOT_BUILT = 5,     /* render/intersect before rtx_build */
    RTX_ERR_UNSUPPORTED = 6,   /* e.g. a depth above RTX_MAX_DEPTH that the scene would really use, more than 2^28 primitives */
    RTX_ERR_OUT_OF_MEMORY = 7
} rtx_status;

/* Deepest recursion rtx_render* follows. The reference takes any usize (src/main.rs:65); a larger depth is
 * accepted (and changes nothing) when no material reflects or refracts, and rejected otherwise. */
#define RTX_MAX_DEPTH 32u
#define RTX_MATERIAL_NONE 0xFFFFFFFFu /* absent slot of a poly's material list */

/* Material: src/scene/material.rs:86-94. 14 f32 = 56 bytes. The reference's parser clamps colours to [0,1]
 * (parser/mod.rs:111-114 -> material.rs:17-27); rtx_scene_set does the same to whatever it is given. */
typedef struct rtx_material {
    float diffuse[3];
    float ambient[3];
    float specular[3];
    float emissive[3];   /* parsed, never read by the shading path */
    float shininess;
    float transparency;
} rtx_material;

/* Light: src/scene/mod.rs:19-24,62-120. */
enum { RTX_LIGHT_POINT = 0, RTX_LIGHT_AREA = 1, RTX_LIGHT_DIRECTIONAL = 2 };
typedef struct rtx_light {
    uint32_t type;
    float a[3];          /* point: pos; area: min (first `position`); directional: dir */
    float b[3];          /* area: max (second `position`); otherwise unused */
    float color[3];      /* intensity, clamped to [0,1] */
} rtx_light;

/* Camera: src/scene/mod.rs:122-129. */
typedef struct rtx_camera {
    float pos[3];
    float view_dir[3];
    float ortho_up[3];
    float focal_dist;    /* parsed, unused */
    float vertical_fov;  /* radians */
} rtx_camera;

/* Triangle flags (Poly::vertex_normal / vertex_material, shapes/poly.rs:54-55). */
enum { RTX_TRI_VERTEX_NORMAL = 1, RTX_TRI_VERTEX_MATERIAL = 2 };

/* Flattened scene = what a host-side `impl` produces from `Scene`
 * (src/scene/mod.rs:156-160). All arrays are host memory, copied by
 * rtx_scene_set. Triangles and spheres are two arrays; `*_prim_index` gives the
 * reference primitive index of each element (NULL = triangles 0..n_tris-1 then
 * spheres n_tris..). The union of both index arrays must be a permutation of
 * 0..n_tris+n_spheres-1. */
typedef struct rtx_scene_desc {
    uint32_t n_tris;
    const float*    tri_pos;        /* [n_tris][3][3]  vertex positions v0,v1,v2 */
    const float*    tri_normals;    /* [n_tris][3][3]  vertex normals; may be NULL if no tri has RTX_TRI_VERTEX_NORMAL */
    const uint32_t* tri_mat;        /* [n_tris][3]     global ids of the poly's de-duplicated material list
                                                       (poly.materials[0..2], parser/mod.rs:266-291); absent = RTX_MATERIAL_NONE */
    const uint8_t*  tri_flags;      /* [n_tris]        RTX_TRI_* ; may be NULL (= 0) */
    const uint32_t* tri_prim_index; /* [n_tris]        may be NULL */
    uint32_t n_spheres;
    const float*    sph_center_radius; /* [n_spheres][4] cx,cy,cz,r (axes/lengths are ignored by the reference, sphere.rs:12-17) */
    const uint32_t* sph_mat;           /* [n_spheres]    global id of materials[0] */
    const uint32_t* sph_prim_index;    /* [n_spheres]    may be NULL */
    uint32_t n_materials;
    const rtx_material* materials;
    uint32_t n_lights;
    const rtx_light* lights;
    rtx_camera camera;
} rtx_scene_desc;

typedef struct rtx_build_stats {
    uint32_t n_prims;
    uint32_t n_nodes;        /* traversal nodes emitted (0 when the exhaustive kernel is used) */
    uint32_t n_leaves;
    uint32_t max_depth;      /* deepest traversal-node level */
    uint32_t uses_bvh;       /* 0: scene is small, exhaustive kernel; 1: LBVH */
    uint32_t key_bits;       /* Morton key width used */
    float    build_ms;       /* device time of the whole build */
    float    upload_ms;      /* host->device copy of the scene arrays in rtx_scene_set */
    uint32_t n_always;       /* triangles without an area to speak of (two edges within 1e-3 rad of parallel, or a zero edge):
                                the reference's test answers rounding noise for them (poly.rs:149-153), no box can promise to
                                contain that, so EVERY ray tests each of them before it enters the tree. Usually 0; a mesh with
                                thousands of them renders correctly but slowly (rtx_last_error then holds a note after rtx_build) */
    uint32_t reserved;
} rtx_build_stats;

/* Ray counts follow the reference's definition: one ray = one call of
 * `scene.intersects` (src/lib.rs:117,232,243,264). `skipped_rays` are rays the
 * reference traces and then discards because the child would be shaded at depth
 * 0 (src/lib.rs:232-234 with depth-1 == 0); the GPU does not trace them. */
typedef struct rtx_render_stats {
    uint64_t primary_rays;
    uint64_t shadow_rays;       /* includes continuation rays through transparent hits */
    uint64_t reflect_rays;
    uint64_t refract_rays;
    uint64_t skipped_rays;      /* reference-counted but not traced here */
    uint64_t shaded_nodes;
    uint64_t panic_pixels;      /* shaded nodes where the reference would have panicked (poly.rs:84-86,93) */
    float    render_ms;         /* device time of the frame, scene resident, excludes the copy of the image to the host */
    float    closest_ms;        /* device time inside the closest-hit traversal kernel (primary + secondary rays) */
    float    shadow_ms;         /* device time inside the shadow-ray traversal kernel */
    uint32_t closest_launches;  /* launches behind closest_ms */
    uint32_t shadow_launches;   /* launches behind shadow_ms */
    uint32_t batches;           /* pixel batches the frame was split into */
    uint32_t reserved;
} rtx_render_stats;

typedef struct rtx_context rtx_context;

/* ---- lifetime ------------------------------------------------------------
 * No counterpart in the reference (it has no device). */
int  rtx_abi_version(void);
/* 16 hex digits naming the sources this library was built from (kernels, headers, build flags): measurements quote
 * counters only for the build they were collected on. */
const char* rtx_source_hash(void);
int  rtx_create(int device, rtx_context** out_ctx);
void rtx_destroy(rtx_context* ctx);
/* Message of the last failing call on `ctx` (or of the last failing rtx_create
 * when ctx is NULL). Never NULL. */
const char* rtx_last_error(const rtx_context* ctx);

/* Tuning knobs (no counterpart in the reference):
 *   "force_mode"   0 = automatic, 1 = exhaustive kernel, 2 = BVH (takes effect at the next rtx_build)
 *   "batch_pixels" pixels rendered per pass of the ray queues (default 2^24; the workspace is sized by min(this, pixels of the frame))
 *   "pool_factor"  retained shading nodes per batch pixel (default 4)
 *   "graphs"        1 (default): frames rendered through the _async entry points that need no host decision between
 *                   their launches (one batch, no reflective or transparent material, "kernel_timing" 0) are recorded once as a hipGraph
 *                   and replayed, which costs the host one call per frame instead of about ten (one recording per frame size, output
 *                   buffer and in-flight slot; at most 32 are kept, so cycle through a few output buffers, not many); 0: always plain launches
 *   "kernel_timing" 1 (default): the traversal launches carry timing events (closest_ms, shadow_ms of the statistics);
 *                   0: no events, a frame's launches follow each other a few microseconds closer
 *   "tune0"         lanes still busy below which a wavefront's idle lanes start taking subtrees from busy ones (default RTX_DEFAULT_TUNE0 = 48; 0 = never)
 *   "tune1"         work units (64 rays) a traversal wavefront should expect, in quarters: byte 0 for closest-hit
 *                   launches (default 32 = 8 units), byte 1 for shadow launches (default 10 = 2.5 units), RTX_DEFAULT_TUNE1; small
 *                   launches get fewer workgroups per CU accordingly. Two launches do not grow with their work but with its
 *                   square root: a shadow launch gets at most round(sqrt(units / (4 * CUs))) workgroups per CU, and the
 *                   primary-ray launch of a frame ceil(sqrt(units * share / (4 * CUs))), share = the part of the previous
 *                   frame's primary rays that hit something (byte 0 then only sizes the launches of deeper levels)
 *   "tune2"         most traversal workgroups per CU (default RTX_DEFAULT_TUNE2 = 7; values above the number a kernel is compiled to keep
 *                   resident -- 4 to 7, by kernel -- change nothing, and the seventh is only used by launches of about
 *                   two million rays or more)
 *   "tune3"         primitives per leaf: fixed at 1 (any other value is RTX_ERR_INVALID_ARG)
 *   "cost_order"    1 (default) = level-0 traversal launches hand out their work units most expensive first, using the
 *                   durations recorded by the previous frame of the same size; 0 = always index order. Never changes results.
 *   "level_hints"   1 (default) = a frame whose ray tree has several levels (reflective or transparent materials) enqueues as many
 *                   levels as the previous frame of the same shape used, plus one, without asking the device in between (one
 *                   wait per frame instead of one per level; such frames stay in flight behind the _async entry points), and is
 *                   rendered again level by level when its tree turned out deeper; 0 = always level by level. Never changes results.
 *   "verify_build"  1 = rtx_build recomputes, in a launch of its own, every inner node's box from its children's and fails
 *                   (RTX_ERR_HIP) when one differs from what the one-launch bottom-up refit handed over (tests; default 0)
 *   "stamps"        1 = traversal kernels record per-wavefront time stamps (see rtx_debug_stamps); at the next rtx_build */
int rtx_set_option(rtx_context* ctx, const char* key, int64_t value);
/* The defaults of "tune0" .. "tune2", in one place: the library initialises its options from these, and tests and
 * tools that restore a default read them here. */
#define RTX_DEFAULT_TUNE0 48
#define RTX_DEFAULT_TUNE1 (32 + (10 << 8)) /* 8 units per closest-hit wavefront, 2.5 per shadow wavefront */
#define RTX_DEFAULT_TUNE2 7

/* ---- scene ---------------------------------------------------------------
 * Replaces RayTracer::set_scene (src/lib.rs:64-67): copies the flattened scene
 * to the device. The camera basis of setup_camera (src/lib.rs:69-84) is
 * computed per frame because it depends on width/height. */
int rtx_scene_set(rtx_context* ctx, const rtx_scene_desc* scene);
/* Same, for a scene whose triangle and sphere arrays (tri_*, sph_*) already live in DEVICE memory of the
 * context's GPU -- e.g. the receive buffers of an RCCL broadcast -- so that a replicated scene does not travel
 * GPU -> host -> GPU on every rank. `materials` and `lights` stay HOST arrays (they are small and the host reads
 * them). The arrays are copied device-to-device on the context's stream; the caller keeps ownership. A rejected
 * descriptor (either call) leaves the context's previous scene in place. No counterpart in the reference. */
int rtx_scene_set_device(rtx_context* ctx, const rtx_scene_desc* scene);
/* A second context on the same GPU that SHARES the scene arrays and the built tree of `src` (reference-counted,
 * read-only) and has its own stream, render workspace and traversal scratch: frames of several such contexts
 * overlap on the GPU without a second copy of the scene or a second build. Options are copied. A later
 * rtx_scene_set / rtx_build on either context only affects that context. No counterpart in the reference. */
int rtx_clone(rtx_context* src, rtx_context** out_ctx);

/* Replaces Tree::init (src/scene/bvh.rs:82-113), the reference's `-b` build:
 * bounds -> Morton keys -> LDS radix sort -> Karras hierarchy -> refit ->
 * traversal nodes. Results of every query equal the exhaustive loop
 * Scene::intersects (src/scene/mod.rs:181-201), NOT the reference BVH's
 * right-wins-ties rule (bvh.rs:134). `stats` may be NULL. */
int rtx_build(rtx_context* ctx, rtx_build_stats* stats);

/* ---- queries -------------------------------------------------------------
 * Batched IntersectableScene::intersects (src/scene/mod.rs:153,181-201).
 * ori3/dir3: [n][3] f32. prim_id: [n] i32 (-1 = Missed). t: [n] f32 (the
 * `point` of ShapeIntersection::Hit; undefined where prim_id < 0). Host
 * pointers. */
int rtx_intersect_batch(rtx_context* ctx, uint64_t n, const float* ori3, const float* dir3,
                        int32_t* prim_id, float* t);
/* Same with device pointers; runs on the context's stream and returns after
 * the kernel has been enqueued and synchronised. */
int rtx_intersect_batch_device(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                               int32_t* d_prim_id, float* d_t);
/* No counterpart in the reference. For callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the query of
 * rtx_intersect_batch_device with the overflow area of the per-lane traversal stacks in a static
 * device array of the library (24 MiB per GPU, held until the process ends) instead of the context's
 * own allocation. The call allocates and frees nothing and does not synchronise: the kernel is
 * enqueued on the context's stream (rtx_stream) and the caller waits for it there. Limits, rejected
 * with RTX_ERR_UNSUPPORTED: n <= RTX_NOALLOC_MAX_RAYS per call, and a tree whose traversal needs at
 * most 48 stack entries beyond the 16 kept in LDS (about 20 levels of 4-wide nodes; every bundled and
 * synthetic scene up to 64 M primitives fits). All calls on one GPU share the array: calls that
 * overlap on one GPU (two contexts or streams) race; calls that follow each other need no clearing. */
#define RTX_NOALLOC_MAX_RAYS (1ull << 26)
int rtx_intersect_batch_device_noalloc(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                                       int32_t* d_prim_id, float* d_t);
/* Same query through the exhaustive device kernel regardless of scene size
 * (the device restatement of the loop at src/scene/mod.rs:186-199). */
int rtx_intersect_batch_exhaustive(rtx_context* ctx, uint64_t n, const float* ori3, const float* dir3,
                                   int32_t* prim_id, float* t);

/* Primary rays only: compute_ray (src/lib.rs:96-103) for every pixel, then the
 * closest hit. Row 0 is the top row of the image, i.e. pixel (x,y) is the ray
 * through (x, height-y-1) as in trace_rays (src/lib.rs:263). Any of the output
 * pointers may be NULL. Host pointers: prim_id [h][w], t [h][w], ray_dir [h][w][3]. */
int rtx_primary_hits(rtx_context* ctx, uint32_t width, uint32_t height,
                     int32_t* prim_id, float* t, float* ray_dir);

/* ---- frame ---------------------------------------------------------------
 * Replaces RayTracer::init + trace_rays (src/lib.rs:55,257-277) followed by
 * Color::as_pixel (src/scene/material.rs:57-63). rgb: host buffer [h][w][3] u8,
 * row 0 = top row; missed pixels are 0,0,0 (bmp::Image::new default).
 * `samples` is num_samples (area lights only). `stats` may be NULL. */
int rtx_render(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
               uint8_t* rgb, rtx_render_stats* stats);

/* Same frame, left on the device: d_rgb is DEVICE memory, [h][w][3] u8 row-major. */
int rtx_render_device(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                      uint8_t* d_rgb, rtx_render_stats* stats);

/* Frames in flight (no counterpart in the reference, whose trace_rays returns the finished image):
 * the _async calls enqueue a frame and return; rtx_render_wait collects the OLDEST enqueued frame
 * (its image is complete, its statistics are returned). At most two frames may be waiting, so a
 * caller alternates "enqueue frame k+1, collect frame k" and the device never idles between frames.
 * Frames that need host decisions while they render (several batches or ray-tree levels) are simply
 * rendered inside the _async call. Each frame in flight needs its own output buffer. */
int rtx_render_device_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                            uint8_t* d_rgb);
int rtx_render_tiles_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                           uint32_t rank, uint32_t world, uint8_t* d_rgb_tiles);
int rtx_render_wait(rtx_context* ctx, rtx_render_stats* stats);

/* Tile-sharded frame for one process per GPU. The image is cut into 8x8-pixel
 * tiles numbered row-major; this call renders tiles with
 * (tile_index % world) == rank. d_rgb_tiles is DEVICE memory receiving the
 * rendered tiles packed in increasing tile order, 8*8*3 bytes per tile (edge
 * tiles are padded with zeros); capacity must be rtx_tiles_per_rank() tiles.
 * The image stays on the device so the caller can hand it to a collective. */
int rtx_render_tiles(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                     uint32_t rank, uint32_t world, uint8_t* d_rgb_tiles, rtx_render_stats* stats);
/* Number of tiles every rank's slab holds (the maximum over ranks, so slabs
 * have equal size for an all-gather). */
uint64_t rtx_tiles_per_rank(uint32_t width, uint32_t height, uint32_t world);
/* Scatters gathered slabs ([world][tiles_per_rank][8*8*3] u8, device memory)
 * into a row-major [h][w][3] image (device memory). */
int rtx_assemble_tiles(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t world,
                       const uint8_t* d_slabs, uint8_t* d_rgb);
/* Same, but returns as soon as the work is enqueued on the library's stream (rtx_stream): the
 * caller orders it against its own streams and collectives with events on that stream. */
int rtx_assemble_tiles_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t world,
                             const uint8_t* d_slabs, uint8_t* d_rgb);

/* Batches of frames (no counterpart in the reference, which renders one frame per process; for callers that render
 * a camera path over one scene, and for tile-sharded rendering, where one rank's share of one frame is too little
 * work to fill a GPU). n_frames (1 .. RTX_MAX_BATCH_FRAMES) frames of the same size are rendered as ONE set of
 * launches: frame f uses cameras[f] (RayTracer::setup_camera per frame, src/lib.rs:69-84), or the scene's camera
 * for every frame when cameras is NULL. Each frame's pixels are exactly those of a separate rtx_render_device /
 * rtx_render_tiles call with that camera (area-light draws are keyed by pixel and do not depend on the batch).
 *   row_major != 0 (world must be 1): d_out receives [n_frames][h][w][3];
 *   otherwise d_out receives the rank's slabs, [n_frames][rtx_tiles_per_rank()][8*8*3].
 * All frames' rays must fit one ray batch (the "batch_pixels" option, default 16 Mi rays): RTX_ERR_INVALID_ARG
 * otherwise. A batch whose ray tree (reflective and transparent materials, deep recursion) outgrows the node pool
 * ("pool_factor") is rendered again one frame after the other, each in as many passes as it needs: the same images, later. The _async form enqueues and returns like rtx_render_tiles_async; rtx_render_wait collects the batch
 * (ray counts are totals over its frames). */
#define RTX_MAX_BATCH_FRAMES 64u
int rtx_render_frames(rtx_context* ctx, uint32_t n_frames, const rtx_camera* cameras, uint32_t width, uint32_t height,
                      uint32_t depth, uint32_t samples, uint32_t rank, uint32_t world, uint8_t* d_out, int row_major,
                      rtx_render_stats* stats);
int rtx_render_frames_async(rtx_context* ctx, uint32_t n_frames, const rtx_camera* cameras, uint32_t width,
                            uint32_t height, uint32_t depth, uint32_t samples, uint32_t rank, uint32_t world,
                            uint8_t* d_out, int row_major);
/* Gathered slabs of a batch, [world][n_frames][tiles_per_rank][8*8*3] (an all-gather of the ranks' d_out), into
 * [n_frames][h][w][3]; enqueued on rtx_stream like rtx_assemble_tiles_async. */
int rtx_assemble_frames_async(rtx_context* ctx, uint32_t n_frames, uint32_t width, uint32_t height, uint32_t world,
                              const uint8_t* d_slabs, uint8_t* d_rgb);

/* ---- exchange steps of a tile-sharded frame (no counterpart in the reference, which has no second device) -----
 * One process per GPU, one context per process. The scene is replicated once from a root rank into every rank's
 * HBM; every frame, each rank renders its interleaved tiles (rtx_render_tiles) and the slabs are gathered to the
 * root, which assembles the image. Both steps run over RCCL (xGMI inside a node) on the context's stream; librccl
 * is loaded when rtx_comm_unique_id / rtx_comm_init is first called (RTX_ERR_UNSUPPORTED if it cannot be).
 *   rtx_comm_unique_id  fills 128 bytes on ONE rank; the caller hands them to the other ranks by any host channel
 *                       (a file, a socket, MPI, torch.distributed)
 *   rtx_comm_init       collective: every rank calls it with the same id, its rank and the number of ranks
 *   rtx_scene_broadcast collective: the root passes its flattened scene (host arrays, as for rtx_scene_set), the
 *                       other ranks pass NULL; afterwards every rank holds the scene as if it had called
 *                       rtx_scene_set, and calls rtx_build itself (building is cheaper than moving the tree)
 *   rtx_gather_frame    collective: d_rgb_tiles = this rank's slab from rtx_render_tiles (device memory); on the root
 *                       d_rgb receives the row-major [h][w][3] image (device memory; ignored elsewhere). Returns
 *                       when this rank's part is complete. A gather: each slab crosses one link once. */
#define RTX_COMM_ID_BYTES 128
int rtx_comm_unique_id(uint8_t* id128);
int rtx_comm_init(rtx_context* ctx, uint32_t rank, uint32_t world, const uint8_t* id128);
int rtx_comm_destroy(rtx_context* ctx);
int rtx_scene_broadcast(rtx_context* ctx, const rtx_scene_desc* scene, uint32_t root);
int rtx_gather_frame(rtx_context* ctx, uint32_t width, uint32_t height, const uint8_t* d_rgb_tiles, uint8_t* d_rgb,
                     uint32_t root);
/* The same for a batch (rtx_render_frames with slabs): d_rgb_tiles = this rank's [n_frames][tiles_per_rank][8*8*3],
 * d_rgb on the root receives [n_frames][h][w][3]; one exchange for the whole batch. */
int rtx_gather_frames(rtx_context* ctx, uint32_t n_frames, uint32_t width, uint32_t height, const uint8_t* d_rgb_tiles,
                      uint8_t* d_rgb, uint32_t root);

/* Diagnostics (no counterpart in the reference): with the "stamps" option set, every
 * wavefront of the last closest-hit launch (kind 0) or shadow launch (kind 1) has
 * recorded 4 x u64: start and end on the 100 MHz constant clock, work units of 64
 * rays taken, and its hardware placement (XCC_ID | SE_ID << 8 | CU_ID << 16).
 * Copies up to `capacity` records of 4 u64 to `out` (host) and returns how many
 * wavefront slots exist (0 when stamps are off). Slots of wavefronts that did
 * not run hold zeros. */
uint64_t rtx_debug_stamps(rtx_context* ctx, int kind, uint64_t* out, uint64_t capacity);

/* Streams. Every context works on a HIP stream of its own, created non-blocking: it waits for no other
 * stream, the null stream included. Device buffers handed to an entry point must therefore be complete
 * when the call is made (a fill or copy the caller enqueued on another stream has to have finished, or
 * the caller makes rtx_stream wait for it with an event), and results written by an _async call are
 * ordered against the caller's streams with events on rtx_stream.
 * rtx_stream: the context's stream as a void* (hipStream_t). */
void* rtx_stream(rtx_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
