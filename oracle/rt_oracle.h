/*
 * rt_oracle.h -- C API of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY. This is a single-threaded C++ restatement of the
 * reference renderer (sondrele/rust-raytracer), written to be checked against,
 * never shipped: only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load it. The product library (librtx_hip.so) never links,
 * loads or calls anything declared here.
 *
 * Parity pinning: the Rust reference cannot be built in this environment, so
 * the oracle is pinned by the known-answer values held by the reference's own
 * unit tests (vec.rs, ray.rs, lib.rs, sphere.rs, poly.rs, shapes/mod.rs,
 * scene/mod.rs, bvh.rs, material.rs, parser/test/mod.rs); tests/test_oracle_kat.py
 * replays every one of them through the orc_kat_* entry points. The shading
 * functions (src/lib.rs:105-255) and reflect/refract rays
 * (src/scene/intersection.rs:42-81) have no test in the reference: for those
 * the restatement itself is the oracle ("parity unpinned" beyond code review).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_stats {
    uint64_t primary_rays, shadow_rays, reflect_rays, refract_rays;
    uint64_t shaded_nodes;       /* shade_intersection calls with depth > 0 */
    uint64_t last_level_rays;    /* reflect/refract rays whose child is shaded at depth 0 (traced, result discarded) */
    uint64_t panic_pixels;       /* pixels during which weighted_areas / material indexing would have panicked */
    uint64_t prim_tests;         /* Primitive::intersects calls */
    double   seconds;            /* wall time of the render loop */
} orc_stats;

const char* orc_last_error(void);

/* SceneParser::new(path).parse_scene()  (src/scene/parser/mod.rs:20,314-346) */
orc_scene* orc_scene_parse(const char* path);
/* Builds the same in-memory scene from a flattened descriptor. */
orc_scene* orc_scene_from_desc(const rtx_scene_desc* desc);
void orc_scene_free(orc_scene* s);

void orc_scene_counts(const orc_scene* s, uint32_t* n_prims, uint32_t* n_tris, uint32_t* n_spheres,
                      uint32_t* n_materials, uint32_t* n_lights);
/* Writes the flattened form (array sizes from orc_scene_counts). Any pointer may be NULL. */
void orc_scene_flatten(const orc_scene* s, float* tri_pos, float* tri_normals, uint32_t* tri_mat,
                       uint8_t* tri_flags, uint32_t* tri_prim_index, float* sph_center_radius,
                       uint32_t* sph_mat, uint32_t* sph_prim_index, rtx_material* materials,
                       rtx_light* lights, rtx_camera* camera);

/* Scene::intersects (src/scene/mod.rs:181-201) for n rays; use_bvh != 0 routes
 * through the restated reference BVH (src/scene/bvh.rs:88-143) instead. */
void orc_intersect_batch(orc_scene* s, uint64_t n, const float* ori3, const float* dir3,
                         int32_t* prim_id, float* t, int use_bvh);

/* RayTracer::init + set_scene + trace_rays (src/lib.rs:55-67,257-277).
 * Only pixels with x % stride_x == 0 and y % stride_y == 0 are rendered
 * (stride 1 = whole frame); untouched pixels keep their previous contents.
 * Outputs (any may be NULL): rgb [h][w][3], prim_id [h][w] (-1 = miss),
 * t [h][w], panic [h][w], ray_dir [h][w][3]. threads <= 1 is the faithful
 * single-threaded loop; threads > 1 splits rows over std::thread workers
 * (results are identical; used to shorten tests). */
void orc_render(orc_scene* s, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                uint32_t stride_x, uint32_t stride_y, int threads, int use_bvh,
                uint8_t* rgb, int32_t* prim_id, float* t, uint8_t* panic, float* ray_dir,
                orc_stats* stats);

/* ---- known-answer hooks: one per unit test of the reference -------------- */
void  orc_kat_vec_ops(const float a[3], const float b[3], float add[3], float sub[3], float mul[3],
                      float cross[3], float* dot, float* length_a, float normalized_a[3], float* distance);
float orc_kat_vec_area(const float a[3], const float b[3], const float c[3]);
/* returns 1 on hit and stores t */
int   orc_kat_sphere_intersect(const float center[3], float radius, const float ori[3], const float dir[3], float* t);
int   orc_kat_poly_intersect(const float v[9], const float ori[3], const float dir[3], float* t);
void  orc_kat_sphere_bbox(const float center[3], float radius, float bmin[3], float bmax[3]);
void  orc_kat_poly_bbox(const float v[9], float bmin[3], float bmax[3]);
int   orc_kat_bbox_intersect(const float bmin[3], const float bmax[3], const float ori[3], const float dir[3]);
void  orc_kat_bbox_centroid(const float bmin[3], const float bmax[3], float c[3]);
/* BoundingBox::lt via centroid + Vec3::lt (shapes/mod.rs:114-116, vec.rs:26-38) */
int   orc_kat_bbox_less(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3]);
void  orc_kat_color_init(float r, float g, float b, float out[3]);
void  orc_kat_compute_ray(const rtx_camera* cam, uint32_t width, uint32_t height, float x, float y,
                          float ori[3], float dir[3]);
/* Builds the reference BVH over n unit-material spheres and reports, for node
 * path `path` (string of 'L'/'R' from the root, "" = root): kind (0 empty,
 * 1 member, 2 leaf), its bbox and, for leaves, the index (into the input
 * array) of the sphere it holds. */
int   orc_kat_bvh_node(const float* centers_radius4, uint32_t n, const char* path,
                       float bmin[3], float bmax[3], int32_t* leaf_index);
/* Tree::intersects on that tree: returns the input index of the sphere hit or -1. */
int   orc_kat_bvh_intersect(const float* centers_radius4, uint32_t n, const float ori[3],
                            const float dir[3], float* t);
/* Refraction / reflection rays from a hit on a unit sphere at the origin
 * (no reference test; exported so tests can compare device and oracle). */
int   orc_kat_secondary_rays(const orc_scene* s, const float ori[3], const float dir[3], int in_vacuum,
                             float refl_ori[3], float refl_dir[3], int* has_refr,
                             float refr_ori[3], float refr_dir[3]);

#ifdef __cplusplus
}
#endif
#endif
