// rstracer.hpp -- host side above the C ABI: the reference crate's scene types,
// parser and render entry point, mirrored in C++ because no Rust toolchain exists
// here. Names, argument meaning and error behaviour follow the reference; where
// the reference panics these functions throw rstracer::Error.
//
//   SceneParser::new / parse_scene     src/scene/parser/mod.rs:20,314-346
//   Scene, Camera, Light, Primitive    src/scene/mod.rs:19-160, src/scene/shapes/*.rs
//   IntersectableScene                 src/scene/mod.rs:148-154
//   RayTracer::init/set_scene/trace_rays   src/lib.rs:55,64,257
//
// There is no CPU renderer behind these types: every query goes to
// librtx_hip.so, and fails if that library or a GPU is missing.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rtx.h"

namespace rstracer {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Vec3 {   // src/vec.rs:5-10
    float x = 0, y = 0, z = 0;
};
struct Color {  // src/scene/material.rs:5-27: clamps on construction
    float r = 0, g = 0, b = 0;
    static Color init(float r, float g, float b);
};
struct Material {  // src/scene/material.rs:86-94
    Color diffuse, ambient, specular, emissive;
    float shininess = 0, transparency = 0;
};
struct Vertex {    // src/scene/shapes/poly.rs:9-15
    uint32_t mat_index = 0;
    bool has_normal = false;
    Vec3 position, normal;
};
struct Poly {      // src/scene/shapes/poly.rs:50-56
    std::vector<Material> materials;
    Vertex vertices[3];
    bool vertex_material = false, vertex_normal = false;
};
struct Sphere {    // src/scene/shapes/sphere.rs:7-18
    std::vector<Material> materials;
    Vec3 origin;
    float radius = 0;
    Vec3 xaxis, yaxis, zaxis;
    float xlength = 0, ylength = 0, zlength = 0;
};
struct Primitive { // src/scene/shapes/mod.rs:131-135
    enum Kind { PolyKind, SphereKind } kind = PolyKind;
    Poly poly;
    Sphere sphere;
};
struct Light {     // src/scene/mod.rs:19-24
    enum Kind { Point = RTX_LIGHT_POINT, Area = RTX_LIGHT_AREA, Directional = RTX_LIGHT_DIRECTIONAL } kind = Point;
    Vec3 pos;      // Point: pos; Area: min; Directional: dir
    Vec3 max;      // Area only
    Color intensity;
};
struct Camera {    // src/scene/mod.rs:122-129
    Vec3 pos, view_dir;
    float focal_dist = 0;
    Vec3 ortho_up;
    float vertical_fov = 0;
};
struct Scene {     // src/scene/mod.rs:156-160
    Camera camera;
    std::vector<Light> lights;
    std::vector<Primitive> primitives;
};

struct Ray {       // src/ray.rs:5-10
    Vec3 ori, dir;
};
struct SceneIntersection {   // src/scene/mod.rs:143-146 + intersection.rs:7-11
    bool intersected = false;
    float point = 0;          // the ray parameter t
    int32_t primitive = -1;   // index into Scene::primitives
};

struct FlatScene;

// SceneParser (src/scene/parser/mod.rs:12-352). Strict keyword order; any
// mismatch throws Error with the message the reference would panic with.
class SceneParser {
public:
    explicit SceneParser(const std::string& path);   // SceneParser::new
    ~SceneParser();
    Scene parse_scene();
    // The same grammar, errors and primitive numbering in ONE pass from the file to the flattened arrays, without
    // the object graph in between (parse_scene holds three Vertex objects and a std::vector<Material> per triangle:
    // several hundred bytes and a heap allocation each, which is what a scene of millions of triangles cannot
    // afford). A poly_set's materials enter the material table once instead of once per triangle.
    FlatScene parse_flat();
    // pieces exposed for the parser tests (parser/test/mod.rs)
    std::string next_token();
    std::string peak();
    bool has_next_token() const { return !finished_; }
    float next_f32();
    Vec3 parse_vec3(const char* name);
    Color parse_color(const char* name);
    Light parse_light();
    Material parse_material();
    Sphere parse_sphere();
    Poly parse_poly(bool has_normal, bool has_material);
    std::vector<Poly> parse_polyset();
    Camera parse_camera();

private:
    struct Tok { const char* p; size_t n; };
    Tok next_tok();
    void expect(const char* token);
    long next_int();
    float parse_f32(const char* name);
    bool parse_bool(const char* name, const char* flag);
    Vertex parse_vertex(bool has_normal, bool has_material);
    void flat_polyset(FlatScene& f, uint32_t& n_prims);
    const char* data_ = nullptr;
    size_t size_ = 0, pos_ = 0;
    bool mapped_ = false, finished_ = false, peaked_ = false;
    Tok last_{nullptr, 0};
};

// Flattened arrays behind an rtx_scene_desc (the descriptor points into them).
struct FlatScene {
    std::vector<float> tri_pos, tri_normals, sph_center_radius;
    std::vector<uint32_t> tri_mat, tri_prim_index, sph_mat, sph_prim_index;
    std::vector<uint8_t> tri_flags;
    std::vector<rtx_material> materials;
    std::vector<rtx_light> lights;
    rtx_camera camera{};
    rtx_scene_desc desc() const;
};
FlatScene flatten(const Scene& scene);
Camera camera_of(const FlatScene& flat);
std::vector<Light> lights_of(const FlatScene& flat);

// The drop-in implementation of IntersectableScene (src/scene/mod.rs:148-154):
// owns an rtx_context with the scene uploaded and built.
class GpuScene {
public:
    // force_bvh: what the reference's `-b` flag selects (BvhScene instead of Scene, src/main.rs:70-74). Here the tree is
    // then built even for the few-primitive scenes the library would otherwise sweep exhaustively; results are the same.
    explicit GpuScene(const Scene& scene, int device = 0, bool force_bvh = false);
    explicit GpuScene(const FlatScene& flat, int device = 0, bool force_bvh = false);   // a scene read with parse_flat
    ~GpuScene();
    GpuScene(const GpuScene&) = delete;
    GpuScene& operator=(const GpuScene&) = delete;
    const Camera& get_camera() const { return camera_; }
    const std::vector<Light>& get_lights() const { return lights_; }
    SceneIntersection intersects(const Ray& ray) const;                      // one ray (slow: one device round trip)
    std::vector<SceneIntersection> intersects(const std::vector<Ray>& rays) const;   // batched
    rtx_context* context() const { return ctx_; }
    const rtx_build_stats& build_stats() const { return build_; }

private:
    void upload(const FlatScene& flat, int device, bool force_bvh);
    rtx_context* ctx_ = nullptr;
    Camera camera_;
    std::vector<Light> lights_;
    rtx_build_stats build_{};
};

struct Image {   // stands in for bmp::Image: row 0 is the top row
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    void save(const std::string& path) const;   // 24-bpp BMP, like img.save (src/main.rs:78)
    static Image load(const std::string& path);
};

class RayTracer {   // src/lib.rs:24-278
public:
    static RayTracer init(uint32_t width, uint32_t height, size_t depth, size_t num_samples);   // lib.rs:55-62
    void set_scene(std::unique_ptr<GpuScene> scene);                                             // lib.rs:64-67
    Image trace_rays();                                                                          // lib.rs:257-277
    const rtx_render_stats& last_stats() const { return stats_; }
    uint32_t width = 0, height = 0;
    size_t num_samples = 1, depth = 0;

private:
    std::unique_ptr<GpuScene> scene_;
    rtx_render_stats stats_{};
};

}  // namespace rstracer
