// rstracer.cpp -- see rstracer.hpp.
#include "rstracer.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rstracer {

static float clamp_unit(float v) {   // Color::r/g/b setters, material.rs:23-27
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}
Color Color::init(float r, float g, float b) {
    Color c;
    c.r = clamp_unit(r);
    c.g = clamp_unit(g);
    c.b = clamp_unit(b);
    return c;
}

// ------------------------------------------------------------- tokenizer ----
// The reference pulls one byte at a time from Bytes<File> (parser/mod.rs:53-82);
// here the file is memory-mapped and tokens are (pointer, length) views, which
// matters for 100 MB+ scenes. Whitespace = char::is_whitespace on a Latin-1 byte.
static inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0; }

SceneParser::SceneParser(const std::string& path) {
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Error("file error: " + std::string(std::strerror(errno)) + ", path: " + path);   // parser/mod.rs:29-34
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
        ::close(fd);
        throw Error("file error: fstat failed, path: " + path);
    }
    size_ = (size_t)sb.st_size;
    if (size_ > 0) {
        void* p = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            ::close(fd);
            throw Error("file error: mmap failed, path: " + path);
        }
        data_ = (const char*)p;
        mapped_ = true;
    }
    ::close(fd);
}
SceneParser::~SceneParser() {
    if (mapped_) ::munmap((void*)data_, size_);
}

SceneParser::Tok SceneParser::next_tok() {
    if (peaked_) {
        peaked_ = false;
        return last_;
    }
    while (pos_ < size_ && is_space((unsigned char)data_[pos_])) ++pos_;
    const size_t start = pos_;
    while (pos_ < size_ && !is_space((unsigned char)data_[pos_])) ++pos_;
    Tok t{data_ + start, pos_ - start};
    if (pos_ >= size_) finished_ = true;   // the reference flags `finished` when the byte stream ends inside a token read
    else ++pos_;                           // consume the delimiter, as the byte loop does
    return t;
}
std::string SceneParser::next_token() {
    Tok t = next_tok();
    return std::string(t.p, t.n);
}
std::string SceneParser::peak() {
    if (!peaked_) {
        last_ = next_tok();
        peaked_ = true;
    }
    return std::string(last_.p, last_.n);
}
void SceneParser::expect(const char* token) {   // check_and_consume, parser/mod.rs:96-99
    Tok t = next_tok();
    if (std::strlen(token) != t.n || std::memcmp(token, t.p, t.n) != 0)
        throw Error("assertion failed: `(left == right)` (left: `" + std::string(t.p, t.n) + "`, right: `" + token + "`)");
}
float SceneParser::next_f32() {                 // next_num::<f32>, parser/mod.rs:84-90
    Tok t = next_tok();
    char buf[64];
    if (t.n == 0 || t.n >= sizeof(buf)) throw Error("Could not represent token as num: '" + std::string(t.p, t.n) + "'");
    std::memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    char* end = nullptr;
    const float v = std::strtof(buf, &end);
    if (end != buf + t.n) throw Error("Could not represent token as num: '" + std::string(buf) + "'");
    return v;
}
long SceneParser::next_int() {
    Tok t = next_tok();
    char buf[32];
    if (t.n == 0 || t.n >= sizeof(buf)) throw Error("Could not represent token as num: '" + std::string(t.p, t.n) + "'");
    std::memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    char* end = nullptr;
    const long v = std::strtol(buf, &end, 10);
    if (end != buf + t.n) throw Error("Could not represent token as num: '" + std::string(buf) + "'");
    return v;
}
float SceneParser::parse_f32(const char* name) { expect(name); return next_f32(); }
Vec3 SceneParser::parse_vec3(const char* name) {
    expect(name);
    Vec3 v;
    v.x = next_f32(); v.y = next_f32(); v.z = next_f32();
    return v;
}
Color SceneParser::parse_color(const char* name) {
    expect(name);
    const float r = next_f32(), g = next_f32(), b = next_f32();
    return Color::init(r, g, b);
}
bool SceneParser::parse_bool(const char* name, const char* flag) {
    expect(name);
    Tok t = next_tok();
    return std::strlen(flag) == t.n && std::memcmp(flag, t.p, t.n) == 0;
}
Light SceneParser::parse_light() {   // parser/mod.rs:124-148
    const std::string kw = next_token();
    expect("{");
    Light l;
    if (kw == "point_light") {
        l.kind = Light::Point;
        l.pos = parse_vec3("position");
        l.intensity = parse_color("color");
    } else if (kw == "area_light") {
        l.kind = Light::Area;
        l.pos = parse_vec3("position");
        l.max = parse_vec3("position");
        l.intensity = parse_color("color");
    } else if (kw == "directional_light") {
        l.kind = Light::Directional;
        l.pos = parse_vec3("direction");
        l.intensity = parse_color("color");
    } else {
        throw Error("LightType is not valid: " + kw);
    }
    expect("}");
    return l;
}
Material SceneParser::parse_material() {   // parser/mod.rs:150-165
    expect("material");
    expect("{");
    Material m;
    m.diffuse = parse_color("diffColor");
    m.ambient = parse_color("ambColor");
    m.specular = parse_color("specColor");
    m.emissive = parse_color("emisColor");
    m.shininess = parse_f32("shininess");
    m.transparency = parse_f32("ktran");
    expect("}");
    return m;
}
Sphere SceneParser::parse_sphere() {   // parser/mod.rs:167-193
    expect("sphere");
    expect("{");
    expect("name");
    next_tok();
    expect("numMaterials");
    long nm = next_int();
    Sphere s;
    for (; nm > 0; --nm) s.materials.push_back(parse_material());
    s.origin = parse_vec3("origin");
    s.radius = parse_f32("radius");
    s.xaxis = parse_vec3("xaxis");
    s.xlength = parse_f32("xlength");
    s.yaxis = parse_vec3("yaxis");
    s.ylength = parse_f32("ylength");
    s.zaxis = parse_vec3("zaxis");
    s.zlength = parse_f32("zlength");
    expect("}");
    return s;
}
Vertex SceneParser::parse_vertex(bool has_normal, bool has_material) {   // parser/mod.rs:195-214
    Vertex v;
    v.position = parse_vec3("pos");
    if (has_normal) {
        v.normal = parse_vec3("norm");
        v.has_normal = true;
    }
    if (has_material) {
        expect("materialIndex");
        v.mat_index = (uint32_t)next_int();
    }
    return v;
}
Poly SceneParser::parse_poly(bool has_normal, bool has_material) {   // parser/mod.rs:216-234
    expect("poly");
    expect("{");
    expect("numVertices");
    next_tok();   // always 3
    Poly p;
    for (int i = 0; i < 3; ++i) p.vertices[i] = parse_vertex(has_normal, has_material);
    p.vertex_material = has_material;
    p.vertex_normal = has_normal;
    expect("}");
    return p;
}
std::vector<Poly> SceneParser::parse_polyset() {   // parser/mod.rs:236-298
    expect("poly_set");
    expect("{");
    expect("name");
    next_tok();
    expect("numMaterials");
    long nm = next_int();
    std::vector<Material> materials;
    for (; nm > 0; --nm) materials.push_back(parse_material());
    expect("type");
    next_tok();
    const bool per_vertex_normal = parse_bool("normType", "PER_VERTEX_NORMAL");
    const bool material_binding = parse_bool("materialBinding", "PER_VERTEX_MATERIAL");
    expect("hasTextureCoords");
    next_tok();
    expect("rowSize");
    next_tok();
    expect("numPolys");
    long np = next_int();
    std::vector<Poly> set;
    // the count is only a hint for the allocation: a file cannot hold more polys than it has bytes for (a poly is at
    // least 16 bytes of text), whatever its header claims
    if (np > 0) set.reserve(std::min<size_t>((size_t)np, (size_ - pos_) / 16 + 1));
    auto mat = [&](uint32_t i) -> const Material& {
        if (i >= materials.size()) throw Error("index out of bounds: the len is " + std::to_string(materials.size()) + " but the index is " + std::to_string(i));
        return materials[i];
    };
    for (; np > 0; --np) {
        Poly poly = parse_poly(per_vertex_normal, material_binding);
        if (material_binding) {   // per-poly de-duplicated material list, parser/mod.rs:266-287
            const uint32_t i0 = poly.vertices[0].mat_index, i1 = poly.vertices[1].mat_index, i2 = poly.vertices[2].mat_index;
            poly.materials.push_back(mat(i0));
            poly.vertices[0].mat_index = (uint32_t)poly.materials.size() - 1;
            if (i1 != i0) {
                poly.materials.push_back(mat(i1));
                poly.vertices[1].mat_index = (uint32_t)poly.materials.size() - 1;
            } else {
                poly.vertices[1].mat_index = 0;
            }
            if (i2 != i1 && i2 != i0) {
                poly.materials.push_back(mat(i2));
                poly.vertices[2].mat_index = (uint32_t)poly.materials.size() - 1;
            } else if (i2 == i1 && i2 != i0) {
                poly.vertices[2].mat_index = 1;
            } else {
                poly.vertices[2].mat_index = 0;
            }
        } else {
            poly.materials.push_back(mat(0));
        }
        set.push_back(std::move(poly));
    }
    expect("}");
    return set;
}
Camera SceneParser::parse_camera() {   // parser/mod.rs:300-312
    expect("camera");
    expect("{");
    Camera c;
    c.pos = parse_vec3("position");
    c.view_dir = parse_vec3("viewDirection");
    c.focal_dist = parse_f32("focalDistance");
    c.ortho_up = parse_vec3("orthoUp");
    c.vertical_fov = parse_f32("verticalFOV");
    expect("}");
    return c;
}
Scene SceneParser::parse_scene() {   // parser/mod.rs:314-346
    expect("Composer");
    expect("format");
    expect("2.1");
    expect("ascii");
    Scene scene;
    std::string tkn = peak();
    while (has_next_token()) {
        if (tkn == "camera") {
            scene.camera = parse_camera();
        } else if (tkn == "sphere") {
            Primitive p;
            p.kind = Primitive::SphereKind;
            p.sphere = parse_sphere();
            scene.primitives.push_back(std::move(p));
        } else if (tkn == "poly_set") {
            std::vector<Poly> set = parse_polyset();
            while (!set.empty()) {   // `polyset.pop()`: polys enter the scene in reverse file order (parser/mod.rs:333-338)
                Primitive p;
                p.kind = Primitive::PolyKind;
                p.poly = std::move(set.back());
                set.pop_back();
                scene.primitives.push_back(std::move(p));
            }
        } else if (tkn.size() >= 5 && tkn.compare(tkn.size() - 5, 5, "light") == 0) {
            scene.lights.push_back(parse_light());
        } else {
            throw Error("Unexpected token: " + tkn);
        }
        tkn = peak();
    }
    return scene;
}

// -------------------------------------------------------------- flatten ----
static void put3(std::vector<float>& v, const Vec3& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }
static rtx_material to_rtx(const Material& m) {
    rtx_material o;
    const Color* src[4] = {&m.diffuse, &m.ambient, &m.specular, &m.emissive};
    float* dst[4] = {o.diffuse, o.ambient, o.specular, o.emissive};
    for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->r; dst[k][1] = src[k]->g; dst[k][2] = src[k]->b; }
    o.shininess = m.shininess;
    o.transparency = m.transparency;
    return o;
}
static rtx_light to_rtx(const Light& l) {
    rtx_light o{};
    o.type = (uint32_t)l.kind;
    o.a[0] = l.pos.x; o.a[1] = l.pos.y; o.a[2] = l.pos.z;
    o.b[0] = l.max.x; o.b[1] = l.max.y; o.b[2] = l.max.z;
    o.color[0] = l.intensity.r; o.color[1] = l.intensity.g; o.color[2] = l.intensity.b;
    return o;
}
static rtx_camera to_rtx(const Camera& c) {
    rtx_camera o{};
    o.pos[0] = c.pos.x; o.pos[1] = c.pos.y; o.pos[2] = c.pos.z;
    o.view_dir[0] = c.view_dir.x; o.view_dir[1] = c.view_dir.y; o.view_dir[2] = c.view_dir.z;
    o.ortho_up[0] = c.ortho_up.x; o.ortho_up[1] = c.ortho_up.y; o.ortho_up[2] = c.ortho_up.z;
    o.focal_dist = c.focal_dist;
    o.vertical_fov = c.vertical_fov;
    return o;
}
Camera camera_of(const FlatScene& flat) {
    const rtx_camera& c = flat.camera;
    Camera o;
    o.pos = Vec3{c.pos[0], c.pos[1], c.pos[2]};
    o.view_dir = Vec3{c.view_dir[0], c.view_dir[1], c.view_dir[2]};
    o.ortho_up = Vec3{c.ortho_up[0], c.ortho_up[1], c.ortho_up[2]};
    o.focal_dist = c.focal_dist;
    o.vertical_fov = c.vertical_fov;
    return o;
}
std::vector<Light> lights_of(const FlatScene& flat) {
    std::vector<Light> out;
    for (const rtx_light& l : flat.lights) {
        Light o;
        o.kind = (Light::Kind)l.type;
        o.pos = Vec3{l.a[0], l.a[1], l.a[2]};
        o.max = Vec3{l.b[0], l.b[1], l.b[2]};
        o.intensity.r = l.color[0]; o.intensity.g = l.color[1]; o.intensity.b = l.color[2];   // already clamped
        out.push_back(o);
    }
    return out;
}

// ------------------------------------------------- file -> arrays, one pass ----
// A poly_set (parser/mod.rs:236-298) straight into the triangle arrays. The set's polys enter the scene in
// reverse file order (`polyset.pop()`, parser/mod.rs:333-338), so the set is appended in file order and its
// range of the arrays is reversed at the end. A poly's material list (parser/mod.rs:266-291) is positional:
// [vertex 0's material, vertex 1's if it differs, vertex 2's if it differs from both]; without per-vertex
// binding it is [materials[0]]. The set's materials are appended to the table once, on first use.
void SceneParser::flat_polyset(FlatScene& f, uint32_t& n_prims) {
    expect("poly_set");
    expect("{");
    expect("name");
    next_tok();
    expect("numMaterials");
    long nm = next_int();
    std::vector<rtx_material> set_mats;
    for (; nm > 0; --nm) set_mats.push_back(to_rtx(parse_material()));
    std::vector<uint32_t> table(set_mats.size(), RTX_MATERIAL_NONE);   // index in the scene's material table, once used
    auto mat = [&](uint32_t i) -> uint32_t {
        if (i >= set_mats.size()) throw Error("index out of bounds: the len is " + std::to_string(set_mats.size()) + " but the index is " + std::to_string(i));
        if (table[i] == RTX_MATERIAL_NONE) {
            table[i] = (uint32_t)f.materials.size();
            f.materials.push_back(set_mats[i]);
        }
        return table[i];
    };
    expect("type");
    next_tok();
    const bool vertex_normal = parse_bool("normType", "PER_VERTEX_NORMAL");
    const bool vertex_material = parse_bool("materialBinding", "PER_VERTEX_MATERIAL");
    expect("hasTextureCoords");
    next_tok();
    expect("rowSize");
    next_tok();
    expect("numPolys");
    long np = next_int();
    const size_t first = f.tri_flags.size();
    if (np > 0) {   // a hint only: bounded by what the rest of the file can hold (a poly is at least 16 bytes of text)
        const size_t hint = std::min<size_t>((size_t)np, (size_ - pos_) / 16 + 1);
        f.tri_pos.reserve(f.tri_pos.size() + hint * 9);
        f.tri_normals.reserve(f.tri_normals.size() + hint * 9);
        f.tri_mat.reserve(f.tri_mat.size() + hint * 3);
        f.tri_flags.reserve(first + hint);
    }
    const uint8_t flags = (uint8_t)((vertex_normal ? RTX_TRI_VERTEX_NORMAL : 0) | (vertex_material ? RTX_TRI_VERTEX_MATERIAL : 0));
    for (; np > 0; --np) {
        expect("poly");                     // parse_poly, parser/mod.rs:216-234
        expect("{");
        expect("numVertices");
        next_tok();                         // always 3
        float pos[9], nrm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t mi[3] = {0, 0, 0};
        for (int v = 0; v < 3; ++v) {       // parse_vertex, parser/mod.rs:195-214
            expect("pos");
            for (int k = 0; k < 3; ++k) pos[3 * v + k] = next_f32();
            if (vertex_normal) {
                expect("norm");
                for (int k = 0; k < 3; ++k) nrm[3 * v + k] = next_f32();
            }
            if (vertex_material) {
                expect("materialIndex");
                mi[v] = (uint32_t)next_int();
            }
        }
        expect("}");
        uint32_t list[3] = {RTX_MATERIAL_NONE, RTX_MATERIAL_NONE, RTX_MATERIAL_NONE};
        int n_list = 0;
        if (vertex_material) {
            list[n_list++] = mat(mi[0]);
            if (mi[1] != mi[0]) list[n_list++] = mat(mi[1]);
            if (mi[2] != mi[1] && mi[2] != mi[0]) list[n_list++] = mat(mi[2]);
        } else {
            list[n_list++] = mat(0);
        }
        f.tri_pos.insert(f.tri_pos.end(), pos, pos + 9);
        f.tri_normals.insert(f.tri_normals.end(), nrm, nrm + 9);
        f.tri_mat.insert(f.tri_mat.end(), list, list + 3);
        f.tri_flags.push_back(flags);
    }
    expect("}");
    const size_t count = f.tri_flags.size() - first;
    for (size_t a = 0, b = count; a + 1 < b; ++a) {   // reverse the set's range
        --b;
        std::swap_ranges(f.tri_pos.begin() + (first + a) * 9, f.tri_pos.begin() + (first + a) * 9 + 9, f.tri_pos.begin() + (first + b) * 9);
        std::swap_ranges(f.tri_normals.begin() + (first + a) * 9, f.tri_normals.begin() + (first + a) * 9 + 9, f.tri_normals.begin() + (first + b) * 9);
        std::swap_ranges(f.tri_mat.begin() + (first + a) * 3, f.tri_mat.begin() + (first + a) * 3 + 3, f.tri_mat.begin() + (first + b) * 3);
    }                                                 // (the flags of a set are all the same)
    f.tri_prim_index.reserve(first + count);
    for (size_t k = 0; k < count; ++k) f.tri_prim_index.push_back(n_prims++);
}
FlatScene SceneParser::parse_flat() {   // parse_scene (parser/mod.rs:314-346) emitting arrays
    expect("Composer");
    expect("format");
    expect("2.1");
    expect("ascii");
    FlatScene f;
    uint32_t n_prims = 0;
    std::string tkn = peak();
    while (has_next_token()) {
        if (tkn == "camera") {
            f.camera = to_rtx(parse_camera());
        } else if (tkn == "sphere") {
            const Sphere s = parse_sphere();
            put3(f.sph_center_radius, s.origin);
            f.sph_center_radius.push_back(s.radius);
            if (!s.materials.empty()) {
                const rtx_material m = to_rtx(s.materials[0]);
                // spheres that follow each other with the same material share its entry
                if (!f.sph_mat.empty() && f.sph_mat.back() != RTX_MATERIAL_NONE && std::memcmp(&f.materials[f.sph_mat.back()], &m, sizeof(m)) == 0) {
                    f.sph_mat.push_back(f.sph_mat.back());
                } else {
                    f.sph_mat.push_back((uint32_t)f.materials.size());
                    f.materials.push_back(m);
                }
            } else {
                f.sph_mat.push_back(RTX_MATERIAL_NONE);
            }
            f.sph_prim_index.push_back(n_prims++);
        } else if (tkn == "poly_set") {
            flat_polyset(f, n_prims);
        } else if (tkn.size() >= 5 && tkn.compare(tkn.size() - 5, 5, "light") == 0) {
            f.lights.push_back(to_rtx(parse_light()));
        } else {
            throw Error("Unexpected token: " + tkn);
        }
        tkn = peak();
    }
    return f;
}

FlatScene flatten(const Scene& scene) {
    FlatScene f;
    for (size_t id = 0; id < scene.primitives.size(); ++id) {
        const Primitive& p = scene.primitives[id];
        if (p.kind == Primitive::PolyKind) {
            const Poly& q = p.poly;
            for (int k = 0; k < 3; ++k) put3(f.tri_pos, q.vertices[k].position);
            for (int k = 0; k < 3; ++k) put3(f.tri_normals, q.vertices[k].normal);
            for (size_t k = 0; k < 3; ++k) {   // poly.materials[0..2]; each poly's list is appended to the global table
                if (k < q.materials.size()) {
                    f.tri_mat.push_back((uint32_t)f.materials.size());
                    f.materials.push_back(to_rtx(q.materials[k]));
                } else {
                    f.tri_mat.push_back(RTX_MATERIAL_NONE);
                }
            }
            f.tri_flags.push_back((uint8_t)((q.vertex_normal ? RTX_TRI_VERTEX_NORMAL : 0) | (q.vertex_material ? RTX_TRI_VERTEX_MATERIAL : 0)));
            f.tri_prim_index.push_back((uint32_t)id);
        } else {
            const Sphere& s = p.sphere;
            put3(f.sph_center_radius, s.origin);
            f.sph_center_radius.push_back(s.radius);
            if (!s.materials.empty()) {
                f.sph_mat.push_back((uint32_t)f.materials.size());
                f.materials.push_back(to_rtx(s.materials[0]));
            } else {
                f.sph_mat.push_back(RTX_MATERIAL_NONE);
            }
            f.sph_prim_index.push_back((uint32_t)id);
        }
    }
    for (const Light& l : scene.lights) f.lights.push_back(to_rtx(l));
    f.camera = to_rtx(scene.camera);
    return f;
}
rtx_scene_desc FlatScene::desc() const {
    rtx_scene_desc d{};
    d.n_tris = (uint32_t)tri_flags.size();
    d.tri_pos = tri_pos.data();
    d.tri_normals = tri_normals.data();
    d.tri_mat = tri_mat.data();
    d.tri_flags = tri_flags.data();
    d.tri_prim_index = tri_prim_index.data();
    d.n_spheres = (uint32_t)sph_mat.size();
    d.sph_center_radius = sph_center_radius.data();
    d.sph_mat = sph_mat.data();
    d.sph_prim_index = sph_prim_index.data();
    d.n_materials = (uint32_t)materials.size();
    d.materials = materials.data();
    d.n_lights = (uint32_t)lights.size();
    d.lights = lights.data();
    d.camera = camera;
    return d;
}

// ------------------------------------------------------------- GpuScene ----
static void check(rtx_context* ctx, int rc, const char* what) {
    if (rc != RTX_OK) throw Error(std::string(what) + ": " + rtx_last_error(ctx));
}
GpuScene::GpuScene(const Scene& scene, int device, bool force_bvh) : camera_(scene.camera), lights_(scene.lights) {
    upload(flatten(scene), device, force_bvh);
}
GpuScene::GpuScene(const FlatScene& flat, int device, bool force_bvh) : camera_(camera_of(flat)), lights_(lights_of(flat)) {
    upload(flat, device, force_bvh);
}
void GpuScene::upload(const FlatScene& flat, int device, bool force_bvh) {
    int rc = rtx_create(device, &ctx_);
    if (rc != RTX_OK) throw Error(std::string("rtx_create: ") + rtx_last_error(nullptr));
    if (force_bvh) (void)rtx_set_option(ctx_, "force_mode", 2);
    const rtx_scene_desc d = flat.desc();
    rc = rtx_scene_set(ctx_, &d);
    if (rc == RTX_OK) rc = rtx_build(ctx_, &build_);
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error(ctx_);
        rtx_destroy(ctx_);
        ctx_ = nullptr;
        throw Error("GpuScene: " + msg);
    }
}
GpuScene::~GpuScene() { rtx_destroy(ctx_); }
std::vector<SceneIntersection> GpuScene::intersects(const std::vector<Ray>& rays) const {
    std::vector<float> o(rays.size() * 3), d(rays.size() * 3), t(rays.size());
    std::vector<int32_t> prim(rays.size());
    for (size_t i = 0; i < rays.size(); ++i) {
        o[3 * i] = rays[i].ori.x; o[3 * i + 1] = rays[i].ori.y; o[3 * i + 2] = rays[i].ori.z;
        d[3 * i] = rays[i].dir.x; d[3 * i + 1] = rays[i].dir.y; d[3 * i + 2] = rays[i].dir.z;
    }
    check(ctx_, rtx_intersect_batch(ctx_, rays.size(), o.data(), d.data(), prim.data(), t.data()), "rtx_intersect_batch");
    std::vector<SceneIntersection> out(rays.size());
    for (size_t i = 0; i < rays.size(); ++i) {
        out[i].intersected = prim[i] >= 0;
        out[i].primitive = prim[i];
        out[i].point = t[i];
    }
    return out;
}
SceneIntersection GpuScene::intersects(const Ray& ray) const { return intersects(std::vector<Ray>{ray})[0]; }

// ---------------------------------------------------------------- Image ----
static void put_le(std::vector<uint8_t>& v, uint32_t x, int bytes) {
    for (int i = 0; i < bytes; ++i) v.push_back((uint8_t)(x >> (8 * i)));
}
void Image::save(const std::string& path) const {
    const uint32_t row = (width * 3 + 3) & ~3u;
    std::vector<uint8_t> out;
    out.reserve(54 + (size_t)row * height);
    out.push_back('B'); out.push_back('M');
    put_le(out, 54 + row * height, 4); put_le(out, 0, 4); put_le(out, 54, 4);
    put_le(out, 40, 4); put_le(out, width, 4); put_le(out, height, 4); put_le(out, 1, 2); put_le(out, 24, 2);
    put_le(out, 0, 4); put_le(out, row * height, 4); put_le(out, 2835, 4); put_le(out, 2835, 4); put_le(out, 0, 4); put_le(out, 0, 4);
    for (uint32_t y = height; y-- > 0;) {   // BMP rows run bottom-up, pixels are B,G,R
        for (uint32_t x = 0; x < width; ++x) {
            const uint8_t* p = &rgb[((size_t)y * width + x) * 3];
            out.push_back(p[2]); out.push_back(p[1]); out.push_back(p[0]);
        }
        for (uint32_t k = width * 3; k < row; ++k) out.push_back(0);
    }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot open " + path);
    const size_t n = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (n != out.size()) throw Error("short write to " + path);
}
Image Image::load(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw Error("cannot open " + path);
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    auto le = [&](size_t off, int bytes) { uint32_t v = 0; for (int i = 0; i < bytes; ++i) v |= (uint32_t)buf.at(off + i) << (8 * i); return v; };
    if (buf.size() < 54 || buf[0] != 'B' || buf[1] != 'M' || le(28, 2) != 24 || le(30, 4) != 0) throw Error("not a 24-bpp uncompressed BMP: " + path);
    Image img;
    const uint32_t off = le(10, 4);
    img.width = le(18, 4);
    img.height = le(22, 4);
    const uint64_t row = ((uint64_t)img.width * 3 + 3) & ~(uint64_t)3;      // 64-bit: a header may claim any width
    if (img.width == 0 || img.height == 0 || img.width > (1u << 24) || img.height > (1u << 24) || off < 54 || off > buf.size() ||
        row * img.height > buf.size() - off)
        throw Error("truncated BMP: " + path);
    img.rgb.resize((size_t)img.width * img.height * 3);
    for (uint32_t y = 0; y < img.height; ++y)
        for (uint32_t x = 0; x < img.width; ++x) {
            const uint8_t* p = &buf[off + (size_t)(img.height - 1 - y) * row + x * 3];
            uint8_t* q = &img.rgb[((size_t)y * img.width + x) * 3];
            q[0] = p[2]; q[1] = p[1]; q[2] = p[0];
        }
    return img;
}

// ------------------------------------------------------------ RayTracer ----
RayTracer RayTracer::init(uint32_t width, uint32_t height, size_t depth, size_t num_samples) {
    RayTracer rt;
    rt.width = width;
    rt.height = height;
    rt.depth = depth;
    rt.num_samples = num_samples;
    return rt;
}
void RayTracer::set_scene(std::unique_ptr<GpuScene> scene) { scene_ = std::move(scene); }
Image RayTracer::trace_rays() {
    if (!scene_) throw Error("RayTracer has not been assigned any Scene");   // lib.rs:275
    Image img;
    img.width = width;
    img.height = height;
    img.rgb.assign((size_t)width * height * 3, 0);
    check(scene_->context(), rtx_render(scene_->context(), width, height, (uint32_t)depth, (uint32_t)num_samples, img.rgb.data(), &stats_),
          "rtx_render");
    return img;
}

}  // namespace rstracer

// ------------------------------------------------------------------ C API ---
// Thin C view of the parser + flattening for bindings (ctypes); the arrays stay
// owned by the handle.
extern "C" {
struct rst_scene {
    rstracer::FlatScene flat;
    rtx_scene_desc desc;
};
static thread_local std::string g_rst_error;
const char* rst_last_error(void) { return g_rst_error.c_str(); }
rst_scene* rst_scene_parse(const char* path) {
    try {
        rstracer::SceneParser parser(path ? path : "");
        auto* s = new rst_scene();
        s->flat = parser.parse_flat();
        s->desc = s->flat.desc();
        return s;
    } catch (const std::exception& e) {
        g_rst_error = e.what();
        return nullptr;
    }
}
/* The same scene through the object graph (SceneParser::parse_scene + flatten): what the tests hold parse_flat against. */
rst_scene* rst_scene_parse_objects(const char* path) {
    try {
        rstracer::SceneParser parser(path ? path : "");
        auto* s = new rst_scene();
        s->flat = rstracer::flatten(parser.parse_scene());
        s->desc = s->flat.desc();
        return s;
    } catch (const std::exception& e) {
        g_rst_error = e.what();
        return nullptr;
    }
}
const rtx_scene_desc* rst_scene_desc(const rst_scene* s) { return s ? &s->desc : nullptr; }
void rst_scene_free(rst_scene* s) { delete s; }
/* Whole CLI pipeline in one call: parse -> GpuScene -> RayTracer -> BMP. */
int rst_render_file(const char* scene_path, const char* out_bmp, uint32_t width, uint32_t height, uint32_t depth,
                    uint32_t samples, int device) {
    try {
        rstracer::SceneParser parser(scene_path ? scene_path : "");
        const rstracer::FlatScene scene = parser.parse_flat();
        rstracer::RayTracer rt = rstracer::RayTracer::init(width, height, depth, samples);
        rt.set_scene(std::make_unique<rstracer::GpuScene>(scene, device));
        rt.trace_rays().save(out_bmp ? out_bmp : "img.bmp");
        return 0;
    } catch (const std::exception& e) {
        g_rst_error = e.what();
        return 1;
    }
}
}
