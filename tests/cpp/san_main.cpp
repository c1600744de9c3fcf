// san_main.cpp -- drives the CPU-side code that reads untrusted files (the product's scene parser and BMP reader,
// rstracer.cpp) and the CPU oracle (oracle/rt_oracle.cpp) in ONE executable built with -fsanitize=address,undefined
// (tests/test_sanitizers.py compiles and runs it; never on the GPU: no device call is made).
//
//   san_main <scenes dir> <parser testdata prefix> <output dir>
//
// 1. every bundled scene through both parser paths (objects + flatten, one-pass) -- arrays must agree;
// 2. malformed inputs: the reference's error cases, empty files, every prefix of a scene cut at 211-byte steps,
//    scenes with bytes flipped -- each must end in rstracer::Error (or parse), never in a sanitizer report;
// 3. BMP writer / reader round trip, and the reader on truncated and lying headers;
// 4. the oracle: test05 and test10 at 100x100 (written to <output dir>/<name>.rgb for the caller to compare with the
//    golden frames), a batch of rays through the exhaustive loop and through the restated reference BVH.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rstracer.hpp"
#include "rt_oracle.h"

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { std::printf("FAIL: %s\n", what); ++failures; } } while (0)

static std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void spit(const std::string& path, const std::string& data) {
    std::ofstream f(path, std::ios::binary);
    f.write(data.data(), (std::streamsize)data.size());
}
// 0: parsed, 1: rejected with rstracer::Error
static int try_parse(const std::string& path, bool objects) {
    try {
        rstracer::SceneParser p(path);
        if (objects) (void)rstracer::flatten(p.parse_scene());
        else (void)p.parse_flat();
        return 0;
    } catch (const rstracer::Error&) {
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { std::printf("usage: san_main <scenes dir> <testdata prefix> <out dir>\n"); return 2; }
    const std::string scenes = argv[1], prefix = argv[2], out = argv[3];
    const char* names[] = {"sphere", "test00", "test01", "test02", "test03", "test04", "test05", "test06", "test07",
                           "test08", "test09", "test10", "test11", "area"};
    // 1. both parser paths on every bundled scene
    for (const char* n : names) {
        const std::string path = scenes + "/" + n + ".ascii";
        rstracer::SceneParser a(path), b(path);
        const rstracer::FlatScene fa = rstracer::flatten(a.parse_scene()), fb = b.parse_flat();
        CHECK(fa.tri_pos == fb.tri_pos && fa.tri_prim_index == fb.tri_prim_index && fa.sph_center_radius == fb.sph_center_radius &&
              fa.tri_flags == fb.tri_flags && fa.lights.size() == fb.lights.size(), n);
        const rtx_scene_desc d = fb.desc();
        CHECK(d.n_tris * 9 == fb.tri_pos.size() && d.n_spheres * 4 == fb.sph_center_radius.size(), "descriptor sizes");
    }
    // the reference's parser fixtures, whole
    for (const char* f : {"camera.txt", "light.txt", "material.txt", "polyset.txt", "scene.txt", "sphere.txt", "vec3.txt", "color.txt"}) {
        const std::string path = prefix + f;
        if (slurp(path).empty()) continue;
        (void)try_parse(path, true);
        (void)try_parse(path, false);
    }
    // 2. malformed inputs
    {
        const std::string bad = out + "/bad.ascii";
        const char* cases[] = {"", "Composer", "Composer format 2.1 ascii\n", "Composer format 2.1 ascii\ncamera {\n  position 0 0 zero\n}\n",
                               "Composer format 2.1 ascii\nteapot {\n}\n", "Composer format 2.1 ascii\ncamera {",
                               "Composer format 2.1 ascii\npoly_set {\n name \"x\"\n numMaterials 99999999\n",
                               "Composer format 2.1 ascii\npoly_set {\n name \"x\"\n numMaterials 0\n type POLYSET_TRI_MESH\n normType PER_FACE_NORMAL\n materialBinding PER_OBJECT_MATERIAL\n hasTextureCoords FALSE\n rowSize 0\n numPolys 4000000000\n",
                               "Composer format 2.1 ascii\nsphere {\n name \"s\"\n numMaterials -1\n"};
        for (const char* c : cases) {
            spit(bad, c);
            const int r1 = try_parse(bad, true), r2 = try_parse(bad, false);
            CHECK(r1 == r2, "both parser paths accept / reject the same malformed file");
        }
        CHECK(try_parse(out + "/does-not-exist.ascii", false) == 1, "missing file is an Error");
        const std::string whole = slurp(scenes + "/test06.ascii");     // per-vertex normals and materials
        for (size_t cut = 0; cut < whole.size(); cut += 211) {
            spit(bad, whole.substr(0, cut));
            const int r1 = try_parse(bad, true), r2 = try_parse(bad, false);
            CHECK(r1 == r2, "prefix of test06: both paths agree");
        }
        unsigned long long seed = 0x5EEDull;
        for (int k = 0; k < 200; ++k) {      // a few bytes replaced at random places
            std::string m = whole.substr(0, 60000);
            for (int j = 0; j < 3; ++j) {
                seed = seed * 6364136223846793005ull + 1442695040888963407ull;
                m[(size_t)(seed >> 33) % m.size()] = " 0-{}e.\nx"[(seed >> 20) % 9];
            }
            spit(bad, m);
            const int r1 = try_parse(bad, true), r2 = try_parse(bad, false);
            CHECK(r1 == r2, "mutated test06: both paths agree");
        }
    }
    // 3. BMP round trip and hostile headers
    {
        rstracer::Image img;
        img.width = 5; img.height = 3;       // row padding: 15 bytes -> 16
        img.rgb.resize(45);
        for (size_t i = 0; i < img.rgb.size(); ++i) img.rgb[i] = (uint8_t)(i * 7 + 1);
        const std::string path = out + "/t.bmp";
        img.save(path);
        const rstracer::Image back = rstracer::Image::load(path);
        CHECK(back.width == 5 && back.height == 3 && back.rgb == img.rgb, "BMP round trip");
        const std::string bytes = slurp(path);
        for (size_t cut : {(size_t)0, (size_t)1, (size_t)10, (size_t)25, (size_t)53, (size_t)60, bytes.size() - 1}) {
            spit(path, bytes.substr(0, cut));
            try { (void)rstracer::Image::load(path); CHECK(cut == bytes.size(), "truncated BMP must be rejected"); } catch (const rstracer::Error&) {}
        }
        std::string lie = bytes;
        lie[18] = (char)0xff; lie[19] = (char)0xff; lie[20] = (char)0xff; lie[21] = 0x7f;      // width 2^31 - 1
        spit(path, lie);
        try { (void)rstracer::Image::load(path); CHECK(false, "BMP with a lying width must be rejected"); } catch (const rstracer::Error&) {}
        lie = bytes;
        lie[10] = (char)0xf0; lie[11] = (char)0xff; lie[12] = (char)0xff; lie[13] = 0x7f;      // pixel offset far past the end
        spit(path, lie);
        try { (void)rstracer::Image::load(path); CHECK(false, "BMP with a lying offset must be rejected"); } catch (const rstracer::Error&) {}
    }
    // 4. the oracle
    for (const char* n : {"test05", "test10"}) {
        orc_scene* s = orc_scene_parse((scenes + "/" + n + ".ascii").c_str());
        CHECK(s != nullptr, "oracle parses the scene");
        if (!s) continue;
        const uint32_t w = 100, h = 100;
        std::vector<uint8_t> rgb(w * h * 3), panic(w * h);
        std::vector<int32_t> prim(w * h);
        std::vector<float> t(w * h), dir(w * h * 3);
        orc_stats st{};
        orc_render(s, w, h, 10, 10, 1, 1, 8, 0, rgb.data(), prim.data(), t.data(), panic.data(), dir.data(), &st);
        spit(out + "/" + n + ".rgb", std::string((const char*)rgb.data(), rgb.size()));
        CHECK(st.primary_rays == w * h, "oracle traced every pixel");
        // the same rays as a batch, exhaustive and through the restated reference BVH
        uint32_t np = 0, nt = 0, ns = 0, nm = 0, nl = 0;
        orc_scene_counts(s, &np, &nt, &ns, &nm, &nl);
        rtx_camera cam{};
        orc_scene_flatten(s, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &cam);
        std::vector<float> ori(w * h * 3);
        for (uint32_t i = 0; i < w * h; ++i) for (int k = 0; k < 3; ++k) ori[i * 3 + k] = cam.pos[k];
        std::vector<int32_t> p2(w * h), p3(w * h);
        std::vector<float> t2(w * h), t3(w * h);
        orc_intersect_batch(s, w * h, ori.data(), dir.data(), p2.data(), t2.data(), 0);
        orc_intersect_batch(s, w * h, ori.data(), dir.data(), p3.data(), t3.data(), 1);
        CHECK(p2 == prim, "batched exhaustive loop equals the frame's primary hits");
        size_t same = 0;
        for (uint32_t i = 0; i < w * h; ++i) same += p3[i] == p2[i];
        CHECK(same * 100 >= (size_t)w * h * 99, "reference BVH agrees with the loop except on ties");
        // from a descriptor (the path bench.py's CPU baseline takes)
        std::vector<float> tp((size_t)nt * 9), tn((size_t)nt * 9), sc((size_t)ns * 4);
        std::vector<uint32_t> tm((size_t)nt * 3), ti(nt), sm(ns), si(ns);
        std::vector<uint8_t> tf(nt);
        std::vector<rtx_material> mats(nm);
        std::vector<rtx_light> lights(nl);
        orc_scene_flatten(s, tp.data(), tn.data(), tm.data(), tf.data(), ti.data(), sc.data(), sm.data(), si.data(), mats.data(), lights.data(), &cam);
        rtx_scene_desc d{};
        d.n_tris = nt; d.tri_pos = tp.data(); d.tri_normals = tn.data(); d.tri_mat = tm.data(); d.tri_flags = tf.data(); d.tri_prim_index = ti.data();
        d.n_spheres = ns; d.sph_center_radius = sc.data(); d.sph_mat = sm.data(); d.sph_prim_index = si.data();
        d.n_materials = nm; d.materials = mats.data(); d.n_lights = nl; d.lights = lights.data(); d.camera = cam;
        orc_scene* s2 = orc_scene_from_desc(&d);
        CHECK(s2 != nullptr, "oracle scene from a descriptor");
        if (s2) {
            std::vector<uint8_t> rgb2(w * h * 3);
            orc_render(s2, w, h, 10, 10, 1, 1, 8, 0, rgb2.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
            CHECK(rgb2 == rgb, "descriptor scene renders the same frame");
            orc_scene_free(s2);
        }
        orc_scene_free(s);
    }
    std::printf(failures ? "%d checks failed\n" : "sanitizer drive: all checks passed\n", failures);
    return failures ? 1 : 0;
}
