// san_parse_files.cpp -- parses every file named on stdin (one path per line) with both parser paths of rstracer.cpp and
// flattens what parses; built with -fsanitize=address,undefined by tests/test_sanitizers.py for files produced by a
// token-level mutator (numbers replaced by huge, negative, fractional and non-numeric tokens, lines dropped, doubled and
// swapped, blocks cut). A file may parse or be rejected with rstracer::Error; anything else is a finding.
#include <cstdio>
#include <iostream>
#include <string>

#include "rstracer.hpp"

int main() {
    std::string path;
    unsigned long parsed = 0, rejected = 0, both_differ = 0;
    while (std::getline(std::cin, path)) {
        if (path.empty()) continue;
        int a = 0, b = 0;
        size_t na = 0, nb = 0;
        try {
            rstracer::SceneParser p(path);
            const rstracer::FlatScene f = rstracer::flatten(p.parse_scene());
            na = f.tri_pos.size() + f.sph_center_radius.size();
        } catch (const rstracer::Error&) { a = 1; }
        try {
            rstracer::SceneParser p(path);
            const rstracer::FlatScene f = p.parse_flat();
            nb = f.tri_pos.size() + f.sph_center_radius.size();
            (void)f.desc();
        } catch (const rstracer::Error&) { b = 1; }
        if (a != b || (a == 0 && na != nb)) { ++both_differ; std::printf("paths disagree on %s: %d %d %zu %zu\n", path.c_str(), a, b, na, nb); }
        (a ? rejected : parsed)++;
    }
    std::printf("parsed %lu rejected %lu disagreements %lu\n", parsed, rejected, both_differ);
    return both_differ ? 1 : 0;
}
