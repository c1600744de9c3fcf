#!/usr/bin/env python3
"""Token-level mutation fuzzing of the scene parser (CPU only). Writes mutated copies of the small bundled scenes into a
directory and prints their paths; tests/cpp/san_parse_files.cpp, built with ASan + UBSan, parses them all.
usage: python tools/fuzz_parser.py <out dir> <count> [seed]  |  san_parse_files"""
import os, random, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, count = sys.argv[1], int(sys.argv[2])
rnd = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
os.makedirs(out, exist_ok=True)
bases = [open(os.path.join(ROOT, "scenes", n + ".ascii")).read() for n in ("sphere", "test00", "test01", "test02", "test03", "test04", "test05", "test06", "area")]
NUMS = ["0", "-1", "1", "2", "3", "4294967295", "4294967296", "99999999999999999999", "-0", "1e39", "-1e39", "1e-46", "nan", "inf", "-inf", "0x10", "1.5",
        "", "{", "}", "TRUE", "FALSE", "abc", "1e", "+", "-", ".", "1..2", "999999999", "2147483648", "-2147483649"]
WORDS = ["camera", "point_light", "directional_light", "area_light", "sphere", "poly_set", "material", "poly", "numPolys", "numMaterials", "numVertices",
         "pos", "norm", "materialIndex", "type", "normType", "materialBinding", "PER_VERTEX_NORMAL", "PER_FACE_NORMAL", "PER_VERTEX_MATERIAL", "PER_OBJECT_MATERIAL",
         "POLYSET_TRI_MESH", "POLYSET_FACE_SET", "POLYSET_QUAD_MESH", "name", "NULL", "rowSize", "hasTextureCoords", "Composer", "format", "2.1", "ascii"]
num = re.compile(r"-?\d+\.?\d*(?:e-?\d+)?")
for i in range(count):
    s = rnd.choice(bases)
    for _ in range(rnd.randint(1, 6)):
        k = rnd.random()
        if k < 0.4:       # replace one number
            m = list(num.finditer(s))
            if m:
                x = rnd.choice(m)
                s = s[:x.start()] + rnd.choice(NUMS) + s[x.end():]
        elif k < 0.55:    # drop, double or swap lines
            lines = s.split("\n")
            j = rnd.randrange(len(lines))
            op = rnd.random()
            if op < 0.4: del lines[j]
            elif op < 0.7: lines.insert(j, lines[j])
            else:
                j2 = rnd.randrange(len(lines)); lines[j], lines[j2] = lines[j2], lines[j]
            s = "\n".join(lines)
        elif k < 0.7:     # replace a word
            toks = list(re.finditer(r"[A-Za-z_]+", s))
            if toks:
                x = rnd.choice(toks)
                s = s[:x.start()] + rnd.choice(WORDS + NUMS) + s[x.end():]
        elif k < 0.8:     # cut
            j = rnd.randrange(len(s) + 1)
            s = s[:j]
        elif k < 0.9:     # splice a block of another scene
            o = rnd.choice(bases)
            a, b = sorted((rnd.randrange(len(o)), rnd.randrange(len(o))))
            j = rnd.randrange(len(s) + 1)
            s = s[:j] + o[a:b] + s[j:]
        else:             # odd bytes
            j = rnd.randrange(len(s) + 1)
            s = s[:j] + rnd.choice(["\0", "\xff", "\r", "\t\t", "/*", "#", "\n\n\n", " " * 1000, "9" * 400]) + s[j:]
    p = os.path.join(out, "m%06d.ascii" % i)
    with open(p, "w", encoding="latin-1") as f:
        f.write(s)
    print(p)
