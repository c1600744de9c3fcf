"""integration/rust/src/ffi.rs against include/rtx.h: every struct has the same fields in the same order with
matching types, every function of the header is declared with the same number of arguments, every constant has
the header's value. (The Rust files cannot be compiled here; this keeps them from drifting.)"""
import os
import re

from conftest import ROOT

C_TO_RUST = {"uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "float": "f32",
             "int": "c_int", "char": "c_char", "void": "c_void"}


def _header():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _rust():
    text = open(os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")).read()
    return re.sub(r"//[^\n]*", "", text)


def _c_structs(h):
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", h, flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(\w+)\s*(\[(\d+)\])?$", decl)
            assert m, decl
            const, ctype, ptr, fname, _, arr = m.groups()
            rtype = C_TO_RUST.get(ctype, ctype)
            if ptr:
                rtype = ("*const " if const else "*mut ") + rtype
            if arr:
                rtype = "[%s; %s]" % (rtype, arr)
            fields.append((fname, rtype))
        out[name] = fields
    return out


def _rust_structs(r):
    out = {}
    for name, body in re.findall(r"pub struct (\w+) \{(.*?)\n\}", r, flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(",\n") if d.strip()]:
            m = re.match(r"pub (\w+): (.+?),?$", decl)
            assert m, decl
            fields.append((m.group(1).rstrip("_"), m.group(2).strip()))
        out[name] = fields
    return out


def test_structs_match_the_header():
    cs, rs = _c_structs(_header()), _rust_structs(_rust())
    assert set(cs) == {"rtx_material", "rtx_light", "rtx_camera", "rtx_scene_desc", "rtx_build_stats", "rtx_render_stats"}
    for name, fields in cs.items():
        assert name in rs, "ffi.rs lacks struct " + name
        assert rs[name] == fields, "%s: header %s, ffi.rs %s" % (name, fields, rs[name])


def _c_functions(h):
    out = {}
    for ret, name, args in re.findall(r"^\s*(?:const\s+)?(\w+\s*\*?)\s*(rtx_\w+)\s*\(([^)]*)\)\s*;", h, flags=re.M):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust_functions(r):
    block = re.search(r'extern "C" \{(.*)\n\}', r, flags=re.S).group(1)
    out = {}
    for name, args in re.findall(r"pub fn (rtx_\w+)\(([^)]*)\)", block, flags=re.S):
        args = args.strip()
        out[name] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_function_is_declared_with_the_same_arity(rtx):
    cf, rf = _c_functions(_header()), _rust_functions(_rust())
    assert sorted(cf) == sorted(rtx.EXPORTED_SYMBOLS)
    assert cf == rf


def test_constants_match_the_header():
    h, r = _header(), _rust()
    enum_vals = dict(re.findall(r"(RTX_\w+)\s*=\s*(\d+)", h))
    defines = {k: v for k, v in re.findall(r"#define\s+(RTX_\w+)\s+\(?([0-9xA-Fa-f]+)u?(?:ll)?", h) if k != "RTX_H"}
    rust_consts = {k: v.replace("_", "") for k, v in re.findall(r"pub const (RTX_\w+): \w+ = ([^;]+);", r)}
    for k, v in enum_vals.items():
        assert int(rust_consts[k], 0) == int(v), k
    for k in ("RTX_ABI_VERSION", "RTX_MAX_DEPTH", "RTX_MATERIAL_NONE"):
        assert int(rust_consts[k], 0) == int(defines[k], 0), k
    assert rust_consts["RTX_NOALLOC_MAX_RAYS"] == "1 << 26" and "RTX_NOALLOC_MAX_RAYS (1ull << 26)" in open(os.path.join(ROOT, "include", "rtx.h")).read()
