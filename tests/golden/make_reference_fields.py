#!/usr/bin/env python3
"""Writes tests/golden/reference_struct_fields.json: for every struct and enum of the reference crate that the Rust
binding (integration/rust/src/gpu_scene.rs) touches, its field names with their types, its enum variants, and the
names of its methods -- names only, no code. Run in the container that holds /root/reference:

    python3 tests/golden/make_reference_fields.py

tests/test_rust_shim.py checks every member access of gpu_scene.rs against this list, so the binding cannot drift
from the reference's real names again (the crate cannot be compiled here: no Rust toolchain)."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src"
FILES = ["vec.rs", "ray.rs", "lib.rs", "scene/mod.rs", "scene/material.rs", "scene/intersection.rs", "scene/shapes/mod.rs",
         "scene/shapes/poly.rs", "scene/shapes/sphere.rs"]
out = {"structs": {}, "enums": {}, "methods": {}, "traits": {}, "derives": {}, "source": {}}
for rel in FILES:
    text = open(os.path.join(REF, rel)).read()
    text = re.sub(r"//[^\n]*", "", text)
    for m in re.finditer(r"#\[derive\(([^)]*)\)\]\s*pub (?:struct|enum) (\w+)", text):
        out["derives"][m.group(2)] = [d.strip() for d in m.group(1).split(",")]
    for m in re.finditer(r"pub struct (\w+)(?:<[^>]*>)?\s*\{(.*?)\n\}", text, flags=re.S):
        fields = {}
        for f in re.finditer(r"(pub\s+)?(\w+)\s*:\s*([^,\n]+)", m.group(2)):
            fields[f.group(2)] = {"type": f.group(3).strip(), "public": bool(f.group(1))}
        out["structs"][m.group(1)] = fields
        out["source"][m.group(1)] = rel
    for m in re.finditer(r"pub enum (\w+)(?:<[^>]*>)?\s*\{(.*?)\n\}", text, flags=re.S):
        variants = {}
        for v in re.finditer(r"(\w+)\s*(?:\(([^)]*)\))?\s*,?", m.group(2)):
            if v.group(1):
                variants[v.group(1)] = (v.group(2) or "").strip()
        out["enums"][m.group(1)] = variants
        out["source"][m.group(1)] = rel
    for m in re.finditer(r"\nimpl(?:<[^>]*>)?\s+(?:(\w+)(?:<[^>]*>)?\s+for\s+)?(\w+)(?:<[^>]*>)?\s*\{(.*?)\n\}", text, flags=re.S):
        names = re.findall(r"fn (\w+)", m.group(3))
        out["methods"].setdefault(m.group(2), [])
        for n in names:
            if n not in out["methods"][m.group(2)]:
                out["methods"][m.group(2)].append(n)
    for m in re.finditer(r"pub trait (\w+)(?:<[^>]*>)?\s*\{(.*?)\n\}", text, flags=re.S):
        out["traits"][m.group(1)] = re.findall(r"fn (\w+)", m.group(2))
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_struct_fields.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "-", len(out["structs"]), "structs,", len(out["enums"]), "enums")
