#!/usr/bin/env python3
"""One frame of tests/fuzz_oracle_long.py again, with details of where library and oracle differ.
usage: python tests/fuzz_oracle_repro.py --family 3 --seed 100932 [--size 64]"""
import argparse, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_amd as rtx
from oracle import oracle_ffi as orc
ap = argparse.ArgumentParser()
ap.add_argument("--family", type=int, required=True)
ap.add_argument("--seed", type=int, required=True)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--depth", type=int, default=-1)
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()
orc.build()
spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
gen = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]]
spec.loader.exec_module(gen)
f, kind, depth, samples = gen.make(a.seed, a.family)
w = h = a.size
if a.depth >= 0:
    depth = a.depth
print("kind", kind, "prims", f.n_prims, "tris", f.n_tris, "spheres", f.n_spheres, "depth", depth, "samples", samples)
if not a.quiet: print("camera", f.camera.pos[:], f.camera.view_dir[:], f.camera.ortho_up[:], f.camera.vertical_fov)
if not a.quiet: print("lights", f.lights)
if not a.quiet: print("materials", np.asarray(f.materials))
ref = orc.OracleScene.from_flat(f).render(w, h, depth, samples, threads=16)
for mode in (2 if f.n_prims > 4 else 0,):
    with rtx.Context(0) as ctx:
        ctx.set_option("force_mode", mode)
        ctx.set_scene(f)
        ctx.build()
        prim, t, _ = ctx.primary_hits(w, h)
        rgb, st = ctx.render(w, h, depth, samples)
    diff = np.abs(rgb.astype(int) - ref["rgb"].astype(int)).max(axis=2)
    ys, xs = np.nonzero(diff > 1)
    print("mode", mode, "ids equal", np.array_equal(prim, ref["prim_id"]), "pixels off by more than 1:", len(ys), "max", diff.max(),
          "stats", (st.shadow_rays, st.reflect_rays, st.refract_rays, st.skipped_rays, st.shaded_nodes), ref["stats"])
    for y, x in list(zip(ys, xs))[:8]:
        print("   pixel", x, y, "prim", prim[y, x], "t", t[y, x], "lib", rgb[y, x], "oracle", ref["rgb"][y, x],
              "mat", (f.tri_mat[prim[y, x]] if 0 <= prim[y, x] < f.n_tris else None))
    prims_off = sorted(set(prim[ys, xs].tolist()))
    print("   primitives of those pixels:", prims_off[:20])
