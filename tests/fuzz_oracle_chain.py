#!/usr/bin/env python3
"""The shadow recursion of lib.rs:105-141 for single pixels of a fuzz scene, step by step in numpy float32 on top of the
library's own intersections (which equal the oracle's on these rays, tests/fuzz_oracle_rays.py), for point lights and
scenes so far from the origin that the 1e-4 offsets vanish. usage: --family 3 --seed 100932 --depth 2 --pixels 22,0;63,0"""
import argparse, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_amd as rtx
ap = argparse.ArgumentParser()
ap.add_argument("--family", type=int, required=True)
ap.add_argument("--seed", type=int, required=True)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--depth", type=int, default=2)
ap.add_argument("--pixels", default="22,0")
a = ap.parse_args()
spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
gen = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]]
spec.loader.exec_module(gen)
f, kind, depth, samples = gen.make(a.seed, a.family)
w = h = a.size
f32 = np.float32
mats = np.asarray(f.materials)
def kt_of(pid):
    m = int(f.tri_mat[pid][0]) if pid < f.n_tris else int(f.sph_mat[pid - f.n_tris])
    return f32(mats[m][5]) if m < len(mats) else f32(0.0)
def dist(p, q):
    d = (p - q).astype(f32)
    return np.sqrt(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]), dtype=f32)
with rtx.Context(0) as ctx:
    ctx.set_option("force_mode", 1)
    ctx.set_scene(f)
    ctx.build()
    prim, t, dirs = ctx.primary_hits(w, h)
    dirs = np.asarray(dirs, f32).reshape(h, w, 3)
    cam = np.array(f.camera.pos[:], f32)
    for px in a.pixels.split(";"):
        x, y = (int(v) for v in px.split(","))
        P = (cam + dirs[y, x] * f32(t[y, x])).astype(f32)
        print("pixel", x, y, "prim", int(prim[y, x]), "point", P.tolist())
        for j, l in enumerate(f.lights):
            L = np.array(l[1], f32)
            o, rem, factor, term = P, a.depth, f32(1.0), None
            while True:
                dv = (L - o).astype(f32)
                d = (dv / dist(L, o)).astype(f32)
                pid, tt = ctx.intersect_batch(o[None, :], d[None, :])
                pid, tt = int(pid[0]), f32(tt[0])
                if pid < 0:
                    term = f32(1.0); print("   light", j, "ray from", o.tolist(), "missed"); break
                hp = (o + d * tt).astype(f32)
                kt = kt_of(pid)
                print("   light", j, "ray from", o.tolist(), "hit", pid, "t", float(tt), "kt", float(kt), "dist hit", float(dist(o, hp)), "dist light", float(dist(o, L)))
                if kt == 0:
                    term = f32(1.0) if dist(o, hp) > dist(o, L) else f32(0.0); break
                if rem - 1 == 0:
                    term = f32(0.0); break
                factor = f32(factor * kt)      # valid while every inner term is 0 or 1 (n = 1): kt * clamp(term)
                rem -= 1
                o = hp
            print("   light", j, "shadow term", float(factor * term))
