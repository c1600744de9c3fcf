#!/usr/bin/env python3
"""Library against oracle on the rays a far-away scene's shadow recursion is made of: rays that start ON surfaces (the
1e-4 offset of lib.rs:111 is below half an ulp at coordinates of 1e4) towards the lights, over several generations.
usage: python tests/fuzz_oracle_rays.py --family 3 --seed 100932"""
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
a = ap.parse_args()
orc.build()
spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
gen = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]]
spec.loader.exec_module(gen)
f, kind, depth, samples = gen.make(a.seed, a.family)
w = h = a.size
osc = orc.OracleScene.from_flat(f)
f32 = np.float32
with rtx.Context(0) as ctx:
    ctx.set_option("force_mode", 1)
    ctx.set_scene(f)
    ctx.build()
    prim, t, dirs = ctx.primary_hits(w, h)
    cam = np.array(f.camera.pos[:], f32)
    hit = prim.reshape(-1) >= 0
    d0 = np.asarray(dirs, f32).reshape(-1, 3)[hit]
    t0 = t.reshape(-1)[hit].astype(f32)
    pts = (cam[None, :] + d0 * t0[:, None]).astype(f32)          # Intersection::point
    lights = [np.array(l[1], f32) for l in f.lights if int(l[0]) == 0]
    for gen_no in range(4):
        if len(pts) == 0 or not lights:
            break
        o = np.concatenate([pts for _ in lights])
        dv = np.concatenate([(L[None, :] - pts).astype(f32) for L in lights])
        ln = np.sqrt((dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1]) + dv[:, 2] * dv[:, 2]).astype(f32)
        ok = ln > 0
        o, dv, ln = o[ok], dv[ok], ln[ok]
        d = (dv / ln[:, None]).astype(f32)
        pl, tl = ctx.intersect_batch(o, d)
        po, to = osc.intersect_batch(o, d)
        both = (pl >= 0) & (po >= 0)
        bad = (pl != po) | (both & (tl.view(np.uint32) != to.view(np.uint32)))
        print("generation", gen_no, "rays", len(o), "hits", int((po >= 0).sum()), "differ", int(bad.sum()), flush=True)
        for i in np.flatnonzero(bad)[:6]:
            print("   o", o[i].tolist(), "d", d[i].tolist(), "lib", int(pl[i]), float(tl[i]), "oracle", int(po[i]), float(to[i]),
                  "kinds", ["sphere" if 0 <= p and p >= f.n_tris else "tri" for p in (int(pl[i]), int(po[i]))])
        keep = po >= 0
        pts = (o[keep] + d[keep] * to[keep][:, None].astype(f32)).astype(f32)
        pts = pts[:200000]
