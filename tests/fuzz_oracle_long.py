#!/usr/bin/env python3
"""Long form of test_gpu_fuzz.py::test_frames_of_random_scenes_equal_the_oracle (not collected by pytest): frames of random
dressed scenes (tools/fuzz_render.py's generator, families 0-3) from the library against the CPU oracle for --seconds per
family: primary hits bit for bit, every channel within 1/255 on >= 99.9 % of a frame's pixels, ray counts equal.
usage: python tests/fuzz_oracle_long.py [--seconds 200] [--first-seed 5000] [--families 0,1,2,3] [--size 48]"""
import argparse, importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtx_amd as rtx
from oracle import oracle_ffi as orc

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=200.0)
ap.add_argument("--first-seed", type=int, default=5000)
ap.add_argument("--families", default="0,1,2,3")
ap.add_argument("--size", type=int, default=48)
ap.add_argument("--threads", type=int, default=16)
a = ap.parse_args()
orc.build()
spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
gen = importlib.util.module_from_spec(spec)
sys.argv = [sys.argv[0]]          # the generator module parses its own command line on import
spec.loader.exec_module(gen)
bad = 0
for family in (int(x) for x in a.families.split(",")):
    t_end = time.time() + a.seconds
    seed, done, failed = a.first_seed, 0, 0
    while time.time() < t_end:
        f, kind, depth, samples = gen.make(seed, family)
        seed += 1
        if f is None or f.n_prims > 3000:
            continue
        w = h = a.size
        ref = orc.OracleScene.from_flat(f).render(w, h, depth, samples, threads=a.threads)
        with rtx.Context(0) as ctx:
            ctx.set_option("force_mode", 2 if f.n_prims > 4 else 0)
            ctx.set_scene(f)
            ctx.build()
            prim, t, _ = ctx.primary_hits(w, h)
            rgb, st = ctx.render(w, h, depth, samples)
        hit = prim >= 0
        diff = np.abs(rgb.astype(int) - ref["rgb"].astype(int)).max(axis=2)
        rs = ref["stats"]
        ok = (np.array_equal(prim, ref["prim_id"]) and np.array_equal(t.view(np.uint32)[hit], ref["t"].view(np.uint32)[hit])
              and (diff <= 1).mean() >= 0.999
              and (st.shadow_rays, st.reflect_rays, st.refract_rays, st.skipped_rays, st.shaded_nodes) ==
                  (rs["shadow_rays"], rs["reflect_rays"], rs["refract_rays"], rs["last_level_rays"], rs["shaded_nodes"]))
        done += 1
        if not ok:
            failed += 1
            print("family", family, "seed", seed - 1, "kind", kind, "prims", f.n_prims, "depth", depth, "samples", samples,
                  "ids differ", int((prim != ref["prim_id"]).sum()), "max colour diff", int(diff.max()), "within 1: %.4f" % (diff <= 1).mean(), flush=True)
        if done % 100 == 0:
            print("family", family, "frames", done, "failed", failed, flush=True)
    print("family", family, "seeds", a.first_seed, "..", seed - 1, "frames compared:", done, "frames that differ:", failed, flush=True)
    bad += failed
sys.exit(1 if bad else 0)
