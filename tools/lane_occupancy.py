#!/usr/bin/env python3
"""Lane occupancy of the traversal loop on bench.py's frame, from a diagnostic build:
  tools/build_variant.sh diag -DRTX_DIAG && python tools/lane_occupancy.py [--width W --height H]"""
import argparse, ctypes, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=708)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--tune", type=str, default="")
ap.add_argument("--depth", type=int, default=10)
ap.add_argument("--spec", type=float, default=0.0)
ap.add_argument("--small-spheres", type=int, default=0)
ap.add_argument("--lib", type=str, default="librtx_hip_diag.so")
args = ap.parse_args()
os.environ["RTX_LIB"] = os.path.join(ROOT, "rust-raytracer_amd", args.lib)
import torch
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
lib = ctypes.CDLL(os.environ["RTX_LIB"])
flat = synth.heightfield_scene(args.grid, spec=args.spec, small_spheres=args.small_spheres)
with rtx.Context(0) as ctx:
    ctx.set_scene(flat)
    for i, v in enumerate([x for x in args.tune.split(",") if x]):
        ctx.set_option("tune%d" % i, int(v))
    ctx.build()
    img = torch.zeros(args.height * args.width * 3, dtype=torch.uint8, device="cuda:0")
    buf = (ctypes.c_ulonglong * 64)()
    for _ in range(3):
        ctx.render_device(args.width, args.height, args.depth, 10, img.data_ptr())
    lib.rtx_diag_counters(buf)
    ctx.render_device(args.width, args.height, args.depth, 10, img.data_ptr())
    assert lib.rtx_diag_counters(buf) == 0
    for name, off in (("closest", 0), ("any-hit", 32)):
        c = [int(buf[off + k]) for k in range(22)]
        it1, l1, n1, both1, it2, l2, n2, both2, don, q, given, split = c[:12]
        print(json.dumps({"kernel": name, "queries": q,
                          "phase1": {"iterations": it1, "lanes_busy": l1 / max(it1, 1) / 64, "node_share": n1 / max(l1, 1), "mixed_iterations": both1 / max(it1, 1)},
                          "phase2": {"iterations": it2, "lanes_busy": l2 / max(it2, 1) / 64, "node_share": n2 / max(l2, 1), "mixed_iterations": both2 / max(it2, 1),
                                     "donation_rounds": don, "subtrees_given": given},
                          "queries_by_phase2_iterations_0_4_16_64_more": c[12:17], "phase2_iterations_in_those_classes": c[17:22]}))
