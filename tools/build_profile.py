#!/usr/bin/env python3
"""Builds the tree of a throughput configuration a few times (for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 708
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
flat = synth.heightfield_scene(grid)
with rtx.Context(0) as ctx:
    ctx.set_scene(flat)
    for k in range(reps):
        t0 = time.perf_counter()
        b = ctx.build()
        print("build %d: device %.3f ms, host wall %.3f ms, nodes %d" % (k, b["build_ms"], (time.perf_counter() - t0) * 1e3, b["n_nodes"]), flush=True)
