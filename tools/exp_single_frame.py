#!/usr/bin/env python3
"""Experiment (GPU box): one 1080p frame per set of launches through the two async entry points -- the scene's camera
(rtx_render_device_async) against an explicit camera (rtx_render_frames_async with one frame), identical and moving.
Prints wall time per frame, host time inside the enqueue call and the traversal kernels' own times."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
flat = synth.heightfield_scene(708)
ctx = rtx.Context(0)
ctx.set_scene(flat)
ctx.build()
w, h = 1920, 1080
bufs = [torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
torch.cuda.synchronize()
path = synth.camera_path(flat, 0, 400)
same = [path[0]] * 400
variants = (("scene camera, render_device_async", None), ("explicit, identical", same), ("explicit, moving", path),
            ("scene camera again", None))
if len(sys.argv) > 1:      # one variant only (profiling)
    variants = (variants[int(sys.argv[1])],)
for name, cams in variants:
    for rep in range(2):
        tot = {"c": 0.0, "s": 0.0, "n": 0, "host": 0.0}
        t0 = time.perf_counter()
        for k in range(300):
            if k >= 2:
                st = ctx.render_wait()
                tot["c"] += st.closest_ms; tot["s"] += st.shadow_ms; tot["n"] += 1
            h0 = time.perf_counter()
            if cams is None:
                ctx.render_device_async(w, h, 10, 10, bufs[k % 3].data_ptr())
            else:
                ctx.render_frames_async(1, [cams[k]], w, h, 10, 10, 0, 1, bufs[k % 3].data_ptr(), True)
            tot["host"] += time.perf_counter() - h0
        for _ in range(2):
            ctx.render_wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300
    print("%-40s %.4f ms/frame  enqueue %.1f us  closest %.4f  shadow %.4f" % (name, dt * 1e3, tot["host"] / 300 * 1e6, tot["c"] / tot["n"], tot["s"] / tot["n"]), flush=True)
