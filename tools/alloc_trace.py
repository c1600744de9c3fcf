#!/usr/bin/env python3
"""Which HIP allocation calls do the host-buffer entry points make in steady state?
Run under `rocprofv3 --hip-trace` with two different iteration counts and compare the number of
hipMalloc* / hipFree* / hipHostMalloc calls in the two traces (tools/alloc_trace.sh does that):
equal counts = nothing is allocated per call.

  python3 tools/alloc_trace.py <iterations>
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_iter = int(sys.argv[1])
rtx = importlib.import_module("rust-raytracer_amd")
flat = rtx.parse_scene(os.path.join(ROOT, "scenes", "test06.ascii"))
rays = np.random.default_rng(4).normal(size=(4096, 3)).astype(np.float32)
ori = np.tile(np.array(list(flat.camera.pos), np.float32), (4096, 1))
with rtx.Context(0) as ctx:
    ctx.set_scene(flat)
    ctx.build()
    for _ in range(2 + n_iter):      # the first two rounds size the workspace and the staging buffer
        rgb, _ = ctx.render(128, 128, 10, 10)
        ctx.intersect_batch(ori, rays)
        ctx.primary_hits(128, 128)
print("iterations", n_iter, "max", int(rgb.max()))
