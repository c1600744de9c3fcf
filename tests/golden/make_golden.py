#!/usr/bin/env python3
"""Regenerates tests/golden/frames_100x100.npz with the CPU oracle: for each of the
13 deterministic bundled scenes at the reference's default frame (100x100, depth 10,
samples 10; src/main.rs:63-66) the primary hit-id map, hit t, final RGB and the ray
counts by class. The Rust reference cannot be built in this environment, so these
are outputs of the C++ restatement (oracle/rt_oracle.cpp), not of the reference
binary; they pin the oracle against accidental change and let GPU tests run
without re-rendering on the CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi  # noqa: E402

BUNDLED = ["sphere"] + ["test%02d" % i for i in range(12)]
COUNT_KEYS = ["primary_rays", "shadow_rays", "reflect_rays", "refract_rays", "shaded_nodes", "last_level_rays", "panic_pixels"]


def main():
    oracle_ffi.build()
    out = {}
    for name in BUNDLED:
        s = oracle_ffi.OracleScene.parse(os.path.join(ROOT, "scenes", name + ".ascii"))
        r = s.render(100, 100, 10, 10, threads=1)
        out[name + "_prim"] = r["prim_id"].astype(np.int32)
        out[name + "_t"] = r["t"]
        out[name + "_rgb"] = r["rgb"]
        out[name + "_counts"] = np.array([r["stats"][k] for k in COUNT_KEYS], np.uint64)
        print(name, out[name + "_counts"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames_100x100.npz"), **out)


if __name__ == "__main__":
    main()
