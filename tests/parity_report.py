"""Parity report of the HIP path against the CPU oracle on every bundled scene (run by name:
`python tests/parity_report.py [--size 100]`; not collected by pytest). For each scene: pixels whose
primary ray hits a different primitive, hit pixels whose t differs in any bit, share of pixels whose
RGB is within 1/255 per channel, largest channel difference, and whether the ray counts by class are
equal. Written as JSON to gpurun_out/parity_report_<size>.json."""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from conftest import BUNDLED, scene_path  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=100)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--samples", type=int, default=10)
    args = ap.parse_args()
    import rtx_amd as rtx
    from oracle import oracle_ffi as orc
    orc.build()
    w = h = args.size
    report = {"width": w, "height": h, "depth": args.depth, "samples": args.samples, "scenes": {}}
    for name in BUNDLED:
        flat = rtx.parse_scene(scene_path(name))
        ref = orc.OracleScene.parse(scene_path(name)).render(w, h, args.depth, args.samples, threads=16)
        with rtx.Context(0) as ctx:
            if flat.n_prims > 4:
                ctx.set_option("force_mode", 2)       # the LBVH path even where the library would loop over a few primitives
            ctx.set_scene(flat)
            ctx.build()
            prim, t, rdir = ctx.primary_hits(w, h)
            rgb, st = ctx.render(w, h, args.depth, args.samples)
            uses_bvh = int(ctx.build_stats["uses_bvh"])
        hit = prim >= 0
        diff = np.abs(rgb.astype(np.int32) - ref["rgb"].astype(np.int32)).max(axis=2)
        rs = ref["stats"]
        counts_equal = (st.primary_rays, st.shadow_rays, st.reflect_rays, st.refract_rays, st.skipped_rays, st.shaded_nodes) == \
            (rs["primary_rays"], rs["shadow_rays"], rs["reflect_rays"], rs["refract_rays"], rs["last_level_rays"], rs["shaded_nodes"])
        report["scenes"][name] = {
            "primitives": int(flat.n_prims), "bvh": uses_bvh,
            "primary_ray_direction_bit_mismatches": int((bits(rdir) != bits(ref["ray_dir"])).any(axis=-1).sum()),
            "pixels_hitting_another_primitive": int((prim != ref["prim_id"]).sum()),
            "hit_pixels_with_different_t_bits": int((bits(t)[hit] != bits(ref["t"])[hit]).sum()),
            "pixels_within_1_of_255": float((diff <= 1).mean()), "largest_channel_difference": int(diff.max()),
            "pixels_not_identical": int((diff > 0).sum()), "ray_counts_equal": bool(counts_equal),
            "rays": int(st.reference_rays)}
        print(name, json.dumps(report["scenes"][name]), flush=True)
    sc = report["scenes"].values()
    report["summary"] = {"scenes": len(report["scenes"]),
                         "all_primitive_ids_equal": all(s["pixels_hitting_another_primitive"] == 0 for s in sc),
                         "all_t_bit_equal": all(s["hit_pixels_with_different_t_bits"] == 0 for s in sc),
                         "smallest_share_within_1_of_255": min(s["pixels_within_1_of_255"] for s in sc),
                         "all_ray_counts_equal": all(s["ray_counts_equal"] for s in sc)}
    print(json.dumps(report["summary"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_report_%d.json" % args.size), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
