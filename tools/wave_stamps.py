#!/usr/bin/env python3
"""Per-wavefront timeline of the two traversal kernels on bench.py's workload
(rtx_debug_stamps): when wavefronts start and end inside a launch, how many work
units each takes, and how they spread over XCDs / CUs.

  python tools/wave_stamps.py [--grid 708] [--width 1920] [--height 1080] [--tune a,b,c]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarize(name, rec):
    rec = rec[rec[:, 0] != 0]
    if len(rec) == 0:
        return {"kernel": name, "waves": 0}
    t0 = rec[:, 0].min()
    start = (rec[:, 0] - t0).astype(np.float64) / 100.0      # microseconds (100 MHz clock)
    end = (rec[:, 1] - t0).astype(np.float64) / 100.0
    units = (rec[:, 2] & 0xffff).astype(np.int64)
    step_max = ((rec[:, 2] >> 16) & 0xffffff).astype(np.int64)    # sum over units of the slowest lane's node steps
    step_sum = (rec[:, 2] >> 40).astype(np.int64)                 # node steps of all lanes (diagnostic build only)
    xcc = (rec[:, 3] & 0xff).astype(np.int64)
    se = ((rec[:, 3] >> 8) & 0xff).astype(np.int64)
    cu = ((rec[:, 3] >> 16) & 0xff).astype(np.int64)
    span = end.max()
    prof = None
    if step_sum.sum() or (rec[:, 3] >> 21).any():   # diagnostic build: rec[3] holds the cycle split (units of 256 shader cycles)
        if name == "k_shadow":     # five 12-bit shares: fetch wait, compute, giving work, taking units, ray set-up
            t_mem = (rec[:, 3] & 0xfff).astype(np.float64) * 256
            t_alu = ((rec[:, 3] >> 12) & 0xfff).astype(np.float64) * 256
            t_don = ((rec[:, 3] >> 24) & 0xfff).astype(np.float64) * 256
            t_deq = ((rec[:, 3] >> 36) & 0xfff).astype(np.float64) * 256
            t_setup = ((rec[:, 3] >> 48) & 0xfff).astype(np.float64) * 256
        else:
            t_mem = (rec[:, 3] & 0x1fffff).astype(np.float64) * 256
            t_alu = ((rec[:, 3] >> 21) & 0x1fffff).astype(np.float64) * 256
            t_don = ((rec[:, 3] >> 42) & 0x1fffff).astype(np.float64) * 256
            t_deq = t_setup = t_mem * 0
        life = (end - start) * 1e-6                          # seconds
        tot = t_mem + t_alu + t_don
        ghz = tot.sum() / life.sum() / 1e9                   # lower bound of the shader clock: cycles accounted per second of life
        last = np.argsort(end)[-64:]
        prof = {"mem_wait_share": float(t_mem.sum() / tot.sum()), "compute_share": float(t_alu.sum() / tot.sum()),
                "donation_share": float(t_don.sum() / tot.sum()), "accounted_gcycles_per_s": float(ghz),
                "dequeue_gcycles_per_s": float(t_deq.sum() / life.sum() / 1e9), "setup_gcycles_per_s": float(t_setup.sum() / life.sum() / 1e9),
                "last64_waves": {"mem_wait_share": float(t_mem[last].sum() / tot[last].sum()),
                                 "compute_share": float(t_alu[last].sum() / tot[last].sum()),
                                 "donation_share": float(t_don[last].sum() / tot[last].sum()),
                                 "accounted_gcycles_per_s": float(tot[last].sum() / life[last].sum() / 1e9)}}
        xcc = xcc * 0
    q = lambda a: [float(np.percentile(a, p)) for p in (0, 10, 50, 90, 100)]
    per_xcc = {int(x): {"waves": int((xcc == x).sum()), "units": int(units[xcc == x].sum()),
                        "last_end_us": float(end[xcc == x].max())} for x in np.unique(xcc)}
    places = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
    return {"kernel": name, "waves": int(len(rec)), "span_us": float(span),
            "start_us_p0_10_50_90_100": q(start), "end_us_p0_10_50_90_100": q(end),
            "resident_fraction": float((end - start).sum() / (span * len(rec))),
            "units_total": int(units.sum()), "units_per_wave_p0_10_50_90_100": q(units),
            "idle_waves": int((units == 0).sum()), "cycle_split": prof,
            "steps": None if step_sum.sum() == 0 else {
                "lane_steps_total": int(step_sum.sum()), "wave_steps_total": int(step_max.sum()),
                "lane_utilisation": float(step_sum.sum() / (64.0 * step_max.sum())),
                "wave_steps_p0_10_50_90_100": q(step_max),
                "us_per_wave_step_p10_50_90": [float(np.percentile((end - start) / np.maximum(step_max, 1), p)) for p in (10, 50, 90)],
                "last_waves": [{"end_us": float(end[i]), "units": int(units[i]), "wave_steps": int(step_max[i])}
                               for i in np.argsort(end)[-8:]]}, "distinct_cu_places": places, "per_xcc": per_xcc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--tune", type=str, default="")
    ap.add_argument("--lib", type=str, default="", help="another build of the library (file name in rust-raytracer_amd/)")
    args = ap.parse_args()
    if args.lib:
        os.environ["RTX_LIB"] = os.path.join(ROOT, "rust-raytracer_amd", args.lib)
    import torch
    rtx = importlib.import_module("rust-raytracer_amd")
    synth = importlib.import_module("rust-raytracer_amd.synth")
    flat = synth.heightfield_scene(args.grid)
    with rtx.Context(0) as ctx:
        ctx.set_option("stamps", 1)
        for i, v in enumerate([x for x in args.tune.split(",") if x]):
            ctx.set_option("tune%d" % i, int(v))
        ctx.set_scene(flat)
        for i, v in enumerate([x for x in args.tune.split(",") if x]):
            ctx.set_option("tune%d" % i, int(v))
        ctx.build()
        img = torch.zeros(args.height * args.width * 3, dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            st = ctx.render_device(args.width, args.height, args.depth, 10, img.data_ptr())
        out = {"closest_ms": st.closest_ms, "shadow_ms": st.shadow_ms, "render_ms": st.render_ms,
               "closest": summarize("k_trace_queue", ctx.debug_stamps(0)),
               "shadow": summarize("k_shadow", ctx.debug_stamps(1))}
        if args.lib:   # diagnostic build: per-unit records of the closest-hit launch
            ur = ctx.debug_stamps(2).reshape(-1, 2)
            ur = ur[ur[:, 0] != 0]
            if len(ur):
                t0 = ur[:, 0].min()
                ustart = (ur[:, 0] - t0) / 100.0
                dur = (ur[:, 1] & 0xffffff) / 100.0
                msteps = (ur[:, 1] >> 24) & 0xfffff
                mleaves = ur[:, 1] >> 44
                order = np.argsort(dur)[::-1]
                out["units"] = {
                    "count": int(len(ur)), "duration_us_p50_90_99_100": [float(np.percentile(dur, p)) for p in (50, 90, 99, 100)],
                    "max_lane_iterations_p50_90_99_100": [float(np.percentile(msteps, p)) for p in (50, 90, 99, 100)],
                    "slowest": [{"unit": int(np.nonzero(ur[:, 0] != 0)[0][0] * 0 + i), "start_us": float(ustart[i]), "dur_us": float(dur[i]),
                                 "max_lane_iterations": int(msteps[i]), "lane_utilisation": float(mleaves[i] / (64.0 * max(int(msteps[i]), 1)))}
                                for i in order[:12]],
                    "us_per_iteration_slowest": [float(dur[i] / max(int(msteps[i]), 1)) for i in order[:12]],
                    "total_wave_iterations": int(msteps.sum()), "total_lane_iterations": int(mleaves.sum())}
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "wave_stamps.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
