#!/usr/bin/env python3
"""Summarises the PMC passes of tools/pmc_passes2.sh (gpurun_out/pmc_<tag>_*/) per traversal kernel, mean per launch,
and merges the result into profiles/pmc_traffic.json under the workload the passes ran (the workload key and the hash of
the library's sources are taken from the bench line each pass printed, so bench.py only quotes counters of its own build).

  python3 tools/pmc_summary.py <tag> [--merge]      # --merge: also update profiles/pmc_traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4
# shader cycles a wave64 VALU instruction occupies its SIMD, averaged over the traversal kernels' instruction mix
# (tools/microbench.hip: 1.7 for fma / add / mul / mov / and, 2.6-2.8 for max3 / cmp / cvt_ubyte / shift / cndmask on an SGPR pair)
VALU_CYCLES_PER_INSTRUCTION = 2.2

tag = sys.argv[1]
merge = "--merge" in sys.argv[2:]
out = collections.defaultdict(dict)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % tag, "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "shadow" if "k_shadow" in k else "closest" if "k_trace" in k else None
        if short:   # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch
            agg[short][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for short, d in agg.items():
        for c, per_dispatch in d.items():
            out[short][c] = sum(per_dispatch.values()) / len(per_dispatch)
            out[short]["launches_in_pass"] = len(per_dispatch)
bench = None
for log in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*.log" % tag))):
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"roofline"' in line:
            bench = json.loads(line)
summary = {}
for short, d in out.items():
    g = d.get
    cycles = g("GRBM_GUI_ACTIVE", 0.0) / 8.0            # the counter sums the 8 XCDs
    e = {"kernel": "k_shadow" if short == "shadow" else "k_trace_queue", "counters": {k: v for k, v in sorted(d.items())}}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = g("FETCH_SIZE"), g("WRITE_SIZE")
        e["bytes_per_launch"] = (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0      # 64-byte gathers count x1.00 (calibration in the note)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        e["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
    if g("SQ_INSTS_VALU") and cycles:
        e["valu_instructions_per_launch"] = g("SQ_INSTS_VALU")
        e["valu_issue_frac"] = g("SQ_INSTS_VALU") * VALU_CYCLES_PER_INSTRUCTION / (SIMDS * cycles)
    if g("SQ_INSTS_VMEM_RD"):
        e["fetch_iterations_per_launch"] = g("SQ_INSTS_VMEM_RD") / 4.0
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        e["wave_wait_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("TA_TA_BUSY_sum") and cycles:
        e["ta_busy_frac"] = g("TA_TA_BUSY_sum") / 256 / cycles
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")):
        e["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        e["l1_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
        e["l1_miss_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
    e["kernel_cycles_per_launch"] = cycles
    summary[short] = e
    print(short, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k != "counters"}))
json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "pmc_%s_summary.json" % tag), "w"), indent=1)
if merge:
    if bench is None:
        raise SystemExit("no bench line found in gpurun_out/pmc_%s_*.log: cannot stamp the counters" % tag)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    doc = {"workloads": {}}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
            if "workloads" in old:
                doc = old
        except Exception:
            pass
    doc["note"] = ("Per-launch means of rocprofv3 PMC passes over bench.py (tools/pmc_passes2.sh: separate passes, counters only). "
                   "bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) KiB x 1024: fabric-side traffic of the L2s (Infinity Cache hits "
                   "included). Calibrated with tools/calib_fetch on the same kind of box (profiles/r01_c_fetch_calibration.txt): "
                   "64-byte record gathers (this library's node / record fetches) read FETCH_SIZE x1.00, 16 B/lane coalesced streams "
                   "x2.00; the gather factor is applied. Each workload entry carries the hash of the library sources it was "
                   "collected on; bench.py quotes an entry only when its own source hash matches. With several launches of a kind "
                   "per frame (deeper ray-tree levels) the figures are means over all of them.")
    doc["valu_cycles_per_wave_instruction"] = VALU_CYCLES_PER_INSTRUCTION
    key = bench["roofline"].get("workload_key")
    doc["workloads"][key] = {"source_hash": bench["roofline"]["source_hash"], "tag": tag,
                             "bench_ms_per_frame_under_profiler": bench.get("ms_per_frame"), "kernels": summary}
    json.dump(doc, open(path, "w"), indent=1)
    print("merged into profiles/pmc_traffic.json as", key, "source hash", bench["roofline"]["source_hash"])
