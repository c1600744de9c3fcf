"""bench.py as the driver runs it (defaults: one GPU), shortened: ONE JSON line on stdout with the agreed keys, the roofline
object of the dominant kernel backed by the committed counters of the same library, and the CPU baseline timed beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_keeps_its_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-stride", "120"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("Mrays/sec") and d["unit"] == "Mrays/s" and d["value"] > 1000.0
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["ms_per_step"] > 0.0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "1002528-triangle" in cfg["workload"] and cfg["batch_frames_differ"] is True and "model" not in cfg
    # whole-job rate = rays of the timed steps / their wall time
    assert abs(d["value"] - d["rays_per_frame"] / (d["ms_per_frame"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["kernel"].startswith("k_shadow") and roof["avg_launch_ms"] > 0.0
    # the counters under profiles/ belong to this very library (same source hash), so the fraction is filled in
    assert roof["traffic"] is not None and roof["frac"] is not None, roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "Mrays/s" and cpu["value"] > 0.0 and "pixels" in cpu["sample"]
