"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code that reads untrusted files: the product's scene
parser and BMP reader (rust-raytracer_amd/host/rstracer.cpp; the reference is safe Rust and panics on bad input,
src/scene/parser/mod.rs:84-99) and the CPU oracle (oracle/rt_oracle.cpp). Host code only, never on the GPU: the driver
(tests/cpp/san_main.cpp) makes no device call; `make -C oracle SAN=1` / `make -C rust-raytracer_amd/host SAN=1` build the
instrumented libraries it links."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, SCENES_DIR


@pytest.fixture(scope="module")
def san_exe(tmp_path_factory):
    pkg = os.path.join(ROOT, "rust-raytracer_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "SAN=1"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(pkg, "host"), "SAN=1"])
    exe = tmp_path_factory.mktemp("san") / "san_main"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "san_main.cpp"),
                           "-I", os.path.join(pkg, "host"), "-I", os.path.join(ROOT, "oracle"),
                           "-L", pkg, "-L", os.path.join(ROOT, "oracle"), "-lrstracer_san", "-loracle_san", "-lrtx_hip", "-pthread",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return str(exe)


def test_parser_bmp_and_oracle_are_clean_under_asan_and_ubsan(san_exe, tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([san_exe, SCENES_DIR, os.path.join(GOLDEN_DIR, "parser", "testdata-"), str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "all checks passed" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr and "LeakSanitizer" not in out.stderr
    # the frames the instrumented oracle rendered are the committed golden frames
    golden = np.load(os.path.join(GOLDEN_DIR, "frames_100x100.npz"))
    for name in ("test05", "test10"):
        rgb = np.fromfile(str(tmp_path / (name + ".rgb")), np.uint8).reshape(100, 100, 3)
        assert np.array_equal(rgb, golden[name + "_rgb"]), name


def test_mutated_scene_files_parse_or_are_rejected_under_asan_and_ubsan(tmp_path):
    """2000 token-level mutants of the small bundled scenes (tools/fuzz_parser.py: numbers replaced by huge, negative,
    fractional and non-numeric tokens, keywords swapped, lines dropped and doubled, blocks spliced and cut) through both
    parser paths of the instrumented library: each parses or raises rstracer::Error, the two paths agree, no report."""
    pkg = os.path.join(ROOT, "rust-raytracer_amd")
    subprocess.check_call(["make", "-s", "-C", os.path.join(pkg, "host"), "SAN=1"])
    exe = tmp_path / "san_parse_files"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "san_parse_files.cpp"),
                           "-I", os.path.join(pkg, "host"), "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lrstracer_san", "-lrtx_hip",
                           "-pthread", "-Wl,-rpath," + pkg])
    import sys
    names = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parser.py"), str(tmp_path / "mutants"), "2000", "7"],
                           capture_output=True, text=True, check=True).stdout
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([str(exe)], input=names, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "disagreements 0" in out.stdout and "parsed" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr and "LeakSanitizer" not in out.stderr
    parsed = int(out.stdout.split("parsed")[1].split()[0])
    assert 50 < parsed < 1900      # the mutator produces both kinds
