"""BVH query against the exhaustive loop (Scene::intersects, src/scene/mod.rs:181-201) on random scenes: triangle soups
and spheres of mixed scales, coplanar, duplicated and zero-area triangles, rays from surfaces, along the axes and towards
silhouettes (tools/fuzz_bvh.py runs the same generator for hundreds of scenes). ids and the bits of t must agree.
Seeds 36, 102, 127, 129 and 135 are scenes in which a zero-area triangle reports hits for rays that pass nowhere near it
(its determinant is rounding noise): the reference's loop takes those hits, so the tree has to find them too."""
import importlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("first,count", [(0, 12), (36, 1), (102, 1), (127, 1), (129, 1), (135, 1), (236, 1), (300, 6)])
def test_bvh_equals_exhaustive_on_random_scenes(first, count):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bvh.py"), "--first-seed", str(first), "--scenes", str(count),
                        "--seconds", "100", "--rays", "100000"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "scenes with mismatches: 0" in r.stdout
