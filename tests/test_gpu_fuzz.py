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


@pytest.mark.parametrize("family,first,count", [(0, 0, 12), (0, 36, 1), (0, 102, 1), (0, 127, 1), (0, 129, 1), (0, 135, 1), (0, 236, 1), (0, 300, 6),
                                                (1, 0, 16), (1, 200, 8)])
def test_bvh_equals_exhaustive_on_random_scenes(family, first, count):
    """Family 1: slivers, rays nearly in the planes of triangles and along their edges, scenes far from the origin, rays
    tangent to spheres. Rays for which the triangle test's determinant is rounding noise (rays in a triangle's plane,
    mostly) are the one documented difference; the tool counts them apart and fails on anything else."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bvh.py"), "--family", str(family), "--first-seed", str(first),
                        "--scenes", str(count), "--seconds", "100", "--rays", "100000"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "scenes with mismatches: 0" in r.stdout


def test_tie_between_a_needle_and_its_duplicate():
    """Seed 26634 of family 0 with the tool's default 200 000 rays: a triangle whose edges at v0 are 1.04e-3 rad apart, in an
    axis-aligned plane, with an exact duplicate 928 indices above it. The reference's test reports its hit 4e-5 in front of
    the plane (the error of poly.rs:149-176 grows with 1 / sin of that angle), outside a box padded for ordinary triangles:
    with the duplicate found first the original's box was culled and the higher index won the tie (rays 28061 and 38617).
    The build pads needles in proportion (sliver_pad, rtx_bvh.hip)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bvh.py"), "--family", "0", "--first-seed", "26634",
                        "--scenes", "1", "--seconds", "100"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "scenes with mismatches: 0" in r.stdout


@pytest.mark.parametrize("seed", [804466, 805045])
def test_rays_whose_barycentrics_are_noise_are_counted_apart(seed):
    """Two scenes of slivers (family 1, the tool's default 200 000 rays) with one ray each for which the reference's triangle
    test (poly.rs:149-176) answers with rounding noise although the ray does not lie in the triangle's plane: a needle of
    0.17 degrees at v0, fifty lengths from the ray's origin, and a ray 3e-4 rad off its plane -- the exhaustive loop
    reports a hit two triangle lengths beside the triangle, where no box reaches; and a needle of 4 degrees with a ray
    5e-6 rad off its plane. The tool measures the noise of the barycentric coordinates, 2^-23 |s| |d| max|e| / |det|, and
    counts a differing ray apart when it exceeds 3 % of their size: these two are, and nothing else in the scenes differs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_bvh.py"), "--family", "1", "--first-seed", str(seed),
                        "--scenes", "1", "--seconds", "100"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "scenes with mismatches: 0" in r.stdout
    assert "rays with a determinant of rounding noise that differ: 0" not in r.stdout   # the scene still holds its noisy ray


@pytest.mark.parametrize("family,first,count", [(0, 0, 60), (1, 0, 60), (2, 0, 60), (3, 0, 60)])
def test_frames_of_random_scenes_equal_in_both_modes(family, first, count):
    """Whole frames (tools/fuzz_render.py): the scenes of the generators above with matte, specular and transparent
    materials, vertex materials, point, directional and area lights and a random camera, rendered with the tree and
    with the exhaustive loop: the same bytes and the same ray counts, i.e. every closest-hit and every shadow query
    (any-hit, or closest-hit with the recursion through transparent hits, lib.rs:105-141) of the frame answered alike."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_render.py"), "--family", str(family), "--first-seed", str(first),
                        "--scenes", str(count), "--seconds", "60"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "scenes with differences: 0" in r.stdout


def _frame_fuzzer():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("family,first,count", [(0, 0, 40), (1, 0, 40), (2, 0, 60), (3, 0, 60)])
def test_frames_of_random_scenes_equal_the_oracle(family, first, count, rtx, orc):
    """The same dressed scenes (up to 3000 primitives, so that the oracle's exhaustive loop takes a second) against the CPU
    restatement of the reference: primary hits bit for bit, every colour channel within 1/255 on at least 99.9 % of the
    pixels of each frame, and the ray counts of lib.rs:105-141 and 201-255 -- random materials with vertex blending,
    transparent hits on shadow rays, reflection and refraction to random depths, all three kinds of light. Families 2 and 3
    dress the same scenes oddly (dress_oddly in the tool): material indices past the list, colours outside [0, 1] (clamped on
    the way in, material.rs:17-27), exponents below zero and beyond 65536, transparency above 1, a directional light whose
    direction is far longer than 1 (the specular term then overflows to 0 * inf, and the reference's pixel is that NaN's
    black), no light, depth and samples of zero, cameras with a zero, negative or huge field of view or no view direction."""
    import numpy as np
    gen = _frame_fuzzer()
    done = 0
    for seed in range(first, first + count):
        f, kind, depth, samples = gen.make(seed, family)
        if f is None or f.n_prims > 3000:
            continue
        w = h = 48
        ref = orc.OracleScene.from_flat(f).render(w, h, depth, samples, threads=8)
        with rtx.Context(0) as ctx:
            ctx.set_option("force_mode", 2 if f.n_prims > 4 else 0)
            ctx.set_scene(f)
            ctx.build()
            prim, t, _ = ctx.primary_hits(w, h)
            rgb, st = ctx.render(w, h, depth, samples)
        assert np.array_equal(prim, ref["prim_id"]), (seed, int((prim != ref["prim_id"]).sum()))
        hit = prim >= 0
        assert np.array_equal(t.view(np.uint32)[hit], ref["t"].view(np.uint32)[hit]), seed
        diff = np.abs(rgb.astype(int) - ref["rgb"].astype(int)).max(axis=2)
        assert (diff <= 1).mean() >= 0.999, "seed %d: max diff %d, within-1 %.4f" % (seed, diff.max(), (diff <= 1).mean())
        rs = ref["stats"]
        assert (st.shadow_rays, st.reflect_rays, st.refract_rays, st.skipped_rays, st.shaded_nodes) == \
            (rs["shadow_rays"], rs["reflect_rays"], rs["refract_rays"], rs["last_level_rays"], rs["shaded_nodes"]), seed
        done += 1
    assert done >= count // 3


def test_scenes_with_non_finite_and_enormous_geometry():
    """tools/fuzz_nonfinite.py: one percent of a scene's vertices, centres and radii replaced by NaN, +-Inf, +-1e30, 3e38,
    denormals and signed zeros. Such primitives are tested by every ray (rtx_build_stats.n_always) and kept out of the
    tree's boxes and of the scene's bounds, so the tree still answers like the loop of scene/mod.rs:186-199 -- Hit(inf) of
    an infinite sphere included -- for ray batches and for frames; nothing faults, hangs or returns an error status."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_nonfinite.py"), "--scenes", "150", "--seconds", "60"],
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "with differences or errors: 0" in r.stdout


@pytest.mark.parametrize("family", [0, 1, 3])
def test_every_way_of_asking_for_a_frame_gives_the_same_bytes(family):
    """tools/fuzz_api.py: host-buffer call, device call (the second one on level hints), three frames in flight, tile slabs
    of 2-5 ranks assembled on the device, batches of 2-3 frames with and without cameras of their own, the scene handed over
    as device arrays, a clone -- under random
    options (ray batches of 64 rays up, node pools down to one node per pixel, hints, recorded launches, timing events, cost
    order and work sharing on and off), on random dressed scenes at random odd sizes: always the plain frame's bytes. A batch
    whose ray tree outgrows the pool is rendered frame by frame; only a pool too small for ONE tile's rays is an error."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_api.py"), "--family", str(family), "--first-seed", "0",
                        "--scenes", "80", "--seconds", "40"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "with failures: 0" in r.stdout
