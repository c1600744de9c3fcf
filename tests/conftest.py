import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES_DIR = os.path.join(ROOT, "scenes")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# The 13 deterministic bundled scenes (area.ascii samples a random number generator in the reference).
BUNDLED = ["sphere"] + ["test%02d" % i for i in range(12)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES_DIR, name + ".ascii")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi


@pytest.fixture(scope="session")
def rtx():
    """The product package (ctypes binding of librtx_hip.so)."""
    import rtx_amd
    return rtx_amd


@pytest.fixture(scope="session")
def gpu_ctx_factory(rtx):
    """Creates contexts on cuda:0 and closes them at the end of the session."""
    made = []

    def make(flat, force_mode=0):
        ctx = rtx.Context(0)
        ctx.set_option("verify_build", 1)      # every tree built by the tests has its refit re-checked box by box
        if force_mode:
            ctx.set_option("force_mode", force_mode)
        ctx.set_scene(flat)
        ctx.build()
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()


@pytest.fixture(scope="session", autouse=True)
def _device_buffers_are_ready_before_the_library_sees_them():
    """The library renders on a non-blocking stream of its own, which waits for no other stream: a buffer a test has just
    filled with torch (on torch's stream) could still be receiving that fill while the library writes it. Every entry
    point of the binding that takes a device pointer therefore first waits for torch's current stream -- what a caller
    has to do by contract (include/rtx.h: "Streams"); the library's own asynchrony is not touched by it."""
    try:
        import torch
        import rtx_amd
    except Exception:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    names = ["render_device", "render_device_async", "render_tiles", "render_tiles_async", "assemble_tiles", "assemble_tiles_async",
             "render_frames", "render_frames_async", "assemble_frames_async", "intersect_batch_device", "intersect_batch_device_noalloc",
             "gather_frame", "gather_frames"]
    saved = {}
    for n in names:
        f = getattr(rtx_amd.Context, n, None)
        if f is None:
            continue
        saved[n] = f

        def wrapped(self, *a, _f=f, **k):
            torch.cuda.current_stream().synchronize()
            return _f(self, *a, **k)
        setattr(rtx_amd.Context, n, wrapped)
    yield
    for n, f in saved.items():
        setattr(rtx_amd.Context, n, f)
