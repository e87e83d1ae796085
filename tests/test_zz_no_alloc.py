"""pt_render_noalloc (include/pt_hip.h): the entry point for callers that cannot allocate.  Same kernels as
pt_render, working memory in the library's per-GPU `scratch` array: its image must be pt_render's bit for bit —
on two calls in a row, the second with a smaller input, on both kernels (the streaming kernel is the one that
uses the queue head in scratch), with and without the bounce counter, and past the limit it must refuse.  Also: the
call really allocates nothing (hipMemGetInfo around it; captured into a HIP graph under the strictest capture mode, which
turns any hipMalloc into an error, and replayed), refuses the synchronising forms while a stream is being captured,
and refuses to overlap an earlier call that is still running on another stream of the same GPU."""
import ctypes as C

import numpy as np
import pytest

from pathtracer_amd import hip, scenes
from pathtracer_amd import seeds as seedmod

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def render_both(ctx, sc, W, H, depth, s, **kw):
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(W, H)
    ra = ctx.render(sc, W, H, depth, s, **kw)
    a = ctx.image_read()
    ctx.image_alloc(W, H)
    rb = ctx.render(sc, W, H, depth, s, noalloc=True, **kw)
    b = ctx.image_read()
    return a, b, ra, rb


def test_noalloc_matches_pt_render_twice_in_a_row(ctx):
    stream = seedmod.glibc_rand_stream(300)
    # first call: streaming kernel (500 spheres: queue head + seed table in scratch), 300 samples = two seed chunks
    sc = scenes.dense_random(500)
    a, b, ca, cb = render_both(ctx, sc, 160, 90, 50, stream, count_bounces=True)
    assert a[..., :3].max() > 0
    assert np.array_equal(bits(a), bits(b)) and ca == cb and ca > 0
    # second call, smaller input (fewer seeds, smaller image, the lock-step kernel): nothing of the first call may leak in
    sc = scenes.main_lua_frame(0)
    a, b, _, _ = render_both(ctx, sc, 64, 36, 50, stream[:5])
    assert a[..., :3].max() > 0
    assert np.array_equal(bits(a), bits(b))
    # and the streaming kernel again with fewer pixels than the first call drew from the queue
    sc = scenes.dense_random(100, seed=4)
    a, b, ca, cb = render_both(ctx, sc, 48, 27, 50, stream[:3], count_bounces=True)
    assert np.array_equal(bits(a), bits(b)) and ca == cb


def test_noalloc_continues_partitions_and_refuses_past_its_limits(ctx):
    sc = scenes.glass_lattice()
    s = seedmod.glibc_rand_stream(9)
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(96, 54)
    ctx.render(sc, 96, 54, 32, s)
    full = ctx.image_read()
    ctx.image_alloc(96, 54)
    for r in range(3):
        ctx.render(sc, 96, 54, 32, s[:4], rank=r, world=3, band_rows=5, noalloc=True)
    for r in range(3):
        ctx.render(sc, 96, 54, 32, s[4:], first_frame=4, rank=r, world=3, band_rows=5, noalloc=True)
    assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(full[..., :3]))
    with pytest.raises(hip.PtError, match="PT_NOALLOC_MAX_FRAMES"):
        ctx.render(sc, 96, 54, 32, seedmod.glibc_rand_stream(4097), noalloc=True)
    with pytest.raises(hip.PtError, match="accel"):
        ctx.render(sc, 96, 54, 32, s, noalloc=True, accel=1)
    ctx.image_alloc(8, 8)
    with pytest.raises(hip.PtError):
        ctx.render(sc, 96, 54, 32, s, noalloc=True)      # accumulator size mismatch


def _hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    lib.hipStreamDestroy.argtypes = [C.c_void_p]
    lib.hipStreamSynchronize.argtypes = [C.c_void_p]
    lib.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    lib.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    lib.hipGraphExecDestroy.argtypes = [C.c_void_p]
    lib.hipGraphDestroy.argtypes = [C.c_void_p]
    lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return lib


def _free_bytes(lib):
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert lib.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("scene", ["frame0", "dense500"])      # the lock-step kernel / the streaming kernel (queue head in scratch)
def test_noalloc_allocates_nothing_and_replays_from_a_graph(ctx, scene):
    lib = _hip()
    sc = scenes.main_lua_frame(0) if scene == "frame0" else scenes.dense_random(500)
    s = seedmod.glibc_rand_stream(6)
    W, H = 96, 54
    want = hip.render_scene(ctx, sc, W, H, 50, s)
    ctx.image_alloc(W, H)
    ctx.sync()
    ctx.render(sc, W, H, 50, s, noalloc=True)                  # warm: code objects loaded, nothing left to set up lazily
    before = _free_bytes(lib)
    ctx.render(sc, W, H, 50, s, noalloc=True)
    assert _free_bytes(lib) == before, "pt_render_noalloc changed the device's free memory"
    assert np.array_equal(bits(ctx.image_read()), bits(want))
    # under capture (global mode: an allocation would be an error): only the enqueue-and-return form is accepted
    st = C.c_void_p()
    assert lib.hipStreamCreate(C.byref(st)) == 0
    try:
        assert lib.hipStreamBeginCapture(st, 0) == 0
        try:
            with pytest.raises(hip.PtError, match="captured"):
                ctx.render(sc, W, H, 50, s, noalloc=True, stream=st.value)                       # async = 0 synchronises
            with pytest.raises(hip.PtError, match="captured"):
                ctx.render(sc, W, H, 50, s, noalloc=True, stream=st.value, async_=True, count_bounces=True)
            ctx.render(sc, W, H, 50, s, noalloc=True, stream=st.value, async_=True)
        finally:
            graph = C.c_void_p()
            rc = lib.hipStreamEndCapture(st, C.byref(graph))
        assert rc == 0 and graph.value
        gexec = C.c_void_p()
        assert lib.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        for _ in range(2):
            # another image in the accumulator first (one sample): the replay must write every pixel itself.  (Not a fresh
            # pt_image_alloc: the graph holds the accumulator's address as it was at capture.)
            ctx.render(sc, W, H, 50, s[:1], noalloc=True)
            assert not np.array_equal(bits(ctx.image_read()), bits(want))
            assert lib.hipGraphLaunch(gexec, st) == 0
            assert lib.hipStreamSynchronize(st) == 0
            assert np.array_equal(bits(ctx.image_read()), bits(want))
        lib.hipGraphExecDestroy(gexec)
        lib.hipGraphDestroy(graph)
    finally:
        lib.hipStreamDestroy(st)


def test_noalloc_refuses_to_overlap_a_call_on_another_stream(ctx):
    sc = scenes.main_lua_frame(0)
    s = seedmod.glibc_rand_stream(256)
    other = hip.Context(0)                                       # same GPU, its own stream, the same scratch array
    try:
        for c in (ctx, other):
            c.scene_upload(sc["spheres"])
            c.image_alloc(640, 360)
        ctx.render(sc, 640, 360, 50, s, noalloc=True, async_=True)             # ~10 ms of kernel
        with pytest.raises(hip.PtError, match="still running on another stream"):
            other.render(sc, 640, 360, 50, s[:2], noalloc=True, async_=True)
        ctx.render(sc, 640, 360, 50, s[:2], noalloc=True, async_=True)         # the same stream is ordered: accepted
        ctx.sync()
        other.render(sc, 640, 360, 50, s[:2], noalloc=True)                    # the first call has finished: accepted
        a = other.image_read()
        ctx.sync()
        assert np.array_equal(bits(a), bits(ctx.image_read()))
    finally:
        other.close()
