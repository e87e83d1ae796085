"""pt_render_noalloc (include/pt_hip.h): the entry point for callers that cannot allocate.  Same kernels as
pt_render, working memory in the library's per-GPU `scratch` array: its image must be pt_render's bit for bit —
on two calls in a row, the second with a smaller input, on both kernels (the streaming kernel is the one that
uses the queue head in scratch), with and without the bounce counter, and past the limit it must refuse."""
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
