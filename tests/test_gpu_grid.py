"""The opt-in uniform grid (pt_render_args.accel = 1 / PT_ACCEL=grid; SURVEY.md §8f rank 4, no reference
counterpart): it must return the closest hit of the reference's linear scan — same arithmetic per tested sphere,
lowest index on equal distance — so the image is BIT-IDENTICAL to the brute-force kernels', which are themselves
pinned to the oracle.  Checked on every fixture scene and on dense tables at sizes where the two can disagree
only if the walk skips a cell."""
import os

import numpy as np
import pytest

from cases import CASES
from pathtracer_amd import hip, scenes
from pathtracer_amd import seeds as seedmod

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", sorted(CASES))
def test_grid_matches_reference_fixture(ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    img, bc = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True, accel=1)
    assert np.array_equal(bits(img[..., :3]), bits(g["strict"])), "%d pixels differ" % int((bits(img[..., :3]) != bits(g["strict"])).any(axis=-1).sum())
    _, bs = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
    assert bc == bs


@pytest.mark.parametrize("n,W,H,spp,seed", [(10000, 160, 90, 4, 12345), (100000, 128, 72, 2, 12345), (30000, 96, 54, 3, 77),
                                             (2000, 256, 144, 4, 5)])
def test_grid_matches_linear_scan_on_dense_tables(ctx, n, W, H, spp, seed):
    sc = scenes.dense_random(n, seed=seed)
    s = seedmod.glibc_rand_stream(spp)
    a, ba = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
    b, bb = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True, accel=1)
    ne = (bits(a) != bits(b)).any(axis=-1)
    assert not ne.any(), "%d of %d pixels differ" % (int(ne.sum()), ne.size)
    assert ba == bb


def test_grid_full_size_and_env_switch(ctx):
    sc = scenes.dense_random(10000)
    s = seedmod.glibc_rand_stream(1)
    a = hip.render_scene(ctx, sc, 1920, 1080, 50, s)
    try:
        os.environ["PT_ACCEL"] = "grid"
        b = hip.render_scene(ctx, sc, 1920, 1080, 50, s)
    finally:
        os.environ.pop("PT_ACCEL", None)
    assert np.array_equal(bits(a), bits(b))
    # partitions and continued accumulators work through the grid kernel too
    s3 = seedmod.glibc_rand_stream(3)
    full = hip.render_scene(ctx, sc, 320, 180, 50, s3, accel=1)
    ctx.image_alloc(320, 180)
    for r in range(3):
        ctx.render(sc, 320, 180, 50, s3[:1], rank=r, world=3, band_rows=7, accel=1)
    for r in range(3):
        ctx.render(sc, 320, 180, 50, s3[1:], first_frame=1, rank=r, world=3, band_rows=7, accel=1)
    assert np.array_equal(bits(ctx.image_read()), bits(full))
