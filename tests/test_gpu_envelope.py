"""Float parity against the ENVELOPE of legal reference builds (tests/envelope.py).

The reference pins no OpenCL runtime and builds with -cl-fast-relaxed-math (/root/reference/src/c/renderer.c:13), so
its own float output is a family of images: tests/golden/env_*.npz hold the same scene through five x86 builds of the
reference kernel (strict; its own flags with four legal definitions of sin/cos/sqrt/pow/dot/length/normalize).  Bar:
  * the strict HIP build IS one member (the strict build), bit for bit;
  * the strict AND the relaxed (pt_fast: the reference's own gfx950 arithmetic, pinned bit for bit elsewhere) HIP builds are, against EVERY member, within the stated tolerance — share of
    pixels within 1e-4 on the linear image, PSNR of the 8-bit image (device epilogue vs the reference's host
    post-process), image mean — which is the tolerance the members meet among themselves
    (tests/test_oracle_golden.py::test_stated_tolerance_holds_over_the_envelope_of_legal_reference_builds)."""
import os

import numpy as np
import pytest

import envelope
from pathtracer_amd import hip

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("name", sorted(envelope.ENVELOPE_CASES))
def test_hip_builds_sit_inside_the_envelope(ctx, oracle, name, fast):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    _, bounces = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], fast_math=fast, count_bounces=True)
    img = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], fast_math=fast)
    u8 = ctx.image_read_u8()
    rgb = img[..., :3]
    assert np.isfinite(rgb).all() and rgb.min() >= 0
    if not fast:
        assert np.array_equal(rgb.view(np.uint32), g["strict"].view(np.uint32)), "strict HIP build != strict reference build"
    for k in envelope.ENVELOPE_KINDS:
        member_u8 = oracle.postprocess(np.concatenate([g[k], np.zeros((H, W, 1), np.float32)], axis=2)).reshape(H, W, 3)
        envelope.check_against_member(name, rgb, u8, g[k], member_u8)
        assert envelope.off_pixels(rgb, g[k]) <= envelope.off_pixel_budget(bounces, sc["spheres"].shape[0]), k
