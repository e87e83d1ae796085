"""Differential fuzz against THE REFERENCE KERNEL ITSELF on the GPU (oracle/_ref/ptref_gfx950.hsaco, tests/gpuref.py):
random scenes in general position (1 .. 130 spheres, boxed and open, all four materials, cameras anywhere), rendered by
the reference's own code object with its launch-per-sample loop and by both HIP builds.

What can be asserted on ARBITRARY scenes is relative.  The absolute flip budget of tests/envelope.py (3e-6 per
ray-sphere test) is a property of scenes like the reference's; a random box crowded with mirrors and glass is chaotic —
the reference's hit point on a radius-10^4 wall is only good to ulp(10^4) = 1e-3 (src/cl/ray.cl:80: -b - sqrt(disc)
cancels), specular bounces amplify that, and two x86 builds of the reference itself then differ at 1e-5 .. 1e-4 flips per
test (a dielectric wall: in a third of the pixels).  The strict HIP build IS a legal IEEE build of the reference bit for
bit (tests/test_gpu_fuzz.py), so its distance from the reference kernel's image is the yardstick, scene by scene:
  * the relaxed build is no farther from the reference kernel's image than the strict build is (Poisson slack);
  * over the run, the relaxed build's flipped pixels stay within 1.2 x the strict build's;
  * on reference-like scenes (diffuse or emissive walls, no more than 3 specular/glass spheres) the absolute budget holds too;
  * image means agree (flips are unbiased).
PT_REFFUZZ_CASES / PT_REFFUZZ_SEED: number of scenes and first seed (defaults 120 and 5000); PT_REFFUZZ_OUT: write the
run's totals as JSON (a one-off run of 20 000 scenes is recorded in profiles/r03_gpuref_fuzz.json)."""
import json
import os

import numpy as np
import pytest

import envelope
import gpuref
from pathtracer_amd import hip, seeds as seedmod
from test_gpu_fuzz import SIZES, random_material

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not gpuref.available(), reason="oracle/_ref/ptref_gfx950.hsaco not built (needs /root/reference at build time)")]


def general_scene(rng):
    n = int(rng.choice(SIZES))
    box = rng.rand() < 0.5 and n >= 7
    rows = []
    if box:
        w = 10000.0
        # walls as in the reference's scene: diffuse (or emissive).  A SPECULAR or DIELECTRIC sphere of radius 10^4 is
        # outside any tolerance: the reference's hit point on it is good to ulp(10^4) = 1e-3 while its self-hit offset
        # is 1e-4 (src/cl/ray.cl:119), so whether the refracted / reflected ray re-hits the wall is decided by the last
        # bit of a square root — two x86 builds of the reference differ in a third of the pixels of such a scene
        for c in ((0, w + 6, -10), (0, -(w + 6), -10), (0, 0, -(w + 20)), (0, 0, w + 20), (w + 6, 0, -5), (-(w + 6), 0, -5)):
            m = random_material(rng)
            while m[0] >= 2.0:
                m = random_material(rng)
            rows.append([c[0], c[1], c[2], w] + m)
    while len(rows) < n:
        c = [rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-15, -5)] if box else [rng.uniform(-6, 6) for _ in range(3)]
        rows.append([float(x) for x in c] + [float(rng.uniform(0.2, 2.0 if box else 3.0))] + random_material(rng))
    if box:
        rng.shuffle(rows)
    pos = [rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-14, -6)] if box else [rng.uniform(-8, 8) for _ in range(3)]
    rot = [float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2)), 0.0]
    # exposure * aperture = 1, as in the reference's own scene (1000 * 0.001): pixel values stay below ~10, where the
    # tolerance's absolute 1e-4 means something (a wall that is a light at exposure * aperture = 100 gives values
    # near 1000, whose float spacing alone is 6e-5: there even the x86 builds of the reference differ in every pixel)
    ap = float(rng.choice([0.001, 0.1]))
    cam = [float(x) for x in pos] + rot + [15.0, 20.0, ap, 1.0 / ap]
    bg = [0.0, 0.0, 0.0] if box and rng.rand() < 0.7 else [float(rng.choice([0.0, 0.5, 0.7, 1.0])) for _ in range(3)]
    return {"spheres": np.asarray(rows, np.float64).astype(np.float32), "bg": np.asarray(bg, np.float32),
            "cam": np.asarray(cam, np.float64).astype(np.float32)}


def off_pixels(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=-1)
    return int((~(d <= 1e-4) & ~(np.isnan(a).any(axis=-1) & np.isnan(b).any(axis=-1))).sum())


def reference_like(sc):
    sp = sc["spheres"]
    walls = sp[sp[:, 3] > 100.0]
    small = sp[sp[:, 3] <= 100.0]
    return bool((walls[:, 4] < 2.0).all()) and int((small[:, 4] >= 2.0).sum()) <= 3


def test_random_scenes_against_the_reference_kernel(ctx):
    cases = int(os.environ.get("PT_REFFUZZ_CASES", "120"))
    seed0 = int(os.environ.get("PT_REFFUZZ_SEED", "5000"))
    stream = seedmod.glibc_rand_stream(16)
    ref = gpuref.ReferenceKernel()
    tot = {"strict": 0, "relaxed": 0}
    like = {"strict": 0, "relaxed": 0, "tests": 0.0, "scenes": 0}
    tests_all, mean = 0.0, {"ref": 0.0, "strict": 0.0, "relaxed": 0.0}
    try:
        for case in range(cases):
            rng = np.random.RandomState(seed0 + case)
            sc = general_scene(rng)
            W, H = [(48, 32), (33, 17), (64, 36)][rng.randint(3)]
            spp = int(rng.choice([1, 5, 16]))
            depth = int(rng.choice([8, 50]))
            s = stream[:spp]
            n = len(sc["spheres"])
            want = ref.render(sc, W, H, depth, s)[..., :3]
            off, bounces = {}, 0
            for name, fast in (("strict", False), ("relaxed", True)):
                img, bounces = hip.render_scene(ctx, sc, W, H, depth, s, fast_math=fast, count_bounces=True)
                off[name] = off_pixels(img[..., :3], want)
                tot[name] += off[name]
                mean[name] += float(np.nan_to_num(img[..., :3], nan=0.0, posinf=0.0).mean(dtype=np.float64))
            mean["ref"] += float(np.nan_to_num(want, nan=0.0, posinf=0.0).mean(dtype=np.float64))
            tests = float(bounces) * max(n, 1)
            tests_all += tests
            budget = envelope.off_pixel_budget(bounces, n)
            tag = "seed %d (%d spheres, %dx%d, %d spp, depth %d): pixels off strict %d, relaxed %d, stated for reference-like scenes %d" % (
                seed0 + case, n, W, H, spp, depth, off["strict"], off["relaxed"], budget)
            assert off["relaxed"] <= off["strict"] + 6.0 * np.sqrt(off["strict"] + 4.0) + budget, tag
            if reference_like(sc):
                like["scenes"] += 1
                like["tests"] += tests
                for k in ("strict", "relaxed"):
                    like[k] += off[k]
            if case % 2000 == 1999:
                print("reference fuzz: %d scenes so far" % (case + 1), flush=True)
    finally:
        ref.close()
    print("reference fuzz: %d scenes, %d reference-like; pixels off the reference kernel's image: strict %d, relaxed %d "
          "(%.3g / %.3g per ray-sphere test; reference-like scenes %.3g / %.3g, stated 3e-6)"
          % (cases, like["scenes"], tot["strict"], tot["relaxed"], tot["strict"] / tests_all, tot["relaxed"] / tests_all,
             like["strict"] / max(like["tests"], 1.0), like["relaxed"] / max(like["tests"], 1.0)))
    out = os.environ.get("PT_REFFUZZ_OUT")
    if out:
        json.dump({"cases": cases, "first_seed": seed0, "off_pixels": tot, "tests": tests_all, "reference_like": like,
                   "mean_of_image_means": {k: v / cases for k, v in mean.items()}, "stated_flips_per_test": envelope.FLIPS_PER_TEST},
                  open(out, "w"), indent=1)
    assert tot["relaxed"] <= 1.2 * tot["strict"] + 50
    for k in ("strict", "relaxed"):
        assert abs(mean[k] - mean["ref"]) <= 0.01 * mean["ref"] + 1e-6, (k, mean)
        assert like[k] <= 8 + envelope.FLIPS_PER_TEST * like["tests"], (k, like)
