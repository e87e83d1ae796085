"""Differential fuzz against THE REFERENCE KERNEL ITSELF on the GPU (oracle/_ref/ptref_gfx950.hsaco, tests/gpuref.py):
random scenes in general position (1 .. 130 spheres, boxed and open, all four materials on spheres AND walls, cameras
anywhere), rendered by the reference's own code object with its launch-per-sample loop and by both HIP builds.

  * The relaxed build (the arithmetic of that code object, pathtracer_amd/csrc/pt_device.h) must EQUAL the reference
    kernel's accumulator BIT FOR BIT on every scene, through whichever kernel pt_render picks and through the other one.
    Chaotic scenes — a box crowded with mirrors and glass, a dielectric wall of radius 10^4 whose hit points are only good
    to 1e-3 — are no exception: equal arithmetic takes equal decisions.
  * The strict build is a legal IEEE build of the reference (bit-identical to the oracle, tests/test_gpu_fuzz.py); its
    distance from the reference kernel's image is reported, and on reference-like scenes (diffuse or emissive walls, at
    most 3 specular / glass spheres) it stays within the stated flip budget (tests/envelope.py); image means agree.
PT_REFFUZZ_CASES / PT_REFFUZZ_SEED: number of scenes and first seed (defaults 150 and 5000); PT_REFFUZZ_OUT: write the
run's totals as JSON (one-off runs of 20 000 and of 200 000 scenes are recorded in profiles/r03_gpuref_fuzz.json and r03_gpuref_fuzz_200k.json)."""
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
        for c in ((0, w + 6, -10), (0, -(w + 6), -10), (0, 0, -(w + 20)), (0, 0, w + 20), (w + 6, 0, -5), (-(w + 6), 0, -5)):
            rows.append([c[0], c[1], c[2], w] + random_material(rng))
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


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def off_pixels(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=-1)
    return int((~(d <= 1e-4) & ~(np.isnan(a).any(axis=-1) & np.isnan(b).any(axis=-1))).sum())


def reference_like(sc):
    sp = sc["spheres"]
    walls = sp[sp[:, 3] > 100.0]
    small = sp[sp[:, 3] <= 100.0]
    return bool((walls[:, 4] < 2.0).all()) and int((small[:, 4] >= 2.0).sum()) <= 3


def test_random_scenes_against_the_reference_kernel(ctx):
    cases = int(os.environ.get("PT_REFFUZZ_CASES", "150"))
    seed0 = int(os.environ.get("PT_REFFUZZ_SEED", "5000"))
    stream = seedmod.glibc_rand_stream(16)
    ref = gpuref.ReferenceKernel()
    like = {"off": 0, "tests": 0.0, "scenes": 0}
    off_all, tests_all, mean = 0, 0.0, {"ref": 0.0, "strict": 0.0}
    saved = os.environ.get("PT_KERNEL")
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
            tag = "seed %d (%d spheres, %dx%d, %d spp, depth %d)" % (seed0 + case, n, W, H, spp, depth)
            for kernel in (None, "lds", "stream"):
                os.environ.pop("PT_KERNEL", None)
                if kernel:
                    os.environ["PT_KERNEL"] = kernel
                img = hip.render_scene(ctx, sc, W, H, depth, s, fast_math=True)[..., :3]
                same = (bits(img) == bits(want)).all(axis=-1) | (np.isnan(img).any(axis=-1) & np.isnan(want).any(axis=-1))
                assert same.all(), "%s, kernel %s: %d pixels of the relaxed build differ in bits from the reference kernel's" % (tag, kernel or "default", int((~same).sum()))
            os.environ.pop("PT_KERNEL", None)
            img, bounces = hip.render_scene(ctx, sc, W, H, depth, s, count_bounces=True)
            off = off_pixels(img[..., :3], want)
            tests = float(bounces) * max(n, 1)
            off_all += off
            tests_all += tests
            mean["strict"] += float(np.nan_to_num(img[..., :3], nan=0.0, posinf=0.0).mean(dtype=np.float64))
            mean["ref"] += float(np.nan_to_num(want, nan=0.0, posinf=0.0).mean(dtype=np.float64))
            if reference_like(sc):
                like["scenes"] += 1
                like["tests"] += tests
                like["off"] += off
            if case % 2000 == 1999:
                print("reference fuzz: %d scenes so far" % (case + 1), flush=True)
    finally:
        ref.close()
        os.environ.pop("PT_KERNEL", None)
        if saved is not None:
            os.environ["PT_KERNEL"] = saved
    print("reference fuzz: %d scenes x 3 kernel choices: the relaxed build equals the reference kernel bit for bit; strict build: %d pixels "
          "farther than 1e-4 (%.3g per ray-sphere test; %d reference-like scenes: %.3g, stated 3e-6)"
          % (cases, off_all, off_all / max(tests_all, 1.0), like["scenes"], like["off"] / max(like["tests"], 1.0)))
    out = os.environ.get("PT_REFFUZZ_OUT")
    if out:
        json.dump({"cases": cases, "first_seed": seed0, "relaxed_build_pixels_differing_in_bits": 0,
                   "strict_build": {"off_pixels": off_all, "tests": tests_all, "flips_per_test": off_all / max(tests_all, 1.0), "reference_like": like},
                   "mean_of_image_means": {k: v / cases for k, v in mean.items()}, "stated_flips_per_test": envelope.FLIPS_PER_TEST},
                  open(out, "w"), indent=1)
    assert abs(mean["strict"] - mean["ref"]) <= 0.01 * mean["ref"] + 1e-6, mean
