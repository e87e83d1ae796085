#!/usr/bin/env python3
"""How far apart are LEGAL builds of the reference?  Renders envelope cases with every x86 build of the reference kernel
(oracle/_ref: strict, fast = the reference's own flags, env1..3 = those flags + other builtin definitions) and prints the
pairwise statistics the parity tolerance is stated in: share of pixels within 1e-4, max abs, PSNR of the 8-bit image.
Container-only (needs oracle/_ref)."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
from envelope import ENVELOPE_CASES, psnr_u8, share_within  # noqa: E402
from pathtracer_amd import seeds as seedmod  # noqa: E402


def main():
    kinds = ("strict", "fast", "env1", "env2", "env3")
    libs = {k: oracle_lib.load_ref(k) for k in kinds}
    post = oracle_lib.load_oracle().postprocess
    for name, (factory, W, H, depth, spp) in ENVELOPE_CASES.items():
        sc, s = factory(), seedmod.glibc_rand_stream(spp)
        imgs = {k: libs[k].render(sc, W, H, depth, s)[..., :3].copy() for k in kinds}
        print("%s  %dx%d depth %d spp %d  (%.1f M bounce-ish samples)" % (name, W, H, depth, spp, W * H * spp / 1e6))
        for a, b in itertools.combinations(kinds, 2):
            d = np.abs(imgs[a] - imgs[b]).max(axis=2)
            pa = post(np.concatenate([imgs[a], np.zeros((H, W, 1), np.float32)], axis=2)).reshape(H, W, 3)
            pb = post(np.concatenate([imgs[b], np.zeros((H, W, 1), np.float32)], axis=2)).reshape(H, W, 3)
            print("   %-6s vs %-6s  within 1e-4: %.5f  (%5d px off)  max abs %.3g  PSNR8 %.2f dB  mean %.5f/%.5f" %
                  (a, b, share_within(imgs[a], imgs[b]), int((d > 1e-4).sum()), float(d.max()), psnr_u8(pa, pb),
                   float(imgs[a].mean()), float(imgs[b].mean())))


if __name__ == "__main__":
    main()
