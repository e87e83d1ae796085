#!/usr/bin/env python3
"""How many pixels of the relaxed (and strict) HIP build differ IN BITS from the frozen output of the reference kernel on
MI355X (tests/golden/gpuref_*.npz)?  Debugging aid for the relaxed build's arithmetic (pt_device.h).
    python3 tools/gpuref_bits.py [case ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import envelope
from cases import CASES
from pathtracer_amd import hip
GOLD = os.path.join(ROOT, "tests", "golden")
ctx = hip.Context(0)
names = sys.argv[1:] or (list(CASES) + list(envelope.ENVELOPE_CASES))
tot = 0
for name in names:
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    want = np.load(os.path.join(GOLD, "gpuref_%s.npz" % name))["rgb"]
    out = []
    for fast in (True, False):
        img = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], fast_math=fast)[..., :3]
        same = (np.ascontiguousarray(img).view(np.uint32) == want.view(np.uint32)).all(axis=-1) | (np.isnan(img).any(axis=-1) & np.isnan(want).any(axis=-1))
        d = np.abs(img.astype(np.float64) - want)
        out.append((int((~same).sum()), float(np.nanmax(d)) if d.size else 0.0, int((~(d.max(axis=-1) <= 1e-4)).sum())))
    tot += out[0][0]
    print("%-26s %4dx%-4d spp %2d | relaxed: %5d pixels differ in bits (max abs %.3g, %d beyond 1e-4) | strict: %5d (%.3g, %d)"
          % (name, W, H, spp, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2]), flush=True)
print("relaxed build: %d pixels differ in bits over %d cases" % (tot, len(names)))
