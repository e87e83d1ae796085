"""Fast-math build against the strict reference fixtures: per case, share of pixels within 1e-4 / 1e-5, max abs, bounce counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cases import CASES
from pathtracer_amd import hip
ctx = hip.Context(0)
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in sorted(CASES):
    g = np.load(os.path.join(gold, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    img, bc = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True, fast_math=True)
    _, bs = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
    d = np.abs(img[..., :3] - g["strict"]).max(axis=2)
    mse = float(((np.clip(img[..., :3], 0, 1) - np.clip(g["strict"], 0, 1)) ** 2).mean())
    psnr = 10 * np.log10(1.0 / mse) if mse > 0 else float("inf")
    print("%-22s within 1e-5 %.4f  1e-4 %.4f  max %.3g  psnr %.1f dB  bounces fast/strict %d/%d  nan %d" %
          (name, (d <= 1e-5).mean(), (d <= 1e-4).mean(), d.max(), psnr, bc, bs, int(np.isnan(img).sum())), flush=True)
