import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import hip, scenes, seeds as seedmod
sc = scenes.dense_random(1000)
ctx = hip.Context(0); ctx.scene_upload(sc["spheres"])
for (W, H, spp) in ((64, 1, 16), (64, 1, 64), (1920, 1, 16), (1920, 8, 16), (1920, 64, 16)):
    ctx.image_alloc(W, H)
    s = seedmod.glibc_rand_stream(spp)
    bc = ctx.render(sc, W, H, 50, s, count_bounces=True)
    b = ctx.image_read()[..., 3]
    ms = ctx.render(sc, W, H, 50, s, time_kernel=True)
    ms = ctx.render(sc, W, H, 50, s, time_kernel=True)
    print("%dx%d spp %d: total scans %d, max per pixel %d, kernel %.3f ms -> %.2f us per scan of the slowest pixel" % (W, H, spp, bc, b.max(), ms, ms * 1e3 / b.max()))
