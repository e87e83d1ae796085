import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd import hip, scenes, seeds as seedmod
name = sys.argv[1]; spp = int(sys.argv[2])
sc = {"dense1k": lambda: scenes.dense_random(1000), "dense10k": lambda: scenes.dense_random(10000), "frame0": lambda: scenes.main_lua_frame(0)}[name]()
ctx = hip.Context(0); ctx.scene_upload(sc["spheres"]); ctx.image_alloc(1920, 1080)
bc = ctx.render(sc, 1920, 1080, 50, seedmod.glibc_rand_stream(spp), count_bounces=True)
img = ctx.image_read()
b = img[..., 3]
print("total", bc, "sum of per-pixel", b.sum(), "mean per pixel", b.mean(), "max", b.max(), "p99", np.percentile(b, 99), "p99.9", np.percentile(b, 99.9))
h = np.bincount(np.minimum(b.astype(np.int64).ravel() // spp, 60))
print("histogram of bounces/spp per pixel:", h)
rows = b.reshape(1080, 1920).mean(axis=1)
print("row means (every 90th):", rows[::90])
big = np.argwhere(b > 200)
print("pixels with >200:", len(big), big[:10].tolist())
import time
for mode in ("",):
    t0 = time.perf_counter(); ms = ctx.render(sc, 1920, 1080, 50, seedmod.glibc_rand_stream(spp), time_kernel=True); print("kernel ms", ms)
# time only the top half / bottom half rows via world partition of 2 bands
H = 1080
for r in range(4):
    ctx.image_alloc(1920, 1080)
    ms = ctx.render(sc, 1920, 1080, 50, seedmod.glibc_rand_stream(spp), rank=r, world=4, band_rows=270, time_kernel=True)
    print("rows %d-%d: %.2f ms" % (r * 270, r * 270 + 269, ms))
