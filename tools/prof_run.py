"""Profiling driver: one 1080p render of a named scene through the C ABI, no torch (rocprofv3 target).

    python3 tools/prof_run.py [scene=frame0] [spp=32] [reps=1] [depth=50] [W=1920] [H=1080] [world=1] [rank=0] [band_rows=8]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import hip, scenes, seeds as seedmod

SCENES = {
    "frame0": lambda: scenes.main_lua_frame(0),
    "lattice": scenes.glass_lattice,
    "lattice4": lambda: scenes.glass_lattice(4),
    "lattice5": lambda: scenes.glass_lattice(5),
    "lattice6": lambda: scenes.glass_lattice(6),
    "lattice7": lambda: scenes.glass_lattice(7),
    "dense10k": lambda: scenes.dense_random(10000),
    "dense100k": lambda: scenes.dense_random(100000),
    "dense1k": lambda: scenes.dense_random(1000),
    "dense500": lambda: scenes.dense_random(500),
    "dense100": lambda: scenes.dense_random(100),
}
for _n in (12, 16, 24, 32, 48, 64):      # small open scenes: where the kernel choice of pt_render switches
    SCENES["dense%d" % _n] = (lambda n: lambda: scenes.dense_random(n))(_n)


def main():
    a = sys.argv[1:]
    name = a[0] if len(a) > 0 else "frame0"
    spp = int(a[1]) if len(a) > 1 else 32
    reps = int(a[2]) if len(a) > 2 else 1
    depth = int(a[3]) if len(a) > 3 else 50
    W = int(a[4]) if len(a) > 4 else 1920
    H = int(a[5]) if len(a) > 5 else 1080
    world = int(a[6]) if len(a) > 6 else 1       # render only one rank's bands of a `world`-way split ...
    rank = int(a[7]) if len(a) > 7 else 0         # ... this rank's
    band = int(a[8]) if len(a) > 8 else 8
    fast = os.environ.get("PT_MATH") == "fast"
    sc = SCENES[name]()
    s = seedmod.glibc_rand_stream(spp)
    ctx = hip.Context(0)
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(W, H)
    if os.environ.get("PT_COUNT", "1") != "0":
        cs = s[:min(spp, 2)]
        bc = ctx.render(sc, W, H, depth, cs, count_bounces=True)
        print("bounce iterations per sample: %.3f (over %d spp)" % (bc / (W * H * len(cs)), len(cs)), flush=True)
    for r in range(reps):
        t0 = time.perf_counter()
        ms = ctx.render(sc, W, H, depth, s, time_kernel=True, rank=rank, world=world, band_rows=band)
        dt = time.perf_counter() - t0
        print("%s %dx%d spp %d depth %d: kernel %.3f ms (wall %.3f ms) -> %.1f Msamples/s" %
              (name, W, H, spp, depth, ms, dt * 1e3, W * H * spp / world / ms / 1e3), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
