#!/usr/bin/env python3
"""Time the REFERENCE kernel (oracle/_ref/ptref_gfx950.hsaco) on the GPU with the reference's own host pattern — one
launch per sample, a device-wide sync after each (src/c/renderer.c:170-251) — and, with `compare`, put the HIP builds'
images of the same frame beside its image.  TEST INFRASTRUCTURE / measurement, outside bench.py's timed region.

    python3 tools/ref_gfx950_time.py [scene=frame0] [W=1920] [H=1080] [spp=256] [depth=50] [reps=2] [compare|nocompare] [nosync]

rocprofv3 target as well (`rocprofv3 --kernel-trace --stats -- python3 tools/ref_gfx950_time.py frame0 1920 1080 32 50 1 nocompare`).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gpuref  # noqa: E402
from pathtracer_amd import scenes, seeds as seedmod  # noqa: E402

SCENES = {"frame0": lambda: scenes.main_lua_frame(0), "lattice": scenes.glass_lattice,
          "dense10k": lambda: scenes.dense_random(10000), "dense500": lambda: scenes.dense_random(500)}


def main():
    a = sys.argv[1:]
    name = a[0] if len(a) > 0 else "frame0"
    W = int(a[1]) if len(a) > 1 else 1920
    H = int(a[2]) if len(a) > 2 else 1080
    spp = int(a[3]) if len(a) > 3 else 256
    depth = int(a[4]) if len(a) > 4 else 50
    reps = int(a[5]) if len(a) > 5 else 2
    compare = "compare" in a[6:]
    sync = "nosync" not in a[6:]
    sc = SCENES[name]()
    s = seedmod.glibc_rand_stream(spp)
    ref = gpuref.ReferenceKernel(gpuref.HSACO)
    ref.upload(sc, W, H, depth)
    res = dict(scene=name, W=W, H=H, spp=spp, depth=depth, sync_per_sample=sync, runs=[])
    for r in range(reps):
        t0 = time.perf_counter()
        for k in range(spp):
            ref.launch(k, s[k], sync=sync)
        if not sync:
            gpuref._ck(gpuref._lib().hipDeviceSynchronize(), "hipDeviceSynchronize")
        dt = time.perf_counter() - t0
        res["runs"].append(dict(wall_ms=dt * 1e3, msamples_s=W * H * spp / dt / 1e6))
        print("reference kernel on GPU: %s %dx%d spp %d depth %d: %.2f ms (%.3f ms per launch%s) -> %.1f Msamples/s"
              % (name, W, H, spp, depth, dt * 1e3, dt * 1e3 / spp, " incl. sync" if sync else "", W * H * spp / dt / 1e6), flush=True)
    if compare:
        import envelope
        import oracle_lib
        from pathtracer_amd import hip
        orc = oracle_lib.load_oracle()
        img = ref.read()
        rgb = img[..., :3].copy()
        u8 = lambda x: orc.postprocess(np.concatenate([x, np.zeros(x.shape[:2] + (1,), np.float32)], axis=2)).reshape(x.shape)
        r_u8 = u8(rgb)
        ctx = hip.Context(0)
        for fast in (False, True):
            him, bc = hip.render_scene(ctx, sc, W, H, depth, s, fast_math=fast, count_bounces=True)
            ms = ctx.render(sc, W, H, depth, s, fast_math=fast, time_kernel=True)
            h_u8 = ctx.image_read_u8()
            h = him[..., :3]
            d = np.abs(h.astype(np.float64) - rgb).max(axis=-1)
            key = "hip_fast" if fast else "hip_strict"
            same = (np.ascontiguousarray(h).view(np.uint32) == rgb.view(np.uint32)).all(axis=-1)
            res[key] = dict(kernel_ms=ms, pixels=int(same.size), pixels_differing_in_bits=int((~same).sum()), share_1e4=float((d <= 1e-4).mean()), share_1e3=float((d <= 1e-3).mean()), share_1e2=float((d <= 1e-2).mean()),
                            off_1e4=int((d > 1e-4).sum()), max_abs=float(d.max()), psnr_u8=envelope.psnr_u8(h_u8, r_u8),
                            u8_equal_share=float((h_u8 == r_u8).all(axis=-1).mean()),
                            u8_max_diff=int(np.abs(h_u8.astype(np.int32) - r_u8.astype(np.int32)).max()),
                            mean=float(h.mean(dtype=np.float64)), ref_mean=float(rgb.mean(dtype=np.float64)),
                            bounces_per_sample=bc / (W * H * spp),
                            budget=envelope.off_pixel_budget(bc, sc["spheres"].shape[0]))
            print(key, json.dumps(res[key]), flush=True)
        ctx.close()
    out = os.path.join(ROOT, "gpurun_out", "gpuref")
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "time_%s_%dx%d_%d%s.json" % (name, W, H, spp, "" if sync else "_nosync")), "w"), indent=1)
    ref.close()


if __name__ == "__main__":
    main()
