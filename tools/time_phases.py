"""Debug helper: HIP-event time vs wall-clock time of one render launch on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pathtracer_amd import hip, scenes, seeds as seedmod
sc = scenes.main_lua_frame(0)
ctx = hip.Context(0); ctx.scene_upload(sc["spheres"])
frame = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream(); st = side.cuda_stream
for spp in (8, 32, 128, 256):
    s = seedmod.glibc_rand_stream(spp)
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ms = ctx.render(sc, 1920, 1080, 50, s, d_image=frame.data_ptr(), time_kernel=True)
        t1 = time.perf_counter()
        with torch.cuda.stream(side):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            e0.record(); ctx.render(sc, 1920, 1080, 50, s, d_image=frame.data_ptr(), stream=st, async_=True); e1.record()
            torch.cuda.synchronize(); t3 = time.perf_counter()
        print("spp %3d  hipEvent %.2f ms  wall %.2f ms | torch-event %.2f ms wall %.2f ms" % (spp, ms, (t1 - t0) * 1e3, e0.elapsed_time(e1), (t3 - t2) * 1e3), flush=True)
