"""Debug helper: host-side timing of one render call's phases on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pathtracer_amd import hip, scenes, seeds as seedmod
sc = scenes.main_lua_frame(0); s = seedmod.glibc_rand_stream(256)
ctx = hip.Context(0); ctx.scene_upload(sc["spheres"])
frame = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode in ("torch-stream", "ctx-stream"):
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.render(sc, 1920, 1080, 50, s, d_image=frame.data_ptr(), stream=st if mode == "torch-stream" else None, async_=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); ctx.sync(); t2 = time.perf_counter()
        print(mode, i, "enqueue %.1f ms  total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)

for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ms = ctx.render(sc, 1920, 1080, 50, s, d_image=frame.data_ptr(), time_kernel=True)
    t2 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.render(sc, 1920, 1080, 50, s, d_image=frame.data_ptr(), stream=st, async_=True); e1.record(); torch.cuda.synchronize()
    print("hipEvent kernel_ms %.2f  wall %.2f ms   torch-event %.2f" % (ms, (t2 - t0) * 1e3, e0.elapsed_time(e1)), flush=True)
