"""Wall time of the C scene driver (reference API path, PNG files included) with and without the frame pipeline."""
import os, subprocess, sys, tempfile, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames, W, H, spp = (int(x) for x in (sys.argv[1:5] + ["24", "640", "360", "200"][len(sys.argv) - 1:]))
with tempfile.TemporaryDirectory() as d:
    for rep in range(2):
        for p in (0, 1, 2):
            env = dict(os.environ, PT_PIPELINE=str(p), PT_DRIVER_TIMING="1")
            t0 = time.perf_counter()
            subprocess.check_call([os.path.join(root, "pathtracer_amd", "pt_scene_driver"), "0", str(frames), str(W), str(H), str(spp), "50",
                                   os.path.join(d, "f%d.png")], env=env, cwd=d, stdout=subprocess.DEVNULL, timeout=200)
            dt = time.perf_counter() - t0
            print("PT_PIPELINE=%d: %d frames %dx%d %d spp in %.3f s -> %.1f Msamples/s incl. process start, upload, PNG encode" %
                  (p, frames, W, H, spp, dt, frames * W * H * spp / dt / 1e6), flush=True)
