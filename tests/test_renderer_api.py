"""The outer drop-in boundary: the reference's renderer.h API (include/renderer.h) over the HIP
backend, driven by the C twin of scripts/main.lua (pathtracer_amd/pt_scene_driver)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from png_decode import read_png_rgb8
from test_abi import declared_functions
from pathtracer_amd import scenes
from pathtracer_amd import seeds as seedmod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pathtracer_amd")


def test_renderer_library_exports_the_15_functions():
    lib = ctypes.CDLL(os.path.join(PKG, "libpt_renderer.so"))
    names = declared_functions("renderer.h")
    assert len(names) == 15, names
    for n in names:
        assert hasattr(lib, n), n


def test_seed_stream_is_glibc_default_rand():
    """renderer.c's private seed stream == glibc rand() with the default seed (SURVEY.md §8a lists
    the first 16 values) == pathtracer_amd.seeds."""
    lib = ctypes.CDLL(os.path.join(PKG, "libpt_renderer.so"))
    lib.pt_seed_stream_next.restype = ctypes.c_uint
    got = [lib.pt_seed_stream_next() for _ in range(2000)]
    assert got[:4] == [1804289383, 846930886, 1681692777, 1714636915]
    assert got == [int(x) for x in seedmod.glibc_rand_stream(2000)]
    libc = ctypes.CDLL(None)
    libc.srand(1)
    assert got[:500] == [libc.rand() for _ in range(500)]


def test_fallback_png_bmp_tga_writers(tmp_path):
    lib = ctypes.CDLL(os.path.join(PKG, "libpt_renderer.so"))
    rs = np.random.RandomState(0)
    for (w, h) in ((7, 5), (300, 200), (1, 1)):
        img = rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        p = str(tmp_path / ("t%dx%d.png" % (w, h))).encode()
        assert lib.pt_write_png(p, w, h, img.ctypes.data_as(ctypes.c_void_p)) == 1
        assert np.array_equal(read_png_rgb8(p.decode()), img)
        b = str(tmp_path / "t.bmp").encode()
        assert lib.pt_write_bmp(b, w, h, img.ctypes.data_as(ctypes.c_void_p)) == 1
        raw = open(b.decode(), "rb").read()
        stride = (3 * w + 3) // 4 * 4
        assert raw[:2] == b"BM" and len(raw) == 54 + stride * h
        rows = np.frombuffer(raw[54:], np.uint8).reshape(h, stride)[::-1, :3 * w].reshape(h, w, 3)
        assert np.array_equal(rows[..., ::-1], img)
        t = str(tmp_path / "t.tga").encode()
        assert lib.pt_write_tga(t, w, h, img.ctypes.data_as(ctypes.c_void_p)) == 1
        raw = open(t.decode(), "rb").read()
        assert len(raw) == 18 + 3 * w * h
        assert np.array_equal(np.frombuffer(raw[18:], np.uint8).reshape(h, w, 3)[::-1, :, ::-1], img)


def test_struct_layouts_match_the_reference():
    """Material 48 B / Sphere 80 B / Camera 48 B with the reference's offsets."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "renderer.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu ", sizeof(ImageInfo), sizeof(SceneInfo), sizeof(Material), sizeof(Sphere), sizeof(Camera));
        printf("%zu %zu %zu %zu %zu ", offsetof(Material, color), offsetof(Material, tint), offsetof(Material, fuzzyness), offsetof(Material, refIdx), offsetof(SceneInfo, sphereCount));
        printf("%zu %zu %zu %zu\n", offsetof(Sphere, radius), offsetof(Sphere, material), offsetof(Camera, rot), offsetof(Camera, sensorWidth));
        return 0;
    }'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "l.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "l.c"), "-o", os.path.join(d, "l")])
        out = subprocess.check_output([os.path.join(d, "l")]).split()
    assert [int(x) for x in out] == [16, 32, 48, 80, 48, 16, 32, 36, 40, 16, 16, 32, 16, 32]


def expected_u8(oracle, math, sc, W, H, depth, seeds):
    """The 8-bit image the drop-in must write.  math "strict" (PT_MATH=strict): the CPU oracle's.  math "reference" (the
    default of renderer.c): the REFERENCE KERNEL's own accumulator on this GPU (tests/gpuref.py) through the reference's
    host post-process (src/c/renderer.c:50-56,17-39, restated in the oracle library)."""
    if math == "strict":
        return oracle.postprocess(oracle.render(sc, W, H, depth, seeds)).reshape(H, W, 3)
    import gpuref
    k = gpuref.ReferenceKernel()
    try:
        img = k.render(sc, W, H, depth, seeds).copy()
    finally:
        k.close()
    img[..., 3] = 0
    return oracle.postprocess(img).reshape(H, W, 3)


def math_env(math):
    import gpuref
    if math == "reference" and not gpuref.available():
        pytest.skip("oracle/_ref/ptref_gfx950.hsaco not built (needs /root/reference at build time)")
    return {"PT_MATH": "strict"} if math == "strict" else {}


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["reference", "strict"])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_scene_driver_frames_match_oracle(tmp_path, oracle, pipeline, math):
    """Animation frames through create_renderer/…/render_to_file in a fresh process: the PNGs must hold exactly the 8-bit
    image of the reference kernel on this GPU (the default arithmetic of the drop-in) or, with PT_MATH=strict, of the CPU
    oracle; every frame continues the rand() seed stream.  pipeline=1: the frames are queued to the worker thread
    (PT_PIPELINE) — same files."""
    W, H, spp = 96, 54, 8
    frames = (3, 4, 5, 6, 7) if pipeline else (3, 4)
    env = dict(os.environ, PT_NUM_GPUS="1", PT_PIPELINE=str(pipeline))
    env.pop("PT_MATH", None)
    env.update(math_env(math))
    pat = str(tmp_path / "frame%d.png")
    subprocess.check_call([os.path.join(PKG, "pt_scene_driver"), "3", str(len(frames)), str(W), str(H), str(spp), "50", pat],
                          env=env, cwd=str(tmp_path), timeout=100)
    stream = seedmod.glibc_rand_stream(len(frames) * spp)
    for k, i in enumerate(frames):
        sc = scenes.main_lua_frame(i)
        want = expected_u8(oracle, math, sc, W, H, 50, stream[k * spp:(k + 1) * spp])
        got = read_png_rgb8(pat % i)
        assert np.array_equal(got, want), "frame %d: %d bytes differ" % (i, int((got != want).sum()))
        assert got.max() > 200


def _driver_frames(tmp_path, env_extra, frames, W=96, H=54, spp=8):
    env = dict(os.environ, PT_PIPELINE="0")
    env.pop("PT_MATH", None)
    env.update(env_extra)
    pat = str(tmp_path / "frame%d.png")
    subprocess.check_call([os.path.join(PKG, "pt_scene_driver"), str(frames[0]), str(len(frames)), str(W), str(H), str(spp), "50", pat],
                          env=env, cwd=str(tmp_path), timeout=200)
    return [read_png_rgb8(pat % i) for i in frames]


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["reference", "strict"])
@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_band_ranks_on_one_gpu_merge_to_the_oracle_image(tmp_path, oracle, ranks, math):
    """renderer.c's multi-rank path end to end on ONE GPU (PT_SHARE_GPU=1: PT_NUM_GPUS contexts on GPU 0): per-rank band
    launches with global ids, pt_combine_bands' merge into rank 0's accumulator (device copies here, RCCL transfers
    between distinct GPUs), one conversion, one read-back.  The PNGs must hold the oracle's 8-bit image."""
    W, H, spp, frames = 96, 54, 8, (3, 4)
    got = _driver_frames(tmp_path, dict({"PT_NUM_GPUS": str(ranks), "PT_SHARE_GPU": "1"}, **math_env(math)), frames)
    stream = seedmod.glibc_rand_stream(len(frames) * spp)
    for k, i in enumerate(frames):
        want = expected_u8(oracle, math, scenes.main_lua_frame(i), W, H, 50, stream[k * spp:(k + 1) * spp])
        assert np.array_equal(got[k], want), "ranks %d frame %d: %d bytes differ" % (ranks, i, int((got[k] != want).sum()))


@pytest.mark.gpu
def test_two_gpus_combine_over_rccl(tmp_path, oracle):
    """PT_NUM_GPUS=2 on two DISTINCT GPUs: the bands of GPU 1 reach GPU 0 through pt_combine_bands' RCCL send/recv group.
    Skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
    from pathtracer_amd import hip
    if hip.device_count() < 2:
        pytest.skip("needs two GPUs")
    W, H, spp, frames = 96, 54, 8, (3, 4)
    got = _driver_frames(tmp_path, {"PT_NUM_GPUS": "2", "PT_MATH": "strict"}, frames)
    stream = seedmod.glibc_rand_stream(len(frames) * spp)
    for k, i in enumerate(frames):
        want = oracle.postprocess(oracle.render(scenes.main_lua_frame(i), W, H, 50, stream[k * spp:(k + 1) * spp])).reshape(H, W, 3)
        assert np.array_equal(got[k], want), "frame %d: %d bytes differ" % (i, int((got[k] != want).sum()))


@pytest.mark.gpu
def test_combine_bands_through_the_c_abi(ctx, oracle):
    """pt_combine_bands called directly: three contexts on one GPU render their bands of a frame, the merge lands in the
    root's accumulator; bad arguments are refused."""
    from pathtracer_amd import hip
    sc = scenes.glass_lattice()
    s = seedmod.glibc_rand_stream(5)
    W, H, band = 80, 45, 4
    full = hip.render_scene(ctx, sc, W, H, 32, s)
    ranks = [hip.Context(0) for _ in range(3)]
    try:
        for r, c in enumerate(ranks):
            c.scene_upload(sc["spheres"])
            c.image_alloc(W, H)
            c.render(sc, W, H, 32, s, rank=r, world=3, band_rows=band, async_=True)
        assert hip.combine_bands(ranks, 1, band) == "device-copy"      # one GPU: the packed bands are device copies
        got = ranks[1].image_read()
        assert np.array_equal(got[..., :3].view(np.uint32), full[..., :3].view(np.uint32))
        # the root's own bands were not touched, the peers' came through pack -> one message each -> unpack
        for r, c in enumerate(ranks):
            c.render(sc, W, H, 32, s, rank=r, world=3, band_rows=band, async_=True)
        assert hip.combine_bands(ranks, 0, band) == "device-copy"
        assert np.array_equal(ranks[0].image_read()[..., :3].view(np.uint32), full[..., :3].view(np.uint32))
        with pytest.raises(hip.PtError):
            hip.combine_bands(ranks, 3, band)
        ranks[2].image_alloc(W, H + 1)
        with pytest.raises(hip.PtError):
            hip.combine_bands(ranks, 0, band)
    finally:
        for c in ranks:
            c.close()


@pytest.mark.gpu
def test_multi_process_combine_entry_points_with_one_rank(tmp_path):
    """pt_comm_unique_id / pt_comm_init_rank / pt_combine_bands_rank with a world of one: librccl loads, every symbol
    resolves, a communicator is created and destroyed with the context (in a child process with a time limit: a
    communicator that cannot come up must not take the suite with it)."""
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pathtracer_amd import hip\n"
            "c = hip.Context(0); c.image_alloc(32, 16)\n"
            "c.comm_init_rank(1, 0, hip.comm_unique_id())\n"
            "print('transport', c.combine_bands_rank(0, 8)); c.close()\n") % ROOT
    out = subprocess.check_output([sys.executable, "-c", code], cwd=str(tmp_path), timeout=180)
    assert b"transport none" in out


@pytest.mark.gpu
def test_bench_rccl_path_executes_with_one_rank(tmp_path):
    """bench.py's N > 1 code path (process group, zero-fill, RCCL reduce, barrier, frame check) with a single rank."""
    import sys
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "2", "--warmup", "1",
                                   "--spp", "4", "--no-cpu-baseline", "--no-strict"], cwd=str(tmp_path), timeout=300,
                                  env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617"))
    import json
    line = json.loads(out.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["verified_pixels"] in (0, 1920 * 1080)      # 0: the reference code object is not in the tree
