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


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1])
def test_scene_driver_frames_match_oracle(tmp_path, oracle, pipeline):
    """Animation frames through create_renderer/…/render_to_file in a fresh process: the PNGs must hold
    exactly the oracle's 8-bit image; every frame continues the rand() seed stream.  pipeline=1: the
    frames are queued to the worker thread (PT_PIPELINE) — same files."""
    W, H, spp = 96, 54, 8
    frames = (3, 4, 5, 6, 7) if pipeline else (3, 4)
    env = dict(os.environ, PT_NUM_GPUS="1", PT_PIPELINE=str(pipeline))
    pat = str(tmp_path / "frame%d.png")
    subprocess.check_call([os.path.join(PKG, "pt_scene_driver"), "3", str(len(frames)), str(W), str(H), str(spp), "50", pat],
                          env=env, cwd=str(tmp_path), timeout=100)
    stream = seedmod.glibc_rand_stream(len(frames) * spp)
    for k, i in enumerate(frames):
        sc = scenes.main_lua_frame(i)
        want = oracle.postprocess(oracle.render(sc, W, H, 50, stream[k * spp:(k + 1) * spp])).reshape(H, W, 3)
        got = read_png_rgb8(pat % i)
        assert np.array_equal(got, want), "frame %d: %d bytes differ" % (i, int((got != want).sum()))
        assert got.max() > 200
