"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pt_hip.h
and include/renderer.h declare (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)))


def test_pt_hip_exports_every_declared_symbol():
    from pathtracer_amd import hip
    lib = hip.load()
    names = declared_functions("pt_hip.h")
    assert set(names) == set(hip.EXPORTS), (names, hip.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_missing_library_is_loud(tmp_path, monkeypatch):
    from pathtracer_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        hip.load()
    except hip.PtError as e:
        assert "not found" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from pathtracer_amd import hip
    lib = hip.load()
    assert lib.pt_device_count(None) == -1
    assert b"NULL" in lib.pt_last_error()
    assert lib.pt_ctx_create(0, None) == -1
    assert lib.pt_render(None, None) == -1
    assert lib.pt_image_alloc(None, 4, 4) == -1
    assert lib.pt_scene_upload(None, None, 0) == -1
    assert lib.pt_ctx_destroy(None) == 0


def test_launch_plan_without_gpu():
    """pt_plan_launch (host arithmetic): the rank's pixel count in closed form equals the band walk, the thread count
    stays below 2^31 up to the largest frame pt_image_alloc admits (4 lanes + a 64-lane tail overflowed the
    kernels' 32-bit thread index from ~3.7e8 pixels), and the usual frames get the documented tiers."""
    import itertools
    from pathtracer_amd import hip, partition
    for H, world, band in itertools.product((1, 2, 7, 8, 9, 54, 1080, 2160), (1, 2, 3, 7, 8), (1, 5, 8, 16, 100)):
        for rank in range(world):
            pl = hip.plan_launch(33, H, 16, rank=rank, world=world, band_rows=band)
            assert pl.local_pixels == partition.local_pixel_count(33, H, rank, world, band), (H, world, band, rank)
    pl = hip.plan_launch(1920, 1080, 256)
    assert (pl.lanes_log2, pl.tail_lanes_log2) == (2, 6) and pl.tier_threads % 256 == 0
    assert pl.total_threads == pl.tier_threads + ((pl.local_pixels - pl.tier_pixels) << 6) and pl.blocks == (pl.total_threads + 255) // 256
    assert 0.96 * pl.local_pixels <= pl.tier_pixels < 0.98 * pl.local_pixels
    pl = hip.plan_launch(1920, 1080, 256, rank=3, world=8)
    assert (pl.lanes_log2, pl.tail_lanes_log2) == (4, 6)
    pl = hip.plan_launch(1920, 1080, 1)
    assert (pl.lanes_log2, pl.tail_lanes_log2) == (0, 0) and pl.total_threads == 1920 * 1080
    for w, h in ((32768, 32768), (30000, 20000), (32768, 16384), (20000, 19000)):
        pl = hip.plan_launch(w, h, 256)
        assert pl.local_pixels == w * h and pl.total_threads < 2 ** 31 and pl.tier_threads <= pl.total_threads
        assert pl.total_threads == (pl.tier_pixels << pl.lanes_log2) + ((pl.local_pixels - pl.tier_pixels) << pl.tail_lanes_log2)
    assert hip.plan_launch(32768, 32768, 256).lanes_log2 == 0                 # 2^30 pixels: one lane each is all that fits
    assert hip.plan_launch(20000, 19000, 256).tail_lanes_log2 < 6            # 3.8e8 pixels: the tail tier gives way first
    lib = hip.load()
    a = hip.RenderArgs()
    a.width, a.height, a.world, a.band_rows = 40000, 40000, 1, 8
    assert lib.pt_plan_launch(a, hip.LaunchPlan()) == -1                      # beyond pt_image_alloc's 2^30 pixels
