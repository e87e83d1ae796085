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
