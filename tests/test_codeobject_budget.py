"""Register budget of the built kernels, read from the code objects inside pathtracer_amd/libpt_hip.so.

The lock-step kernel is bound by VALU issue and its speed moves by several percent with what the register allocator
does ("Branches the compiler must keep": the same instructions at 80 VGPRs / 6 waves per SIMD ran
158 ms, at 63 VGPRs / 8 waves 155 ms; the relaxed build spilled under its cap and lost 12 %).  This test pins the
outcome for the instances the benchmark runs: at most 72 VGPRs (7 waves per SIMD: since the path's mask moved to LDS the
strict instance takes 70 and runs faster than under a 64-register cap, 151.4 against 152.6 ms; the relaxed one 60) and no
scratch, in both builds, and no scratch in any render kernel but the one instance known to spill (the relaxed build's
nine equal spheres).  It needs
objcopy and llvm-readelf (ROCm); it is skipped where they are missing.
"""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pathtracer_amd", "libpt_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_metadata(tmp_path):
    fat = str(tmp_path / "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    meta = {}
    pos = 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, i + 24)
        off = i + 32
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and sz:
                co = str(tmp_path / ("k%d.co" % len(meta)))
                open(co, "wb").write(blob[i + o:i + o + sz])
                notes = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
                for block in notes.split("- .agpr_count")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", block).group(1)
                    meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                                  for k in ("vgpr_count", "private_segment_fixed_size", "sgpr_spill_count")}
        pos = i + len(magic)
    return meta


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF) and shutil.which("objcopy")),
                    reason="needs the built library, objcopy and llvm-readelf")
def test_headline_kernels_fit_seven_waves_without_scratch(tmp_path):
    meta = kernel_metadata(tmp_path)
    render = {k: v for k, v in meta.items() if "pt_render_" in k}
    assert len(render) >= 40, sorted(meta)
    for ns in ("9pt_strict", "7pt_fast"):
        name = [k for k in render if ns in k and "pt_render_lds_kernelILb0ELi3ELi4E" in k]
        assert len(name) == 1, (ns, name)
        m = render[name[0]]
        assert m["vgpr_count"] <= 72 and m["private_segment_fixed_size"] == 0 and m["sgpr_spill_count"] == 0, (name[0], m)
    # (SGPR spills go to VGPR lanes, not to memory: the grid variants of the streaming kernel have a few)
    spilling = sorted(k for k, m in render.items() if m["private_segment_fixed_size"])
    assert all("7pt_fast" in k and "ELi3ELi0E" in k for k in spilling), spilling
