"""The running mean of /root/reference/src/cl/main.cl:29 divides by frame + 1.  The lock-step HIP kernel does it, when
several lanes share a pixel, as q = x * r, q' = fma(fma(-q, f1, x), r, q) with r = RN(1 / f1)
(pathtracer_amd/csrc/pt_render_kernels.h mean_step).  Claim: q' == RN(x / f1) for every float x and every integer
f1 < 2^24 as long as nothing under- or overflows (the kernel's mean_safe() range).  tools/probe/meandiv_probe.c checks
all 2^23 significands of x per divisor (the quotient commutes with scaling x by a power of two); the full walk
f1 = 1 .. 65536 takes 3 minutes on 8 cores (0 mismatches, recorded in DESIGN.md), this test runs a slice of it and a
strided pass up to 2^24.  The device's own arithmetic is checked by tests/test_gpu_parity.py (pt_selftest op 12)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_fma_divide_by_the_frame_count_is_correctly_rounded(tmp_path):
    exe = str(tmp_path / "meandiv_probe")
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-o", exe, os.path.join(ROOT, "tools/probe/meandiv_probe.c"), "-lm"],
                   check=True)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    for args in (("1", "768"), ("769", "16777215", "32771"), ("16776960", "16777215")):
        out = subprocess.run([exe, *args], env=env, capture_output=True, text=True)
        assert out.returncode == 0 and ": 0 mismatches" in out.stdout, out.stdout[-500:]
        # not vacuous: the bare product x * r is wrong for a large share of the operands
        assert int(out.stdout.rsplit("differs", 1)[1].split()[0]) > 1000, out.stdout
