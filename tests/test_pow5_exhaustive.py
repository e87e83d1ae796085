"""pow(1 - cosTheta, 5) of the Schlick term (/root/reference/src/cl/ray.cl:61) is the one libm call on the shading
path.  The x86 reference build gets it from `(float)pow((double)x, 5.0)` (oracle/ref_shim.cpp, restated by
oracle/pt_oracle.c pow_f); the HIP kernels compute (d*d)*(d*d)*d in f64 and narrow (pt_device.h pow5).  Claim: both
are the CORRECTLY ROUNDED binary32 value of the exact x^5 for EVERY float x, hence equal to each other.

Proof by enumeration against an exact 128-bit integer arbiter (oracle/pt_oracle.c pow5_exact):
 * normal results: the f64 chain, f64 -> f32 narrowing and the arbiter all commute with scaling x by a power of two
   as long as the result stays a normal float, so all 2^23 significands of ONE binade decide every binade; three
   are walked;
 * x^5 denormal, or rounding to zero: x in [2^-31, 2^-25) walked completely (below 2^-31 both give 0: x^5 < 2^-155);
 * overflow: x in [2^24, 2^26) walked completely (x^5 crosses 2^128 inside it; above, both give +inf);
 * zero and denormal x: exponent fields 0 and 1 walked (x^5 rounds to 0).
The device's own pow5 is checked on the same inputs by tests/test_gpu_parity.py::test_pow5_is_correctly_rounded."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import oracle_lib

EXP_FIELDS = [126, 127, 128] + list(range(96, 103)) + [151, 152] + [0, 1]


def test_libm_pow_and_the_f64_chain_are_the_correctly_rounded_fifth_power():
    lib = oracle_lib.load_oracle().lib
    f = lib.pto_pow5_mismatches
    f.restype = C.c_uint64
    f.argtypes = [C.c_int, C.c_int]
    jobs = [(e, w) for e in EXP_FIELDS for w in (0, 1)]
    with ThreadPoolExecutor(8) as ex:
        bad = list(ex.map(lambda j: f(*j), jobs))
    wrong = {j: b for j, b in zip(jobs, bad) if b}
    assert not wrong, "(exponent field, 0 = libm / 1 = chain): mismatches %r" % wrong
