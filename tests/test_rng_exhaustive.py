"""rand_float's float for EVERY state: the reference converts `(float)((double)state / 2147483647.0)`
(/root/reference/src/cl/rng.cl:11-14); the HIP kernels multiply by RN(1/2147483647.0) instead of dividing
(pathtracer_amd/csrc/pt_device.h rand_float).  The state after the update is any of 0 .. 2^31 - 2 (0 only from
the stuck state 0); this walks all of 0 .. 2^31 - 1 through the oracle library's own divide and counts the
states whose narrowed float differs.  ~10 s on 8 threads."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import oracle_lib


def test_multiply_by_reciprocal_equals_the_f64_divide_on_every_state():
    lib = oracle_lib.load_oracle().lib
    f = lib.pto_rand_float_mul_mismatches
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_uint64]
    n, parts = 1 << 31, 64
    edges = [n * i // parts for i in range(parts + 1)]
    with ThreadPoolExecutor(8) as ex:      # ctypes releases the GIL during the call
        bad = sum(ex.map(lambda i: f(edges[i], edges[i + 1]), range(parts)))
    assert bad == 0, "%d states where state * RN(1/M) != state / M after narrowing to float" % bad
    # the comparison can fail: a reciprocal that only has binary32 accuracy does differ somewhere
    import numpy as np
    s = np.arange(1, 1 << 22, dtype=np.float64) * 511.0
    wrong = float(np.float32(1.0 / 2147483647.0))
    assert ((s / 2147483647.0).astype(np.float32) != (s * wrong).astype(np.float32)).any()
