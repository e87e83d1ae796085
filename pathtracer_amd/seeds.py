"""Per-sample seed stream of the reference's host loop.

/root/reference/src/c/renderer.c:172 takes `seed = rand()` once per sample, from a
process that never calls srand(): the k-th sample of the k-th render() call uses the
k-th value of glibc's default-seeded rand() stream, which continues across render()
calls.  The C host (pathtracer_amd/csrc/renderer.c) calls libc rand() itself; this
module restates glibc's TYPE_3 additive-feedback generator (seed 1) so that Python
callers (bench.py, tests) get the same stream without touching process-global libc
state.  Checked against libc and against the 16 values listed in SURVEY.md §8a.
"""
import numpy as np


def glibc_rand_stream(n, seed=1):
    """First `n` values of glibc rand() after srand(seed) (default seed 1)."""
    r = [0] * (344 + n)
    r[0] = seed if seed != 0 else 1
    for i in range(1, 31):
        hi, lo = divmod(r[i - 1], 127773)
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        r[i] = w
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + n):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return np.asarray([r[344 + k] >> 1 for k in range(n)], dtype=np.uint64)
