#!/usr/bin/env python3
"""The split running mean's divide (pt_render_kernels.h mean_step: q = x*r, q' = fma(fma(-q, f1, x), r, q) with
r = RN(1/f1)) against the IEEE quotient x / f1 for EVERY pair of significands, on the device's own arithmetic
(pt_selftest op 12).  Both forms are invariant under powers of two in x and in f1 (inside the range mean_safe() admits),
so the 2^23 integer divisors f1 in [2^23, 2^24) x the 2^23 significands of one binade of x are ALL cases: 2^46 quotients,
about two minutes of one MI355X.  Writes gpurun_out/meandiv_walk.json; run once per kernel change of mean_step.

    python3 tools/meandiv_walk.py [first=2^23] [last=2^24] [per_launch=2^17]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import hip  # noqa: E402


def main():
    a = sys.argv[1:]
    first = int(a[0]) if len(a) > 0 else 1 << 23
    last = int(a[1]) if len(a) > 1 else 1 << 24
    per = int(a[2]) if len(a) > 2 else 1 << 17
    ctx = hip.Context(0)
    # the canary first: a reciprocal 256 ulps off must show up
    ou, _ = ctx.selftest(12, a=np.asarray([1000003, 9999991], np.uint64), b=np.asarray([127 | 0x100] * 2, np.uint64))
    assert int(ou[0]) > 1000 and int(ou[1]) > 1000, "canary did not fire"
    bad, done, t0 = 0, 0, time.perf_counter()
    worst = []
    for lo in range(first, last, per):
        hi = min(lo + per, last)
        div = np.arange(lo, hi, dtype=np.uint64)
        ou, _ = ctx.selftest(12, a=div, b=np.full(len(div), 127, np.uint64))
        nb = int(ou.sum())
        if nb:
            worst += [(int(d), int(c)) for d, c in zip(div[ou > 0][:8], ou[ou > 0][:8])]
        bad += nb
        done += len(div)
        print("divisors %d..%d: %d mismatches so far (%.0f s)" % (first, hi - 1, bad, time.perf_counter() - t0), flush=True)
    res = dict(first=first, last=last, divisors=done, significands_per_divisor=1 << 23, quotients=done * (1 << 23),
               mismatches=bad, examples=worst[:16], seconds=time.perf_counter() - t0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "meandiv_walk.json"), "w"), indent=1)
    print(json.dumps(res))
    ctx.close()
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
