import numpy as np, sys
sys.path.insert(0, '.')
from pathtracer_amd import hip
ctx = hip.Context(0)
n = 1 << 16
a = (np.arange(n, dtype=np.uint64) << np.uint64(16))
b = np.full(n, 1 << 16, np.uint64)
ou, _ = ctx.selftest(6, a=a, b=b)
hi = int((ou & np.uint64(0xFFFFFFFF)).sum()); lo = int((ou >> np.uint64(32)).sum())
print("raw v_sqrt above correctly rounded:", hi, " below:", lo, " of", (0x7f800000 - 0x0f800000))
