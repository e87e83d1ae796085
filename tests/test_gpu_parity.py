"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and
the committed reference outputs.  Bar: the linear float accumulator is BIT-EXACT (the kernels keep
the reference's operation order under strict IEEE-754 rules, see pathtracer_amd/csrc/pt_device.h);
the integer RNG stream is bit-exact by construction."""
import os

import numpy as np
import pytest

import envelope
from cases import CASES
from pathtracer_amd import hip, scenes
from pathtracer_amd import seeds as seedmod

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def mismatch_report(a, b):
    ne = (bits(a) != bits(b)).any(axis=-1)
    return "%d of %d pixels differ, max abs %.3g" % (int(ne.sum()), ne.size, float(np.abs(a - b).max()))


def test_rng_bit_exact(ctx, oracle):
    k = np.load(os.path.join(GOLD, "kat.npz"))
    ou, _ = ctx.selftest(0, a=k["init2_a"], b=k["init2_b"])
    assert np.array_equal(ou, k["init2_out"])
    rs = np.random.RandomState(11)
    a = np.concatenate([rs.randint(0, 2**31 - 1, size=20000), np.arange(0, 4096)]).astype(np.uint64)
    b = np.concatenate([rs.randint(0, 2**31 - 1, size=20000), seedmod.glibc_rand_stream(4096)]).astype(np.uint64)
    ou, _ = ctx.selftest(0, a=a, b=b)
    want = np.asarray([oracle.init_rng_2(int(x), int(y)) for x, y in zip(a, b)], np.uint64)
    assert np.array_equal(ou, want)
    # rand_float: states and float bits
    st = np.concatenate([k["rand_state_in"], rs.randint(0, 2**31 - 1, size=50000).astype(np.uint64),
                         np.asarray([2147483646 - 64, 2147483646 - 192, 2147483646], np.uint64)])
    ou, of = ctx.selftest(1, a=st)
    want = [oracle.rand_float(int(x)) for x in st]
    assert np.array_equal(ou, np.asarray([w[0] for w in want], np.uint64))
    assert np.array_equal(bits(of[:len(st)]), bits(np.asarray([w[1] for w in want])))
    ou, of = ctx.selftest(2, a=st[:5000])
    want = [oracle.random_unit_vector(int(x)) for x in st[:5000]]
    assert np.array_equal(ou, np.asarray([w[0] for w in want], np.uint64))
    assert np.array_equal(bits(of[:15000].reshape(-1, 3)), bits(np.stack([w[1] for w in want])))


def test_divide_sqrt_correctly_rounded(ctx, oracle):
    rs = np.random.RandomState(5)
    n = 200000
    x = np.abs(rs.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rs.randint(-30, 30, size=n).astype(np.float32)
    y = (rs.standard_normal(n).astype(np.float32)) * np.float32(10.0) ** rs.randint(-8, 8, size=n).astype(np.float32)
    x[:8] = [0.0, 1e-45, 1e-40, 1.0, 4.0, 2.0, 3.4e38, 1e-38]
    inp = np.stack([x, y], axis=1).reshape(-1)
    _, of = ctx.selftest(3, in_f=inp, n=n)
    of = of.reshape(n, 4)
    with np.errstate(all="ignore"):
        assert np.array_equal(bits(of[:, 0]), bits(x / y))
        assert np.array_equal(bits(of[:, 1]), bits(np.sqrt(x)))
    # reflectance / pow5 on the range the kernel feeds them
    c = rs.uniform(-1.5, 1.0, size=n).astype(np.float32)
    r = rs.uniform(0.3, 3.0, size=n).astype(np.float32)
    _, of = ctx.selftest(3, in_f=np.stack([c, r], axis=1).reshape(-1), n=n)
    of = of.reshape(n, 4)
    want = np.asarray([oracle.reflectance(float(a), float(b)) for a, b in zip(c[:20000], r[:20000])], np.float32)
    # pow(x,5): libm on the oracle's side, an f64 product chain on the device — both correctly rounded for every
    # float (tests/test_pow5_exhaustive.py, test_pow5_is_correctly_rounded below): no tolerance
    assert np.array_equal(bits(of[:20000, 2]), bits(want))


def test_short_sqrt_is_sqrtf_on_every_float(ctx):
    """The kernels' five-instruction correctly-rounded sqrt (pt_device.h: v_rsq_f32 + one Markstein step) and its guarded
    forms against the compiler's correctly rounded sqrtf on ALL 2^32 bit patterns, on the device (sqrtf itself is pinned
    against the CPU in test_divide_sqrt_correctly_rounded): sqrt_cr everywhere, sqrt_z on +-0 / |x| >= 2^-96 / inf / NaN,
    sqrt_pos on 2^-96 <= x < inf and the negatives, sqrt_01 on [2^-96, 1] and +-0."""
    n = 1 << 16
    a = (np.arange(n, dtype=np.uint64) << np.uint64(16))
    b = np.full(n, 1 << 16, np.uint64)
    ou, _ = ctx.selftest(4, a=a, b=b)
    m = np.uint64((1 << 21) - 1)
    assert int((ou & m).sum()) == 0, "sqrt_cr != sqrtf somewhere"
    assert int(((ou >> np.uint64(21)) & m).sum()) == 0, "sqrt_z != sqrtf inside its domain"
    assert int(((ou >> np.uint64(42)) & m).sum()) == 0, "sqrt_pos / sqrt_01 != sqrtf inside their domains"


def test_normalize_reciprocal_length_on_every_float(ctx):
    """normalize()'s 1 / length(v) from x = dot(v, v) (pt_device.h inv_len: ONE v_rsq_f32 serves the root and seeds the
    reciprocal of the rounded root) == 1.0f / sqrtf(x) on all 2^32 patterns of x (pt_selftest op 11)."""
    n = 1 << 16
    a = (np.arange(n, dtype=np.uint64) << np.uint64(16))
    b = np.full(n, 1 << 16, np.uint64)
    ou, _ = ctx.selftest(11, a=a, b=b)
    bad = np.nonzero(ou)[0]
    assert len(bad) == 0, "inv_len(x) != 1/sqrtf(x) in %d chunks, first at bits 0x%08x" % (len(bad), int(bad[0]) << 16)


def test_running_mean_divide_on_every_significand(ctx):
    """The split running mean divides by frame + 1 as q = x * r, q' = fma(fma(-q, f1, x), r, q), r = RN(1 / f1)
    (pt_render_kernels.h mean_step).  On the device, against x / f1: every significand of x for f1 = 1 .. 4096 in one
    binade, and at both ends of the range mean_safe() allows x (2^-64, 2^65) for divisors of every shape up to 2^24 - 1.
    tools/probe/meandiv_probe.c walks f1 = 1 .. 65536 the same way on the CPU (tests/test_meandiv_exhaustive.py runs a
    slice)."""
    div = list(range(1, 4097))
    exp = [127] * len(div)
    shapes = sorted({(1 << k) + d for k in range(1, 24) for d in (-1, 0, 1)} | {(1 << 24) - 1, 4097, 65535, 65537,
                    1000003, 9999991, 16777213, 12345679, 11184811, 5592405, 3 * 5 * 7 * 11 * 13 * 17})
    for e in (63, 191):
        div += shapes
        exp += [e] * len(shapes)
    assert max(div) < (1 << 24)
    div += [1000003, 9999991]                    # canary: a reciprocal 256 ulps off must be noticed
    exp += [127 | 0x100] * 2
    ou, _ = ctx.selftest(12, a=np.asarray(div, np.uint64), b=np.asarray(exp, np.uint64))
    assert ou[-1] > 1000 and ou[-2] > 1000, ou[-2:]
    bad = np.nonzero(ou[:-2])[0]
    assert len(bad) == 0, "mean_step != x / f1 for %d divisors, first f1 = %d (%d significands)" % (
        len(bad), div[bad[0]], int(ou[bad[0]]))


def test_short_reciprocal_is_the_divide_on_every_float(ctx):
    """pt_device.h rcp_cr (v_rcp_f32 + two FMA steps, full divide outside the normal range) == 1.0f / x on all 2^32 patterns."""
    n = 1 << 16
    a = (np.arange(n, dtype=np.uint64) << np.uint64(16))
    b = np.full(n, 1 << 16, np.uint64)
    ou, _ = ctx.selftest(5, a=a, b=b)
    bad = np.nonzero(ou)[0]
    assert len(bad) == 0, "rcp_cr != 1/x in %d chunks, first at bits 0x%08x" % (len(bad), int(bad[0]) << 16)


def test_short_divide_is_the_divide_on_its_range(ctx):
    """finish_scan's divide (pt_device.h div_nt: the compiler's expansion without range scaling and special-case
    fix-up) against the compiler's correctly rounded divide on 2^32 operand pairs of the range the scan guarantees
    (2^-80 <= num <= 2^80, 2^-40 <= den <= 2^40: pseudo-random, plus the 16 corners per work item), on the device
    (pt_selftest op 7).  Not vacuous: the op reports how many pairs it compared, and the same comparison FAILS for
    the bare num * v_rcp_f32(den) (op 8)."""
    n, per = 1 << 16, 1 << 16
    a, b = np.arange(n, dtype=np.uint64), np.full(n, per, np.uint64)
    ou, _ = ctx.selftest(7, a=a, b=b)
    assert int((ou >> np.uint64(32)).sum()) == n * per, "op 7 compared %d pairs, expected %d" % (int((ou >> np.uint64(32)).sum()), n * per)
    assert int((ou & np.uint64(0xFFFFFFFF)).sum()) == 0, "div_nt != / on %d pairs" % int((ou & np.uint64(0xFFFFFFFF)).sum())
    # the lock-step kernel's form of the same divide: den = a = RN(RN(sqrt(x))^2), reciprocal seeded with v_rsq_f32(x)^2 (op 10)
    ou, _ = ctx.selftest(10, a=a, b=b)
    assert int((ou >> np.uint64(32)).sum()) == n * per
    assert int((ou & np.uint64(0xFFFFFFFF)).sum()) == 0, "div_by_a != / on %d pairs" % int((ou & np.uint64(0xFFFFFFFF)).sum())
    ou, _ = ctx.selftest(8, a=a[:256], b=b[:256])
    assert int((ou >> np.uint64(32)).sum()) == 256 * per
    bad = int((ou & np.uint64(0xFFFFFFFF)).sum())
    assert bad > 256 * per // 100, "canary: num * rcp(den) should differ from the divide on a good share of the pairs, got %d" % bad


def test_pow5_is_correctly_rounded(ctx, oracle):
    """pt_device.h pow5 on the device against the exact integer arbiter (oracle/pt_oracle.c pow5_exact) on every input
    class of tests/test_pow5_exhaustive.py: all 2^23 significands of three normal binades, every float whose x^5 is a
    binary32 denormal or rounds to zero, the overflow edge, zero and denormal x.  Zero tolerance."""
    import ctypes as C
    from test_pow5_exhaustive import EXP_FIELDS
    f = oracle.lib.pto_pow5_exact
    f.argtypes = [C.POINTER(C.c_uint32), C.c_long, C.POINTER(C.c_float)]
    mant = np.arange(1 << 23, dtype=np.uint32)
    for e in EXP_FIELDS:
        xb = (np.uint32(e) << np.uint32(23)) | mant
        _, of = ctx.selftest(9, a=xb.astype(np.uint64))
        want = np.empty(len(xb), np.float32)
        f(xb.ctypes.data_as(C.POINTER(C.c_uint32)), len(xb), want.ctypes.data_as(C.POINTER(C.c_float)))
        ne = bits(of[:len(xb)]) != bits(want)
        assert not ne.any(), "exponent field %d: %d mismatches, first x bits 0x%08x" % (e, int(ne.sum()), int(xb[np.argmax(ne)]))
    # negative x and the specials
    xb = np.asarray([0x80000000, 0xbf800000, 0xc0490fdb, 0x7f800000, 0xff800000, 0x7fc00000, 0x00000001, 0x807fffff], np.uint32)
    _, of = ctx.selftest(9, a=xb.astype(np.uint64))
    want = np.empty(len(xb), np.float32)
    f(xb.ctypes.data_as(C.POINTER(C.c_uint32)), len(xb), want.ctypes.data_as(C.POINTER(C.c_float)))
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(of[:len(xb)]), nan) and np.array_equal(bits(of[:len(xb)])[~nan], bits(want)[~nan])


def test_kernel_timing_ring(ctx):
    sc = scenes.main_lua_frame(0)
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(64, 36)
    s = seedmod.glibc_rand_stream(2)
    for _ in range(3):
        ctx.render(sc, 64, 36, 50, s, async_=True, record_timing=True)
    t = ctx.kernel_times()
    assert len(t) == 3 and all(0 < x < 1000 for x in t)
    assert ctx.kernel_times() == []


@pytest.mark.parametrize("name", sorted(CASES))
def test_render_matches_oracle_and_reference_fixture(ctx, oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    img, bounces = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
    want, wb = oracle.render(sc, W, H, depth, g["seeds"], count_bounces=True)
    assert not np.isnan(img).any()
    assert np.array_equal(bits(img[..., :3]), bits(want[..., :3])), "HIP != oracle: " + mismatch_report(img[..., :3], want[..., :3])
    assert np.array_equal(bits(img[..., :3]), bits(g["strict"])), "HIP != reference (strict build) fixture"
    assert bounces == wb
    # against the reference built with its own fast-math flags: the tolerance two builds of the reference meet
    d = np.abs(img[..., :3] - g["fast"]).max(axis=2)
    if name.startswith("dense"):
        assert envelope.off_pixels(img[..., :3], g["fast"]) <= envelope.off_pixel_budget(bounces, sc["spheres"].shape[0])      # flip budget, tests/envelope.py
    else:
        assert d.max() <= 1e-5
    # the non-counting kernel variant gives the same image
    img2 = hip.render_scene(ctx, sc, W, H, depth, g["seeds"])
    assert np.array_equal(bits(img2[..., :3]), bits(img[..., :3]))
    # counting launches leave each pixel's own bounce count in the (otherwise unused) 4th lane
    assert int(img[..., 3].astype(np.float64).sum()) == bounces and not img2[..., 3].any()


def test_frames_compose_and_partition(ctx, oracle):
    g = np.load(os.path.join(GOLD, "frame77_96x54.npz"))
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    s = g["seeds"]
    full = hip.render_scene(ctx, sc, 96, 54, 50, s)
    # chunked accumulation: 3 + 5 samples continue the same running mean
    ctx.image_alloc(96, 54)
    ctx.render(sc, 96, 54, 50, s[:3])
    ctx.render(sc, 96, 54, 50, s[3:], first_frame=3)
    assert np.array_equal(bits(ctx.image_read()), bits(full))
    # row-band partition: ranks write disjoint pixels of one zeroed buffer -> identical image
    for world, band in ((2, 16), (3, 5), (8, 4), (4, 54), (7, 100)):
        ctx.image_alloc(96, 54)
        for r in range(world):
            ctx.render(sc, 96, 54, 50, s, rank=r, world=world, band_rows=band)
        assert np.array_equal(bits(ctx.image_read()), bits(full)), (world, band)
    # one rank alone leaves the other ranks' rows zero
    ctx.image_alloc(96, 54)
    ctx.render(sc, 96, 54, 50, s, rank=1, world=2, band_rows=16)
    part = ctx.image_read()
    rows = (np.arange(54) // 16) % 2 == 1
    assert np.array_equal(bits(part[rows]), bits(full[rows]))
    assert not part[~rows].any()
    # a rank that owns no band at all (one band, two ranks): nothing is written, and the outputs still get their values
    ctx.image_alloc(96, 54)
    assert ctx.render(sc, 96, 54, 50, s, rank=1, world=2, band_rows=100, count_bounces=True) == 0
    assert ctx.render(sc, 96, 54, 50, s, rank=1, world=2, band_rows=100, time_kernel=True) == 0.0
    assert ctx.render(sc, 96, 54, 50, s, rank=1, world=2, band_rows=100, noalloc=True, count_bounces=True) == 0
    assert not ctx.image_read().any()


def test_sample_interleaved_lanes_are_bit_identical(ctx, oracle):
    """2 ... 64 lanes per pixel (samples k = sub mod S, colours accumulated in order after each round) give
    the bits of 1 lane per pixel: odd sample counts, fewer samples than lanes, continued accumulators."""
    g = np.load(os.path.join(GOLD, "lattice_96x54.npz"))
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    stream = seedmod.glibc_rand_stream(40)
    try:
        os.environ["PT_LANES_PER_PIXEL"] = "1"
        ref = {n: hip.render_scene(ctx, sc, 96, 54, 32, stream[:n], count_bounces=True) for n in (1, 3, 8, 13, 40)}
        want = oracle.render(sc, 96, 54, 32, stream[:13])
        assert np.array_equal(bits(ref[13][0][..., :3]), bits(want[..., :3]))
        for lanes in (2, 4, 8, 16, 32, 64):
            os.environ["PT_LANES_PER_PIXEL"] = str(lanes)
            for n in (1, 3, 8, 13, 40):
                img, bc = hip.render_scene(ctx, sc, 96, 54, 32, stream[:n], count_bounces=True)
                assert np.array_equal(bits(img), bits(ref[n][0])), (lanes, n)      # incl. the per-pixel counts
                assert bc == ref[n][1]
            ctx.image_alloc(96, 54)
            ctx.render(sc, 96, 54, 32, stream[:5])
            ctx.render(sc, 96, 54, 32, stream[5:13], first_frame=5)
            assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(ref[13][0][..., :3])), lanes
    finally:
        os.environ.pop("PT_LANES_PER_PIXEL", None)


def test_split_running_mean_and_its_fallback(ctx, oracle):
    """With 4 or more lanes per pixel the ordered running mean (main.cl:29) is split by channel and divides by
    frame + 1 as two FMAs around RN(1 / (frame + 1)) (pt_render_kernels.h mean_step; exhaustively checked on the CPU by
    tools/probe/meandiv_probe.c), for colours that are 0 or in [2^-40, 2^40]; a wave that meets anything else goes back to
    the plain divide for the rest of the launch.  Dark walls (albedo 0.02: a path that finds the light after k bounces
    carries 10 * 0.02^k, down to denormals and 0) make waves switch at different rounds; extreme exposures make every wave
    switch at once; a continued accumulator checks the values loaded from the image."""
    base = scenes.main_lua_frame(0)
    dark = {k: v.copy() for k, v in base.items()}
    dark["spheres"][:6, 5:8] *= np.float32(0.02)
    cases = [("dark", dark)]
    for name, expo in (("tiny", 1e-20), ("huge", 1e25), ("denormal", 1e-36)):
        sc = {k: v.copy() for k, v in base.items()}
        sc["cam"][9] = np.float32(expo)
        cases.append((name, sc))
    s = seedmod.glibc_rand_stream(48)
    try:
        for name, sc in cases:
            want = oracle.render(sc, 96, 54, 50, s)
            for lanes in (4, 16, 64):
                os.environ["PT_LANES_PER_PIXEL"] = str(lanes)
                img = hip.render_scene(ctx, sc, 96, 54, 50, s)
                assert np.array_equal(bits(img[..., :3]), bits(want[..., :3])), (name, lanes)
                ctx.image_alloc(96, 54)
                ctx.render(sc, 96, 54, 50, s[:19])
                ctx.render(sc, 96, 54, 50, s[19:], first_frame=19)
                assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(want[..., :3])), (name, lanes, "continued")
        v = want[..., :3]
        assert np.isfinite(v).all()
    finally:
        os.environ.pop("PT_LANES_PER_PIXEL", None)


def test_two_tier_launch_is_bit_identical(ctx, oracle):
    """pt_render gives the pixels whose blocks are dispatched last more lanes each (shorter blocks at the end of the
    launch; PT_TAIL_PCT sets their share, 0 = one tier).  Any share must give the bits of the one-tier launch, also for
    a rank's bands and for a continued accumulator."""
    sc = scenes.main_lua_frame(0)
    W, H = 1920, 1080
    s = seedmod.glibc_rand_stream(24)
    try:
        os.environ["PT_TAIL_PCT"] = "0"
        ref, bc = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
        want = oracle.render(sc, W, H, 50, s, id_begin=1079 * W, id_end=1080 * W)
        assert np.array_equal(bits(ref[1079, :, :3]), bits(want[1079, :, :3]))
        for pct in (8, 37, 100):
            os.environ["PT_TAIL_PCT"] = str(pct)
            img, b2 = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
            assert np.array_equal(bits(img), bits(ref)) and b2 == bc, pct
            ctx.image_alloc(W, H)
            for r in range(3):
                ctx.render(sc, W, H, 50, s[:16], rank=r, world=3, band_rows=7)
            for r in range(3):
                ctx.render(sc, W, H, 50, s[16:], first_frame=16, rank=r, world=3, band_rows=7)
            assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(ref[..., :3])), pct
        # a small launch: the first tier already has 16 lanes per pixel, the second a whole wave
        s = seedmod.glibc_rand_stream(64)
        os.environ["PT_TAIL_PCT"] = "0"
        ref, bc = hip.render_scene(ctx, sc, 640, 360, 50, s, count_bounces=True)
        want = oracle.render(sc, 640, 360, 50, s, id_begin=358 * 640, id_end=360 * 640)
        assert np.array_equal(bits(ref[358:, :, :3]), bits(want[358:, :, :3]))
        for pct in (8, 50):
            os.environ["PT_TAIL_PCT"] = str(pct)
            img, b2 = hip.render_scene(ctx, sc, 640, 360, 50, s, count_bounces=True)
            assert np.array_equal(bits(img), bits(ref)) and b2 == bc, pct
    finally:
        os.environ.pop("PT_TAIL_PCT", None)


def test_u8_epilogue(ctx, oracle):
    g = np.load(os.path.join(GOLD, "frame0_64x36.npz"))
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    img = hip.render_scene(ctx, sc, 64, 36, 50, g["seeds"])
    u8 = ctx.image_read_u8()
    assert np.array_equal(u8.reshape(-1, 3), oracle.postprocess(img))
    assert u8.max() == 255 and u8.min() == 0


def test_u8_epilogue_on_every_float_that_matters(ctx, oracle):
    """pt_to_u8_kernel (gamma 2 as sqrtf, clamp, (int)(x * 255) truncation, include/pt_hip.h pt_image_read_u8) against
    the oracle's restatement of renderer.c:50-56 (the DOUBLE sqrt narrowed to float) on EVERY float in [0, 1] and a bit
    above (bit patterns 0 .. 0x3f800100: where the 255 thresholds lie), and on every 4099th pattern of the rest
    (larger values, negatives, infinities, NaNs: the clamp and the conversion's special cases).  PT_U8_ALL=1 walks all 2^32."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    P = 1 << 22                                         # pixels per launch: 3 patterns each
    # the caller-owned device image comes straight from the HIP runtime the library itself uses (torch would bring a
    # second copy of the runtime into this process when it is imported after the library)
    rt = C.CDLL("libamdhip64.so.7")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    dev = C.c_void_p()
    assert rt.hipMalloc(C.byref(dev), 16 * P) == 0

    def check(pats):
        n = len(pats) // 3
        img = np.zeros((n, 4), np.float32)
        img[:, :3] = pats[:3 * n].view(np.float32).reshape(n, 3)
        assert rt.hipMemcpy(dev, img.ctypes.data_as(C.c_void_p), img.nbytes, 1) == 0       # hipMemcpyHostToDevice
        u8 = ctx.image_read_u8(d_image=dev, width=n, height=1).reshape(n, 3)
        parts = np.array_split(np.arange(n), 16)
        with ThreadPoolExecutor(16) as ex:
            want = list(ex.map(lambda ix: oracle.postprocess(img[ix[0]:ix[-1] + 1]), [ix for ix in parts if len(ix)]))
        want = np.concatenate(want)
        bad = np.nonzero((u8 != want).any(axis=1))[0]
        assert len(bad) == 0, "pixel values %r: HIP %r, oracle %r" % (img[bad[0], :3], u8[bad[0]], want[bad[0]])

    everything = os.environ.get("PT_U8_ALL") == "1"
    hi = (1 << 32) if everything else 0x3f800100
    for start in range(0, hi, 3 * P):
        check(np.arange(start, min(start + 3 * P, hi), dtype=np.uint64).astype(np.uint32))
    if not everything:
        rest = np.arange(0x3f800100, 1 << 32, 4099, dtype=np.uint64).astype(np.uint32)
        check(rest[:len(rest) // 3 * 3])
    assert rt.hipFree(dev) == 0


def test_error_paths(ctx):
    sc = scenes.main_lua_frame(0)
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(8, 8)
    with pytest.raises(hip.PtError):
        ctx.render(sc, 16, 16, 50, seedmod.glibc_rand_stream(1))       # accumulator size mismatch
    with pytest.raises(hip.PtError):
        ctx.render(sc, 8, 8, 50, seedmod.glibc_rand_stream(1), rank=2, world=2)
    with pytest.raises(hip.PtError):
        ctx.render(sc, 8, 8, -1, seedmod.glibc_rand_stream(1))
    assert ctx.render(sc, 8, 8, 50, np.zeros(0, np.uint64)) is None     # zero samples: no-op


def test_full_size_properties(ctx, oracle):
    """BASELINE config 2 size (1920x1080, frame-0 scene) at reduced spp: size-independent checks —
    sampled rows against the oracle bit for bit, the bounce count, partition invariance."""
    sc = scenes.main_lua_frame(0)
    W, H, spp = 1920, 1080, 4
    s = seedmod.glibc_rand_stream(spp)
    img, bounces = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
    assert not np.isnan(img).any() and img.min() >= 0
    rows = [0, 1, 539, 540, 777, 1079]
    for r in rows:
        want = oracle.render(sc, W, H, 50, s, id_begin=r * W, id_end=(r + 1) * W)
        assert np.array_equal(bits(img[r, :, :3]), bits(want[r, :, :3])), "row %d" % r
    assert 40.0 < bounces / (W * H * spp) < 50.0
    ctx.image_alloc(W, H)
    for r in range(8):
        ctx.render(sc, W, H, 50, s, rank=r, world=8, band_rows=16)
    assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(img[..., :3]))
    u8 = ctx.image_read_u8()
    assert np.array_equal(u8[540], oracle.postprocess(img[540]))


def test_headline_frame_at_its_full_sample_count(ctx, oracle):
    """BASELINE config 2 exactly as bench.py times it: 1920x1080, 256 spp, depth 50, the first 256 glibc seeds —
    64 rounds of the four-lanes-per-pixel group accumulation.  Sampled rows against the oracle at the full 256 spp,
    bit for bit; the 8-bit image of those rows; the bounce count the roofline figure is computed from."""
    sc = scenes.main_lua_frame(0)
    W, H, spp = 1920, 1080, 256
    s = seedmod.glibc_rand_stream(spp)
    img, bounces = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
    u8 = ctx.image_read_u8()
    for r in (0, 333, 540, 1079):
        want = oracle.render(sc, W, H, 50, s, id_begin=r * W, id_end=(r + 1) * W, nthreads=16)
        assert np.array_equal(bits(img[r, :, :3]), bits(want[r, :, :3])), "row %d: %s" % (r, mismatch_report(img[r, :, :3], want[r, :, :3]))
        assert np.array_equal(u8[r], oracle.postprocess(want[r]))
    assert abs(bounces / (W * H * spp) - 47.47) < 0.05
    # the non-counting kernel (the one that is timed) gives the same frame
    img2 = hip.render_scene(ctx, sc, W, H, 50, s)
    assert np.array_equal(bits(img2[..., :3]), bits(img[..., :3]))


FULL = os.environ.get("PT_FULL_FRAME", "")


@pytest.mark.skipif(FULL not in ("1", "headline", "lattice", "dense10k"),
                    reason="5-10 minutes of CPU: set PT_FULL_FRAME=headline|lattice|dense10k (one-off runs recorded in DESIGN.md section 2)")
def test_headline_frame_every_pixel(ctx, oracle):
    """The WHOLE frame of a BASELINE config at its full sample count — headline: frame-0 scene, 1920x1080, 256 spp, depth 50
    (5.3e8 samples); lattice: config 5, 64 spp, depth 32; dense10k: config 3, 10^4 spheres, 64 spp — against the x86 build
    of the reference kernel itself (oracle/_ref strict build; the oracle restatement where that library is absent), bit for bit."""
    import time
    import oracle_lib
    chk = oracle_lib.load_ref("strict") if oracle_lib.have_ref("strict") else oracle
    sc, spp, depth = {"lattice": (scenes.glass_lattice(), 64, 32), "dense10k": (scenes.dense_random(10000), 64, 50)}.get(
        FULL, (scenes.main_lua_frame(0), 256, 50))
    W, H = 1920, 1080
    s = seedmod.glibc_rand_stream(spp)
    img = hip.render_scene(ctx, sc, W, H, depth, s)
    t0 = time.time()
    step = 30
    for r0 in range(int(os.environ.get("PT_FULL_FRAME_ROW0", "0")), H, step):      # (a long check can be split over calls)
        r1 = min(H, r0 + step)
        want = chk.render(sc, W, H, depth, s, id_begin=r0 * W, id_end=r1 * W, nthreads=16)
        assert np.array_equal(bits(img[r0:r1, :, :3]), bits(want[r0:r1, :, :3])), "rows %d..%d: %s" % (
            r0, r1, mismatch_report(img[r0:r1, :, :3].reshape(-1, 3), want[r0:r1, :, :3].reshape(-1, 3)))
        print("rows %d..%d of %d identical to %s (%.0f s)" % (r0, r1, H, os.path.basename(chk.path), time.time() - t0), flush=True)


def test_config3_hundred_thousand_spheres_at_1080p(ctx, oracle):
    """BASELINE config 3 at its largest table AND its full image size: 10^5 spheres, 1920x1080, 1 spp.  The oracle needs
    ~5 ms per path here, so segments of pixels are compared (bit for bit), spread over the image."""
    sc = scenes.dense_random(100000)
    W, H = 1920, 1080
    s = seedmod.glibc_rand_stream(1)
    img, bc = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
    assert not np.isnan(img).any() and 3.0 < bc / (W * H) < 8.0
    flat = img.reshape(-1, 4)
    for start in (0, 517 * W + 900, 540 * W + 960, 1079 * W + 1792):
        want = oracle.render(sc, W, H, 50, s, id_begin=start, id_end=start + 128, nthreads=16).reshape(-1, 4)
        assert np.array_equal(bits(flat[start:start + 128, :3]), bits(want[start:start + 128, :3])), "pixels %d.." % start


def test_streaming_kernel_composes_and_partitions(ctx, oracle):
    """The persistent streaming kernel (tables above 16 spheres): continued accumulators, row-band partitions and
    the bounce count behave as on the lock-step kernel."""
    g = np.load(os.path.join(GOLD, "dense500_96x54.npz"))
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    s = seedmod.glibc_rand_stream(7)
    full, bc = hip.render_scene(ctx, sc, 96, 54, 50, s, count_bounces=True)
    want, wb = oracle.render(sc, 96, 54, 50, s, count_bounces=True)
    assert np.array_equal(bits(full[..., :3]), bits(want[..., :3])) and bc == wb
    ctx.image_alloc(96, 54)
    ctx.render(sc, 96, 54, 50, s[:2])
    ctx.render(sc, 96, 54, 50, s[2:3], first_frame=2)
    ctx.render(sc, 96, 54, 50, s[3:], first_frame=3)
    assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(full[..., :3]))
    for world, band in ((2, 8), (5, 3), (8, 100)):
        ctx.image_alloc(96, 54)
        for r in range(world):
            ctx.render(sc, 96, 54, 50, s, rank=r, world=world, band_rows=band)
        assert np.array_equal(bits(ctx.image_read()[..., :3]), bits(full[..., :3])), (world, band)


@pytest.mark.parametrize("n,W,H,spp", [(10000, 64, 36, 2), (100000, 32, 18, 1)])
def test_config3_dense_tables_match_oracle(ctx, oracle, n, W, H, spp):
    """BASELINE config 3 table sizes (10^4 and 10^5 random spheres) at an image the oracle finishes in seconds."""
    sc = scenes.dense_random(n)
    s = seedmod.glibc_rand_stream(spp)
    img, bc = hip.render_scene(ctx, sc, W, H, 50, s, count_bounces=True)
    want, wb = oracle.render(sc, W, H, 50, s, count_bounces=True)
    assert np.array_equal(bits(img[..., :3]), bits(want[..., :3])), mismatch_report(img[..., :3], want[..., :3])
    assert bc == wb and img[..., :3].max() > 0


def test_config3_full_size_rows(ctx, oracle):
    """1920x1080 over 10^4 spheres at 1 spp: sampled rows against the oracle, partition invariance."""
    sc = scenes.dense_random(10000)
    W, H = 1920, 1080
    s = seedmod.glibc_rand_stream(1)
    img = hip.render_scene(ctx, sc, W, H, 50, s)
    for r in (0, 541, 1079):
        want = oracle.render(sc, W, H, 50, s, id_begin=r * W, id_end=r * W + 480)
        assert np.array_equal(bits(img[r, :480, :3]), bits(want[r, :480, :3])), "row %d" % r
    ctx.image_alloc(W, H)
    for r in range(4):
        ctx.render(sc, W, H, 50, s, rank=r, world=4, band_rows=8)
    assert np.array_equal(bits(ctx.image_read()), bits(img))


def test_config4_4k_rows_and_partition(ctx, oracle):
    """BASELINE config 4 size (3840x2160, frame-0 scene) at 2 spp: sampled rows against the oracle bit for bit,
    and the 8-rank band partition reassembles the single-GPU image."""
    sc = scenes.main_lua_frame(0)
    W, H = 3840, 2160
    s = seedmod.glibc_rand_stream(2)
    img = hip.render_scene(ctx, sc, W, H, 50, s)
    assert not np.isnan(img).any()
    for r in (0, 1080, 2159):
        want = oracle.render(sc, W, H, 50, s, id_begin=r * W, id_end=(r + 1) * W)
        assert np.array_equal(bits(img[r, :, :3]), bits(want[r, :, :3])), "row %d" % r
    ctx.image_alloc(W, H)
    for r in range(8):
        ctx.render(sc, W, H, 50, s, rank=r, world=8, band_rows=8)
    assert np.array_equal(bits(ctx.image_read()), bits(img))


def test_config5_lattice_full_size_rows(ctx, oracle):
    """BASELINE config 5 (glass lattice + emitters, depth 32) at 1920x1080, 2 spp: sampled rows against the oracle."""
    sc = scenes.glass_lattice()
    W, H = 1920, 1080
    s = seedmod.glibc_rand_stream(2)
    img, bc = hip.render_scene(ctx, sc, W, H, 32, s, count_bounces=True)
    for r in (3, 540, 900):
        want = oracle.render(sc, W, H, 32, s, id_begin=r * W, id_end=(r + 1) * W)
        assert np.array_equal(bits(img[r, :, :3]), bits(want[r, :, :3])), "row %d" % r
    assert 20.0 < bc / (W * H * 2) < 32.0


def test_config4_one_rank_at_its_full_sample_count(ctx, oracle):
    """BASELINE config 4 as one of eight ranks runs it: 3840x2160, 1024 spp (64 rounds of 16 lanes per pixel, the last
    bands at 32), bands of 8 rows, rank 3.  One row of the rank's bands against the oracle at the full 1024 spp, bit for
    bit; rows of other ranks stay untouched."""
    sc = scenes.main_lua_frame(0)
    W, H, spp = 3840, 2160, 1024
    s = seedmod.glibc_rand_stream(spp)
    ctx.scene_upload(sc["spheres"])
    ctx.image_alloc(W, H)
    ctx.render(sc, W, H, 50, s, rank=3, world=8, band_rows=8)
    img = ctx.image_read()
    r = 8 * (3 + 8 * 20) + 5                       # a row of rank 3's 21st band
    want = oracle.render(sc, W, H, 50, s, id_begin=r * W, id_end=(r + 1) * W, nthreads=16)
    assert np.array_equal(bits(img[r, :, :3]), bits(want[r, :, :3])), mismatch_report(img[r, :, :3], want[r, :, :3])
    assert not img[8 * 4:8 * 11].any()             # bands of ranks 4..7, 0..2: still the cleared accumulator


@pytest.mark.parametrize("k", [4, 5, 6])
def test_larger_closed_scenes(ctx, oracle, k):
    """Glass lattices of 4^3, 5^3 and 6^3 spheres in the closed box (72, 133 and 224 spheres with the walls, light and
    mirror): the first two are rendered by the lock-step kernel's scan loop (pt_render picks it up to 160 spheres in a
    closed scene; the small-sphere mask, 64 trios wide, is walked as runs of trios), the third by the streaming kernel;
    forced the other way round they must give the same bits."""
    sc = scenes.glass_lattice(k)
    s = seedmod.glibc_rand_stream(6)
    want, wb = oracle.render(sc, 96, 54, 32, s, count_bounces=True)
    try:
        for kernel in (None, "lds", "stream"):
            if kernel:
                os.environ["PT_KERNEL"] = kernel
            img, bc = hip.render_scene(ctx, sc, 96, 54, 32, s, count_bounces=True)
            assert np.array_equal(bits(img[..., :3]), bits(want[..., :3])) and bc == wb, (k, kernel)
    finally:
        os.environ.pop("PT_KERNEL", None)


def test_config5_and_config3_at_their_full_sample_counts(ctx, oracle):
    """BASELINE config 5 (glass lattice, depth 32, 1080p, 64 spp) and config 3 (10^4 random spheres, 1080p, 64 spp) as
    DESIGN.md section 6 times them: a row / segments of the frame against the oracle at the full sample count."""
    W, H, spp = 1920, 1080, 64
    s = seedmod.glibc_rand_stream(spp)
    sc = scenes.glass_lattice()
    img = hip.render_scene(ctx, sc, W, H, 32, s)
    want = oracle.render(sc, W, H, 32, s, id_begin=540 * W, id_end=541 * W, nthreads=16)
    assert np.array_equal(bits(img[540, :, :3]), bits(want[540, :, :3])), mismatch_report(img[540, :, :3], want[540, :, :3])
    sc = scenes.dense_random(10000)
    flat = hip.render_scene(ctx, sc, W, H, 50, s).reshape(-1, 4)
    for start in (0, 540 * W + 928, 1079 * W + 1856):
        want = oracle.render(sc, W, H, 50, s, id_begin=start, id_end=start + 64, nthreads=16).reshape(-1, 4)
        assert np.array_equal(bits(flat[start:start + 64, :3]), bits(want[start:start + 64, :3])), "pixels %d.." % start


@pytest.mark.parametrize("kernel", ["lds", "stream"])
@pytest.mark.parametrize("name", sorted(n for n in CASES if not n.startswith(("dense3000", "dense513", "empty"))))
def test_both_kernels_on_every_small_fixture(ctx, kernel, name):
    """The kernel choice is a matter of speed only: the lock-step kernel (default up to 12 spheres, 160 in a closed scene) and the persistent
    streaming kernel (default beyond) must both reproduce the strict reference fixture on every scene either can hold
    (depth 0 and 1, the stuck RNG of pixel 0, duplicates and ties included)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    try:
        os.environ["PT_KERNEL"] = kernel
        img, bc = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
    finally:
        os.environ.pop("PT_KERNEL", None)
    assert np.array_equal(bits(img[..., :3]), bits(g["strict"])), mismatch_report(img[..., :3], g["strict"])
    assert int(img[..., 3].astype(np.float64).sum()) == bc


@pytest.mark.parametrize("name", sorted(n for n in CASES if not n.startswith(("dense", "empty"))))
def test_small_sphere_tests_in_the_scan_change_nothing(ctx, name):
    """The lock-step kernel's wave-level "can any lane hit it" tests (trios holding a sphere much smaller than the
    largest one, pt_render's trioSkipMask) are a speed heuristic: with them switched off (PT_NO_TRIO_SKIP, the other
    template instance of the kernel) the image is the same strict reference fixture."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    try:
        os.environ["PT_NO_TRIO_SKIP"] = "1"
        os.environ["PT_KERNEL"] = "lds"
        img, _ = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
    finally:
        os.environ.pop("PT_NO_TRIO_SKIP", None)
        os.environ.pop("PT_KERNEL", None)
    assert np.array_equal(bits(img[..., :3]), bits(g["strict"])), mismatch_report(img[..., :3], g["strict"])


@pytest.mark.parametrize("name", ["frame0_64x36", "frame77_96x54", "frame211_80x45", "depth1_32x32", "frame0_scale_m45_48x27", "frame0_scale_p57_48x27"])
@pytest.mark.parametrize("noskip", [False, True])
def test_scan_loop_and_unrolled_scan_agree_on_nine_sphere_tables(ctx, name, noskip):
    """Tables of exactly nine spheres (the reference's scene) run an unrolled instance of the lock-step scan, one per
    small-sphere mask; PT_GENERIC_SCAN forces the scan loop every other table size uses.  Both must give the strict
    reference fixture (the default path is checked by test_render_matches_oracle_and_reference_fixture)."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, depth, spp = (int(x) for x in g["size"])
    sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
    assert sc["spheres"].shape[0] == 9
    try:
        os.environ["PT_GENERIC_SCAN"] = "1"
        if noskip:
            os.environ["PT_NO_TRIO_SKIP"] = "1"
        img, bc = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
        img2 = hip.render_scene(ctx, sc, W, H, depth, g["seeds"])
    finally:
        os.environ.pop("PT_GENERIC_SCAN", None)
        os.environ.pop("PT_NO_TRIO_SKIP", None)
    assert np.array_equal(bits(img[..., :3]), bits(g["strict"])), mismatch_report(img[..., :3], g["strict"])
    assert np.array_equal(bits(img2[..., :3]), bits(g["strict"]))
