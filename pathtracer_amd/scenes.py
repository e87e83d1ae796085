"""Scene descriptions used by bench.py, the tests and the C scene driver's Python twin.

A scene is a plain dict of numpy arrays in the interchange layout every layer of this
repo shares (oracle/, the C ABI in include/pt_hip.h, tests/):

    spheres : float32 [N, 12] = cx cy cz radius | type colr colg colb | tint fuzz refIdx 0
              (type 0 light, 1 lambert, 2 metal, 3 dielectric — the codes of
              /root/reference/src/c/renderer.c:119-133)
    bg      : float32 [3]     background colour   (renderer.c:115-117)
    cam     : float32 [10]    pos3 rot3 sensorWidth focalLength aperture exposure (renderer.c:147-151)

Insertion order is kept: the closest-hit scan breaks ties towards the lowest index
(/root/reference/src/cl/ray.cl:101-110).
"""
import math

import numpy as np

LIGHT, LAMBERT, METAL, DIELECTRIC = 0, 1, 2, 3


def _sphere(x, y, z, r, mat):
    t, cr, cg, cb, tint, fuzz, ref = mat
    return [x, y, z, r, float(t), cr, cg, cb, tint, fuzz, ref, 0.0]


def lambertian(r, g, b):
    return (LAMBERT, r, g, b, 0.0, 0.0, 0.0)


def metal(r, g, b, tint, fuzz):
    return (METAL, r, g, b, tint, fuzz, 0.0)


def dielectric(r, g, b, tint, fuzz, ref_idx):
    return (DIELECTRIC, r, g, b, tint, fuzz, ref_idx)


def light_source(r, g, b):
    return (LIGHT, r, g, b, 0.0, 0.0, 0.0)


def _pack(spheres, bg, cam):
    # Lua numbers are doubles, narrowed to float at the binding
    # (/root/reference/src/c/lua_wrapper.c:166-188) — np.float32() does the same narrowing.
    return {
        "spheres": np.asarray(spheres, dtype=np.float64).astype(np.float32).reshape(-1, 12),
        "bg": np.asarray(bg, dtype=np.float64).astype(np.float32),
        "cam": np.asarray(cam, dtype=np.float64).astype(np.float32),
    }


def main_lua_frame(i=0, frames=300):
    """Scene of animation frame `i` of /root/reference/scripts/main.lua:2-94."""
    wall = 10000.0
    white, red, green = lambertian(1, 1, 1), lambertian(1, 0, 0), lambertian(0, 1, 0)
    light = light_source(10, 10, 10)
    mirror = metal(0, 0, 0, 0, 0)
    green_glass = dielectric(0, 1, 0, 0.8, 0, 0.95)
    s = [
        _sphere(0, wall + 6, -10, wall, white),
        _sphere(0, -(wall + 6), -10, wall, white),
        _sphere(0, 0, -(wall + 20), wall, white),
        _sphere(0, 0, (wall + 20), wall, white),
        _sphere(wall + 6, 0, -5, wall, green),
        _sphere(-(wall + 6), 0, -5, wall, red),
        _sphere(-1, 0, -10, 1, green_glass),
    ]
    cx, cy, cz = -1.0, 0.0, -10.0
    ph = 2 * math.pi * (i / frames)
    s.append(_sphere(cx + 2 * math.cos(ph), cy + math.sin(ph), cz + 2 * math.sin(ph), 0.5, light))
    ph2 = -4 * math.pi * (i / frames)
    s.append(_sphere(cx + 3.5 * math.cos(ph2), cy + 0, cz + 3.5 * math.sin(ph2), 1, mirror))
    start = 3 * math.pi / 4
    cam = [cx + 0, cy + 5 * math.cos(ph + start), cz + 5 * math.sin(ph + start),
           ph - math.pi / 2 + start, 0, 0, 15, 20, 0.001, 1000]
    return _pack(s, [0, 0, 0], cam)


class _Lcg64:
    """Fixed 64-bit LCG (Knuth MMIX constants) — our own generator, so the synthetic
    scenes are reproducible without libc."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def u01(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return (self.s >> 11) / float(1 << 53)

    def uniform(self, a, b):
        return a + (b - a) * self.u01()


def dense_random(n=10000, seed=12345):
    """Config 3 — intersection stress: n random spheres in [-50,50]^3 under a sky."""
    g = _Lcg64(seed)
    s = []
    for _ in range(n):
        x, y, z = g.uniform(-50, 50), g.uniform(-50, 50), g.uniform(-50, 50)
        r = g.uniform(0.1, 1.0)
        m = g.u01()
        cr, cg, cb = g.u01(), g.u01(), g.u01()
        if m < 0.70:
            mat = lambertian(cr, cg, cb)
        elif m < 0.85:
            mat = metal(cr, cg, cb, g.u01(), 0.3 * g.u01())
        elif m < 0.95:
            mat = dielectric(cr, cg, cb, 0.5 * g.u01(), 0.0, 1.5)
        else:
            mat = light_source(10 * cr, 10 * cg, 10 * cb)
        s.append(_sphere(x, y, z, r, mat))
    cam = [0, 0, 120, 0, 0, 0, 15, 20, 0.001, 1000]
    return _pack(s, [0.5, 0.7, 1.0], cam)


def glass_lattice(k=3):
    """Config 5 — divergence stress: the frame-0 box with the glass sphere replaced by a
    3x3x3 lattice of dielectric spheres and four small emissive spheres (depth 32).
    k > 3: the same 1.8-unit cube filled with k^3 proportionally smaller spheres (closed scenes with larger tables)."""
    base = main_lua_frame(0)
    keep = [row for j, row in enumerate(base["spheres"].tolist()) if j != 6]
    glass = dielectric(1, 1, 1, 0.2, 0, 1.5)
    step = 0.9 if k == 3 else 1.8 / (k - 1)
    rad = 0.4 if k == 3 else step * 0.4 / 0.9
    half = (k - 1) / 2.0
    for ix in range(k):
        for iy in range(k):
            for iz in range(k):
                keep.append(_sphere(-1 + step * (ix - half), step * (iy - half), -10 + step * (iz - half), rad, glass))
    for (x, y, z) in ((-3.5, 2.5, -11), (2.0, -2.5, -9), (-2.5, -2.0, -13), (1.5, 3.0, -12)):
        keep.append(_sphere(x, y, z, 0.3, light_source(10, 8, 6)))
    return {"spheres": np.asarray(keep, dtype=np.float32), "bg": base["bg"], "cam": base["cam"]}


def scaled(scene, log2_scale):
    """The same scene in other units: every length times 2^log2_scale (exact), exposure divided by it.  At
    2^-45 .. 2^-60 the discriminants and hit parameters leave the ranges the kernels' short arithmetic is exact on
    (everything goes to the reference scan); at 2^50 they approach overflow, at 2^57 they overflow to inf/NaN.
    The reference's fixed 0.0001 offset (src/cl/ray.cl:119) is not scaled, so these are different images."""
    k = np.float32(2.0) ** np.float32(log2_scale)
    sp = scene["spheres"].copy()
    sp[:, :4] *= k
    cam = scene["cam"].copy()
    cam[0:3] *= k
    cam[6:9] *= k          # sensorWidth, focalLength, aperture
    cam[9] /= k            # exposure: colour = colour * exposure * aperture (src/cl/main.cl:27)
    return {"spheres": sp, "bg": scene["bg"].copy(), "cam": cam}


def material_micro(kind):
    """One sphere of one material under a bright sky — one scene per shading branch."""
    mats = {
        "light": light_source(4, 3, 2),
        "lambert": lambertian(0.8, 0.6, 0.4),
        "metal": metal(0.9, 0.7, 0.5, 0.6, 0.25),
        "dielectric": dielectric(0.6, 0.9, 0.7, 0.5, 0.1, 1.5),
        "dielectric_low": dielectric(0.2, 1.0, 0.3, 0.8, 0.0, 0.95),
        # directions of length ~2^25 and ~2^70 after the first bounce: a = |d|^2 outside the scan's [2^-40, 2^40], and overflowing
        "metal_fuzz25": metal(0.9, 0.7, 0.5, 0.6, 2.0 ** 25),
        "metal_fuzz70": metal(0.9, 0.7, 0.5, 0.6, 2.0 ** 70),
        "dielectric_idx30": dielectric(0.6, 0.9, 0.7, 0.5, 0.1, 2.0 ** 30),
        # a material code that none of the kernel's branches matches (/root/reference/src/cl/ray.cl:130-171; the code
        # comes straight from the Lua table, lua_wrapper.c:41-42): the ray is left as it is and re-hits until the depth runs out
        "unknown7": (7, 0.9, 0.1, 0.1, 0.5, 0.1, 1.5),
        "unknown_neg": (-2, 0.9, 0.1, 0.1, 0.5, 0.1, 1.5),
    }
    s = [
        _sphere(0, 0, -6, 1.5, mats[kind]),
        _sphere(0, -1001.5, -6, 1000, lambertian(0.5, 0.5, 0.5)),
    ]
    cam = [0, 0.5, 0, 0.05, 0.1, 0.3, 4, 5, 0.01, 100]
    return _pack(s, [0.6, 0.7, 0.9], cam)


def degenerate():
    """Ties and touching surfaces for the closest-hit scan: exact duplicates (the lowest index must
    win, src/cl/ray.cl:104 strict <), concentric shells, tangent spheres, a zero-radius sphere, a sphere
    around the camera (near root only: never hit from inside) and one far behind it."""
    s = [
        _sphere(0, -1001, -8, 1000, lambertian(0.7, 0.7, 0.7)),
        _sphere(-1.5, 0, -8, 1, metal(0.9, 0.2, 0.2, 0.7, 0.1)),
        _sphere(-1.5, 0, -8, 1, lambertian(0.2, 0.9, 0.2)),            # duplicate of the metal sphere
        _sphere(1.5, 0, -8, 1, dielectric(0.9, 0.9, 1.0, 0.3, 0.0, 1.5)),
        _sphere(1.5, 0, -8, 0.5, light_source(6, 5, 4)),               # concentric, inside the glass
        _sphere(1.5, 0, -8, 1, light_source(1, 2, 3)),                 # duplicate of the glass shell
        _sphere(0, 1.5, -8, 0.5, lambertian(0.3, 0.3, 0.9)),
        _sphere(0, 2.5, -8, 0.5, metal(0.8, 0.8, 0.8, 0.5, 0.0)),      # tangent to the one below
        _sphere(0, 0.5, -6, 0.0, light_source(9, 9, 9)),               # zero radius
        _sphere(0, 0, 0, 3, lambertian(0.5, 0.1, 0.1)),                # camera inside
        _sphere(0, 0, 40, 5, light_source(3, 3, 3)),                   # behind the camera
        _sphere(0, -1001, -8, 1000, light_source(5, 5, 5)),            # duplicate of the floor
    ]
    cam = [0, 0.5, 0, 0.0, 0.0, 0.0, 6, 7, 0.02, 50]
    return _pack(s, [0.4, 0.5, 0.7], cam)


def on_surface():
    """The camera sits EXACTLY on a sphere (|pos - centre|^2 == r^2 in binary32), looking into it: every primary ray meets
    that sphere at distance exactly +0, which is no hit (`distance > 0`, src/cl/ray.cl:82) — the one value the kernels'
    key scans cannot order and hand to their serial fallback.  A light and a mirror inside, a diffuse shell behind."""
    s = [
        _sphere(0, 0, -5, 5, lambertian(0.8, 0.3, 0.3)),               # the camera is on this one
        _sphere(0.5, 0.25, -3, 1, light_source(4, 4, 3)),
        _sphere(-1.5, -0.5, -4, 1, metal(0.9, 0.9, 0.9, 0.5, 0.0)),
        _sphere(0, 0, -5, 8, lambertian(0.5, 0.6, 0.7)),               # concentric shell: seen from inside never (near root), from the gap yes
        _sphere(0, 3, 2, 3, dielectric(0.9, 1.0, 0.9, 0.3, 0.0, 1.5)), # the camera is on this one too: 0 + 9 + 4 == 13 != 9 -> not exactly; plain neighbour
    ]
    cam = [0, 0, 0, 0.0, 0.0, 0.0, 6, 7, 0.02, 50]
    return _pack(s, [0.4, 0.5, 0.7], cam)


def empty():
    """No spheres: every path misses on its first bounce and returns the background."""
    cam = [0, 0, 0, 0, 0, 0, 15, 20, 0.001, 1000]
    return {"spheres": np.zeros((0, 12), np.float32), "bg": np.asarray([0.25, 0.5, 0.75], np.float32),
            "cam": np.asarray(cam, np.float32)}
