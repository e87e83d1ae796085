"""ctypes bindings for the CPU oracle (oracle/libpt_oracle.so) and, where it was built,
the x86 build of the reference kernel (oracle/_ref/libptref_{strict,fast}.so).

TEST INFRASTRUCTURE: imported only by tests/, tools/make_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libpt_oracle.so")
REF_SO = {k: os.path.join(ROOT, "oracle", "_ref", "libptref_%s.so" % k) for k in ("strict", "fast", "env1", "env2", "env3")}

_f = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)


def _fp(a):
    return a.ctypes.data_as(_f)


class _Lib:
    """Common wrapper: `prefix` is 'pto_' (restatement) or 'ptref_' (reference build)."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        g = lambda n: getattr(self.lib, prefix + n)
        g("init_rng_1").restype = C.c_uint64
        g("init_rng_1").argtypes = [C.c_uint64]
        g("init_rng_2").restype = C.c_uint64
        g("init_rng_2").argtypes = [C.c_uint64, C.c_uint64]
        g("rand_float").restype = C.c_float
        g("rand_float").argtypes = [_u64p]
        g("random_unit_vector").argtypes = [_u64p, _f]
        g("rotate_vector").argtypes = [_f, _f, _f]
        g("reflection_dir").argtypes = [_f, _f, _f]
        g("reflectance").restype = C.c_float
        g("reflectance").argtypes = [C.c_float, C.c_float]
        g("refraction_dir").argtypes = [_f, _f, C.c_float, _f]
        g("hit_sphere").restype = C.c_int
        g("hit_sphere").argtypes = [_f, _f, _f, C.c_float, _f]
        g("cast_ray").argtypes = [_f, _f, C.c_int, _f, _f, C.c_int, _u64p, _f]
        g("render").restype = C.c_int
        base = [_f, C.c_int, C.c_int, C.c_int, _f, _f, _f, C.c_int, _u64p, C.c_int, C.c_int,
                C.c_long, C.c_long, C.c_int]
        g("render").argtypes = base + ([_u64p] if prefix == "pto_" else [])
        if prefix == "pto_":
            self.lib.pto_postprocess.argtypes = [_f, C.c_int, C.POINTER(C.c_ubyte)]

    def _g(self, n):
        return getattr(self.lib, self.prefix + n)

    # --- scalar pieces -------------------------------------------------------------
    def init_rng_1(self, a):
        return int(self._g("init_rng_1")(a))

    def init_rng_2(self, a, b):
        return int(self._g("init_rng_2")(a, b))

    def rand_float(self, state):
        s = C.c_uint64(state)
        f = self._g("rand_float")(C.byref(s))
        return int(s.value), np.float32(f)

    def random_unit_vector(self, state):
        s = C.c_uint64(state)
        out = np.zeros(3, np.float32)
        self._g("random_unit_vector")(C.byref(s), _fp(out))
        return int(s.value), out

    def _v3(self, name, *vecs, scalar=None):
        out = np.zeros(3, np.float32)
        args = [_fp(np.ascontiguousarray(v, np.float32)) for v in vecs]
        keep = [np.ascontiguousarray(v, np.float32) for v in vecs]
        args = [_fp(k) for k in keep]
        if scalar is not None:
            args.append(C.c_float(scalar))
        self._g(name)(*args, _fp(out))
        return out

    def rotate_vector(self, vec, rot):
        return self._v3("rotate_vector", vec, rot)

    def reflection_dir(self, i, n):
        return self._v3("reflection_dir", i, n)

    def refraction_dir(self, i, n, rel):
        return self._v3("refraction_dir", i, n, scalar=rel)

    def reflectance(self, cos_theta, rel):
        return np.float32(self._g("reflectance")(cos_theta, rel))

    def hit_sphere(self, o, d, c, radius):
        o, d, c = (np.ascontiguousarray(v, np.float32) for v in (o, d, c))
        t = C.c_float(0)
        h = self._g("hit_sphere")(_fp(o), _fp(d), _fp(c), C.c_float(radius), C.byref(t))
        return int(h), np.float32(t.value)

    def cast_ray(self, scene, o, d, max_depth, state):
        sp = np.ascontiguousarray(scene["spheres"], np.float32)
        bg = np.ascontiguousarray(scene["bg"], np.float32)
        o, d = (np.ascontiguousarray(v, np.float32) for v in (o, d))
        s = C.c_uint64(state)
        out = np.zeros(3, np.float32)
        self._g("cast_ray")(_fp(bg), _fp(sp), sp.shape[0], _fp(o), _fp(d), max_depth, C.byref(s), _fp(out))
        return int(s.value), out

    # --- the host loop -------------------------------------------------------------
    def render(self, scene, W, H, max_depth, seeds, first_frame=0, image=None, id_begin=0, id_end=None,
               nthreads=None, count_bounces=False):
        """Returns the linear float accumulator [H, W, 4] (4th lane untouched = 0)."""
        sp = np.ascontiguousarray(scene["spheres"], np.float32).reshape(-1, 12)
        bg = np.ascontiguousarray(scene["bg"], np.float32)
        cam = np.ascontiguousarray(scene["cam"], np.float32)
        seeds = np.ascontiguousarray(seeds, np.uint64)
        if image is None:
            image = aligned_zeros(W * H * 4)
        if id_end is None:
            id_end = W * H
        if nthreads is None:
            nthreads = min(8, os.cpu_count() or 1)
        args = [_fp(image), W, H, max_depth, _fp(bg), _fp(cam), _fp(sp), sp.shape[0],
                seeds.ctypes.data_as(_u64p), first_frame, len(seeds), id_begin, id_end, nthreads]
        bounces = C.c_uint64(0)
        if self.prefix == "pto_":
            args.append(C.byref(bounces))
        rc = self._g("render")(*args)
        if rc != 0:
            raise RuntimeError("%srender failed: %d" % (self.prefix, rc))
        img = image.reshape(H, W, 4)
        return (img, int(bounces.value)) if count_bounces else img

    def postprocess(self, image):
        img = np.ascontiguousarray(image, np.float32).reshape(-1, 4)
        out = np.zeros((img.shape[0], 3), np.uint8)
        self.lib.pto_postprocess(_fp(img), img.shape[0], out.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out


def aligned_zeros(nfloats, align=64):
    raw = np.zeros(nfloats + align // 4, np.float32)
    off = (-raw.ctypes.data % align) // 4
    return raw[off:off + nfloats]


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        raise RuntimeError("oracle/libpt_oracle.so missing — run `make -C oracle oracle` (or __graft_entry__.build())")
    return _Lib(ORACLE_SO, "pto_")


def have_ref(kind="strict"):
    return os.path.exists(REF_SO[kind])


def load_ref(kind="strict"):
    return _Lib(REF_SO[kind], "ptref_")
