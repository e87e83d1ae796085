"""ctypes binding of libpt_hip.so (include/pt_hip.h) — plumbing for bench.py and the tests.

The product is the shared library and the C host above it (pathtracer_amd/csrc/renderer.c);
this module only hands numpy / torch buffers to the C ABI.  There is NO fallback: if the
library is missing or a call fails, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_HIP_LIB: debugging aid for A/B timing of kernel variants (tools/ab_build.sh): another build of the same library
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "libpt_hip.so")

PT_SPHERE_FLOATS = 12


class PtError(RuntimeError):
    pass


class RenderArgs(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("max_depth", C.c_int),
        ("background", C.c_float * 3),
        ("cam_pos", C.c_float * 3), ("cam_rot", C.c_float * 3),
        ("sensor_width", C.c_float), ("focal_length", C.c_float), ("aperture", C.c_float), ("exposure", C.c_float),
        ("seeds", C.POINTER(C.c_uint64)),
        ("first_frame", C.c_int), ("n_frames", C.c_int),
        ("rank", C.c_int), ("world", C.c_int), ("band_rows", C.c_int),
        ("d_image", C.c_void_p), ("stream", C.c_void_p),
        ("bounce_count", C.POINTER(C.c_uint64)),
        ("async_", C.c_int),
        ("kernel_ms", C.POINTER(C.c_float)),
        ("record_timing", C.c_int),
        ("fast_math", C.c_int),
        ("accel", C.c_int),
    ]


_lib = None

# every symbol include/pt_hip.h declares
EXPORTS = ["pt_last_error", "pt_device_count", "pt_ctx_create", "pt_ctx_destroy", "pt_ctx_reload_env", "pt_image_alloc",
           "pt_scene_upload", "pt_plan_launch", "pt_render", "pt_render_noalloc", "pt_combine_bands", "pt_combine_transport",
           "pt_comm_unique_id", "pt_comm_init_rank", "pt_combine_bands_rank",
           "pt_kernel_times", "pt_sync", "pt_image_read", "pt_image_read_u8", "pt_selftest"]
COMM_ID_BYTES = 128
# the PT_* switches a context reads once at creation (include/pt_hip.h pt_ctx_reload_env)
ENV_SWITCHES = ("PT_MATH", "PT_ACCEL", "PT_KERNEL", "PT_GENERIC_SCAN", "PT_NO_TRIO_SKIP", "PT_LANES_PER_PIXEL",
                "PT_TAIL_PCT", "PT_TAIL_LANES", "PT_COMBINE")


class LaunchPlan(C.Structure):
    _fields_ = [("local_pixels", C.c_longlong), ("lanes_log2", C.c_int), ("tail_lanes_log2", C.c_int),
                ("tier_pixels", C.c_longlong), ("tier_threads", C.c_longlong), ("total_threads", C.c_longlong),
                ("blocks", C.c_uint)]


def _env_state():
    return tuple(os.environ.get(k) for k in ENV_SWITCHES)


def load():
    """Load libpt_hip.so (once).  Raises PtError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError("%s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_combine_transport.restype = C.c_char_p
    lib.pt_ctx_reload_env.argtypes = [C.c_void_p]
    lib.pt_plan_launch.argtypes = [C.POINTER(RenderArgs), C.POINTER(LaunchPlan)]
    lib.pt_comm_unique_id.argtypes = [C.c_void_p, C.c_int]
    lib.pt_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pt_combine_bands_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pt_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.pt_ctx_destroy.argtypes = [C.c_void_p]
    lib.pt_image_alloc.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.pt_render.argtypes = [C.c_void_p, C.POINTER(RenderArgs)]
    lib.pt_render_noalloc.argtypes = [C.c_void_p, C.POINTER(RenderArgs)]
    lib.pt_combine_bands.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    lib.pt_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    lib.pt_sync.argtypes = [C.c_void_p]
    lib.pt_image_read.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.pt_image_read_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ubyte)]
    lib.pt_selftest.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
    for name in EXPORTS:
        if name not in ("pt_last_error", "pt_combine_transport"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise PtError("%s failed (%d): %s" % (what, rc, load().pt_last_error().decode()))


def device_count():
    n = C.c_int(0)
    _check(load().pt_device_count(C.byref(n)), "pt_device_count")
    return n.value


class Context:
    """One backend context = one GPU + one stream (pt_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib.pt_ctx_create(device, C.byref(h)), "pt_ctx_create")
        self.h = h
        self.size = None
        self._env = _env_state()

    def close(self):
        if self.h:
            self.lib.pt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def image_alloc(self, width, height):
        _check(self.lib.pt_image_alloc(self.h, width, height), "pt_image_alloc")
        self.size = (width, height)

    def scene_upload(self, spheres):
        sp = np.ascontiguousarray(spheres, np.float32).reshape(-1, PT_SPHERE_FLOATS)
        _check(self.lib.pt_scene_upload(self.h, sp.ctypes.data_as(C.POINTER(C.c_float)), sp.shape[0]), "pt_scene_upload")

    def render(self, scene, width, height, max_depth, seeds, first_frame=0, rank=0, world=1, band_rows=16,
               d_image=None, stream=None, count_bounces=False, async_=False, time_kernel=False, record_timing=False, fast_math=False, accel=0,
               noalloc=False):
        """scene: dict with 'bg' and 'cam' (pathtracer_amd.scenes layout); the spheres must have
        been uploaded with scene_upload().  Returns the bounce count (count_bounces) or the
        HIP-event kernel time in ms (time_kernel) if asked, else None."""
        seeds = np.ascontiguousarray(seeds, np.uint64)
        if _env_state() != self._env:      # the tests flip PT_* switches between renders; the library reads them once per context
            self._env = _env_state()
            _check(self.lib.pt_ctx_reload_env(self.h), "pt_ctx_reload_env")
        a = RenderArgs()
        a.width, a.height, a.max_depth = width, height, max_depth
        bg = np.asarray(scene["bg"], np.float32)
        cam = np.asarray(scene["cam"], np.float32)
        a.background[:] = [float(x) for x in bg]
        a.cam_pos[:] = [float(x) for x in cam[0:3]]
        a.cam_rot[:] = [float(x) for x in cam[3:6]]
        a.sensor_width, a.focal_length, a.aperture, a.exposure = (float(x) for x in cam[6:10])
        a.seeds = seeds.ctypes.data_as(C.POINTER(C.c_uint64))
        a.first_frame, a.n_frames = first_frame, len(seeds)
        a.rank, a.world, a.band_rows = rank, world, band_rows
        a.d_image = d_image
        a.stream = stream
        bc = C.c_uint64(0xdeadbeef)      # the library writes the count, also for a rank that owns no band
        a.bounce_count = C.pointer(bc) if count_bounces else None
        a.async_ = 1 if async_ else 0
        ms = C.c_float(-1.0)
        a.kernel_ms = C.pointer(ms) if time_kernel else None
        a.record_timing = 1 if record_timing else 0
        a.fast_math = 1 if fast_math else 0
        a.accel = int(accel)
        if noalloc:
            _check(self.lib.pt_render_noalloc(self.h, C.byref(a)), "pt_render_noalloc")
        else:
            _check(self.lib.pt_render(self.h, C.byref(a)), "pt_render")
        if time_kernel:
            return float(ms.value)
        return int(bc.value) if count_bounces else None

    def kernel_times(self, cap=256):
        """Durations (ms) of the launches made with record_timing=True since the last call."""
        ms = (C.c_float * cap)()
        n = C.c_int(0)
        _check(self.lib.pt_kernel_times(self.h, ms, cap, C.byref(n)), "pt_kernel_times")
        return [float(ms[i]) for i in range(n.value)]

    def sync(self):
        _check(self.lib.pt_sync(self.h), "pt_sync")

    def comm_init_rank(self, world, rank, comm_id):
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        _check(self.lib.pt_comm_init_rank(self.h, world, rank, buf), "pt_comm_init_rank")

    def combine_bands_rank(self, root, band_rows, d_image=None, width=0, height=0, stream=None):
        _check(self.lib.pt_combine_bands_rank(self.h, root, band_rows, d_image, width, height, stream), "pt_combine_bands_rank")
        return self.lib.pt_combine_transport().decode()

    def image_read(self):
        w, h = self.size
        out = np.empty((h, w, 4), np.float32)
        _check(self.lib.pt_image_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float))), "pt_image_read")
        return out

    def image_read_u8(self, d_image=None, width=0, height=0):
        if d_image is None:
            width, height = self.size
        out = np.empty((height, width, 3), np.uint8)
        _check(self.lib.pt_image_read_u8(self.h, d_image, width, height, out.ctypes.data_as(C.POINTER(C.c_ubyte))),
               "pt_image_read_u8")
        return out

    def selftest(self, op, a=None, b=None, in_f=None, n=None):
        """Returns (out_u64[n], out_f[n,4]) — see pt_selftest in include/pt_hip.h."""
        if n is None:
            n = len(a) if a is not None else len(in_f) // 2
        pa = np.ascontiguousarray(a if a is not None else np.zeros(n), np.uint64)
        pb = np.ascontiguousarray(b if b is not None else np.zeros(n), np.uint64)
        pf = np.ascontiguousarray(in_f if in_f is not None else np.zeros(2 * n), np.float32)
        ou = np.zeros(n, np.uint64)
        of = np.zeros(4 * n, np.float32)
        u64p, fp = C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        _check(self.lib.pt_selftest(self.h, op, pa.ctypes.data_as(u64p), pb.ctypes.data_as(u64p), pf.ctypes.data_as(fp),
                                    n, ou.ctypes.data_as(u64p), of.ctypes.data_as(fp)), "pt_selftest")
        return ou, of


def combine_bands(ctxs, root, band_rows):
    """pt_combine_bands over a list of Context objects (rank i = ctxs[i]); returns the transport that was used."""
    for c in ctxs:
        if _env_state() != c._env:
            c._env = _env_state()
            _check(c.lib.pt_ctx_reload_env(c.h), "pt_ctx_reload_env")
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    _check(load().pt_combine_bands(arr, len(ctxs), root, band_rows), "pt_combine_bands")
    return load().pt_combine_transport().decode()


def plan_launch(width, height, n_frames, first_frame=0, rank=0, world=1, band_rows=8):
    """pt_plan_launch: the lock-step kernel's launch geometry for these arguments (host arithmetic, no GPU)."""
    a = RenderArgs()
    a.width, a.height, a.n_frames, a.first_frame = width, height, n_frames, first_frame
    a.rank, a.world, a.band_rows = rank, world, band_rows
    out = LaunchPlan()
    _check(load().pt_plan_launch(C.byref(a), C.byref(out)), "pt_plan_launch")
    return out


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().pt_comm_unique_id(buf, COMM_ID_BYTES), "pt_comm_unique_id")
    return buf.raw


def render_scene(ctx, scene, width, height, max_depth, seeds, **kw):
    """Convenience for tests: upload, allocate, render all samples, read back [H,W,4] float32."""
    ctx.scene_upload(scene["spheres"])
    ctx.image_alloc(width, height)
    bc = ctx.render(scene, width, height, max_depth, seeds, **kw)
    img = ctx.image_read()
    return (img, bc) if kw.get("count_bounces") else img
