"""The reference kernel itself on the GPU: oracle/_ref/ptref_gfx950.hsaco through hipModuleLoad.

TEST INFRASTRUCTURE (the float pin of DESIGN.md §2): imported only by tests/ and by the tools that freeze or time
it (tools/make_gpuref_golden.py, tools/ref_gfx950_time.py).  Nothing under pathtracer_amd/ and nothing inside
bench.py's timed region loads it.

The code object is /root/reference/src/cl/main.cl compiled VERBATIM for gfx950 by `make -C oracle ref_gfx950` with the
reference's own build options (src/c/renderer.c:13) and ROCm's own OpenCL device library — every sin/cos/sqrt/pow/
dot/length/normalize is AMD's, none is ours.  This module is the reference's host loop around it
(src/c/renderer.c:155-179,251): the AoS sphere table, the by-value ImageInfo / SceneInfo / Camera kernel arguments,
ONE launch per sample with global size W*H, `frame` and `seed` per launch, a device-wide sync after each.

Launch geometry: OpenCL's `local = NULL` leaves the work-group size to the runtime; nothing in the kernel depends on
it (no LDS, no barrier; `get_global_id(0)` only).  We use the code object's max_flat_workgroup_size (256) and, where
W*H is not a multiple of it, allocate the image with that many pixels of padding: the surplus work-items compute rows
>= H into the padding, which is never read.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HSACO = os.path.join(ROOT, "oracle", "_ref", "ptref_gfx950.hsaco")
HSACO_STRICT = os.path.join(ROOT, "oracle", "_ref", "ptref_gfx950_strict.hsaco")
BLOCK = 256


class _Float3(C.Structure):            # cl_float3: 16 bytes
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class ImageInfo(C.Structure):          # src/cl/ray.cl:6-9, 16 bytes
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("maxRayDepth", C.c_int), ("_pad", C.c_int)]


class SceneInfo(C.Structure):          # src/cl/ray.cl:11-14, 32 bytes
    _fields_ = [("color", _Float3), ("sphereCount", C.c_int), ("_pad", C.c_int * 3)]


class Camera(C.Structure):             # src/cl/ray.cl:30-37, 48 bytes
    _fields_ = [("pos", _Float3), ("rot", _Float3), ("sensorWidth", C.c_float), ("focalLength", C.c_float),
                ("aperture", C.c_float), ("exposure", C.c_float)]


assert C.sizeof(ImageInfo) == 16 and C.sizeof(SceneInfo) == 32 and C.sizeof(Camera) == 48


def spheres_aos(spheres):
    """Interchange rows [N,12] -> the reference's 80-byte Sphere records (src/cl/ray.cl:16-28):
    center@0, radius@16, material@32 = {type@0 (int), color@16, tint@32, fuzzyness@36, refIdx@40}."""
    sp = np.ascontiguousarray(spheres, np.float32).reshape(-1, 12)
    out = np.zeros((max(sp.shape[0], 1), 20), np.float32)
    n = sp.shape[0]
    out[:n, 0:3] = sp[:, 0:3]
    out[:n, 4] = sp[:, 3]
    out[:n, 8] = sp[:, 4].astype(np.int32).view(np.float32)
    out[:n, 12:15] = sp[:, 5:8]
    out[:n, 16:19] = sp[:, 8:11]
    return out


class HipError(RuntimeError):
    pass


_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipGetErrorString.argtypes = [C.c_int]
        _hip.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        _hip.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
        _hip.hipModuleUnload.argtypes = [C.c_void_p]
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p,
                                                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    return _hip


def _ck(rc, what):
    if rc != 0:
        raise HipError("%s: %s" % (what, _lib().hipGetErrorString(rc).decode()))


def available(path=HSACO):
    return os.path.exists(path)


class ReferenceKernel:
    """The loaded code object + one device image; `render()` is src/c/renderer.c:153-271 without the post-process."""

    def __init__(self, path=HSACO, device=0):
        hip = _lib()
        _ck(hip.hipSetDevice(device), "hipSetDevice")
        self.mod = C.c_void_p()
        _ck(hip.hipModuleLoad(C.byref(self.mod), path.encode()), "hipModuleLoad(%s)" % path)
        self.fn = C.c_void_p()
        _ck(hip.hipModuleGetFunction(C.byref(self.fn), self.mod, b"pt_main"), "hipModuleGetFunction(pt_main)")
        self.d_image = C.c_void_p()
        self.d_spheres = C.c_void_p()
        self.pixels = 0

    def close(self):
        hip = _lib()
        for p in (self.d_image, self.d_spheres):
            if p:
                hip.hipFree(p)
        self.d_image = C.c_void_p()
        self.d_spheres = C.c_void_p()
        if self.mod:
            hip.hipModuleUnload(self.mod)
            self.mod = C.c_void_p()

    def _alloc(self, pixels):
        hip = _lib()
        padded = (pixels + BLOCK - 1) // BLOCK * BLOCK
        if padded != self.pixels:
            if self.d_image:
                hip.hipFree(self.d_image)
            self.d_image = C.c_void_p()
            _ck(hip.hipMalloc(C.byref(self.d_image), padded * 16), "hipMalloc(image)")
            self.pixels = padded
        return padded

    def upload(self, scene, W, H, max_depth):
        """clCreateBuffer + clEnqueueWriteBuffer of the sphere table and the five launch-invariant arguments
        (src/c/renderer.c:155-163)."""
        hip = _lib()
        aos = spheres_aos(scene["spheres"])
        if self.d_spheres:
            hip.hipFree(self.d_spheres)
        self.d_spheres = C.c_void_p()
        _ck(hip.hipMalloc(C.byref(self.d_spheres), aos.nbytes), "hipMalloc(spheres)")
        _ck(hip.hipMemcpy(self.d_spheres, aos.ctypes.data, aos.nbytes, 1), "hipMemcpy(spheres)")
        self.padded = self._alloc(W * H)
        self.W, self.H = W, H
        bg = [float(x) for x in np.asarray(scene["bg"], np.float32)]
        cam = [float(x) for x in np.asarray(scene["cam"], np.float32)]
        self.info = ImageInfo(W, H, max_depth, 0)
        self.scene = SceneInfo(_Float3(bg[0], bg[1], bg[2], 0.0), int(np.asarray(scene["spheres"]).reshape(-1, 12).shape[0]))
        self.cam = Camera(_Float3(cam[0], cam[1], cam[2], 0.0), _Float3(cam[3], cam[4], cam[5], 0.0), cam[6], cam[7], cam[8], cam[9])

    def launch(self, frame, seed, sync=True):
        """One sample: clSetKernelArg 5/6 + clEnqueueNDRangeKernel + clFinish (src/c/renderer.c:172-179,251)."""
        hip = _lib()
        image, spheres = C.c_void_p(self.d_image.value), C.c_void_p(self.d_spheres.value)
        fr, sd = C.c_int(frame), C.c_uint64(int(seed))
        params = (C.c_void_p * 7)(*[C.cast(C.byref(x), C.c_void_p) for x in (image, self.info, self.scene, self.cam, spheres, fr, sd)])
        _ck(hip.hipModuleLaunchKernel(self.fn, self.padded // BLOCK, 1, 1, BLOCK, 1, 1, 0, None, params, None),
            "hipModuleLaunchKernel(pt_main)")
        if sync:
            _ck(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def read(self):
        out = np.empty((self.padded, 4), np.float32)
        _ck(_lib().hipMemcpy(out.ctypes.data, self.d_image, out.nbytes, 2), "hipMemcpy(image)")
        return out[:self.W * self.H].reshape(self.H, self.W, 4)

    def render(self, scene, W, H, max_depth, seeds, first_frame=0, poison=True):
        """Linear float accumulator [H, W, 4] (4th lane: whatever the float3 store left there)."""
        self.upload(scene, W, H, max_depth)
        if first_frame == 0 and poison:
            # the reference's buffer is uninitialised (src/c/renderer.c:103); sample 0 multiplies it by 0
            _ck(_lib().hipMemset(self.d_image, 0, self.padded * 16), "hipMemset")
        for k, s in enumerate(np.asarray(seeds, np.uint64)):
            self.launch(first_frame + k, s)
        return self.read()
