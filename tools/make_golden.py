#!/usr/bin/env python3
"""Generate tests/golden/ from the x86 build of the REFERENCE kernel (oracle/_ref).

Runs only in the build container (needs /root/reference to have built oracle/_ref via
`make -C oracle ref`).  What is committed is data: inputs (scene arrays, seeds) and the
reference's outputs (linear float accumulators) for each case of tests/cases.py, plus
single-function known-answer vectors.  No reference source is stored.

  strict : kernel compiled without fast-math flags     -> the oracle must match bit for bit
  fast   : kernel compiled with the reference's own PROGRAM_ARGS flags
           (/root/reference/src/c/renderer.c:13)        -> the oracle must match within 1e-5
  env1-3 : (envelope cases only, tests/envelope.py) those flags + three other legal definitions of the OpenCL builtins
           (oracle/ref_shim_variants.cpp)               -> what the float tolerance is stated against
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from cases import CASES  # noqa: E402
from envelope import ENVELOPE_CASES, ENVELOPE_KINDS  # noqa: E402
from pathtracer_amd import seeds as seedmod  # noqa: E402


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    strict, fast = oracle_lib.load_ref("strict"), oracle_lib.load_ref("fast")
    stream = seedmod.glibc_rand_stream(64)
    only = set(sys.argv[1:])     # optional: case names to (re)write; "kat" for the single-function vectors
    for name, (factory, W, H, depth, spp) in CASES.items():
        if only and name not in only:
            continue
        sc = factory()
        s = stream[:spp]
        a = strict.render(sc, W, H, depth, s)
        b = fast.render(sc, W, H, depth, s)
        np.savez_compressed(os.path.join(out, name + ".npz"), spheres=sc["spheres"], bg=sc["bg"], cam=sc["cam"],
                            seeds=s, size=np.asarray([W, H, depth, spp], np.int32),
                            strict=a[..., :3].copy(), fast=b[..., :3].copy())
        print("%-22s %dx%d depth %d spp %d  mean %.5f  |strict-fast|max %.3g" %
              (name, W, H, depth, spp, float(a[..., :3].mean()), float(np.abs(a - b).max())))

    # envelope cases: the same scene through every legal build of the reference (oracle/ref_shim_variants.cpp)
    if not only or "envelope" in only:
        libs = {k: oracle_lib.load_ref(k) for k in ENVELOPE_KINDS}
        for name, (factory, W, H, depth, spp) in ENVELOPE_CASES.items():
            sc = factory()
            s = stream[:spp]
            imgs = {k: libs[k].render(sc, W, H, depth, s)[..., :3].copy() for k in ENVELOPE_KINDS}
            np.savez_compressed(os.path.join(out, name + ".npz"), spheres=sc["spheres"], bg=sc["bg"], cam=sc["cam"],
                                seeds=s, size=np.asarray([W, H, depth, spp], np.int32), **imgs)
            print("%-24s %dx%d depth %d spp %d  %d KB" % (name, W, H, depth, spp, os.path.getsize(os.path.join(out, name + ".npz")) // 1024))

    if only and "kat" not in only:
        return
    # single-function KATs from the strict build
    rng = np.random.RandomState(7)
    kat = {}
    ids = np.asarray([0, 1, 2, 63, 64, 65535, 65536, 2073599, 8294399, 2**31 - 1], np.uint64)
    sds = np.concatenate([stream[:6], np.asarray([0, 1, 2147483646, 2147483647], np.uint64)])
    kat["init2_a"] = np.repeat(ids, len(sds))
    kat["init2_b"] = np.tile(sds, len(ids))
    kat["init2_out"] = np.asarray([strict.init_rng_2(int(a), int(b)) for a, b in zip(kat["init2_a"], kat["init2_b"])], np.uint64)
    st = np.concatenate([np.asarray([0, 1, 2, 16807, 381647767, 2147483646, 2147483583, 2147483455], np.uint64),
                         rng.randint(1, 2**31 - 1, size=200).astype(np.uint64)])
    kat["rand_state_in"] = st
    rs = [strict.rand_float(int(x)) for x in st]
    kat["rand_state_out"] = np.asarray([r[0] for r in rs], np.uint64)
    kat["rand_float"] = np.asarray([r[1] for r in rs], np.float32)
    ru = [strict.random_unit_vector(int(x)) for x in st]
    kat["ruv_state_out"] = np.asarray([r[0] for r in ru], np.uint64)
    kat["ruv"] = np.stack([r[1] for r in ru]).astype(np.float32)
    vecs = rng.uniform(-2, 2, size=(64, 3)).astype(np.float32)
    nrms = rng.uniform(-1, 1, size=(64, 3)).astype(np.float32)
    nrms /= np.linalg.norm(nrms, axis=1, keepdims=True).astype(np.float32)
    rots = rng.uniform(-3.2, 3.2, size=(64, 3)).astype(np.float32)
    kat["vecs"], kat["nrms"], kat["rots"] = vecs, nrms, rots
    kat["rotate"] = np.stack([strict.rotate_vector(v, r) for v, r in zip(vecs, rots)])
    kat["reflect"] = np.stack([strict.reflection_dir(v, n) for v, n in zip(vecs, nrms)])
    rel = rng.uniform(0.5, 2.0, size=64).astype(np.float32)
    cs = rng.uniform(-1.2, 1.0, size=64).astype(np.float32)
    kat["rel"], kat["cos"] = rel, cs
    kat["refract"] = np.stack([strict.refraction_dir(v, n, float(r)) for v, n, r in zip(vecs, nrms, rel)])
    kat["reflectance"] = np.asarray([strict.reflectance(float(c), float(r)) for c, r in zip(cs, rel)], np.float32)
    ctr = rng.uniform(-3, 3, size=(64, 3)).astype(np.float32)
    rad = rng.uniform(0.2, 4, size=64).astype(np.float32)
    org = rng.uniform(-1, 1, size=(64, 3)).astype(np.float32)
    hs = [strict.hit_sphere(o, d, c, float(r)) for o, d, c, r in zip(org, vecs, ctr, rad)]
    kat["hit_org"], kat["hit_ctr"], kat["hit_rad"] = org, ctr, rad
    kat["hit_flag"] = np.asarray([h[0] for h in hs], np.int32)
    kat["hit_t"] = np.asarray([h[1] if h[0] else 0 for h in hs], np.float32)
    np.savez_compressed(os.path.join(out, "kat.npz"), **kat)
    print("kat.npz written (%d arrays)" % len(kat))


if __name__ == "__main__":
    main()
