#!/usr/bin/env python3
"""Freeze the output of the REFERENCE kernel on MI355X (oracle/_ref/ptref_gfx950.hsaco, tests/gpuref.py) and measure
how far the HIP builds and the x86 builds of the reference are from it.  Runs ON THE GPU BOX:

    python3 tools/make_gpuref_golden.py            # writes gpurun_out/gpuref/gpuref_<case>.npz + stats.json + stats.md

The .npz files are then copied to tests/golden/ (data: the reference's own outputs for the inputs already stored in
tests/golden/<case>.npz).  TEST INFRASTRUCTURE.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import envelope  # noqa: E402
import gpuref  # noqa: E402
import oracle_lib  # noqa: E402
from cases import CASES  # noqa: E402
from pathtracer_amd import hip  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def distance(a, b, a_u8, b_u8):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max(axis=-1)
    ma, mb = float(a.mean(dtype=np.float64)), float(b.mean(dtype=np.float64))
    return dict(share=float((d <= 1e-4).mean()), off=int((d > 1e-4).sum()), max_abs=float(d.max()),
                identical=float((a.view(np.uint32) == b.view(np.uint32)).all(axis=-1).mean()),
                psnr=envelope.psnr_u8(a_u8, b_u8), mean_rel=abs(ma - mb) / max(mb, 1e-12))


def main():
    out = os.path.join(ROOT, "gpurun_out", "gpuref")
    os.makedirs(out, exist_ok=True)
    orc = oracle_lib.load_oracle()
    ref = gpuref.ReferenceKernel(gpuref.HSACO)
    ref_nf = gpuref.ReferenceKernel(gpuref.HSACO_STRICT) if gpuref.available(gpuref.HSACO_STRICT) else None
    ctx = hip.Context(0)
    u8 = lambda rgb: orc.postprocess(np.concatenate([rgb, np.zeros(rgb.shape[:2] + (1,), np.float32)], axis=2)).reshape(rgb.shape)
    stats = {}
    names = list(CASES) + list(envelope.ENVELOPE_CASES)
    only = set(sys.argv[1:])
    for name in names:
        if only and name not in only:
            continue
        g = np.load(os.path.join(GOLD, name + ".npz"))
        W, H, depth, spp = (int(x) for x in g["size"])
        sc = {"spheres": g["spheres"], "bg": g["bg"], "cam": g["cam"]}
        r = ref.render(sc, W, H, depth, g["seeds"])[..., :3].copy()
        r2 = ref.render(sc, W, H, depth, g["seeds"], poison=False)[..., :3].copy()   # second run on a dirty buffer
        rerun_equal = bool(np.array_equal(r.view(np.uint32), r2.view(np.uint32)))
        keep = dict(rgb=r)
        if ref_nf is not None:
            keep["rgb_noflags"] = ref_nf.render(sc, W, H, depth, g["seeds"])[..., :3].copy()
        np.savez_compressed(os.path.join(out, "gpuref_%s.npz" % name), **keep)
        r_u8 = u8(r)
        st = dict(size=[W, H, depth, spp], n_spheres=int(sc["spheres"].shape[0]), rerun_equal=rerun_equal,
                  finite=bool(np.isfinite(r).all()), mean=float(r.mean(dtype=np.float64)))
        img_s, bounces = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], count_bounces=True)
        hs_u8 = ctx.image_read_u8()
        img_f = hip.render_scene(ctx, sc, W, H, depth, g["seeds"], fast_math=True)
        hf_u8 = ctx.image_read_u8()
        st["bounces"] = int(bounces)
        st["budget"] = envelope.off_pixel_budget(bounces, sc["spheres"].shape[0])
        st["hip_strict"] = distance(img_s[..., :3], r, hs_u8, r_u8)
        st["hip_fast"] = distance(img_f[..., :3], r, hf_u8, r_u8)
        for k in envelope.ENVELOPE_KINDS:
            if k in g.files:
                st["x86_" + k] = distance(g[k], r, u8(np.ascontiguousarray(g[k])), r_u8)
        if "rgb_noflags" in keep:
            st["gpu_noflags"] = distance(keep["rgb_noflags"], r, u8(keep["rgb_noflags"]), r_u8)
        stats[name] = st
        print("%-24s %4dx%-4d spp %2d N %5d | strict: share %.5f off %4d psnr %6.2f | fast: share %.5f off %4d psnr %6.2f | budget %d rerun_equal %s"
              % (name, W, H, spp, st["n_spheres"], st["hip_strict"]["share"], st["hip_strict"]["off"], st["hip_strict"]["psnr"],
                 st["hip_fast"]["share"], st["hip_fast"]["off"], st["hip_fast"]["psnr"], st["budget"], rerun_equal), flush=True)
    json.dump(stats, open(os.path.join(out, "stats.json"), "w"), indent=1)
    with open(os.path.join(out, "stats.md"), "w") as f:
        f.write("# Distance to the reference kernel on MI355X (oracle/_ref/ptref_gfx950.hsaco), per case\n\n")
        f.write("share = pixels with all channels within 1e-4 (linear float image); off = pixels farther; psnr = 8-bit image\n\n")
        cols = ["hip_strict", "hip_fast", "x86_strict", "x86_fast", "x86_env1", "x86_env2", "x86_env3", "gpu_noflags"]
        f.write("| case | size | N | " + " | ".join("%s share / off / max / psnr" % c for c in cols) + " | budget |\n")
        f.write("|---|---|---|" + "---|" * len(cols) + "---|\n")
        for name, st in stats.items():
            cells = []
            for c in cols:
                d = st.get(c)
                cells.append("—" if d is None else "%.5f / %d / %.3g / %.2f" % (d["share"], d["off"], d["max_abs"], d["psnr"]))
            f.write("| %s | %dx%d d%d s%d | %d | %s | %d |\n" % (name, st["size"][0], st["size"][1], st["size"][2], st["size"][3],
                                                               st["n_spheres"], " | ".join(cells), st["budget"]))
    ref.close()
    ctx.close()


if __name__ == "__main__":
    main()
