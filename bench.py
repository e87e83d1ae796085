#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on N MI355X GPUs.

Workload (BASELINE.json configs[1]): the frame-0 scene of the reference's scripts/main.lua,
1920x1080, 256 spp, max depth 50, seeds = the first 256 values of glibc rand() — exactly what the
reference's render() would run (src/c/renderer.c:153-262).  One step = one whole render of that
image: W*H*spp samples.  Scene, seeds and accumulator are resident on the GPU when the timed
region starts; the per-step host->device traffic is the 2 KB seed table.

What is timed is the build the drop-in runs by default (renderer.c): the reference kernel's own
arithmetic as its OpenCL build for gfx950 performs it (pt_render_args.fast_math = 1).
After the timed region EVERY pixel of the last frame is compared, in bits, with the frame the
reference kernel itself (oracle/_ref/ptref_gfx950.hsaco through tests/gpuref.py, the checker) renders
on this GPU from the same scene and seeds.  The strict-IEEE build (bit-identical to the CPU oracle,
rows checked against it) is measured beside it, untimed, as `strict_build`.

N = 1 : python bench.py [--steps K --warmup W]
N > 1 : python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
        one process per GPU; the SAME image is cut into 8-row bands dealt round-robin to the ranks
        (work-item ids stay global, so every pixel is bit-identical to the 1-GPU image); each rank
        renders its bands and THE PRODUCT'S combine (pt_combine_bands_rank, include/pt_hip.h — what
        renderer.c runs as pt_combine_bands) brings them to rank 0 inside the timed region: every rank
        packs its bands into ONE RCCL message over xGMI, rank 0 unpacks; bytes moved per step = the
        image x (N-1)/N, nothing is zero-filled.  torch.distributed only carries the communicator id,
        the barrier and the max-over-ranks timing.  If the communicator cannot be set up on every rank
        the run falls back to zero-filled frames + one torch reduce(sum) per step and says so in
        config.combine.  Fixed total work: "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 256, 50
BAND_ROWS = 8
PEAK_FP32_VALU_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
MBAR_DEFAULT_SCENE = 30.0          # mean material FLOPs per bounce, default scene


def flops_per_sample(bounces_per_sample, n_spheres, mbar=MBAR_DEFAULT_SCENE):
    """FLOPs/sample = B*(18*N + 34 + Mbar) + 60 (algorithmic, reference arithmetic)."""
    return bounces_per_sample * (18.0 * n_spheres + 34.0 + mbar) + 60.0


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(scene, seeds):
    """The reference kernel (x86 build, oracle/_ref — built in the build container from the reference's sources and
    shipped as a .so like our own libraries) or, where that is absent, the oracle port, timed on the
    host cores this job may use, on a bounded sample of the same workload: every pixel of the 1080p image at the
    first 4 of its 256 samples (8.3 M samples)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    host_cpus = os.cpu_count() or 1
    # the GPU box exports the job's CPU share (16 per GPU) as OMP_NUM_THREADS; os.cpu_count() is the whole machine
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = min(host_cpus, share) if share > 0 else host_cpus
    spp = 4
    if oracle_lib.have_ref("fast"):
        lib, kind = oracle_lib.load_ref("fast"), "reference"
    else:
        lib, kind = oracle_lib.load_oracle(), "port"
    t0 = time.perf_counter()
    lib.render(scene, WIDTH, HEIGHT, DEPTH, seeds[:spp], nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": WIDTH * HEIGHT * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": kind,
            "host_cpus": host_cpus, "cpu_model": _cpu_model(),
            "sample": "all %dx%d pixels at the first %d of %d spp (%.1f M samples, %.1f s) on %d threads (the job's CPU share; "
                      "the machine has %d)" % (WIDTH, HEIGHT, spp, SPP, WIDTH * HEIGHT * spp / 1e6, dt, cores, host_cpus)}


VERIFY_ROWS = (0, 1, 269, 540, 777, 1079)


def verify_frame(frame_rows, scene, seeds, rows):
    """Rows of the frame the timed loop produced against the CPU oracle at the FULL sample count, bit for bit.
    Checker only: runs after the timed region.  Raises SystemExit on a mismatch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle = oracle_lib.load_oracle()
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    for r, got in zip(rows, frame_rows):
        want = oracle.render(scene, WIDTH, HEIGHT, DEPTH, seeds, id_begin=r * WIDTH, id_end=(r + 1) * WIDTH, nthreads=threads)[r, :, :3]
        if not np.array_equal(np.ascontiguousarray(got[:, :3]).view(np.uint32), np.ascontiguousarray(want).view(np.uint32)):
            bad = int((np.ascontiguousarray(got[:, :3]).view(np.uint32) != np.ascontiguousarray(want).view(np.uint32)).any(axis=1).sum())
            raise SystemExit("bench.py: row %d of the benchmarked frame differs from the oracle in %d pixels (max abs %.3g)"
                             % (r, bad, float(np.abs(got[:, :3] - want).max())))
    return len(rows)


def verify_against_reference_kernel(frame, scene, seeds, W, H, depth):
    """Every pixel of `frame` ([H, W, 4] numpy) against the reference kernel's own render of the same scene and seeds on
    this GPU, bit for bit.  Checker only, after the timed region.  Returns a description, or None when the code object
    is not in the tree (it is built only where /root/reference exists)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpuref
    if not gpuref.available():
        return None
    k = gpuref.ReferenceKernel()
    try:
        want = k.render(scene, W, H, depth, seeds)[..., :3]
    finally:
        k.close()
    got = np.ascontiguousarray(frame[..., :3])
    diff = int((got.view(np.uint32) != np.ascontiguousarray(want).view(np.uint32)).any(axis=-1).sum())
    if diff:
        raise SystemExit("bench.py: %d pixels of the benchmarked frame differ in bits from the reference kernel's frame" % diff)
    return "all %d pixels bit-identical to the reference kernel's own frame on this GPU (%d launches of its gfx950 code object)" % (W * H, len(seeds))


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the newest PMC summary of it in profiles/ (WRITE_SIZE + FETCH_SIZE,
    KiB per dispatch as rocprofv3 reports them; the table is read from LDS and the reads are not wide streaming loads, so no
    x2 correction applies to the 0.05-0.1 MB that are fetched).  Returns (bytes, file) or (None, None)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_lds_fast*.md")))
    for f in reversed(files):
        text = open(f).read()
        w = re.search(r"\|\s*WRITE_SIZE\s*\|\s*([0-9.e+]+)", text)
        r = re.search(r"\|\s*FETCH_SIZE\s*\|\s*([0-9.e+]+)", text)
        if w and r:
            return (float(w.group(1)) + float(r.group(1))) * 1024.0, os.path.relpath(f, ROOT)
    return None, None


def extras(ctx, stream, torch, hip, scenes, seedmod, budget_s=90.0):
    """The other BASELINE configs on this GPU (after the timed loop, rank 0, N = 1), both builds: kernel time from HIP
    events (mean of two warm launches), bounces per sample counted in-kernel, the roofline fraction by the same
    SURVEY.md §8d formula.  Checks, untimed: the reference-exact build's frame of the first two samples against the
    reference kernel's own (every pixel, in bits; live, tests/gpuref.py), and pixels of the strict build's full frame
    against the CPU oracle at the full sample count, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpuref
    import oracle_lib
    from pathtracer_amd import partition
    oracle = oracle_lib.load_oracle()
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    dev = torch.device("cuda", torch.cuda.current_device())
    cases = [
        ("configs[0]: frame-0 scene 256x256 16 spp depth 50", "pt_render_lds_kernel", lambda: scenes.main_lua_frame(0), 256, 256, 16, 50, 1, 0, None),
        ("configs[2]: 10^4 random spheres 1920x1080 64 spp depth 50, linear scan", "pt_render_stream_kernel", lambda: scenes.dense_random(10000), 1920, 1080, 64, 50, 1, 0, (540 * 1920 + 928, 64)),
        ("configs[3]: frame-0 scene 3840x2160 1024 spp, rank 3 of 8 (8-row bands)", "pt_render_lds_kernel", lambda: scenes.main_lua_frame(0), 3840, 2160, 1024, 50, 8, 3, ((24 * 8 + 3) * 8 * 3840 + 1900, 8)),
        ("configs[4]: glass lattice + emitters (39 spheres) 1920x1080 64 spp depth 32", "pt_render_lds_kernel (scan loop)", scenes.glass_lattice, 1920, 1080, 64, 32, 1, 0, (500 * 1920 + 900, 64)),
    ]
    refk = gpuref.ReferenceKernel() if gpuref.available() else None
    out, t_begin = [], time.perf_counter()
    for label, kernel, factory, W, H, spp, depth, world, rank, probe in cases:
        if time.perf_counter() - t_begin > budget_s:
            out.append({"workload": label, "skipped": "time budget of the extras spent"})
            continue
        sc = factory()
        n = sc["spheres"].shape[0]
        s = seedmod.glibc_rand_stream(spp)
        ctx.scene_upload(sc["spheres"])
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        kw = dict(rank=rank, world=world, band_rows=BAND_ROWS, d_image=frame.data_ptr(), stream=stream, async_=True)
        sync_kw = {k: v for k, v in kw.items() if k != "async_"}
        bounces = ctx.render(sc, W, H, depth, s, count_bounces=True, fast_math=True, **sync_kw)
        for fast in (True, False):
            for _ in range(3):
                ctx.render(sc, W, H, depth, s, record_timing=True, fast_math=fast, **kw)
        torch.cuda.synchronize()
        kt = ctx.kernel_times()
        ms, strict_ms = float(np.mean(kt[1:3])), float(np.mean(kt[4:6]))
        samples = partition.local_pixel_count(W, H, rank, world, BAND_ROWS) * spp
        bps = bounces / samples
        fps = flops_per_sample(bps, n)
        # the strict build's frame (left by the last timed launch) against the oracle
        b, cnt = probe if probe else (0, W * H)
        want = oracle.render(sc, W, H, depth, s, id_begin=b, id_end=b + cnt, nthreads=threads).reshape(-1, 4)[b:b + cnt, :3]
        got = frame.reshape(-1, 4)[b:b + cnt, :3].cpu().numpy()
        if not np.array_equal(np.ascontiguousarray(got).view(np.uint32), np.ascontiguousarray(want).view(np.uint32)):
            raise SystemExit("bench.py extras: %s: pixels %d..%d of the strict build differ from the oracle" % (label, b, b + cnt))
        # the reference-exact build against the reference kernel: the first two samples of every pixel this rank owns
        checked_ref = "reference code object not in the tree"
        if refk is not None:
            k2 = min(spp, 2)
            frame.zero_()
            ctx.render(sc, W, H, depth, s[:k2], fast_math=True, **sync_kw)
            mine = frame[..., :3].cpu().numpy()
            ref = refk.render(sc, W, H, depth, s[:k2])[..., :3]
            rows = partition.rows_of_rank(H, rank, world, BAND_ROWS)
            diff = int((np.ascontiguousarray(mine[rows]).view(np.uint32) != np.ascontiguousarray(ref[rows]).view(np.uint32)).any(axis=-1).sum())
            if diff:
                raise SystemExit("bench.py extras: %s: %d pixels differ in bits from the reference kernel's frame" % (label, diff))
            checked_ref = "all %d pixels of this rank bit-identical to the reference kernel's frame at the first %d samples" % (len(rows) * W, k2)
        e = {"workload": label, "spheres": int(n), "kernel": kernel,
             "kernel_ms": ms, "msamples_s": samples / ms / 1e3, "bounces_per_sample": round(bps, 3),
             "frac": fps * samples / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
             "checked": checked_ref,
             "strict_build_kernel_ms": strict_ms, "strict_build_frac": fps * samples / (strict_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
             "strict_build_checked": "pixels %d..%d bit-identical to the CPU oracle at %d spp" % (b, b + cnt - 1, spp)}
        if n >= 1000:
            e["gtests_s"] = samples * bps * n / (ms * 1e-3) / 1e9      # ray-sphere tests per second
        out.append(e)
        del frame
    if refk is not None:
        refk.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the benchmarked frame (about 3 s of CPU)")
    ap.add_argument("--no-strict", "--no-fast-math", dest="no_strict", action="store_true",
                    help="skip the extra, untimed measurement of the strict-IEEE build")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed measurements of the other BASELINE configs (about 20 s)")
    ap.add_argument("--combine", choices=("product", "torch"), default="product",
                    help="N > 1: the product's packed RCCL combine (default) or the zero-fill + torch reduce fallback")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: run the multi-rank code path (process group, zero-fill, RCCL reduce, barrier) even with one rank")
    ap.add_argument("--spp", type=int, default=SPP, help="debug only: any value but %d is not the benchmark" % SPP)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pathtracer_amd import hip, scenes, seeds as seedmod

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible — this benchmark has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", rank=rank, world_size=world)

    spp = args.spp
    scene = scenes.main_lua_frame(0)
    seeds = seedmod.glibc_rand_stream(spp)
    n_spheres = scene["spheres"].shape[0]

    ctx = hip.Context(local_rank)
    ctx.scene_upload(scene["spheres"])
    frame = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
    # One explicit (non-null) HIP stream carries everything: the render kernel (launched by the C ABI
    # on this handle), the zero-fill, the RCCL reduce and the timing events.  torch's default stream
    # has handle 0, which the C ABI reads as "use the context's own stream".
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0

    # N > 1: the product's combine needs one RCCL communicator over the ranks' contexts; torch.distributed carries its id
    packed = False
    if multi:
        ok = 0
        if args.combine == "product":
            try:
                box = [hip.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ctx.comm_init_rank(world, rank, box[0])
                ok = 1
            except Exception as e:      # library, symbol or communicator trouble on this rank: every rank falls back together
                print("bench.py: rank %d: product combine unavailable (%s); falling back to torch reduce" % (rank, e), file=sys.stderr, flush=True)
        agree = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        packed = bool(int(agree[0]) == 1)

    def combine(buf):
        if packed:
            ctx.combine_bands_rank(0, BAND_ROWS, d_image=buf.data_ptr(), width=WIDTH, height=HEIGHT, stream=stream)
            return None
        return dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, async_op=True)

    def step(count=False):
        if multi and not packed:
            frame.zero_()
        bc = ctx.render(scene, WIDTH, HEIGHT, DEPTH, seeds, rank=rank, world=world, band_rows=BAND_ROWS,
                        d_image=frame.data_ptr(), stream=stream, count_bounces=count, async_=True, fast_math=True)
        if multi:
            w = combine(frame)
            if w is not None:
                w.wait()
        return bc

    # untimed: bounce-loop iterations of this rank's share (deterministic given the seeds) for the FLOP count
    my_bounces = step(count=True)
    for _ in range(args.warmup):
        step()

    def barrier():
        if multi:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    if args.steps > 256:
        raise SystemExit("bench.py: at most 256 timed steps (size of the C ABI's kernel-timing ring)")
    barrier()
    t0 = time.perf_counter()
    # N > 1, product combine: render -> pack -> one send per rank, all enqueued on this rank's stream; rank 0 receives and
    # unpacks on its own.  Fallback: two zero-filled frame buffers, so that the torch reduce of step k overlaps the render
    # of step k+1 (a buffer is zeroed again only after the reduce that read it has been waited for)
    frames = [frame, torch.zeros_like(frame)] if (multi and not packed) else [frame]
    pending = [None, None]
    for k in range(args.steps):
        buf = frames[k % len(frames)]
        if multi and not packed:
            if pending[k % 2] is not None:
                pending[k % 2].wait()
            buf.zero_()
        # record_timing: the C ABI brackets the kernel with HIP events on the launch stream, no host sync
        ctx.render(scene, WIDTH, HEIGHT, DEPTH, seeds, rank=rank, world=world, band_rows=BAND_ROWS,
                   d_image=buf.data_ptr(), stream=stream, async_=True, record_timing=True, fast_math=True)
        if multi:
            pending[k % 2] = combine(buf)
    for w in pending:
        if w is not None:
            w.wait()
    if multi:
        frame = frames[(args.steps - 1) % len(frames)]       # the last step's combined image, checked below
    barrier()
    dt = time.perf_counter() - t0

    kern_times = ctx.kernel_times()
    assert len(kern_times) == args.steps, (len(kern_times), args.steps)
    kern_ms = float(np.mean(kern_times))

    # outside the timed region: the frame the last timed step left (rank 0 holds the whole image, after the combine when
    # N > 1) against the reference kernel's own frame, every pixel, bit for bit
    verified = None
    if rank == 0 and not args.no_verify:
        torch.cuda.synchronize()
        verified = verify_against_reference_kernel(frame.cpu().numpy(), scene, seeds, WIDTH, HEIGHT, DEPTH)

    # untimed extra (rank-local, N=1 only): the strict-IEEE build of the same kernel (bit-identical to the CPU oracle:
    # six rows of its frame are checked against it at the full sample count)
    strict_ms, strict_rows = None, 0
    if world == 1 and not multi and not args.no_strict:
        for _ in range(1 + min(args.steps, 3)):
            ctx.render(scene, WIDTH, HEIGHT, DEPTH, seeds, d_image=frame.data_ptr(), stream=stream, async_=True,
                       record_timing=True, fast_math=False)
        torch.cuda.synchronize()
        strict_ms = float(np.mean(ctx.kernel_times()[1:]))
        if not args.no_verify:
            rows = [frame[r].cpu().numpy() for r in VERIFY_ROWS]
            strict_rows = verify_frame(rows, scene, seeds, VERIFY_ROWS)
    tb = torch.tensor([float(my_bounces), dt, kern_ms], dtype=torch.float64, device=dev)
    if multi:
        tsum = tb.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        tmax = tb.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_bounces, dt, kern_ms_max = float(tsum[0]), float(tmax[1]), float(tmax[2])
    else:
        total_bounces, kern_ms_max = float(my_bounces), kern_ms

    if rank == 0:
        traffic, traffic_file = measured_traffic() if world == 1 else (None, None)
        samples = WIDTH * HEIGHT * spp
        bps = total_bounces / samples
        fps = flops_per_sample(bps, n_spheres)
        # dominant kernel = pt_render_kernel; this rank's launch processes its share of the samples
        my_samples = samples if world == 1 else float(my_bounces) / bps
        achieved = fps * my_samples / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "Msamples/sec at 1080p",
            "value": samples * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "main.lua frame-0 scene 1920x1080 %d spp depth 50 (BASELINE configs[1])" % spp,
                       "spheres": n_spheres, "seeds": "glibc rand() stream, first %d values" % spp,
                       "partition": "1 GPU" if world == 1 else "%d-row bands round-robin over %d ranks" % (BAND_ROWS, world),
                       "combine": "none (1 GPU)" if not multi else
                                  ("pt_combine_bands_rank: one packed RCCL message per rank to rank 0, %.1f MB per step in %d messages"
                                   % (16.0 * WIDTH * HEIGHT * (world - 1) / world / 1e6, world - 1)) if packed else
                                  "fallback: zero-filled full frames + torch.distributed reduce(sum), %.1f MB per rank per step" % (16.0 * WIDTH * HEIGHT / 1e6),
                       "bounces_per_sample": round(bps, 3),
                       "arithmetic": "the reference kernel's own: its float operations as its OpenCL build for gfx950 performs them "
                                     "(pt_render_args.fast_math = 1, the default of the renderer.h drop-in; DESIGN.md section 2)",
                       "verified_pixels": WIDTH * HEIGHT if verified else 0,
                       "verified": verified or ("skipped" if args.no_verify else "reference code object not in the tree: not checked in this run")},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         # HBM bytes per launch: the counter reading of the newest PMC summary of this kernel in profiles/
                         # (same frame: the accumulator write does not depend on spp), not a formula; the algorithmic
                         # bytes (16 B/pixel written once + the tables and seeds read) are given beside it
                         "traffic": traffic, "traffic_source": traffic_file,
                         "traffic_model": 16.0 * WIDTH * HEIGHT / max(world, 1) + 432 + 8 * spp,
                         "kernel": "pt_fast::pt_render_lds_kernel", "kernel_ms": kern_ms,
                         "flops_per_sample": round(fps, 1),
                         "note": "FP32 vector roofline (no dense contraction on this path, SURVEY.md §8d); "
                                 "algorithmic FLOPs = B*(18N+34+30)+60 per sample with B measured in-kernel"},
        }
        if strict_ms is not None:
            out["strict_build"] = {"value": samples / (strict_ms * 1e-3) / 1e6, "unit": "Msamples/s", "kernel_ms": strict_ms,
                                   "frac": fps * samples / (strict_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                                   "verified_rows": strict_rows,
                                   "note": "the strict-IEEE build of the same kernel (pt_render_args.fast_math = 0 / PT_MATH=strict): IEEE binary32 "
                                           "in the reference's source order, no contraction, correctly rounded sqrt and divide; bit-identical to "
                                           "the CPU oracle (rows %s of its frame checked at %d spp), within the stated tolerance of the reference "
                                           "kernel's frame; kernel time only, not the reported value" % (list(VERIFY_ROWS), spp)}
        if world == 1 and not multi and not args.no_extras and spp == SPP:
            out["extras"] = extras(ctx, stream, torch, hip, scenes, seedmod)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, seedmod.glibc_rand_stream(SPP))
        print(json.dumps(out), flush=True)
    if multi and rank == 0:
        # the reduced frame on rank 0 is the whole image: the counting launch left every pixel's bounce count in
        # the 4th lane only in the untimed step, so check the colour lanes are populated in every band
        rows = frame[::BAND_ROWS, :, :3].abs().sum(dim=(1, 2))
        assert bool((rows > 0).all()), "bench.py: a row band is empty after the reduce"
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
