/* pt_hip.h — C ABI of libpt_hip.so, the MI355X (gfx950) path-tracing backend.
 *
 * This is the inner boundary of the drop-in: the thin C host (pathtracer_amd/csrc/renderer.c,
 * which keeps the reference's 15-function renderer.h API) calls these entry points
 * where the reference's host called OpenCL.  Plain C: pointers, sizes, ints.  Every
 * function returns 0 on success or a negative PT_E_* code; pt_last_error() gives the text.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference checkout).
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK            0
#define PT_E_INVALID    -1   /* bad argument (NULL, non-positive size, partition out of range) */
#define PT_E_DEVICE     -2   /* HIP runtime error (no device, launch failure, ...) */
#define PT_E_NOMEM      -3   /* device or host allocation failed */
#define PT_E_STATE      -4   /* call order: no scene / no image set */

/* Interchange sphere record, 12 floats (48 B), used by every layer of this repo:
 *   cx cy cz radius | type colr colg colb | tint fuzzyness refIdx pad
 * `type` holds the reference's material code as a float: 0 light, 1 lambertian,
 * 2 metal, 3 dielectric (src/c/renderer.c:119-133).  It is the flattened form of the
 * reference's 80-byte `Sphere` (src/c/renderer.h:32-44); the backend re-lays it out as
 * structure-of-arrays device tables. */
#define PT_SPHERE_FLOATS 12

typedef struct pt_ctx pt_ctx;

/* Arguments of one render call = the reference's per-sample launch loop
 * (src/c/renderer.c:159-179: kernel args 1..3, 5, 6) folded into ONE launch. */
typedef struct pt_render_args {
	int width, height;        /* ImageInfo.size          (src/c/renderer.h:22-25) */
	int max_depth;            /* ImageInfo.maxRayDepth */
	float background[3];      /* SceneInfo.color         (src/c/renderer.h:27-30) */
	float cam_pos[3];         /* Camera                  (src/c/renderer.h:46-53) */
	float cam_rot[3];
	float sensor_width, focal_length, aperture, exposure;
	const uint64_t *seeds;    /* HOST array, n_frames entries: seeds[k] is the `seed` kernel arg of
	                             sample first_frame+k (src/c/renderer.c:172,176) */
	int first_frame;          /* `frame` kernel arg of the first sample (src/c/renderer.c:175);
	                             0 overwrites the accumulator, >0 continues its running mean */
	int n_frames;             /* samples to run in this call */
	/* Row-band partition for multi-GPU (no reference counterpart): image rows are
	 * cut into bands of band_rows rows, band b is rendered by rank b % world.  Work-item ids stay
	 * global, so every pixel is bit-identical to the single-GPU image.  Pixels of other ranks are
	 * not touched.  rank=0, world=1 renders everything. */
	int rank, world, band_rows;
	void *d_image;            /* device pointer to width*height float4 accumulators (the reference's
	                             `global float3 *image`, 16 B per pixel, src/cl/main.cl:4), or NULL to
	                             use the context's own buffer (pt_image_alloc) */
	void *stream;             /* hipStream_t to launch on, or NULL for the context's stream */
	uint64_t *bounce_count;   /* optional HOST out: bounce-loop iterations executed (src/cl/ray.cl:96)
	                             over all samples of this call — the B of SURVEY.md §8d.  Forces a
	                             synchronisation.  NULL: not counted. */
	int async;                /* 0: return after the kernel has finished (the reference's clFinish,
	                             src/c/renderer.c:251).  1: return after enqueue. */
	float *kernel_ms;         /* optional HOST out: duration of the render kernel in ms, from HIP events
	                             recorded on the launch stream around the launch.  Forces a
	                             synchronisation.  NULL: not timed. */
	int record_timing;        /* 1: bracket the render kernel with a pair of HIP events on the launch stream
	                             WITHOUT synchronising; collect the durations later with pt_kernel_times().
	                             At most 256 launches between two collections.  Ignored when kernel_ms is set. */
	int fast_math;            /* Which arithmetic.  1: THE REFERENCE'S OWN — the float operations of its kernel as its OpenCL
	                             build for this GPU performs them (the reference builds with -cl-mad-enable
	                             -cl-fast-relaxed-math, src/c/renderer.c:13; ROCm's compiler then fuses products into
	                             sums, uses v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 bare and AMD's length / normalize / sin /
	                             cos; every such choice is written out in pathtracer_amd/csrc/pt_device.h): the image
	                             equals the reference kernel's output on MI355X BIT FOR BIT.  This is what the renderer.h
	                             drop-in passes.  0: strict IEEE binary32 in the reference's SOURCE order, no fusion,
	                             correctly rounded sqrt and divide — bit-identical to the CPU oracle, within the stated
	                             tolerance of the reference kernel's image.  Same integer RNG stream either way.  The
	                             environment variable PT_MATH=fast|strict (read by pt_ctx_create) overrides this field. */
	int accel;                /* 0 (default): the reference's linear scan over all spheres (src/cl/ray.cl:101-110).
	                             1: opt-in extension with no reference counterpart — the same closest hit (same
	                             arithmetic per tested sphere, lowest index on equal distance) found through a
	                             uniform grid built from the uploaded scene; meant for 10^4-10^5 sphere tables.
	                             The environment variable PT_ACCEL=grid|none overrides this field. */
} pt_render_args;

/* Text of the last error on this thread ("" if none). */
const char *pt_last_error(void);

/* Number of visible HIP devices.  Replaces clGetPlatformIDs/clGetDeviceIDs (src/c/renderer.c:68,72). */
int pt_device_count(int *count);

/* Create a backend context on `device`: selects the device, creates its stream.  Replaces
 * clCreateContext / clCreateCommandQueueWithProperties / clCreateProgramWithSource /
 * clBuildProgram / clCreateKernel (src/c/renderer.c:75-94) — the kernels are compiled ahead
 * of time into this library, so there is no run-time build step that can fail. */
int pt_ctx_create(int device, pt_ctx **out);

/* Destroy the context and free its device memory.  Replaces clRelease* (src/c/renderer.c:295-303). */
int pt_ctx_destroy(pt_ctx *ctx);

/* The PT_* environment switches (PT_MATH, PT_ACCEL, PT_KERNEL, PT_GENERIC_SCAN, PT_NO_TRIO_SKIP, PT_LANES_PER_PIXEL,
 * PT_TAIL_PCT, PT_TAIL_LANES, PT_COMBINE: opt-ins and debugging aids) are read ONCE, by pt_ctx_create; no render call
 * reads the environment.  This re-reads them for a context whose process changed them afterwards (the tests do).
 * No reference counterpart (the reference has no run-time switches). */
int pt_ctx_reload_env(pt_ctx *ctx);

/* (Re)allocate the context's own accumulator, width*height float4, zero-filled.
 * Replaces clCreateBuffer(imageBuff) in set_image_properties (src/c/renderer.c:103). */
int pt_image_alloc(pt_ctx *ctx, int width, int height);

/* Upload the scene: n interchange records (HOST) -> SoA device tables.  n may be 0.
 * Replaces clCreateBuffer(sphereBuff) + clEnqueueWriteBuffer (src/c/renderer.c:155-156). */
int pt_scene_upload(pt_ctx *ctx, const float *spheres, int n);

/* Render args->n_frames samples of every pixel this rank owns.  Replaces the sample loop
 * clSetKernelArg ×7 + clEnqueueNDRangeKernel + clFinish (src/c/renderer.c:159-179,251). */
int pt_render(pt_ctx *ctx, const pt_render_args *args);

/* Launch geometry pt_render would use for the lock-step kernel on these arguments (width, height, first_frame,
 * n_frames, rank, world, band_rows; the rest is ignored): lanes per pixel of the two tiers, the tier boundary, the total
 * thread count (kept below 2^31 — the kernels index threads with a 32-bit int — by giving up the tail tier and then
 * lanes per pixel on frames above ~3.7e8 pixels per rank) and the number of 256-thread blocks.  Host arithmetic only:
 * needs no GPU.  No reference counterpart (its NDRange is W*H work-items, src/c/renderer.c:168); for the tests. */
typedef struct pt_launch_plan {
	long long local_pixels;       /* pixels of this rank */
	int lanes_log2;               /* first tier: 2^lanes_log2 lanes per pixel */
	int tail_lanes_log2;          /* second tier (the pixels dispatched last) */
	long long tier_pixels;        /* pixels in the first tier */
	long long tier_threads;       /* threads of the first tier = tier_pixels << lanes_log2 */
	long long total_threads;      /* < 2^31 */
	unsigned blocks;
} pt_launch_plan;
int pt_plan_launch(const pt_render_args *args, pt_launch_plan *out);

/* pt_render for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its host allocates a sphere buffer per render(),
 * src/c/renderer.c:155).  Same arguments, same kernels, same image as pt_render, but the call makes no hipMalloc,
 * hipFree or host allocation: the working memory pt_render keeps in buffers of the context (the per-sample seed
 * table, the streaming kernel's pixel-queue head, the bounce counter) lives in a file-scope device array of
 * 512 + 8 * PT_NOALLOC_MAX_FRAMES bytes (33 280 B) that every GPU the library runs on holds until the process
 * ends; the seeds reach it as kernel arguments.  The caller still passes the scene (pt_scene_upload, beforehand)
 * and the accumulator (args->d_image, or the context's own from pt_image_alloc).
 * Limits, rejected with PT_E_INVALID and a message: n_frames <= PT_NOALLOC_MAX_FRAMES per call (continue with
 * first_frame for more); accel must be 0 (building the grid allocates; PT_ACCEL is ignored here).
 * All contexts on one GPU share that GPU's array, so calls on one GPU must not overlap in time.  Calls on one stream
 * are ordered and safe (each call re-initialises what it needs: queue head and counter are cleared, the seed table is
 * rewritten); a call made while an earlier one is still running on ANOTHER stream of the same GPU is refused with
 * PT_E_STATE (one event per GPU tracks the last call; calls that replay from a captured graph are not tracked — keep
 * such graphs on one stream).
 * Under stream capture (hipStreamBeginCapture on args->stream) only the enqueue-and-return form is legal: async must be
 * 1 and bounce_count, kernel_ms and record_timing unset — each of those synchronises or records timing events — else
 * PT_E_INVALID.  What is captured: the seed staging kernel(s), one 256-byte memset for the streaming kernel, the
 * render kernel.  The seeds are baked into the graph as kernel arguments. */
#define PT_NOALLOC_MAX_FRAMES 4096
int pt_render_noalloc(pt_ctx *ctx, const pt_render_args *args);

/* Multi-GPU combine, for one process driving several GPUs (pathtracer_amd/csrc/renderer.c with PT_NUM_GPUS > 1).
 * NO COUNTERPART IN THE REFERENCE (single device, src/c/renderer.c:72); SURVEY.md §8e.  ctxs[i] is rank i of world n:
 * each has rendered its own row bands (pt_render with rank = i, world = n, band_rows) into its context accumulator
 * (pt_image_alloc, same size everywhere).  On return the work is ENQUEUED that brings every other rank's bands into
 * ctxs[root]'s accumulator, each band once, as ONE message per peer: the peer packs its bands into a compact staging
 * buffer (in the order the render kernel numbers its own pixels) on its stream, after its render; one RCCL
 * ncclSend / ncclRecv pair per peer, all in one group, moves it over that peer's xGMI link; the root unpacks on its
 * stream — pt_sync(ctxs[root]) or pt_image_read*(ctxs[root], ...) waits for all of it.  Nothing needs to be zero
 * anywhere; bytes moved = the image minus the root's own bands, in n - 1 messages.  librccl is loaded on first use;
 * communicators are created once per device list and kept until the process ends.  If RCCL cannot be set up (library,
 * symbol or communicator creation fails) the same packed buffers travel as peer copies ordered by events, with a line
 * on stderr (PT_COMBINE=copy asks for that, PT_COMBINE=rccl forbids the fallback).  n <= 8; the ranks' GPUs must be
 * pairwise distinct — or all the same one (a debugging arrangement: several contexts on one GPU, which cannot be RCCL
 * peers), in which case the messages are device-to-device copies. */
int pt_combine_bands(pt_ctx *const *ctxs, int n, int root, int band_rows);

/* Which transport the last pt_combine_bands / pt_combine_bands_rank call on this thread used:
 * "rccl", "peer-copy", "device-copy", or "none" (one rank).  No reference counterpart. */
const char *pt_combine_transport(void);

/* The same combine for one process PER GPU (bench.py --gpus N under torch.distributed.run; an MPI-style launcher).
 * NO COUNTERPART IN THE REFERENCE.  Rank 0 obtains an id with pt_comm_unique_id (ncclGetUniqueId; PT_COMM_ID_BYTES
 * bytes) and hands it to the other processes by whatever channel the launcher offers; every rank then calls
 * pt_comm_init_rank (ncclCommInitRank; collective: all ranks must call it) on its context, and after each render
 * pt_combine_bands_rank: a non-root rank packs its bands of d_image (NULL: the context's accumulator) and sends them
 * as one message; the root receives one message per peer (one group) and unpacks them into its d_image.  Everything
 * is enqueued on `stream` (NULL: the context's), ordered after the render that was launched there.  Bytes moved and
 * message count as pt_combine_bands.  The communicator is destroyed with the context. */
#define PT_COMM_ID_BYTES 128
int pt_comm_unique_id(void *id, int bytes);
int pt_comm_init_rank(pt_ctx *ctx, int world, int rank, const void *id);
int pt_combine_bands_rank(pt_ctx *ctx, int root, int band_rows, void *d_image, int width, int height, void *stream);

/* Durations (ms) of the render kernels launched with record_timing since the last call, oldest first:
 * waits for those launches, writes up to `cap` values to ms[], stores the count in *n and empties the
 * ring.  No reference counterpart (the reference times with gettimeofday around its loop,
 * src/c/renderer.c:166,254-261); bench.py uses it for the per-launch kernel time of the roofline figure. */
int pt_kernel_times(pt_ctx *ctx, float *ms, int cap, int *n);

/* Block until everything enqueued on the context's stream has finished (clFinish, src/c/renderer.c:273). */
int pt_sync(pt_ctx *ctx);

/* Copy the context's accumulator to HOST memory, width*height*4 floats.
 * Replaces clEnqueueReadBuffer (src/c/renderer.c:271). */
int pt_image_read(pt_ctx *ctx, float *host_rgba);

/* Device epilogue + copy: gamma (sqrt), clamp to [0,1], (int)(x*255) truncation -> width*height*3
 * bytes RGB, row-major, in HOST memory.  Replaces gamma_correct_data + convert_to_image
 * (src/c/renderer.c:50-56,17-39) and shrinks the read-back from 16 to 3 bytes per pixel.
 * d_image: accumulator to convert (NULL: the context's own, sized by pt_image_alloc). */
int pt_image_read_u8(pt_ctx *ctx, const void *d_image, int width, int height, unsigned char *host_rgb);

/* Self-test hooks used by tests/ to pin the device arithmetic against the oracle, one function
 * of the reference kernel at a time.  All pointers are HOST; n items each.
 *   op 0: init_rng_2(a[i], b[i])                 -> out_u64[i]                (src/cl/rng.cl:7-9)
 *   op 1: state a[i]: rand_float                  -> out_u64[i]=state, out_f[i] (src/cl/rng.cl:11-14)
 *   op 2: state a[i]: random_unit_vector          -> out_u64[i]=state, out_f[3i..] (src/cl/rng.cl:20-34)
 *   op 3: in_f[2i]=x, in_f[2i+1]=y: x/y, sqrt(x), reflectance(x,y), pow5(x) -> out_f[4i..]
 *   op 4: float bit patterns a[i] .. a[i]+b[i]-1, the kernels' short correctly-rounded sqrt forms against sqrtf:
 *         out_u64[i] = #patterns where the fully guarded form differs anywhere
 *         | (#patterns where the +-0/inf-guarded form differs inside its stated domain) << 21
 *         | (#patterns where the unguarded forms differ inside their stated domains) << 42
 *   op 5: same ranges: out_u64[i] = #patterns where the kernels' short reciprocal differs from 1.0f / x
 *   op 6: same ranges (positive normal floats >= 2^-96): out_u64[i] = #patterns where raw v_sqrt_f32 is one ulp
 *         above the correctly rounded root | (#patterns one ulp below) << 32 — a histogram, no pass/fail
 *   op 7: b[i] (num, den) pairs from seed a[i] — the first 16 the corners of the range, the rest pseudo-random with
 *         2^-80 <= num < 2^80, 2^-40 <= den < 2^40: out_u64[i] = #pairs where the scan's short divide differs
 *         from num / den | (#pairs compared) << 32
 *   op 8: as op 7 for the bare num * v_rcp_f32(den) — the canary that the comparison can fail
 *   op 9: pow5 of the float with bit pattern a[i] -> out_f[i]   (src/cl/ray.cl:61 pow(1 - cosTheta, 5))
 *   op 10: as op 7 for the lock-step kernel's form of the divide: den = a = |d|^2 formed from x = dot(d, d) as the
 *         kernel forms it (src/cl/ray.cl:73), the reciprocal seeded with v_rsq_f32(x)^2
 *   op 11: same ranges as op 4: out_u64[i] = #patterns x where normalize()'s 1 / length (one v_rsq_f32 for root and
 *         reciprocal) differs from 1.0f / sqrtf(x)
 *   op 12: out_u64[i] = #significands x of the binade with exponent field b[i] where the running mean's divide by
 *         f1 = a[i] (src/cl/main.cl:29 with frame + 1 = a[i] < 2^24) in its two-FMA form differs from x / f1
 *         (b[i] | 0x100: the same with a reciprocal 256 ulps off, a canary that must report differences) */
int pt_selftest(pt_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, const float *in_f,
                int n, uint64_t *out_u64, float *out_f);

#ifdef __cplusplus
}
#endif
#endif
