// pt_kernels.hip — gfx950 path-tracing kernels and the C ABI of libpt_hip.so (include/pt_hip.h).
//
// Replaces the reference's OpenCL kernel `main` (src/cl/main.cl:4-30, src/cl/ray.cl,
// src/cl/rng.cl) and the device side of its host loop (src/c/renderer.c:153-262).
// Not a translation: one launch renders ALL samples of a pixel with the accumulator in
// registers (the reference re-launches per sample and read-modify-writes 32 B per sample
// per pixel), the scene is structure-of-arrays (geometry table staged in LDS and read as
// wave-uniform broadcasts, material table gathered only for the winning sphere), the RNG is
// 32-bit with Mersenne folding instead of 64-bit remainders, and per-launch uniform
// trigonometry is folded on the host.  The render kernels themselves are in
// pt_render_kernels.h (built here as pt_strict, in pt_kernels_fast.hip as pt_fast).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math
//        -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize  (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <mutex>
#include <vector>

#define PT_NS pt_strict
#define PT_FAST_MATH 0
#include "pt_render_kernels.h"
#include "pt_hip.h"

using namespace pt_strict;

#define PT_TIMING_SLOTS 256

// ---------------------------------------------------------------------------------------
// Epilogue: gamma (sqrt), clamp, (int)(x*255) truncation -> packed RGB8
// (src/c/renderer.c:50-56, 17-39).  sqrtf == (float)sqrt((double)x) for every float.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_to_u8_kernel(const float4 *__restrict__ image, int pixels, unsigned char *__restrict__ out) {
	int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if(i >= pixels) return;
	float4 v = image[i];
	float c[3] = {v.x, v.y, v.z};
#pragma unroll
	for(int k = 0; k < 3; k++) {
		float x = sqrtf(c[k]);
		if(x < 0.0f) x = 0.0f;
		if(x > 1.0f) x = 1.0f;
		out[3 * (size_t)i + k] = (unsigned char)(int)(x * 255.0f);
	}
}

// ---------------------------------------------------------------------------------------
// Self-test kernel (pt_selftest in pt_hip.h)
// ---------------------------------------------------------------------------------------
__global__ void pt_selftest_kernel(int op, const uint64_t *a, const uint64_t *b, const float *inf_, int n, uint64_t *ou, float *of) {
	int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if(i >= n) return;
	if(op == 0) {
		ou[i] = init_rng_2((uint32_t)a[i], 16807ull * b[i]);
	} else if(op == 1) {
		uint32_t s = (uint32_t)a[i];
		of[i] = rand_float(s);
		ou[i] = s;
	} else if(op == 2) {
		uint32_t s = (uint32_t)a[i];
		f3 r = random_unit_vector(s);
		of[3 * i] = r.x; of[3 * i + 1] = r.y; of[3 * i + 2] = r.z;
		ou[i] = s;
	} else if(op == 3) {
		float x = inf_[2 * i], y = inf_[2 * i + 1];
		of[4 * i] = x / y;
		of[4 * i + 1] = sqrtf(x);
		of[4 * i + 2] = reflectance(x, schlick_r0(y));
		of[4 * i + 3] = pow5(x);
	} else if(op == 4) {
		// bit patterns a[i] .. a[i] + b[i] - 1 against sqrtf (NaN == NaN): bits 0-20 #sqrt_cr wrong anywhere; bits 21-41
		// #sqrt_z wrong in its domain (+-0, |x| >= 2^-96, inf, NaN); bits 42-62 #sqrt_pos wrong in its domain
		// (2^-96 <= x < inf, negative below -2^-96, NaN) or #sqrt_01 wrong on [2^-96, 1] and +-0
		uint64_t bad_cr = 0, bad_z = 0, bad_pos = 0;
		for(uint64_t k = 0; k < b[i]; k++) {
			const uint32_t u = (uint32_t)(a[i] + k);
			const float x = __uint_as_float(u);
			const float want = sqrtf(x);
			const bool nanw = want != want;
			auto differs = [&](float g) { return nanw ? (g == g) : (__float_as_uint(g) != __float_as_uint(want)); };
			if(differs(sqrt_cr(x))) bad_cr++;
			const uint32_t mag = u & 0x7fffffffu;
			if(!(mag - 1u < PT_SQRT_TINY_BITS - 1u) && differs(sqrt_z(x))) bad_z++;
			if(mag >= PT_SQRT_TINY_BITS && mag != 0x7f800000u && differs(sqrt_pos(x))) bad_pos++;
			if((u == 0u || u == 0x80000000u || (u >= PT_SQRT_TINY_BITS && u <= 0x3f800000u)) && __float_as_uint(sqrt_01(x)) != __float_as_uint(u == 0x80000000u ? 0.0f : want)) bad_pos++;
		}
		ou[i] = bad_cr + (bad_z << 21) + (bad_pos << 42);
	} else if(op == 5) {
		// bit patterns a[i] .. a[i] + b[i] - 1: how many have rcp_cr(x) != 1.0f / x (NaN == NaN)
		uint64_t bad = 0;
		for(uint64_t k = 0; k < b[i]; k++) {
			const float x = __uint_as_float((uint32_t)(a[i] + k));
			const float want = 1.0f / x, got = rcp_cr(x);
			if(want != want ? (got == got) : (__float_as_uint(got) != __float_as_uint(want))) bad++;
		}
		ou[i] = bad;
	} else if(op == 6) {
		// bit patterns a[i] .. a[i] + b[i] - 1 (positive normal floats >= 2^-96): how often is raw v_sqrt_f32 one ulp
		// above (low half) / one ulp below (high half) the correctly rounded result
		uint64_t hi = 0, lo = 0;
		for(uint64_t k = 0; k < b[i]; k++) {
			const uint32_t u = (uint32_t)(a[i] + k);
			if(u < PT_SQRT_TINY_BITS || u >= 0x7f800000u) continue;
			const float x = __uint_as_float(u);
			const uint32_t want = __float_as_uint(sqrtf(x)), raw = __float_as_uint(__builtin_amdgcn_sqrtf(x));
			if(raw > want) hi++;
			if(raw < want) lo++;
		}
		ou[i] = hi + (lo << 32);
	} else if(op == 7 || op == 8) {
		// b[i] pseudo-random (num, den) pairs from seed a[i], 2^-80 <= num <= 2^80, 2^-40 <= den <= 2^40 (finish_scan's
		// guarantee), the first 16 pairs of every item the corners of that range: how many have div_nt(num, den)
		// (op 7) or the bare num * v_rcp_f32(den) (op 8, the canary: it must fail) != num / den.  High half: pairs compared.
		uint64_t bad = 0, done = 0;
		uint64_t st = a[i] * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
		for(uint64_t k = 0; k < b[i]; k++) {
			float num, den;
			if(k < 16) {
				const float ncorner[4] = {0x1p-80f, 0x1p80f, 0x1.fffffep-80f * 2.0f, 0x1.fffffep79f};
				const float dcorner[4] = {0x1p-40f, 0x1p40f, 0x1.fffffep-40f * 2.0f, 0x1.fffffep39f};
				num = ncorner[k & 3]; den = dcorner[k >> 2];
			} else {
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				const uint32_t r1 = (uint32_t)(st >> 32);
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				const uint32_t r2 = (uint32_t)(st >> 32);
				// exponent fields 127-80 .. 127+79 (160 binades) and 127-40 .. 127+39 (80 binades), any mantissa
				num = __uint_as_float(((47u + (r1 >> 23) % 160u) << 23) | (r1 & 0x7fffffu));
				den = __uint_as_float(((87u + (r2 >> 23) % 80u) << 23) | (r2 & 0x7fffffu));
			}
			const float want = num / den;
			const float got = op == 7 ? div_nt(num, den) : num * __builtin_amdgcn_rcpf(den);
			if(__float_as_uint(got) != __float_as_uint(want)) bad++;
			done++;
		}
		ou[i] = bad + (done << 32);
	} else if(op == 9) {
		// pow5 of the float with bit pattern a[i] -> of[i] (tests compare with an exact integer arbiter on the host)
		of[i] = pow5(__uint_as_float((uint32_t)a[i]));
	} else if(op == 10) {
		// the lock-step scan's divide as the kernel forms it: x = dot(d, d) pseudo-random in [2^-40, 2^40] (16 corners
		// first), ln = sqrt_pos_y(x, y), a = ln * ln, num pseudo-random in [2^-80, 2^80]: #pairs with
		// div_by_a(num, a, y) != num / a | (#pairs compared) << 32
		uint64_t bad = 0, done = 0;
		uint64_t st = a[i] * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
		for(uint64_t k = 0; k < b[i]; k++) {
			float num, x;
			if(k < 16) {
				const float ncorner[4] = {0x1p-80f, 0x1p80f, 0x1.fffffep-80f * 2.0f, 0x1.fffffep79f};
				const float dcorner[4] = {0x1p-40f, 0x1p40f, 0x1.fffffep-40f * 2.0f, 0x1.fffffep39f};
				num = ncorner[k & 3]; x = dcorner[k >> 2];
			} else {
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				const uint32_t r1 = (uint32_t)(st >> 32);
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				const uint32_t r2 = (uint32_t)(st >> 32);
				num = __uint_as_float(((47u + (r1 >> 23) % 160u) << 23) | (r1 & 0x7fffffu));
				x = __uint_as_float(((87u + (r2 >> 23) % 80u) << 23) | (r2 & 0x7fffffu));
			}
			float y;
			const float ln = sqrt_pos_y(x, y);
			const float den = ln * ln;
			if(__float_as_uint(div_by_a(num, den, y)) != __float_as_uint(num / den)) bad++;
			done++;
		}
		ou[i] = bad + (done << 32);
	} else if(op == 11) {
		// bit patterns a[i] .. a[i] + b[i] - 1: #patterns x where inv_len(x) (normalize()'s 1 / length from x = dot(v, v),
		// one v_rsq_f32 for both the root and the reciprocal) != 1.0f / sqrtf(x)   (NaN == NaN)
		uint64_t bad = 0;
		for(uint64_t k = 0; k < b[i]; k++) {
			const float x = __uint_as_float((uint32_t)(a[i] + k));
			const float want = 1.0f / sqrtf(x), got = inv_len(x);
			if(want != want ? (got == got) : (__float_as_uint(got) != __float_as_uint(want))) bad++;
		}
		ou[i] = bad;
	} else if(op == 12) {
		// the split running mean's divide (pt_render_kernels.h mean_step) against x / f1: f1 = (float)a[i] (an integer
		// below 2^24), x = every significand of the binade with exponent field b[i]: #x where they differ
		// (b[i] & 0x100: with r 256 ulps too large instead, which must show up — the tests' canary; the form absorbs a few ulps)
		const float f1 = (float)a[i], r = __uint_as_float(__float_as_uint(1.0f / f1) + (uint32_t)(b[i] & 0x100));
		const uint32_t e = (uint32_t)(b[i] & 0xff) << 23;
		uint64_t bad = 0;
		for(uint32_t m = 0; m < (1u << 23); m++) {
			const float x = __uint_as_float(e | m);
			if(__float_as_uint(mean_step(0.0f, x, f1 - 1.0f, f1, r)) != __float_as_uint(x / f1)) bad++;
		}
		ou[i] = bad;
	}
}

// ---------------------------------------------------------------------------------------
// Scratch of pt_render_noalloc (include/pt_hip.h): what pt_render keeps in hipMalloc'ed buffers of the context —
// the pixel-queue head of the streaming kernel, the bounce counter, the per-sample seed table — lives here
// instead, one copy per GPU for the life of the process.  The NAME is only a host-side handle: the device
// address comes from hipGetSymbolAddress (pt_ctx_create), and only that pointer is ever handed to a kernel,
// a memset or a copy.
// ---------------------------------------------------------------------------------------
#define PT_SCRATCH_QUEUE_OFF 0                    // unsigned int, streaming kernel's queue head (256 B line of its own)
#define PT_SCRATCH_COUNTER_OFF 256                // unsigned long long, bounce counter (line of its own)
#define PT_SCRATCH_SEEDS_OFF 512                  // uint64_t[PT_NOALLOC_MAX_FRAMES]: 16807 * seed_k
#define PT_SCRATCH_BYTES (PT_SCRATCH_SEEDS_OFF + 8 * PT_NOALLOC_MAX_FRAMES)
static __device__ __align__(256) unsigned char scratch[PT_SCRATCH_BYTES];

// Calls of pt_render_noalloc on one GPU share that GPU's `scratch`: a call that would overlap an earlier one still
// running on ANOTHER stream is refused (PT_E_STATE) instead of racing.  One event per GPU, created by the first
// pt_ctx_create on it (never inside pt_render_noalloc: creating an event allocates), recorded behind every call.
#define PT_MAX_DEVICES 64
namespace {
struct NoallocSlot { hipEvent_t ev; hipStream_t stream; bool armed; };
NoallocSlot g_noalloc[PT_MAX_DEVICES];
std::mutex g_noallocLock;
}

// Host seeds -> the scratch seed table without any staging allocation: up to 256 seeds ride in the kernel
// arguments (2 KB by value), the kernel folds the sample-uniform 16807 * seed (src/cl/rng.cl:8) and stores them.
struct SeedChunk { uint64_t v[256]; };
__global__ __launch_bounds__(256) void pt_stage_seeds_kernel(const SeedChunk ch, int count, uint64_t *__restrict__ dst) {
	const int i = (int)threadIdx.x;
	if(i < count) dst[i] = 16807ull * ch.v[i];
}

// ---------------------------------------------------------------------------------------
// Host side of the C ABI
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail) {
	snprintf(g_err, sizeof g_err, fmt, detail ? detail : "");
	return code;
}

#define HIP_TRY(expr)                                                                          \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                \
		if(e_ != hipSuccess) {                                                                 \
			snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, hipGetErrorString(e_));      \
			return e_ == hipErrorOutOfMemory ? PT_E_NOMEM : PT_E_DEVICE;                       \
		}                                                                                      \
	} while(0)

// The PT_* environment switches (debugging aids and opt-ins), read ONCE per context (pt_ctx_create) and again only
// when the caller asks (pt_ctx_reload_env): no getenv on the render path.
struct Switches {
	int lanesPerPixel;      // PT_LANES_PER_PIXEL: force 1 .. 64 lanes per pixel; -1: chosen by pt_render
	int tailPct;            // PT_TAIL_PCT: share of the pixels in the second tier; -1: default
	int tailLanes;          // PT_TAIL_LANES: log2 of the second tier's lane limit; -1: default (6)
	bool noTrioSkip;        // PT_NO_TRIO_SKIP
	int kernel;             // PT_KERNEL: 0 chosen by pt_render, 1 "stream", 2 "lds"
	bool genericScan;       // PT_GENERIC_SCAN
	int math;               // PT_MATH: 0 pt_render_args.fast_math decides, 1 "fast", 2 "strict"
	int accel;              // PT_ACCEL: 0 pt_render_args.accel decides, 1 "grid", 2 "none"
	int combine;            // PT_COMBINE: 0 RCCL, peer copies where it cannot be set up; 1 "copy"; 2 "rccl" (no fallback)
};
static Switches read_switches() {
	Switches w;
	const char *v;
	w.lanesPerPixel = (v = getenv("PT_LANES_PER_PIXEL")) ? atoi(v) : -1;
	w.tailPct = (v = getenv("PT_TAIL_PCT")) ? atoi(v) : -1;
	w.tailLanes = (v = getenv("PT_TAIL_LANES")) ? atoi(v) : -1;
	w.noTrioSkip = getenv("PT_NO_TRIO_SKIP") != NULL;
	v = getenv("PT_KERNEL");
	w.kernel = v && strcmp(v, "stream") == 0 ? 1 : v && strcmp(v, "lds") == 0 ? 2 : 0;
	w.genericScan = getenv("PT_GENERIC_SCAN") != NULL;
	v = getenv("PT_MATH");
	w.math = v && strcmp(v, "fast") == 0 ? 1 : v && strcmp(v, "strict") == 0 ? 2 : 0;
	v = getenv("PT_ACCEL");
	w.accel = v && strcmp(v, "grid") == 0 ? 1 : v && strcmp(v, "none") == 0 ? 2 : 0;
	v = getenv("PT_COMBINE");
	w.combine = v && strcmp(v, "copy") == 0 ? 1 : v && strcmp(v, "rccl") == 0 ? 2 : 0;
	return w;
}

struct pt_ctx {
	int device;
	hipStream_t stream;
	Switches sw;
	// derived from the uploaded scene once (pt_scene_upload), not per launch
	unsigned long long sceneTrioMask;     // trios that hold a sphere under an eighth of the largest radius (closest_hit_lds)
	float enclosedPos[3]; int enclosedKnown, enclosed;   // camera_is_enclosed() of the last camera position asked about
	// multi-GPU combine (pt_combine_bands*): packed bands of this rank / of the peers (root), RCCL communicator of a
	// multi-process job (pt_comm_init_rank)
	float4 *stage; size_t capStage;
	void *mpComm; int mpWorld, mpRank;
	float4 *image; int imgW, imgH;
	float4 *geom, *geomR, *matA, *matB, *matC; int nSpheres, capSpheres;      // geomR: {centre, radius}, matC: derived dielectric constants, both for the relaxed build
	uint64_t *seedmul; int capSeeds;
	uint64_t *hostSeedmul; int nHostSeeds;   // what seedmul holds, to skip identical re-uploads
	unsigned long long *counter;
	unsigned int *queue;          // pixel queue head of the streaming kernel
	float4 *hostSpheres;          // cx, cy, cz, radius of the uploaded scene (host copy, for the grid build)
	// opt-in uniform grid (built on first use after a scene upload)
	bool gridBuilt;
	GridTables grid;
	int *dCellStart, *dItems, *dBig; size_t capCells, capItems, capBig;
	unsigned char *u8; size_t capU8;
	hipEvent_t evStart, evStop;
	hipEvent_t ring[2 * PT_TIMING_SLOTS];   // start/stop pairs of launches made with record_timing
	int nTimed;
	unsigned char *scratchPtr;    // device address of `scratch` on this context's GPU (hipGetSymbolAddress)
};

extern "C" const char *pt_last_error(void) { return g_err; }

extern "C" int pt_device_count(int *count) {
	if(!count) return fail(PT_E_INVALID, "pt_device_count: NULL%s", NULL);
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if(e != hipSuccess) { *count = 0; return fail(PT_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
	*count = n;
	return PT_OK;
}

extern "C" int pt_ctx_destroy(pt_ctx *c);

extern "C" int pt_ctx_create(int device, pt_ctx **out) {
	if(!out) return fail(PT_E_INVALID, "pt_ctx_create: NULL out%s", NULL);
	*out = NULL;
	int n = 0;
	HIP_TRY(hipGetDeviceCount(&n));
	if(device < 0 || device >= n) return fail(PT_E_INVALID, "pt_ctx_create: device out of range%s", NULL);
	HIP_TRY(hipSetDevice(device));
	pt_ctx *c = (pt_ctx *)calloc(1, sizeof(pt_ctx));
	if(!c) return fail(PT_E_NOMEM, "pt_ctx_create: host allocation failed%s", NULL);
	c->device = device;
	c->sw = read_switches();
	// every step after the calloc: on failure the half-built context is torn down again (pt_ctx_destroy copes with
	// the members that are still NULL) and the first error is what the caller sees
	hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	const char *what = "hipStreamCreateWithFlags";
	if(e == hipSuccess) { e = hipMalloc((void **)&c->counter, sizeof(unsigned long long)); what = "hipMalloc(counter)"; }
	if(e == hipSuccess) { e = hipMalloc((void **)&c->queue, 256); what = "hipMalloc(queue)"; }
	if(e == hipSuccess) { e = hipEventCreate(&c->evStart); what = "hipEventCreate"; }
	if(e == hipSuccess) { e = hipEventCreate(&c->evStop); what = "hipEventCreate"; }
	for(int i = 0; e == hipSuccess && i < 2 * PT_TIMING_SLOTS; i++) { e = hipEventCreate(&c->ring[i]); what = "hipEventCreate(ring)"; }
	// this GPU's copy of the no-allocation scratch: its device address, looked up with this device current
	if(e == hipSuccess) { e = hipGetSymbolAddress((void **)&c->scratchPtr, HIP_SYMBOL(scratch)); what = "hipGetSymbolAddress(scratch)"; }
	if(e == hipSuccess && device < PT_MAX_DEVICES) {
		std::lock_guard<std::mutex> lock(g_noallocLock);
		if(!g_noalloc[device].ev) { e = hipEventCreateWithFlags(&g_noalloc[device].ev, hipEventDisableTiming); what = "hipEventCreate(noalloc)"; }
	}
	if(e != hipSuccess) {
		char msg[256];
		snprintf(msg, sizeof msg, "%s failed: %s", what, hipGetErrorString(e));
		pt_ctx_destroy(c);
		snprintf(g_err, sizeof g_err, "pt_ctx_create: %s", msg);
		return e == hipErrorOutOfMemory ? PT_E_NOMEM : PT_E_DEVICE;
	}
	*out = c;
	return PT_OK;
}

static void pt_comm_release(pt_ctx *c);

extern "C" int pt_ctx_reload_env(pt_ctx *c) {
	if(!c) return fail(PT_E_INVALID, "pt_ctx_reload_env: NULL%s", NULL);
	c->sw = read_switches();
	return PT_OK;
}

extern "C" int pt_ctx_destroy(pt_ctx *c) {
	if(!c) return PT_OK;
	(void)hipSetDevice(c->device);
	if(c->stream) (void)hipStreamSynchronize(c->stream);
	(void)hipFree(c->image); (void)hipFree(c->geom); (void)hipFree(c->geomR); (void)hipFree(c->matA); (void)hipFree(c->matB); (void)hipFree(c->matC);
	(void)hipFree(c->seedmul); (void)hipFree(c->counter); (void)hipFree(c->queue);
	(void)hipFree(c->dCellStart); (void)hipFree(c->dItems); (void)hipFree(c->dBig);
	free(c->hostSpheres); (void)hipFree(c->u8); (void)hipFree(c->stage);
	if(c->mpComm) pt_comm_release(c);
	if(c->evStart) (void)hipEventDestroy(c->evStart);
	if(c->evStop) (void)hipEventDestroy(c->evStop);
	for(int i = 0; i < 2 * PT_TIMING_SLOTS; i++) if(c->ring[i]) (void)hipEventDestroy(c->ring[i]);
	if(c->stream) (void)hipStreamDestroy(c->stream);
	free(c->hostSeedmul);
	free(c);
	return PT_OK;
}

extern "C" int pt_image_alloc(pt_ctx *c, int width, int height) {
	if(!c || width <= 0 || height <= 0 || (long long)width * height > (1ll << 30))
		return fail(PT_E_INVALID, "pt_image_alloc: bad size%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	if(c->image) { HIP_TRY(hipFree(c->image)); c->image = NULL; }
	size_t bytes = sizeof(float4) * (size_t)width * (size_t)height;
	HIP_TRY(hipMalloc((void **)&c->image, bytes));
	HIP_TRY(hipMemsetAsync(c->image, 0, bytes, c->stream));
	c->imgW = width; c->imgH = height;
	return PT_OK;
}

extern "C" int pt_scene_upload(pt_ctx *c, const float *spheres, int n) {
	if(!c || n < 0 || (n > 0 && !spheres)) return fail(PT_E_INVALID, "pt_scene_upload: bad arguments%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	if(n > c->capSpheres) {
		(void)hipFree(c->geom); (void)hipFree(c->geomR); (void)hipFree(c->matA); (void)hipFree(c->matB); (void)hipFree(c->matC);
		c->geom = c->geomR = c->matA = c->matB = c->matC = NULL; c->capSpheres = 0;
		HIP_TRY(hipMalloc((void **)&c->geom, sizeof(float4) * (size_t)n));
		HIP_TRY(hipMalloc((void **)&c->geomR, sizeof(float4) * (size_t)n));
		HIP_TRY(hipMalloc((void **)&c->matA, sizeof(float4) * (size_t)n));
		HIP_TRY(hipMalloc((void **)&c->matB, sizeof(float4) * (size_t)n));
		HIP_TRY(hipMalloc((void **)&c->matC, sizeof(float4) * (size_t)n));
		c->capSpheres = n;
	}
	c->nSpheres = n;
	c->gridBuilt = false;
	c->sceneTrioMask = 0;
	c->enclosedKnown = 0;
	free(c->hostSpheres); c->hostSpheres = NULL;
	if(n == 0) return PT_OK;
	c->hostSpheres = (float4 *)malloc(sizeof(float4) * (size_t)n);
	if(!c->hostSpheres) return fail(PT_E_NOMEM, "pt_scene_upload: host allocation failed%s", NULL);
	for(int i = 0; i < n; i++) {
		const float *r = spheres + PT_SPHERE_FLOATS * (size_t)i;
		c->hostSpheres[i] = make_float4(r[0], r[1], r[2], r[3]);
	}
	// LDS kernel: trios of the table that hold a sphere much smaller than the largest one get a wave-level "can any
	// lane hit it" test per sphere before the sqrt/select part (closest_hit_lds); a pure speed heuristic, same image.
	// The mask has 64 bits: trios beyond the first 64 (tables above 192 spheres, reachable only with PT_KERNEL=lds)
	// always take the unconditional path: correct, only slower.
	{
		float rmax = 0.0f;
		for(int i = 0; i < n; i++) rmax = fmaxf(rmax, fabsf(c->hostSpheres[i].w));
		for(int i = 0; i < n && i < 192 && i / 3 < n / 3; i++)
			if(fabsf(c->hostSpheres[i].w) < rmax * 0.125f) c->sceneTrioMask |= 1ull << (i / 3);
	}
	float4 *h = (float4 *)malloc(sizeof(float4) * 3 * (size_t)n);
	if(!h) return fail(PT_E_NOMEM, "pt_scene_upload: host allocation failed%s", NULL);
	float4 *hg = h, *ha = h + n, *hb = h + 2 * (size_t)n;
	for(int i = 0; i < n; i++) {
		const float *r = spheres + PT_SPHERE_FLOATS * (size_t)i;
		// r*r is rounded once here exactly as `sphere.radius * sphere.radius` is per test (src/cl/ray.cl:75)
		volatile float r2 = r[3] * r[3];
		hg[i] = make_float4(r[0], r[1], r[2], r2);
		// material codes outside 0..3 match no branch of src/cl/ray.cl:130-171 (the ray is left as it is): one code, -1,
		// for all of them, which shade_hit() handles together with the back face of a light
		int type = (r[4] > -1.0f && r[4] < 4.0f) ? (int)r[4] : -1;
		float tb;
		memcpy(&tb, &type, 4);
		ha[i] = make_float4(r[5], r[6], r[7], tb);
		// Schlick r0 of a front-face hit, ((1 - refIdx) / (1 + refIdx))^2 (src/cl/ray.cl:59-60 with rel = refIdx):
		// the same four correctly rounded binary32 operations the kernel would repeat per hit
		volatile float q = (1.0f - r[10]) / (1.0f + r[10]);
		volatile float r0 = q * q;
		// .y is the factor of random_unit_vector() in the new direction (shade_hit): the fuzzyness for metal and
		// dielectric (ray.cl:145,164,167), 1 for the lambertian, whose `normal + random_unit_vector` (ray.cl:138) is
		// `normal + random_unit_vector * 1.0f` bit for bit and whose own fuzzyness field the reference never reads
		hb[i] = make_float4(r[8], type == 1 ? 1.0f : r[9], r[10], r0);
	}
	// one stream, in order; the staging buffer is freed after the copies have completed
	hipError_t e0 = hipMemcpyAsync(c->geomR, c->hostSpheres, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream);
	hipError_t e1 = hipMemcpyAsync(c->geom, hg, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream);
	hipError_t e2 = hipMemcpyAsync(c->matA, ha, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream);
	hipError_t e3 = hipMemcpyAsync(c->matB, hb, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, c->stream);
	// the relaxed build's derived table, from the table just uploaded, by that build's own instructions
	pt_fast::derive_materials(c->matB, c->matC, n, c->stream);
	hipError_t e5 = hipGetLastError();
	hipError_t e4 = hipStreamSynchronize(c->stream);
	free(h);
	HIP_TRY(e0); HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3); HIP_TRY(e5); HIP_TRY(e4);
	return PT_OK;
}

// ---------------------------------------------------------------------------------------
// Uniform grid build (host).  Spheres whose padded box would cover more than PT_GRID_BIG_CELLS cells of the
// grid go to the `big` list instead (the walls of the reference's scene are spheres of radius 10^4).
// ---------------------------------------------------------------------------------------
#define PT_GRID_BIG_CELLS 512
#define PT_GRID_MAX_AXIS 256

static int grow_int(int **d, size_t *cap, size_t need) {
	if(need <= *cap) return PT_OK;
	if(*d) { (void)hipFree(*d); *d = NULL; *cap = 0; }
	hipError_t e = hipMalloc((void **)d, sizeof(int) * need);
	if(e != hipSuccess) { snprintf(g_err, sizeof g_err, "grid hipMalloc failed: %s", hipGetErrorString(e)); return PT_E_NOMEM; }
	*cap = need;
	return PT_OK;
}

static int build_grid(pt_ctx *c, hipStream_t st) {
	const int n = c->nSpheres;
	const float4 *sp = c->hostSpheres;
	std::vector<char> isBig((size_t)n, 0);
	GridTables G;
	memset(&G, 0, sizeof G);
	std::vector<int> cellStart, items, big;
	// two rounds: the first sizes the grid over everything and finds the spheres that are too large for it,
	// the second sizes it over the rest
	for(int round = 0; round < 3; round++) {
		double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
		long nSmall = 0;
		for(int i = 0; i < n; i++) {
			if(isBig[i]) continue;
			const double cc[3] = {sp[i].x, sp[i].y, sp[i].z}, r = fabs((double)sp[i].w);
			if(!(std::isfinite(cc[0]) && std::isfinite(cc[1]) && std::isfinite(cc[2]) && std::isfinite(r))) { isBig[i] = 1; continue; }
			for(int ax = 0; ax < 3; ax++) { lo[ax] = fmin(lo[ax], cc[ax] - r); hi[ax] = fmax(hi[ax], cc[ax] + r); }
			nSmall++;
		}
		if(nSmall == 0) { G.gx = G.gy = G.gz = 0; break; }
		double ext[3], vol = 1;
		for(int ax = 0; ax < 3; ax++) { ext[ax] = fmax(hi[ax] - lo[ax], 1e-3); vol *= ext[ax]; }
		double cell = cbrt(vol * 2.0 / (double)nSmall);          // about two spheres per cell
		const double emax = fmax(ext[0], fmax(ext[1], ext[2]));
		cell = fmax(cell, emax / PT_GRID_MAX_AXIS);
		const double pad = 0.1 * cell;
		int g[3];
		for(int ax = 0; ax < 3; ax++) {
			lo[ax] -= 2 * pad; hi[ax] += 2 * pad;
			g[ax] = (int)ceil((hi[ax] - lo[ax]) / cell);
			if(g[ax] < 1) g[ax] = 1;
			if(g[ax] > PT_GRID_MAX_AXIS + 8) g[ax] = PT_GRID_MAX_AXIS + 8;
		}
		G.gx = g[0]; G.gy = g[1]; G.gz = g[2];
		G.minx = (float)lo[0]; G.miny = (float)lo[1]; G.minz = (float)lo[2];
		G.cell = (float)cell; G.invCell = 1.0f / G.cell;
		G.maxx = G.minx + G.cell * (float)g[0]; G.maxy = G.miny + G.cell * (float)g[1]; G.maxz = G.minz + G.cell * (float)g[2];
		// cell ranges per sphere, with the same float cell arithmetic the kernel uses and a tenth of a cell of padding
		auto range = [&](int i, int *a0, int *a1) {
			const float cc[3] = {sp[i].x, sp[i].y, sp[i].z}, r = fabsf(sp[i].w);
			const float mn[3] = {G.minx, G.miny, G.minz};
			const float eps = (float)pad + 1e-3f * r + 1e-5f * fmaxf(fabsf(cc[0]), fmaxf(fabsf(cc[1]), fabsf(cc[2])));
			long cells = 1;
			for(int ax = 0; ax < 3; ax++) {
				int l = (int)floorf((cc[ax] - r - eps - mn[ax]) * G.invCell) , h = (int)floorf((cc[ax] + r + eps - mn[ax]) * G.invCell);
				l = l < 0 ? 0 : l; h = h >= g[ax] ? g[ax] - 1 : h;
				if(h < l) h = l;
				a0[ax] = l; a1[ax] = h;
				cells *= (h - l + 1);
			}
			return cells;
		};
		bool changed = false;
		if(round < 2)
			for(int i = 0; i < n; i++) {
				if(isBig[i]) continue;
				int a0[3], a1[3];
				if(range(i, a0, a1) > PT_GRID_BIG_CELLS) { isBig[i] = 1; changed = true; }
			}
		if(changed) continue;
		const size_t nc = (size_t)g[0] * g[1] * g[2];
		cellStart.assign(nc + 1, 0);
		for(int pass = 0; pass < 2; pass++) {
			if(pass == 1) {
				size_t run = 0;
				for(size_t k = 0; k <= nc; k++) { size_t v = (size_t)cellStart[k]; cellStart[k] = (int)run; run += v; }
				items.assign(run, 0);
			}
			std::vector<int> fill;
			if(pass == 1) fill.assign(cellStart.begin(), cellStart.end() - 1);
			for(int i = 0; i < n; i++) {
				if(isBig[i]) continue;
				int a0[3], a1[3];
				range(i, a0, a1);
				for(int z = a0[2]; z <= a1[2]; z++)
					for(int y = a0[1]; y <= a1[1]; y++)
						for(int x = a0[0]; x <= a1[0]; x++) {
							const size_t cellIdx = ((size_t)z * g[1] + y) * g[0] + x;
							if(pass == 0) cellStart[cellIdx]++;
							else items[(size_t)fill[cellIdx]++] = i;
						}
			}
		}
		break;
	}
	for(int i = 0; i < n; i++) if(isBig[i]) big.push_back(i);
	if(G.gx == 0) { cellStart.assign(2, 0); items.clear(); }
	G.nBig = (int)big.size();
	int rc;
	if((rc = grow_int(&c->dCellStart, &c->capCells, cellStart.size() + 1)) != PT_OK) return rc;
	if((rc = grow_int(&c->dItems, &c->capItems, items.size() + 1)) != PT_OK) return rc;
	if((rc = grow_int(&c->dBig, &c->capBig, big.size() + 1)) != PT_OK) return rc;
	// on the stream the render will be launched on (ordered before it), drained before the host vectors go away
	HIP_TRY(hipMemcpyAsync(c->dCellStart, cellStart.data(), sizeof(int) * cellStart.size(), hipMemcpyHostToDevice, st));
	if(!items.empty()) HIP_TRY(hipMemcpyAsync(c->dItems, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice, st));
	if(!big.empty()) HIP_TRY(hipMemcpyAsync(c->dBig, big.data(), sizeof(int) * big.size(), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	G.cellStart = c->dCellStart; G.items = c->dItems; G.big = c->dBig;
	c->grid = G;
	c->gridBuilt = true;
	return PT_OK;
}

// rows r with (r / band) % world == rank
static long long local_rows64(int H, int rank, int world, int band) {
	const long long nb = ((long long)H + band - 1) / band;          // bands of the image, the last one possibly short
	if(rank >= nb) return 0;
	const long long mine = (nb - 1 - rank) / world + 1;               // bands rank, rank + world, ...
	long long rows = mine * band;
	if((nb - 1) % world == rank) rows -= nb * band - H;               // this rank owns the short last band
	return rows;
}

// Speed heuristic for the kernel choice: do rays from `pos` along the 26 axis and diagonal directions all hit a
// sphere (near root, as src/cl/ray.cl:71-87)?  Then the scene is closed around the camera and paths run long.
static bool camera_is_enclosed(const pt_ctx *c, const float *pos) {
	if(!c->hostSpheres || c->nSpheres == 0) return false;
	for(int dx = -1; dx <= 1; dx++) for(int dy = -1; dy <= 1; dy++) for(int dz = -1; dz <= 1; dz++) {
		if(!dx && !dy && !dz) continue;
		const double a = (double)(dx * dx + dy * dy + dz * dz);
		bool hit = false;
		for(int i = 0; i < c->nSpheres && !hit; i++) {
			const float4 s = c->hostSpheres[i];
			const double ox = pos[0] - s.x, oy = pos[1] - s.y, oz = pos[2] - s.z;
			const double bHalf = ox * dx + oy * dy + oz * dz;
			const double disc = bHalf * bHalf - a * (ox * ox + oy * oy + oz * oz - (double)s.w * s.w);
			hit = disc >= 0.0 && -bHalf - sqrt(disc) > 0.0;
		}
		if(!hit) return false;
	}
	return true;
}

// Where one render call keeps its device-side working memory: the context's hipMalloc'ed buffers (pt_render) or
// this GPU's copy of `scratch` (pt_render_noalloc).  The kernels take these pointers as arguments either way.
struct RenderScratch {
	const uint64_t *seedmul;
	unsigned long long *counter;
	unsigned int *queue;
};

// Launch geometry of the lock-step kernel: lanes per pixel for the two tiers of a launch, in 64-bit arithmetic.
// The kernel indexes threads with a 32-bit int (pt_render_kernels.h), so the total must stay below 2^31: a frame
// whose tiers would exceed that (above ~3.7e8 pixels per rank with the default tiers) loses the tail tier first and
// then lanes per pixel; one lane per pixel always fits (pt_image_alloc admits 2^30 pixels).
struct LaunchPlan { int logS, logS2; long long tierPixels, tierThreads, totalThreads; unsigned blocks; };

static int plan_launch(long long lp, int firstFrame, int nFrames, const Switches &sw, LaunchPlan *out) {
	// enough lanes per pixel that the launch has about a dozen rounds of the chip's 8192 wave slots, at most 16
	// (measured: the 1080p headline frame 243 ms at 1 lane per pixel, 231 ms at 4; one rank's eighth of it
	// 32.9 / 28.6 / 26.1 / 25.0 / 24.4 ms at 1 / 2 / 4 / 8 / 16 lanes, 25.1 ms at 32)
	int want = 0;
	const bool forced = sw.lanesPerPixel >= 0;
	if(forced) { while((1 << want) < sw.lanesPerPixel && want < 6) want++; }
	else {
		while(want < 2 && (2 << want) <= nFrames) want++;      // 4 lanes whatever the frame size (4K: 734 -> 727 ms)
		while(want < 4 && (lp << want) < 64ll * 12 * 8192 && (2 << want) <= nFrames) want++;
	}
	const bool floatFrames = (long long)firstFrame + nFrames < (1 << 24);      // the group accumulation counts frames in float
	if(!floatFrames) want = 0;
	// The tail of the launch: blocks last nFrames / S rounds and the chip drains over about one block time at the
	// end, so the pixels whose blocks are dispatched last get more lanes each (shorter blocks; bit-identical).
	// A whole wave (64 lanes) per pixel for the last 3 % of a large launch (headline frame: 157.6 ms against 158.2
	// with 16 lanes for the last 8 %, 162.3 without a second tier), the last 5 % of a small one whose first tier
	// already has 16 (one rank's band of eight 20.23 -> 20.1 ms, main.lua's 640x360x2000 144.7 -> 142.8 ms).
	// PT_TAIL_PCT / PT_TAIL_LANES (debugging aids) override the share and the log2 of the lane limit.
	int want2 = want;
	const int pct = sw.tailPct >= 0 ? sw.tailPct : want >= 4 ? 5 : 3;
	const int cap2 = sw.tailLanes >= 0 ? sw.tailLanes : 6;
	if((!forced || sw.tailPct >= 0) && pct > 0 && floatFrames) while(want2 < cap2 && want2 < 6 && (2 << want2) <= nFrames) want2++;
	for(;;) {
		long long tier = lp;
		if(want2 > want) {
			tier = lp - lp * pct / 100;
			tier -= tier % 64;                      // tier << logS is a multiple of the block size (256) for logS >= 2
			if(want < 2) tier -= tier % 256;
			if(tier < 0) tier = 0;
		}
		const long long total = (tier << want) + ((lp - tier) << want2);
		if(total < (1ll << 31)) {
			out->logS = want; out->logS2 = want2;
			out->tierPixels = tier; out->tierThreads = tier << want; out->totalThreads = total;
			out->blocks = (unsigned)((total + 255) / 256);
			return PT_OK;
		}
		if(want2 > want) want2--;
		else if(want > 0) { want--; want2 = want; }
		else return fail(PT_E_INVALID, "pt_render: more than 2^31 - 1 pixels on one rank%s", NULL);
	}
}

extern "C" int pt_plan_launch(const pt_render_args *a, pt_launch_plan *out) {
	if(!a || !out) return fail(PT_E_INVALID, "pt_plan_launch: NULL%s", NULL);
	if(a->width <= 0 || a->height <= 0 || a->n_frames < 0 || a->first_frame < 0 || (long long)a->width * a->height > (1ll << 30) ||
	   a->world < 1 || a->rank < 0 || a->rank >= a->world || a->band_rows < 1)
		return fail(PT_E_INVALID, "pt_plan_launch: bad arguments%s", NULL);
	const long long lp = local_rows64(a->height, a->rank, a->world, a->band_rows) * a->width;
	LaunchPlan pl;
	memset(&pl, 0, sizeof pl);
	if(lp > 0) { int rc = plan_launch(lp, a->first_frame, a->n_frames, read_switches(), &pl); if(rc != PT_OK) return rc; }
	out->local_pixels = lp;
	out->lanes_log2 = pl.logS; out->tail_lanes_log2 = pl.logS2;
	out->tier_pixels = pl.tierPixels; out->tier_threads = pl.tierThreads; out->total_threads = pl.totalThreads;
	out->blocks = pl.blocks;
	return PT_OK;
}

static int check_render_args(const pt_ctx *c, const pt_render_args *a, const char *who) {
	if(!c || !a) return fail(PT_E_INVALID, "%s: NULL", who);
	if(a->width <= 0 || a->height <= 0 || a->max_depth < 0 || a->n_frames < 0 || a->first_frame < 0)
		return fail(PT_E_INVALID, "%s: bad image/sample arguments", who);
	if((long long)a->width * a->height > (1ll << 30)) return fail(PT_E_INVALID, "%s: image too large", who);
	if(a->accel != 0 && a->accel != 1) return fail(PT_E_INVALID, "%s: unknown accel", who);
	if(a->world < 1 || a->rank < 0 || a->rank >= a->world || a->band_rows < 1)
		return fail(PT_E_INVALID, "%s: bad partition (rank/world/band_rows)", who);
	if(a->n_frames > 0 && !a->seeds) return fail(PT_E_INVALID, "%s: NULL seeds", who);
	if(!a->d_image && (!c->image || c->imgW != a->width || c->imgH != a->height))
		return fail(PT_E_STATE, "%s: no accumulator of this size (call pt_image_alloc)", who);
	return PT_OK;
}

// Kernel choice, launch geometry and the launch itself, shared by pt_render and pt_render_noalloc.  Allocates nothing.
static int launch_frames(pt_ctx *c, const pt_render_args *a, float4 *img, hipStream_t st, const RenderScratch &rs, bool useGrid, const char *who) {
	KParams p;
	p.W = a->width; p.H = a->height; p.maxDepth = a->max_depth; p.n = c->nSpheres;
	p.bgx = a->background[0]; p.bgy = a->background[1]; p.bgz = a->background[2];
	p.px = a->cam_pos[0]; p.py = a->cam_pos[1]; p.pz = a->cam_pos[2];
	// cos/sin of the launch-uniform camera angles: the host libm, as the oracle uses (src/cl/ray.cl:45-49)
	p.cosRx = cosf(a->cam_rot[0]); p.sinRx = sinf(a->cam_rot[0]);
	p.cosRy = cosf(a->cam_rot[1]); p.sinRy = sinf(a->cam_rot[1]);
	p.rotX = a->cam_rot[0]; p.rotY = a->cam_rot[1];
	p.sensorWidth = a->sensor_width; p.focalLength = a->focal_length; p.aperture = a->aperture; p.exposure = a->exposure;
	p.firstFrame = a->first_frame; p.nFrames = a->n_frames;
	p.rank = a->rank; p.world = a->world; p.bandRows = a->band_rows;
	const long long lp = local_rows64(a->height, a->rank, a->world, a->band_rows) * a->width;
	p.localPixels = (int)lp;
	if(lp == 0) {      // this rank owns no band of the image: nothing to launch, the outputs still get their values
		if(a->bounce_count) *a->bounce_count = 0;
		if(a->kernel_ms) *a->kernel_ms = 0.0f;
		return PT_OK;
	}

	// float rules: strict (bit-identical to the oracle) unless the caller, or PT_MATH=fast, asks for the relaxed build
	const bool fast = c->sw.math == 1 || (c->sw.math == 0 && a->fast_math != 0);
	SceneTables sc = {fast ? c->geomR : c->geom, c->matA, c->matB, c->matC};
	LaunchPlan plan;
	if(plan_launch(lp, a->first_frame, a->n_frames, c->sw, &plan) != PT_OK) return PT_E_INVALID;
	p.logS = plan.logS; p.logS2 = plan.logS2;
	p.tierPixels = (int)plan.tierPixels; p.tierThreads = (int)plan.tierThreads;
	p.trioSkipMask = c->sw.noTrioSkip ? 0ull : c->sceneTrioMask;      // derived from the scene at pt_scene_upload
	LaunchArgs la;
	la.image = img; la.p = p; la.sc = sc; la.seedmul = rs.seedmul; la.bounceCounter = rs.counter; la.queue = rs.queue;
	la.stream = st; la.count = a->bounce_count != NULL;
	la.gridPixels = plan.blocks;
	// Up to PT_LOCKSTEP_MAX_SPHERES spheres: the lock-step kernel with the whole table in LDS and a branch-free
	// scan (the reference's own scene has 9).  Beyond that the persistent streaming kernel wins on open scenes even
	// where the table would still fit in LDS, because paths end after a few bounces and its lanes never idle (1080p,
	// 64 spp, random spheres under a sky: 12 spheres 2.2 vs 2.2 ms, 16: 4.0 vs 3.2, 32: 7.0 vs 4.3, 100: 24 vs 7.8).  In a CLOSED scene paths run to the depth limit, no
	// lane idles, and the lock-step kernel stays ahead up to about 190 spheres (glass lattices in the box at 32 spp,
	// depth 32: 39 spheres 31.7 vs 45.9 ms; 72: 54.7 vs 69.3; 133: 92.5 vs 106.4; 224: 158 vs 149; 351: 274 vs 232):
	// it is chosen up to PT_LOCKSTEP_MAX_CLOSED when rays from the camera in all 26 axis and diagonal directions hit
	// something.  Either kernel gives the same image.
	la.kernel = c->nSpheres <= PT_LOCKSTEP_MAX_SPHERES ? PT_K_LDS : PT_K_STREAM;
	if(la.kernel == PT_K_STREAM && c->nSpheres <= PT_LOCKSTEP_MAX_CLOSED && c->sw.kernel == 0) {
		// asked once per camera position (the 26 probes walk the whole host table)
		if(!c->enclosedKnown || memcmp(c->enclosedPos, a->cam_pos, sizeof c->enclosedPos) != 0) {
			c->enclosed = camera_is_enclosed(c, a->cam_pos) ? 1 : 0;
			memcpy(c->enclosedPos, a->cam_pos, sizeof c->enclosedPos);
			c->enclosedKnown = 1;
		}
		if(c->enclosed) la.kernel = PT_K_LDS;
	}
	// PT_KERNEL (debugging aid): "stream" or "lds" overrides the choice
	if(c->sw.kernel == 1 && c->nSpheres > 0) la.kernel = PT_K_STREAM;
	if(c->sw.kernel == 2 && c->nSpheres <= PT_LDS_MAX_SPHERES) la.kernel = PT_K_LDS;
	memset(&la.grid, 0, sizeof la.grid);
	if(useGrid) { la.grid = c->grid; la.kernel = PT_K_GRID; }
	la.ldsBytes = la.kernel != PT_K_LDS ? 0 : fast ? pt_fast::lds_bytes(c->nSpheres) : pt_strict::lds_bytes(c->nSpheres);
	la.genericScan = c->sw.genericScan;
	// streaming kernel: persistent one-wave blocks, at most 32 per CU; surplus pixels come through the queue
	const long long swaves = (lp + 63) / 64;
	la.gridStream = (unsigned)(swaves < 256 * 32 ? swaves : 256 * 32);
	if(la.kernel == PT_K_STREAM || la.kernel == PT_K_GRID) HIP_TRY(hipMemsetAsync(rs.queue, 0, 256, st));     // before the start event: not kernel time
	if(a->bounce_count) {
		HIP_TRY(hipMemsetAsync(rs.counter, 0, sizeof(unsigned long long), st));
		if(fast) pt_fast::launch_render(la); else pt_strict::launch_render(la);
		HIP_TRY(hipGetLastError());
		unsigned long long v = 0;
		HIP_TRY(hipMemcpyAsync(&v, rs.counter, sizeof v, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		*a->bounce_count = (uint64_t)v;
		return PT_OK;
	}
	const bool ringTimed = a->record_timing && !a->kernel_ms;
	if(ringTimed && c->nTimed >= PT_TIMING_SLOTS)
		return fail(PT_E_STATE, "%s: timing ring full (call pt_kernel_times)", who);
	if(a->kernel_ms) HIP_TRY(hipEventRecord(c->evStart, st));
	if(ringTimed) HIP_TRY(hipEventRecord(c->ring[2 * c->nTimed], st));
	if(fast) pt_fast::launch_render(la); else pt_strict::launch_render(la);
	HIP_TRY(hipGetLastError());
	if(ringTimed) { HIP_TRY(hipEventRecord(c->ring[2 * c->nTimed + 1], st)); c->nTimed++; }
	if(a->kernel_ms) {
		HIP_TRY(hipEventRecord(c->evStop, st));
		HIP_TRY(hipEventSynchronize(c->evStop));
		HIP_TRY(hipEventElapsedTime(a->kernel_ms, c->evStart, c->evStop));
		return PT_OK;
	}
	if(!a->async) HIP_TRY(hipStreamSynchronize(st));
	return PT_OK;
}

extern "C" int pt_render(pt_ctx *c, const pt_render_args *a) {
	int rc = check_render_args(c, a, "pt_render");
	if(rc != PT_OK) return rc;
	float4 *img = a->d_image ? (float4 *)a->d_image : c->image;
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = a->stream ? (hipStream_t)a->stream : c->stream;
	if(a->n_frames == 0) return PT_OK;

	if(a->n_frames > c->capSeeds) {
		if(c->seedmul) { HIP_TRY(hipFree(c->seedmul)); c->seedmul = NULL; c->capSeeds = 0; }
		int cap = a->n_frames < 256 ? 256 : a->n_frames;
		HIP_TRY(hipMalloc((void **)&c->seedmul, sizeof(uint64_t) * (size_t)cap));
		c->capSeeds = cap;
		free(c->hostSeedmul); c->hostSeedmul = NULL; c->nHostSeeds = 0;
	}
	uint64_t *hs = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)a->n_frames);
	if(!hs) return fail(PT_E_NOMEM, "pt_render: host allocation failed%s", NULL);
	for(int k = 0; k < a->n_frames; k++) hs[k] = 16807ull * a->seeds[k];     // src/cl/rng.cl:8, folded per sample
	if(c->hostSeedmul && c->nHostSeeds == a->n_frames && memcmp(hs, c->hostSeedmul, sizeof(uint64_t) * (size_t)a->n_frames) == 0) {
		free(hs);   // the device table already holds these seeds: nothing to copy, no host synchronisation
	} else {
		// the table may still be read by an earlier launch on another stream: drain the context first
		hipError_t e0 = hipStreamSynchronize(c->stream);
		hipError_t e1 = st != c->stream ? hipStreamSynchronize(st) : hipSuccess;
		hipError_t ec = hipMemcpyAsync(c->seedmul, hs, sizeof(uint64_t) * (size_t)a->n_frames, hipMemcpyHostToDevice, st);
		hipError_t es = hipStreamSynchronize(st);   // pageable staging buffer: finish the copy before reusing it
		free(c->hostSeedmul);
		c->hostSeedmul = hs; c->nHostSeeds = a->n_frames;
		if(e0 != hipSuccess || e1 != hipSuccess || ec != hipSuccess || es != hipSuccess) c->nHostSeeds = 0;      // the table's content is unknown: never skip the next upload
		HIP_TRY(e0); HIP_TRY(e1); HIP_TRY(ec); HIP_TRY(es);
	}

	// opt-in acceleration structure (no reference counterpart): same closest hit, found through a uniform grid
	const bool accel = (c->sw.accel == 1 || (c->sw.accel == 0 && a->accel == 1)) && c->nSpheres > 0;
	if(accel && !c->gridBuilt) { rc = build_grid(c, st); if(rc != PT_OK) return rc; }
	const RenderScratch rs = {c->seedmul, c->counter, c->queue};
	return launch_frames(c, a, img, st, rs, accel, "pt_render");
}

// pt_render for callers that cannot allocate (include/pt_hip.h).  Same kernels, same image; the working memory is
// this GPU's `scratch` array, reached through the pointer pt_ctx_create looked up.  No hipMalloc / hipFree /
// malloc here or below: the seeds travel as kernel arguments, everything else is a launch, a memset or an event.
extern "C" int pt_render_noalloc(pt_ctx *c, const pt_render_args *a) {
	int rc = check_render_args(c, a, "pt_render_noalloc");
	if(rc != PT_OK) return rc;
	if(a->n_frames > PT_NOALLOC_MAX_FRAMES)
		return fail(PT_E_INVALID, "pt_render_noalloc: n_frames exceeds PT_NOALLOC_MAX_FRAMES (4096) — split the call or use pt_render%s", NULL);
	if(a->accel != 0) return fail(PT_E_INVALID, "pt_render_noalloc: accel=1 builds a grid, which allocates — use pt_render%s", NULL);
	if(!c->scratchPtr) return fail(PT_E_STATE, "pt_render_noalloc: no scratch address on this context%s", NULL);
	float4 *img = a->d_image ? (float4 *)a->d_image : c->image;
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = a->stream ? (hipStream_t)a->stream : c->stream;
	if(a->n_frames == 0) return PT_OK;
	// Under stream capture nothing may synchronise: only the enqueue-and-return form is legal there.
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	HIP_TRY(hipStreamIsCapturing(st, &cap));
	const bool capturing = cap != hipStreamCaptureStatusNone;
	if(capturing && (!a->async || a->bounce_count || a->kernel_ms || a->record_timing))
		return fail(PT_E_INVALID, "pt_render_noalloc: the stream is being captured — async must be 1 and bounce_count, kernel_ms, record_timing unset (they synchronise or record timing events)%s", NULL);
	// One call at a time per GPU on the shared scratch: same stream = ordered; another stream must have finished.
	std::unique_lock<std::mutex> inflight(g_noallocLock, std::defer_lock);
	NoallocSlot *slot = c->device < PT_MAX_DEVICES ? &g_noalloc[c->device] : NULL;
	if(slot && !capturing) {
		inflight.lock();
		if(slot->armed && slot->stream != st) {
			const hipError_t q = hipEventQuery(slot->ev);
			if(q == hipErrorNotReady)
				return fail(PT_E_STATE, "pt_render_noalloc: an earlier call on this GPU is still running on another stream; the calls share one scratch array (include/pt_hip.h)%s", NULL);
			if(q != hipSuccess) { (void)hipGetLastError(); slot->armed = false; }
		}
	}
	unsigned char *const sp = c->scratchPtr;      // the device address, never the array's name
	uint64_t *const dSeeds = (uint64_t *)(sp + PT_SCRATCH_SEEDS_OFF);
	for(int k0 = 0; k0 < a->n_frames; k0 += 256) {
		SeedChunk ch;
		const int cnt = a->n_frames - k0 < 256 ? a->n_frames - k0 : 256;
		for(int k = 0; k < cnt; k++) ch.v[k] = a->seeds[k0 + k];
		for(int k = cnt; k < 256; k++) ch.v[k] = 0;
		hipLaunchKernelGGL(pt_stage_seeds_kernel, dim3(1), dim3(256), 0, st, ch, cnt, dSeeds + k0);
		HIP_TRY(hipGetLastError());
	}
	const RenderScratch rs = {dSeeds, (unsigned long long *)(sp + PT_SCRATCH_COUNTER_OFF), (unsigned int *)(sp + PT_SCRATCH_QUEUE_OFF)};
	rc = launch_frames(c, a, img, st, rs, false, "pt_render_noalloc");
	if(slot && !capturing) {
		if(hipEventRecord(slot->ev, st) == hipSuccess) { slot->stream = st; slot->armed = true; }
		else { (void)hipGetLastError(); slot->armed = false; }
	}
	return rc;
}

// ---------------------------------------------------------------------------------------
// Multi-GPU combine (include/pt_hip.h pt_combine_bands, pt_combine_bands_rank): the root's accumulator receives
// every other rank's row bands, each band exactly once, as ONE message per peer: a rank's bands are packed into a
// compact staging buffer in the order the render kernel itself numbers its pixels (pixel_id(): local pixel t ->
// global work-item id), travel as one RCCL send/recv pair over the peer's own xGMI link, and are unpacked into the
// root's accumulator through the same map.  Bytes moved = the image minus the root's own bands, in world - 1
// messages.  librccl is loaded on first use (dlopen), so a single-GPU user never maps it; where it cannot be set
// up, one process driving several GPUs falls back to peer copies of the same packed buffers.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_bands_kernel(float4 *__restrict__ image, float4 *__restrict__ stage, const KParams p, const int unpack) {
	const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int id = pixel_id(p, t);      // -1 past this rank's pixels
	if(id < 0) return;
	if(unpack) image[id] = stage[t];
	else stage[t] = image[id];
}

// pack (image -> stage) or unpack (stage -> image) the bands of `rank` on stream st; returns the pixel count
static int bands_copy(float4 *image, float4 *stage, int W, int H, int rank, int world, int band, int unpack, hipStream_t st) {
	KParams p;
	memset(&p, 0, sizeof p);
	p.W = W; p.H = H; p.rank = rank; p.world = world; p.bandRows = band;
	const long long lp = local_rows64(H, rank, world, band) * W;
	p.localPixels = (int)lp;
	if(lp == 0) return PT_OK;
	hipLaunchKernelGGL(pt_bands_kernel, dim3((unsigned)((lp + 255) / 256)), dim3(256), 0, st, image, stage, p, unpack);
	HIP_TRY(hipGetLastError());
	return PT_OK;
}

static int grow_stage(pt_ctx *c, size_t pixels) {
	if(pixels <= c->capStage) return PT_OK;
	if(c->stage) { HIP_TRY(hipFree(c->stage)); c->stage = NULL; c->capStage = 0; }
	HIP_TRY(hipMalloc((void **)&c->stage, sizeof(float4) * pixels));
	c->capStage = pixels;
	return PT_OK;
}

namespace {
struct RcclApi {
	void *lib;
	decltype(&ncclCommInitAll) commInitAll;
	decltype(&ncclCommInitRank) commInitRank;
	decltype(&ncclGetUniqueId) getUniqueId;
	decltype(&ncclCommDestroy) commDestroy;
	decltype(&ncclGroupStart) groupStart;
	decltype(&ncclGroupEnd) groupEnd;
	decltype(&ncclSend) send;
	decltype(&ncclRecv) recv;
	decltype(&ncclGetErrorString) errorString;
};
RcclApi g_rccl;
std::mutex g_rcclLock;
struct CommSet { int n; int dev[8]; ncclComm_t comm[8]; };
std::vector<CommSet> g_comms;      // one set per distinct device list, kept for the life of the process
thread_local const char *g_transport = "";

int rccl_load() {
	if(g_rccl.lib) return PT_OK;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *lib = NULL;
	for(const char *nm : names) if((lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
	if(!lib) return fail(PT_E_DEVICE, "cannot load librccl: %s", dlerror());
	RcclApi a;
	a.lib = lib;
	a.commInitAll = (decltype(a.commInitAll))dlsym(lib, "ncclCommInitAll");
	a.commInitRank = (decltype(a.commInitRank))dlsym(lib, "ncclCommInitRank");
	a.getUniqueId = (decltype(a.getUniqueId))dlsym(lib, "ncclGetUniqueId");
	a.commDestroy = (decltype(a.commDestroy))dlsym(lib, "ncclCommDestroy");
	a.groupStart = (decltype(a.groupStart))dlsym(lib, "ncclGroupStart");
	a.groupEnd = (decltype(a.groupEnd))dlsym(lib, "ncclGroupEnd");
	a.send = (decltype(a.send))dlsym(lib, "ncclSend");
	a.recv = (decltype(a.recv))dlsym(lib, "ncclRecv");
	a.errorString = (decltype(a.errorString))dlsym(lib, "ncclGetErrorString");
	if(!a.commInitAll || !a.commInitRank || !a.getUniqueId || !a.commDestroy || !a.groupStart || !a.groupEnd || !a.send || !a.recv || !a.errorString) {
		dlclose(lib);
		return fail(PT_E_DEVICE, "librccl lacks a symbol%s", NULL);
	}
	g_rccl = a;
	return PT_OK;
}

// the communicators of one process driving the GPUs of ctxs[0..n): created once per device list
CommSet *comm_set(pt_ctx *const *ctxs, int n) {
	for(auto &c : g_comms) {
		bool same = c.n == n;
		for(int i = 0; same && i < n; i++) same = c.dev[i] == ctxs[i]->device;
		if(same) return &c;
	}
	CommSet c;
	c.n = n;
	for(int i = 0; i < n; i++) c.dev[i] = ctxs[i]->device;
	const ncclResult_t r = g_rccl.commInitAll(c.comm, n, c.dev);
	if(r != ncclSuccess) { snprintf(g_err, sizeof g_err, "ncclCommInitAll failed: %s", g_rccl.errorString(r)); return NULL; }
	g_comms.push_back(c);
	return &g_comms.back();
}
}   // namespace

#define RCCL_TRY(expr)                                                                               \
	do {                                                                                             \
		ncclResult_t r_ = (expr);                                                                    \
		if(r_ != ncclSuccess) {                                                                      \
			snprintf(g_err, sizeof g_err, "%s failed: %s", #expr, g_rccl.errorString(r_));           \
			return PT_E_DEVICE;                                                                      \
		}                                                                                            \
	} while(0)

extern "C" const char *pt_combine_transport(void) { return g_transport; }

static void pt_comm_release(pt_ctx *c) {
	if(c->mpComm && g_rccl.lib) (void)g_rccl.commDestroy((ncclComm_t)c->mpComm);
	c->mpComm = NULL;
}

extern "C" int pt_combine_bands(pt_ctx *const *ctxs, int n, int root, int band_rows) {
	if(!ctxs || n < 1 || n > 8 || root < 0 || root >= n || band_rows < 1) return fail(PT_E_INVALID, "pt_combine_bands: bad arguments%s", NULL);
	for(int i = 0; i < n; i++) {
		if(!ctxs[i] || !ctxs[i]->image) return fail(PT_E_STATE, "pt_combine_bands: a context has no accumulator (pt_image_alloc)%s", NULL);
		if(ctxs[i]->imgW != ctxs[root]->imgW || ctxs[i]->imgH != ctxs[root]->imgH) return fail(PT_E_STATE, "pt_combine_bands: accumulators differ in size%s", NULL);
	}
	g_transport = "none";
	if(n == 1) return PT_OK;
	pt_ctx *const R = ctxs[root];
	const int W = R->imgW, H = R->imgH;
	bool anyRemote = false, distinct = true;
	for(int i = 0; i < n; i++) {
		if(ctxs[i]->device != R->device) anyRemote = true;
		for(int j = 0; j < i; j++) if(ctxs[i]->device == ctxs[j]->device) distinct = false;
	}
	if(anyRemote && !distinct) return fail(PT_E_INVALID, "pt_combine_bands: ranks must be on distinct GPUs, or all on one%s", NULL);
	// 1. every peer packs its bands on its own stream (ordered after its render); the root makes room for all of them
	size_t cnt[8], off[8], total = 0;
	for(int r = 0; r < n; r++) {
		cnt[r] = (size_t)(local_rows64(H, r, n, band_rows) * W);
		off[r] = total;
		if(r != root) total += cnt[r];
	}
	int rc;
	for(int r = 0; r < n; r++) {
		if(r == root || cnt[r] == 0) continue;
		HIP_TRY(hipSetDevice(ctxs[r]->device));
		if((rc = grow_stage(ctxs[r], cnt[r])) != PT_OK) return rc;
		if((rc = bands_copy(ctxs[r]->image, ctxs[r]->stage, W, H, r, n, band_rows, 0, ctxs[r]->stream)) != PT_OK) return rc;
	}
	HIP_TRY(hipSetDevice(R->device));
	if((rc = grow_stage(R, total)) != PT_OK) return rc;
	// 2. one message per peer.  Distinct GPUs: RCCL send/recv pairs in one group (each peer over its own xGMI link); if
	// RCCL cannot be set up (library, symbols, communicators) peer copies of the same buffers, ordered by events, take
	// its place (PT_COMBINE=copy asks for them, PT_COMBINE=rccl forbids the fallback).  One GPU (PT_SHARE_GPU debugging
	// set-ups, several contexts on one device, which cannot be RCCL peers): device copies.
	bool useRccl = anyRemote && R->sw.combine != 1;
	if(useRccl) {
		std::lock_guard<std::mutex> lock(g_rcclLock);
		CommSet *cs = rccl_load() == PT_OK ? comm_set(ctxs, n) : NULL;
		if(!cs) {
			if(R->sw.combine == 2) return PT_E_DEVICE;      // g_err holds the reason
			fprintf(stderr, "pt_combine_bands: RCCL unavailable (%s); using peer copies\n", g_err);
			useRccl = false;
		} else {
			RCCL_TRY(g_rccl.groupStart());
			for(int r = 0; r < n; r++) {
				if(r == root || cnt[r] == 0) continue;
				// each call with its communicator's device current (one process drives all ranks)
				hipError_t d1 = hipSetDevice(ctxs[r]->device);
				ncclResult_t e1 = g_rccl.send(ctxs[r]->stage, cnt[r] * 4, ncclFloat, root, cs->comm[r], ctxs[r]->stream);
				hipError_t d2 = hipSetDevice(R->device);
				ncclResult_t e2 = g_rccl.recv(R->stage + off[r], cnt[r] * 4, ncclFloat, r, cs->comm[root], R->stream);
				if(d1 != hipSuccess || d2 != hipSuccess) { (void)g_rccl.groupEnd(); return fail(PT_E_DEVICE, "pt_combine_bands: hipSetDevice failed%s", NULL); }
				if(e1 != ncclSuccess || e2 != ncclSuccess) {
					(void)g_rccl.groupEnd();
					snprintf(g_err, sizeof g_err, "pt_combine_bands: ncclSend/ncclRecv failed: %s", g_rccl.errorString(e1 != ncclSuccess ? e1 : e2));
					return PT_E_DEVICE;
				}
			}
			RCCL_TRY(g_rccl.groupEnd());
			g_transport = "rccl";
		}
	}
	if(!useRccl) {
		for(int r = 0; r < n; r++) {
			if(r == root || cnt[r] == 0) continue;
			HIP_TRY(hipSetDevice(ctxs[r]->device));
			HIP_TRY(hipEventRecord(ctxs[r]->evStop, ctxs[r]->stream));
			HIP_TRY(hipSetDevice(R->device));
			HIP_TRY(hipStreamWaitEvent(R->stream, ctxs[r]->evStop, 0));
			if(ctxs[r]->device == R->device)
				HIP_TRY(hipMemcpyAsync(R->stage + off[r], ctxs[r]->stage, cnt[r] * sizeof(float4), hipMemcpyDeviceToDevice, R->stream));
			else
				HIP_TRY(hipMemcpyPeerAsync(R->stage + off[r], R->device, ctxs[r]->stage, ctxs[r]->device, cnt[r] * sizeof(float4), R->stream));
		}
		g_transport = anyRemote ? "peer-copy" : "device-copy";
	}
	// 3. the root scatters every peer's bands into its accumulator (its stream: after the receives)
	HIP_TRY(hipSetDevice(R->device));
	for(int r = 0; r < n; r++) {
		if(r == root || cnt[r] == 0) continue;
		if((rc = bands_copy(R->image, R->stage + off[r], W, H, r, n, band_rows, 1, R->stream)) != PT_OK) return rc;
	}
	return PT_OK;
}

// ---- the same combine for one process PER GPU (bench.py --gpus N under torch.distributed.run) -------------------
extern "C" int pt_comm_unique_id(void *id, int bytes) {
	if(!id || bytes < (int)sizeof(ncclUniqueId)) return fail(PT_E_INVALID, "pt_comm_unique_id: need a buffer of PT_COMM_ID_BYTES%s", NULL);
	std::lock_guard<std::mutex> lock(g_rcclLock);
	int rc = rccl_load();
	if(rc != PT_OK) return rc;
	ncclUniqueId u;
	RCCL_TRY(g_rccl.getUniqueId(&u));
	memset(id, 0, (size_t)bytes);
	memcpy(id, &u, sizeof u);
	return PT_OK;
}

extern "C" int pt_comm_init_rank(pt_ctx *c, int world, int rank, const void *id) {
	if(!c || !id || world < 1 || rank < 0 || rank >= world) return fail(PT_E_INVALID, "pt_comm_init_rank: bad arguments%s", NULL);
	std::lock_guard<std::mutex> lock(g_rcclLock);
	int rc = rccl_load();
	if(rc != PT_OK) return rc;
	HIP_TRY(hipSetDevice(c->device));
	if(c->mpComm) { (void)g_rccl.commDestroy((ncclComm_t)c->mpComm); c->mpComm = NULL; }
	ncclUniqueId u;
	memcpy(&u, id, sizeof u);
	ncclComm_t comm;
	RCCL_TRY(g_rccl.commInitRank(&comm, world, u, rank));
	c->mpComm = comm; c->mpWorld = world; c->mpRank = rank;
	return PT_OK;
}

extern "C" int pt_combine_bands_rank(pt_ctx *c, int root, int band_rows, void *d_image, int width, int height, void *stream) {
	if(!c || band_rows < 1) return fail(PT_E_INVALID, "pt_combine_bands_rank: bad arguments%s", NULL);
	if(!c->mpComm) return fail(PT_E_STATE, "pt_combine_bands_rank: no communicator (pt_comm_init_rank)%s", NULL);
	const int n = c->mpWorld, me = c->mpRank;
	if(root < 0 || root >= n) return fail(PT_E_INVALID, "pt_combine_bands_rank: bad root%s", NULL);
	float4 *img = (float4 *)d_image;
	if(!img) {
		if(!c->image) return fail(PT_E_STATE, "pt_combine_bands_rank: no accumulator (call pt_image_alloc)%s", NULL);
		img = c->image; width = c->imgW; height = c->imgH;
	}
	if(width <= 0 || height <= 0 || (long long)width * height > (1ll << 30)) return fail(PT_E_INVALID, "pt_combine_bands_rank: bad size%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	hipStream_t st = stream ? (hipStream_t)stream : c->stream;
	g_transport = n == 1 ? "none" : "rccl";
	if(n == 1) return PT_OK;
	int rc;
	if(me != root) {
		const size_t cnt = (size_t)(local_rows64(height, me, n, band_rows) * width);
		if(cnt == 0) return PT_OK;
		if((rc = grow_stage(c, cnt)) != PT_OK) return rc;
		if((rc = bands_copy(img, c->stage, width, height, me, n, band_rows, 0, st)) != PT_OK) return rc;
		RCCL_TRY(g_rccl.send(c->stage, cnt * 4, ncclFloat, root, (ncclComm_t)c->mpComm, st));
		return PT_OK;
	}
	size_t cnt[64], off[64], total = 0;
	if(n > 64) return fail(PT_E_INVALID, "pt_combine_bands_rank: more than 64 ranks%s", NULL);
	for(int r = 0; r < n; r++) {
		cnt[r] = (size_t)(local_rows64(height, r, n, band_rows) * width);
		off[r] = total;
		if(r != root) total += cnt[r];
	}
	if((rc = grow_stage(c, total)) != PT_OK) return rc;
	RCCL_TRY(g_rccl.groupStart());
	for(int r = 0; r < n; r++) {
		if(r == root || cnt[r] == 0) continue;
		const ncclResult_t e = g_rccl.recv(c->stage + off[r], cnt[r] * 4, ncclFloat, r, (ncclComm_t)c->mpComm, st);
		if(e != ncclSuccess) { (void)g_rccl.groupEnd(); snprintf(g_err, sizeof g_err, "pt_combine_bands_rank: ncclRecv failed: %s", g_rccl.errorString(e)); return PT_E_DEVICE; }
	}
	RCCL_TRY(g_rccl.groupEnd());
	for(int r = 0; r < n; r++) {
		if(r == root || cnt[r] == 0) continue;
		if((rc = bands_copy(img, c->stage + off[r], width, height, r, n, band_rows, 1, st)) != PT_OK) return rc;
	}
	return PT_OK;
}

extern "C" int pt_kernel_times(pt_ctx *c, float *ms, int cap, int *n) {
	if(!c || !n || (cap > 0 && !ms)) return fail(PT_E_INVALID, "pt_kernel_times: NULL%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	int k = 0;
	for(; k < c->nTimed && k < cap; k++) {
		HIP_TRY(hipEventSynchronize(c->ring[2 * k + 1]));
		HIP_TRY(hipEventElapsedTime(&ms[k], c->ring[2 * k], c->ring[2 * k + 1]));
	}
	*n = k;
	c->nTimed = 0;
	return PT_OK;
}

extern "C" int pt_sync(pt_ctx *c) {
	if(!c) return fail(PT_E_INVALID, "pt_sync: NULL%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return PT_OK;
}

extern "C" int pt_image_read(pt_ctx *c, float *host_rgba) {
	if(!c || !host_rgba) return fail(PT_E_INVALID, "pt_image_read: NULL%s", NULL);
	if(!c->image) return fail(PT_E_STATE, "pt_image_read: no accumulator (call pt_image_alloc)%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipMemcpyAsync(host_rgba, c->image, sizeof(float4) * (size_t)c->imgW * c->imgH, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return PT_OK;
}

extern "C" int pt_image_read_u8(pt_ctx *c, const void *d_image, int width, int height, unsigned char *host_rgb) {
	if(!c || !host_rgb) return fail(PT_E_INVALID, "pt_image_read_u8: NULL%s", NULL);
	const float4 *img = (const float4 *)d_image;
	if(!img) {
		if(!c->image) return fail(PT_E_STATE, "pt_image_read_u8: no accumulator (call pt_image_alloc)%s", NULL);
		img = c->image; width = c->imgW; height = c->imgH;
	}
	if(width <= 0 || height <= 0) return fail(PT_E_INVALID, "pt_image_read_u8: bad size%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	size_t pixels = (size_t)width * height, bytes = 3 * pixels;
	if(bytes > c->capU8) {
		if(c->u8) { HIP_TRY(hipFree(c->u8)); c->u8 = NULL; c->capU8 = 0; }
		HIP_TRY(hipMalloc((void **)&c->u8, bytes));
		c->capU8 = bytes;
	}
	hipLaunchKernelGGL(pt_to_u8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, c->stream, img, (int)pixels, c->u8);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(host_rgb, c->u8, bytes, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return PT_OK;
}

extern "C" int pt_selftest(pt_ctx *c, int op, const uint64_t *a, const uint64_t *b, const float *in_f, int n,
                           uint64_t *out_u64, float *out_f) {
	if(!c || n <= 0 || op < 0 || op > 12) return fail(PT_E_INVALID, "pt_selftest: bad arguments%s", NULL);
	HIP_TRY(hipSetDevice(c->device));
	uint64_t *da = NULL, *db = NULL, *dou = NULL;
	float *dif = NULL, *dof = NULL;
	size_t nu = sizeof(uint64_t) * (size_t)n;
	int rc = PT_OK;
	hipError_t e;
#define ST_TRY(x) do { e = (x); if(e != hipSuccess) { snprintf(g_err, sizeof g_err, "%s failed: %s", #x, hipGetErrorString(e)); rc = PT_E_DEVICE; goto done; } } while(0)
	ST_TRY(hipMalloc((void **)&da, nu));
	ST_TRY(hipMalloc((void **)&db, nu));
	ST_TRY(hipMalloc((void **)&dou, nu));
	ST_TRY(hipMalloc((void **)&dif, sizeof(float) * 2 * (size_t)n));
	ST_TRY(hipMalloc((void **)&dof, sizeof(float) * 4 * (size_t)n));
	// on the stream the kernel runs on (a non-blocking stream is not ordered after the null stream's work)
	ST_TRY(hipMemsetAsync(dou, 0, nu, c->stream));
	ST_TRY(hipMemsetAsync(dof, 0, sizeof(float) * 4 * (size_t)n, c->stream));
	if(a) ST_TRY(hipMemcpy(da, a, nu, hipMemcpyHostToDevice));
	if(b) ST_TRY(hipMemcpy(db, b, nu, hipMemcpyHostToDevice));
	if(in_f) ST_TRY(hipMemcpy(dif, in_f, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(pt_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, op, da, db, dif, n, dou, dof);
	ST_TRY(hipGetLastError());
	ST_TRY(hipStreamSynchronize(c->stream));
	if(out_u64) ST_TRY(hipMemcpy(out_u64, dou, nu, hipMemcpyDeviceToHost));
	if(out_f) ST_TRY(hipMemcpy(out_f, dof, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost));
done:
	(void)hipFree(da); (void)hipFree(db); (void)hipFree(dou); (void)hipFree(dif); (void)hipFree(dof);
	return rc;
#undef ST_TRY
}
