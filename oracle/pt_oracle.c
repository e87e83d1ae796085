/* TEST INFRASTRUCTURE — CPU oracle for the path-tracing hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product path (pathtracer_amd/csrc) never does.
 *
 * A from-scratch scalar C restatement of the reference's per-pixel kernel, with every
 * quirk listed in SURVEY.md §8a kept.  Each function cites the reference lines it
 * follows (paths relative to /root/reference).  Float semantics: strict IEEE-754
 * binary32, operations in the reference's source order, no FMA contraction (build
 * with -ffp-contract=off), OpenCL builtins as defined in oracle/ref_shim.cpp.
 *
 * Parity pin: bit-exact against oracle/_ref/libptref_strict.so (the reference kernel
 * compiled verbatim for x86 without fast-math flags) and within 1e-5 of
 * libptref_fast.so (the reference's own -cl-fast-relaxed-math flags) — checked by
 * tests/test_oracle_golden.py (test_oracle_vs_live_reference_build) here and frozen into tests/golden/ for the GPU box.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PT_M 2147483647ULL

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* OpenCL builtins, as oracle/ref_shim.cpp defines them */
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { float inv = 1.0f / length3(a); return scale(a, inv); }
static inline float pow_f(float x, float y) { return (float)pow((double)x, (double)y); }

/* src/cl/rng.cl:3-5 — left-assoc `%` and `*`, 64-bit wraparound kept */
uint64_t pto_init_rng_1(uint64_t a) {
	uint64_t x = 16807ULL * a;
	return ((x % PT_M) * x) % PT_M;
}

/* src/cl/rng.cl:7-9 */
uint64_t pto_init_rng_2(uint64_t a, uint64_t b) {
	uint64_t x = 16807ULL * a * b;
	return ((x % PT_M) * x) % PT_M;
}

/* src/cl/rng.cl:11-14 — f64 divide, then narrowed to f32 by the return */
float pto_rand_float(uint64_t *rng) {
	*rng = (16807ULL * *rng) % PT_M;
	return (float)((double)*rng / 2147483647.0);
}

/* src/cl/rng.cl:16-18 */
float pto_rand_float_in_range(float a, float b, uint64_t *rng) {
	return a + pto_rand_float(rng) * (b - a);
}

/* src/cl/rng.cl:20-34 — not uniform on the sphere (y >= 0 always); 2 draws, theta first */
static v3 random_unit_vector(uint64_t *rng) {
	float cosTheta = pto_rand_float_in_range(-1, 1, rng);
	float cosPhi = pto_rand_float_in_range(-1, 1, rng);
	float sinTheta = sqrtf(1 - cosTheta * cosTheta);
	float sinPhi = sqrtf(1 - cosPhi * cosPhi);
	return V(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}

/* src/cl/ray.cl:39-52 — only the last assignment of each component survives; all
 * read the ORIGINAL vec.  rot.z is dead. */
static v3 rotate_vector(v3 vec, v3 rot) {
	v3 n;
	n.x = vec.x * cosf(rot.y) + vec.z * sinf(rot.y);
	n.y = vec.y * cosf(rot.x) - vec.z * sinf(rot.x);
	n.z = vec.y * sinf(rot.x) + vec.z * cosf(rot.x);
	return n;
}

/* src/cl/ray.cl:54-56 — (normal*2)*dot, in that order */
static v3 reflection_dir(v3 in, v3 normal) {
	return sub(in, scale(scale(normal, 2), dot3(in, normal)));
}

/* src/cl/ray.cl:58-62 */
static float reflectance(float cosTheta, float rel) {
	float r = (1 - rel) / (1 + rel);
	r = r * r;
	return r + (1 - r) * pow_f(1 - cosTheta, 5);
}

/* src/cl/ray.cl:64-69 */
static v3 refraction_dir(v3 in, v3 normal, float rel) {
	float cosTheta = fminf(dot3(scale(in, -1), normal), 1);
	v3 perp = scale(add(in, scale(normal, cosTheta)), rel);
	v3 par = scale(normal, -sqrtf(fabsf(1 - length3(perp) * length3(perp))));
	return add(par, perp);
}

/* src/cl/ray.cl:73 — a = pow(length(d),2): depends on the ray only, so the bounce loop
 * computes it once per ray; the sqrt-then-square rounding is kept. */
static inline float ray_a(v3 d) { return pow_f(length3(d), 2); }

/* src/cl/ray.cl:71-87 — near root only */
static int hit_sphere(v3 o, v3 d, float a, v3 center, float radius, float *distance) {
	v3 oc = sub(o, center);
	float bHalf = dot3(oc, d);
	float c = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z - radius * radius;
	float disc = bHalf * bHalf - a * c;
	if(disc < 0) return 0;
	*distance = (-bHalf - sqrtf(disc)) / a;
	return *distance > 0 ? 1 : 0;
}

/* interchange sphere record: cx cy cz r | type cr cg cb | tint fuzz refIdx pad */
typedef struct { v3 center; float radius; int type; v3 color; float tint, fuzz, refIdx; } sphere_t;

static void load_spheres(sphere_t *out, const float *s, int n) {
	for(int i = 0; i < n; i++) {
		const float *r = s + 12 * (size_t)i;
		out[i].center = V(r[0], r[1], r[2]); out[i].radius = r[3];
		out[i].type = (int)r[4]; out[i].color = V(r[5], r[6], r[7]);
		out[i].tint = r[8]; out[i].fuzz = r[9]; out[i].refIdx = r[10];
	}
}

/* src/cl/ray.cl:89-175.  *bounces (optional) += bounce-loop iterations executed. */
static v3 cast_ray(v3 bg, const sphere_t *sl, int n, v3 o, v3 d, int maxDepth, uint64_t *rng, uint64_t *bounces) {
	v3 color = V(0, 0, 0), mask = V(1, 1, 1);
	for(int j = 0; j < maxDepth; j++) {
		int hit = 0, idx = -1;
		float distance = 0;
		if(bounces) (*bounces)++;
		float a = ray_a(d);
		for(int i = 0; i < n; i++) {                       /* :101-110, strict < keeps the lowest index */
			float nd;
			if(hit_sphere(o, d, a, sl[i].center, sl[i].radius, &nd)) {
				if(!hit || nd < distance) { hit = 1; distance = nd; idx = i; }
			}
		}
		if(!hit) { color = mul(mask, bg); break; }          /* :112-115 */
		const sphere_t *sp = &sl[idx];
		v3 hitPos = add(o, scale(d, distance));             /* :117 */
		v3 normal = normalize3(sub(hitPos, sp->center));    /* :118 */
		v3 offPos = sub(hitPos, scale(normal, 0.0001f));    /* :119 — nudged INTO the sphere */
		int front;
		if(dot3(d, normal) > 0) { front = 0; normal = scale(normal, -1); } else front = 1;   /* :122-128 */
		if(sp->type == 0) {                                 /* :130-134 light */
			if(front) { color = mul(mask, sp->color); break; }
		} else if(sp->type == 1) {                          /* :136-141 lambert, direction un-normalised */
			if(front) {
				v3 nd = add(normal, random_unit_vector(rng));
				o = offPos; d = nd;
				mask = mul(mask, sp->color);
			}
		} else if(sp->type == 2) {                          /* :143-148 metal */
			if(front) {
				v3 nd = add(reflection_dir(d, normal), scale(random_unit_vector(rng), sp->fuzz));
				o = offPos; d = nd;
				mask = add(scale(mask, 1 - sp->tint), scale(mul(mask, sp->color), sp->tint));
			}
		} else if(sp->type == 3) {                          /* :150-171 dielectric */
			float rel = front ? sp->refIdx : 1 / sp->refIdx;
			float cosTheta = fminf(dot3(scale(d, -1), normal), 1);
			float sinTheta = sqrtf(1 - cosTheta * cosTheta);
			v3 nd;
			/* short-circuit: no RNG draw when rel*sinTheta > 1 (:163) */
			if(rel * sinTheta > 1 || reflectance(cosTheta, rel) > pto_rand_float(rng))
				nd = add(reflection_dir(d, normal), scale(random_unit_vector(rng), sp->fuzz));
			else
				nd = add(refraction_dir(d, normal, rel), scale(random_unit_vector(rng), sp->fuzz));
			o = offPos; d = nd;
			mask = add(scale(mask, 1 - sp->tint), scale(mul(mask, sp->color), sp->tint));
		}
	}
	return color;
}

typedef struct {
	v3 pos, rot; float sensorWidth, focalLength, aperture, exposure;
} camera_t;

/* src/cl/main.cl:4-30 — one work-item, one sample */
static void kernel_main(float *image, int W, int H, int maxDepth, v3 bg, const camera_t *cam,
                        const sphere_t *sl, int n, int frame, uint64_t seed, long id, uint64_t *bounces) {
	uint64_t rng = pto_init_rng_2((uint64_t)id, seed);                                   /* :6 */
	int row = (int)id / W, column = (int)id % W;                                         /* :8-9 */
	float aspect = (float)W / (float)H;                                                  /* :11 */
	float xOff = 2 * (float)(column - W / 2) / (float)W * cam->sensorWidth;              /* :13 */
	float yOff = 2 * (float)(row - H / 2) / (float)H * cam->sensorWidth / aspect;        /* :14 */
	v3 offset = V(-xOff, yOff, cam->focalLength);                                        /* :15 */
	v3 origin = add(cam->pos, rotate_vector(offset, cam->rot));                          /* :17 */
	v3 target = add(cam->pos, scale(random_unit_vector(&rng), cam->aperture));           /* :19 */
	v3 dir = normalize3(sub(target, origin));                                            /* :21 */
	v3 color = cast_ray(bg, sl, n, cam->pos, dir, maxDepth, &rng, bounces);              /* :23-25 ray starts at pos */
	color = scale(scale(color, cam->exposure), cam->aperture);                           /* :27 */
	float *px = image + 4 * id;                                                          /* 16-byte float3 */
	float f = (float)frame, f1 = (float)(frame + 1);
	px[0] = (px[0] * f + color.x) / f1;                                                  /* :29 */
	px[1] = (px[1] * f + color.y) / f1;
	px[2] = (px[2] * f + color.z) / f1;
}

typedef struct {
	float *image; int W, H, maxDepth; v3 bg; camera_t cam; const sphere_t *sl; int n;
	const uint64_t *seeds; int first_frame, nframes; long id_begin, id_end; uint64_t bounces;
} job_t;

static void *worker(void *arg) {
	job_t *j = (job_t *)arg;
	for(long id = j->id_begin; id < j->id_end; id++)
		for(int k = 0; k < j->nframes; k++)
			kernel_main(j->image, j->W, j->H, j->maxDepth, j->bg, &j->cam, j->sl, j->n,
			            j->first_frame + k, j->seeds[k], id, &j->bounces);
	return NULL;
}

/* The host loop of src/c/renderer.c:153-262 over work-items id_begin..id_end-1 and
 * frames first_frame..first_frame+nframes-1 (seeds[k] is the seed of frame first_frame+k).
 * image: W*H*4 floats; frame 0 overwrites (stale*0 — callers pass a zeroed buffer).
 * total_bounces (optional): bounce-loop iterations over all samples. */
int pto_render(float *image, int W, int H, int maxDepth, const float *bg, const float *cam,
               const float *spheres, int n, const uint64_t *seeds, int first_frame, int nframes,
               long id_begin, long id_end, int nthreads, uint64_t *total_bounces) {
	sphere_t *sl = (sphere_t *)malloc(sizeof(sphere_t) * (size_t)(n > 0 ? n : 1));
	load_spheres(sl, spheres, n);
	job_t base;
	memset(&base, 0, sizeof base);
	base.image = image; base.W = W; base.H = H; base.maxDepth = maxDepth;
	base.bg = V(bg[0], bg[1], bg[2]);
	base.cam.pos = V(cam[0], cam[1], cam[2]); base.cam.rot = V(cam[3], cam[4], cam[5]);
	base.cam.sensorWidth = cam[6]; base.cam.focalLength = cam[7]; base.cam.aperture = cam[8]; base.cam.exposure = cam[9];
	base.sl = sl; base.n = n; base.seeds = seeds; base.first_frame = first_frame; base.nframes = nframes;
	if(nthreads < 1) nthreads = 1;
	if(nthreads > 256) nthreads = 256;
	/* contiguous id ranges, sized in 64-pixel granules */
	long total = id_end - id_begin;
	long per = (total + nthreads - 1) / nthreads;
	job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)nthreads);
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
	for(int t = 0; t < nthreads; t++) {
		jobs[t] = base;
		long b = id_begin + per * t, e = b + per;
		if(b > id_end) b = id_end;
		if(e > id_end) e = id_end;
		jobs[t].id_begin = b; jobs[t].id_end = e;
		pthread_create(&th[t], NULL, worker, &jobs[t]);
	}
	uint64_t tb = 0;
	for(int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); tb += jobs[t].bounces; }
	if(total_bounces) *total_bounces = tb;
	free(jobs); free(th); free(sl);
	return 0;
}

/* ---- single-function entry points for known-answer tests ---- */
void pto_random_unit_vector(uint64_t *state, float *out) { v3 r = random_unit_vector(state); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void pto_rotate_vector(const float *vec, const float *rot, float *out) {
	v3 r = rotate_vector(V(vec[0], vec[1], vec[2]), V(rot[0], rot[1], rot[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void pto_reflection_dir(const float *in, const float *nrm, float *out) {
	v3 r = reflection_dir(V(in[0], in[1], in[2]), V(nrm[0], nrm[1], nrm[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float pto_reflectance(float cosTheta, float rel) { return reflectance(cosTheta, rel); }
void pto_refraction_dir(const float *in, const float *nrm, float rel, float *out) {
	v3 r = refraction_dir(V(in[0], in[1], in[2]), V(nrm[0], nrm[1], nrm[2]), rel); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int pto_hit_sphere(const float *o, const float *d, const float *c, float radius, float *t) {
	v3 dv = V(d[0], d[1], d[2]);
	return hit_sphere(V(o[0], o[1], o[2]), dv, ray_a(dv), V(c[0], c[1], c[2]), radius, t);
}
void pto_cast_ray(const float *bg, const float *spheres, int n, const float *o, const float *d,
                  int maxDepth, uint64_t *state, float *out) {
	sphere_t *sl = (sphere_t *)malloc(sizeof(sphere_t) * (size_t)(n > 0 ? n : 1));
	load_spheres(sl, spheres, n);
	v3 r = cast_ray(V(bg[0], bg[1], bg[2]), sl, n, V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), maxDepth, state, NULL);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
	free(sl);
}

/* src/c/renderer.c:50-56 + :17-39 — gamma 2 (sqrt), clamp to [0,1], (int)(x*255) truncation.
 * in: W*H*4 floats (not modified), out: W*H*3 bytes RGB row-major. */
void pto_postprocess(const float *image, int pixels, unsigned char *out) {
	for(int i = 0; i < pixels; i++)
		for(int c = 0; c < 3; c++) {
			float x = (float)sqrt((double)image[4 * (size_t)i + c]);   /* sqrt() is the double one in C (:52) */
			if(x < 0) x = 0;
			if(x > 1) x = 1;
			out[3 * (size_t)i + c] = (unsigned char)(int)(x * 255);
		}
}

/* ---- exhaustive-check helpers (test infrastructure for tests/test_rng_exhaustive.py, tests/test_pow5_exhaustive.py
 * and the GPU pow5 test).  Nothing below restates reference code; these are arbiters. ---- */

/* States lo <= s < hi: how often does rand_float's narrowed float, computed with the reference's f64 DIVIDE
 * (pto_rand_float above, src/cl/rng.cl:13), differ from the same state times RN(1/2147483647.0) — the form the HIP
 * kernels use (pathtracer_amd/csrc/pt_device.h rand_float).  `s` is the state AFTER the update, i.e. the value
 * that is converted. */
uint64_t pto_rand_float_mul_mismatches(uint64_t lo, uint64_t hi) {
	const double inv = 1.0 / 2147483647.0;
	uint64_t bad = 0;
	for(uint64_t s = lo; s < hi; s++) {
		float a = (float)((double)s / 2147483647.0);
		float b = (float)((double)s * inv);
		uint32_t ua, ub;
		memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
		bad += ua != ub;
	}
	return bad;
}

/* The correctly rounded binary32 value of x^5, by exact integer arithmetic (a 24-bit significand to the fifth power
 * has at most 120 bits: unsigned __int128), round-to-nearest-even into normal or denormal binary32. */
typedef unsigned __int128 u128_t;
static float pow5_exact(float x) {
	if(x == 0 || isnan(x) || isinf(x)) return x * x * x * x * x;
	int neg = x < 0, e;
	float fr = frexpf(fabsf(x), &e);                 /* |x| = fr * 2^e, fr in [0.5, 1) */
	uint32_t m = (uint32_t)ldexpf(fr, 24);            /* |x| = m * 2^(e-24), exact */
	int E = 5 * (e - 24);
	u128_t p = (u128_t)m * m; p *= m; p *= m; p *= m;
	int top = 127;
	while(!((p >> top) & 1)) top--;
	int lead = top + E, keep = 24;
	if(lead > 127) return neg ? -INFINITY : INFINITY;
	if(lead < -126) keep = 24 - (-126 - lead);        /* denormal result: fewer bits kept */
	float r;
	if(keep < 0) r = 0.0f;
	else if(keep == 0) {                              /* value in [2^-150, 2^-149): a tie only at exactly 2^-150 */
		u128_t rest = p & ((((u128_t)1) << top) - 1);
		r = rest ? ldexpf(1.0f, -149) : 0.0f;
	} else {
		int drop = top + 1 - keep;
		u128_t q = drop > 0 ? p >> drop : p << -drop;
		if(drop > 0) {
			u128_t rem = p & ((((u128_t)1) << drop) - 1), half = ((u128_t)1) << (drop - 1);
			if(rem > half || (rem == half && (q & 1))) q++;
		}
		r = ldexpf((float)(uint64_t)q, E + drop);     /* q <= 2^24: exact; overflow to inf is the correct rounding */
	}
	return neg ? -r : r;
}

/* out[i] = exact-arbiter x^5 of the float with bit pattern bits[i] */
void pto_pow5_exact(const uint32_t *bits, long n, float *out) {
	for(long i = 0; i < n; i++) { float x; memcpy(&x, &bits[i], 4); out[i] = pow5_exact(x); }
}

/* All 2^23 significands at biased exponent `expfield`: mismatches against the arbiter of
 * which = 0: pow_f(x, 5) above — what the oracle's reflectance() calls (libm pow in f64, narrowed);
 * which = 1: the three-multiply f64 chain the HIP kernels use (pt_device.h pow5), restated. */
uint64_t pto_pow5_mismatches(int expfield, int which) {
	uint64_t bad = 0;
	for(uint32_t m = 0; m < (1u << 23); m++) {
		uint32_t u = ((uint32_t)expfield << 23) | m, ua, ub;
		float x; memcpy(&x, &u, 4);
		float want = pow5_exact(x), got;
		if(which == 0) got = pow_f(x, 5);
		else { double d = (double)x, d2 = d * d, d4 = d2 * d2; got = (float)(d4 * d); }
		memcpy(&ua, &want, 4); memcpy(&ub, &got, 4);
		bad += ua != ub;
	}
	return bad;
}
